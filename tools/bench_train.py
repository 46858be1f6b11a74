#!/usr/bin/env python3
"""Training-step timing on one MI355X: forward (with tape) + PIT SI-SDR loss + backward (+ Adam), shipped config,
3 s utterances.  Prints utt/s and the per-entry-point time of one profiled step (HIP events around every C-ABI call).

    python tools/bench_train.py --batch 16 --steps 3
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd import engine, training
from tss_with_dprnn_amd.training import PITSiSdrLoss


class _All:
    def __contains__(self, k):
        return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--samples", type=int, default=24000)
    ap.add_argument("--budget-gb", type=float, default=None)
    ap.add_argument("--trainer-step", type=int, default=0, help="1: time trainer.train_step instead of the explicit call sequence")
    ap.add_argument("--keep-max-first", type=int, default=None, help="override training.KEEP_MAX_FIRST (0 / 1)")
    ap.add_argument("--model", default="bss", choices=["bss", "spe"], help="DPRNNTasNet, or DPRNNSpeTasNet ('att' fusion, config_tss.yaml)")
    ap.add_argument("--groups", type=int, default=None, help="training.TRAIN_STREAM_GROUPS")
    ap.add_argument("--min-group", type=int, default=None, help="training.MIN_TRAIN_GROUP")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam + clip_grad_norm_ instead of training.FusedAdam")
    a = ap.parse_args()
    if a.keep_max_first is not None:
        training.KEEP_MAX_FIRST = bool(a.keep_max_first)
    if a.budget_gb is not None:
        training.SAVE_BUDGET_BYTES = int(a.budget_gb * (1 << 30))
    if a.groups is not None:
        training.TRAIN_STREAM_GROUPS = a.groups
    if a.min_group is not None:
        training.MIN_TRAIN_GROUP = a.min_group
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    if a.model == "bss":
        model = pkg.DPRNNTasNet(64, 128, 128, 250, 2, 125, 6, True, "LSTM", "ln", "sigmoid", 0).to(dev).train()
    else:
        model = pkg.DPRNNSpeTasNet(64, 128, 128, 250, 2, 125, 6, True, "LSTM", "ln", "sigmoid", 0, None, 128, 256, 128, 251,
                                   "att").to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3) if a.torch_adam else training.FusedAdam(model.parameters(), lr=1e-3)
    lossm = PITSiSdrLoss()
    g = torch.Generator(device="cpu").manual_seed(1)
    src = 0.07 * torch.randn(a.batch, 2, a.samples, generator=g)
    mix = src.sum(1).to(dev)
    tgt = src.to(dev)
    aux = (0.07 * torch.randn(a.batch, a.samples, generator=g)).to(dev)
    spk = torch.randint(0, 251, (a.batch,), generator=g).to(dev)
    ref_len = torch.tensor(float(a.samples), device=dev)

    def step():
        if a.trainer_step and a.model == "bss":           # trainer.train_step: micro-batches with the loss inside
            from tss_with_dprnn_amd import trainer
            return trainer.train_step(model, lossm, opt, mix, tgt, clip_norm=5)[0]
        opt.zero_grad()
        if a.model == "bss":
            out = model(mix)
            loss = lossm(out, tgt)
        else:                                   # trainer_spe.py:36-45
            est, logits = model(mix, aux, ref_len)
            loss = lossm(est.unsqueeze(1), tgt[:, :1, : est.shape[-1]]) + 0.5 * torch.nn.functional.cross_entropy(logits, spk)
        loss.backward()
        if a.torch_adam:
            torch.nn.utils.clip_grad_norm_(model.parameters(), 5)
            opt.step()
        else:
            opt.step(clip_norm=5)
        return loss

    l0 = step().item()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"batch {a.batch}: {dt * 1e3:.1f} ms/step, {a.batch / dt:.1f} utt/s, loss {l0:.4f} -> {loss.item():.4f}, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    # forward alone (inference path, for the ratio)
    model.eval()
    fwd = (lambda: model(mix)) if a.model == "bss" else (lambda: model(mix, aux, ref_len))
    with torch.no_grad():
        fwd(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fwd()
        torch.cuda.synchronize()
    fdt = (time.perf_counter() - t0) / a.steps
    print(f"inference forward: {fdt * 1e3:.1f} ms ({a.batch / fdt:.1f} utt/s); training step = {dt / fdt:.2f} x forward")
    model.train()
    engine.PROFILE = {"names": _All(), "events": {}}
    step()
    torch.cuda.synchronize()
    prof, engine.PROFILE = engine.PROFILE, None
    rows = sorted(((sum(e0.elapsed_time(e1) for e0, e1 in ev), len(ev), n) for n, ev in prof["events"].items()), reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"profiled step: {tot:.1f} ms inside C-ABI calls")
    for ms, cnt, n in rows[:16]:
        print(f"  {n:34s} {cnt:4d} calls {ms:9.2f} ms {100 * ms / tot:5.1f} %")
    # fp32-MFMA roofline of the three dominant kernels (peak 157.3 TFLOP/s): algorithmic FLOPs, DESIGN.md section 8
    geo = engine.make_geometry(a.batch, a.samples, 64, 250, 125, 2, 1)
    rws = geo.rows
    lstm = 12 * rws * 2 * (2 * 256 * 512)                  # 12 paths x 2 directions, [x|h] (256) x 4 gates (512) per row
    wg = 12 * (2 * rws * 2 * 512 * 256 + rws * 2 * 128 * 256)     # LSTM dW (both directions) + Linear dW per path
    byname = {n: ms for ms, _, n in rows}
    for names, fl in ((("dprnn_lstm_fwd_train", "dprnn_rnn_fwd_train"), lstm), (("dprnn_lstm_bwd", "dprnn_rnn_bwd"), lstm),
                      (("dprnn_wgrad",), wg)):
        n = next((x for x in names if x in byname), None)
        if n is None:
            continue
        tf = fl / (byname[n] * 1e-3) / 1e12
        print(f"roofline {n:22s} {fl / 1e12:6.2f} TFLOP in {byname[n]:7.2f} ms = {tf:6.1f} TFLOP/s = {tf / 157.3:.2f} of the fp32 MFMA peak")


if __name__ == "__main__":
    main()
