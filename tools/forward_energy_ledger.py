#!/usr/bin/env python3
"""Where the joules of one DPRNN-TasNet forward (64 utterances of 3 s, the headline) go, by kernel class -- the forward runs at the
board's power limit, so this is also where its time goes ("Round 4").

Measured with the socket's energy counter, each item queued back to back for a few seconds on an otherwise idle chip:
  * the whole forward (the product path: graph replay of model(x), B = 64);
  * the recurrence launches of one 32-utterance pipeline in the forward's own shapes: intra-chunk (194 workgroups of 64 sequences,
    250 steps) with and without the pending norm folded in, inter-chunk (250 workgroups, 194 steps), h leaving as piece quads;
  * the projection that follows every recurrence (Linear 256 -> 128 + GroupNorm sums on h piece quads, 32 utterances).
Ledger: 2 pipelines x (1 + 5 intra + 6 inter recurrences + 12 projections) against the whole forward; the remainder is the prologue
(encoder + bottleneck, segmentation), the mask head and the decoder.  Stand-alone items run at their own clock and power, so the
ledger is a budget in joules, not a timeline."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from energy_budget import _hwmon_energy_file, read_energy_uj, read_power_clock


def measure_calls(label, call, seconds, ms_hint=None):
    """Queue `call()` for ~`seconds` after a warm-up third; joules, milliseconds and watts per call."""
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms_est = e0.elapsed_time(e1) / 10
    n = max(20, int(seconds * 1e3 / ms_est))
    for _ in range(n // 3):
        call()
    torch.cuda.synchronize()
    src = _hwmon_energy_file()
    uj0, how = read_energy_uj(src)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    mid = read_power_clock()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    uj1, _ = read_energy_uj(src)
    gpu_ms = e0.elapsed_time(e1)
    r = {"label": label, "calls": n, "ms_per_call": gpu_ms / n, "joules_per_call": (uj1 - uj0) / 1e6 / n,
         "mean_watts": (uj1 - uj0) / 1e6 / (t1 - t0), "mid_run_watts_sclk": mid, "energy_source": how}
    print(json.dumps(r), flush=True)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--fold", choices=["off", "auto"], default="auto",
                    help="the forward and the per-launch items with the Linear folded into the recurrence (time-split, two launches per path) or not")
    a = ap.parse_args()
    import tss_with_dprnn_amd as pkg
    from tss_with_dprnn_amd import engine
    from tss_with_dprnn_amd.synthetic import full_cfg, make_mixture, make_state
    engine.LIN_FOLD = a.fold
    dev = torch.device("cuda:0")
    ctor = ["input_size", "feature_size", "hidden_size", "chunk_length", "kernel_size", "hop_length", "n_repeats",
            "bidirectional", "rnn_type", "norm_type", "activation_type", "dropout", "stride"]
    cfg = full_cfg()
    model = pkg.DPRNNTasNet(**{k: cfg[k] for k in ctor}).eval()
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state("bss", cfg, 0).items()})
    model = model.to(dev)
    x = torch.from_numpy(make_mixture(64, 24000, seed=1)).to(dev)
    res = {}
    with torch.no_grad():
        res["forward"] = measure_calls("whole forward, B = 64 (graph replay)", lambda: model(x), a.seconds)
    engine.drop_graphs()
    time.sleep(1.0)
    uj0, _ = read_energy_uj(_hwmon_energy_file()); t0 = time.perf_counter()
    time.sleep(1.0)
    uj1, _ = read_energy_uj(_hwmon_energy_file()); idle_w = (uj1 - uj0) / 1e6 / (time.perf_counter() - t0)

    st = engine.Stage(dev)
    g = torch.Generator(device="cuda").manual_seed(1)
    B, S, K = 32, 194, 250
    rows = B * S * K
    X = torch.randn(rows, 128, device="cuda", generator=g) * 0.5
    Z = torch.randn(rows, 128, device="cuda", generator=g)
    X2 = torch.empty_like(X)
    H = torch.empty(rows, 256, device="cuda")
    stats = torch.zeros(B, 2, device="cuda", dtype=torch.float64)
    stats[:, 1] = S * K * 128
    gamma, beta = torch.ones(128, device="cuda"), torch.zeros(128, device="cuda")
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device="cuda")
    bsum = torch.empty(2 * 512, device="cuda")
    w = [torch.randn(512, 128, device="cuda", generator=g) * 0.08, torch.randn(512, 128, device="cuda", generator=g) * 0.08,
         torch.randn(512, device="cuda", generator=g) * 0.05, torch.randn(512, device="cuda", generator=g) * 0.05]
    p = lambda t: t.data_ptr()
    st.call("dprnn_rnn_pack_ex", 3, 0, *[p(t) for t in w], *[p(t) for t in w], p(packed), p(bsum))
    intra, inter = (B * S, K, B * S, 0, K, 1), (B * K, S, K, S * K, 1, K)
    fn = "dprnn_rnn_fwd_hp_ex"
    def rec(addr, per, fuse):
        if fuse:
            args = (3, 0, p(X), p(Z), p(stats), p(gamma), p(beta), 1e-8, per, S * K, p(X2), p(packed), p(bsum), p(H), *addr, 2, 2064)
        else:
            args = (3, 0, p(X), 0, 0, 0, 0, 0.0, 1, 1, 0, p(packed), p(bsum), p(H), *addr, 2, 2064)
        return lambda: st.call(fn, *args)
    res["intra_first"] = measure_calls("recurrence, intra-chunk, first path (no pending norm)", rec(intra, S, 0), a.seconds)
    res["intra"] = measure_calls("recurrence, intra-chunk, pending norm folded in", rec(intra, S, 1), a.seconds)
    res["inter"] = measure_calls("recurrence, inter-chunk, pending norm folded in", rec(inter, K, 1), a.seconds)
    Wp = torch.randn(128, 256, device="cuda", generator=g) / 16
    bp = torch.randn(128, device="cuda", generator=g)
    Zo = torch.empty(rows, 128, device="cuda")
    stats2 = torch.zeros(B, 2, device="cuda", dtype=torch.float64)
    res["proj"] = measure_calls("projection 256 -> 128 + GroupNorm sums on h piece quads",
                                lambda: st.call("dprnn_proj_stats_hp_fwd", p(H), p(Wp), p(bp), p(Zo), p(stats2), rows, S * K, 0, 0, 256), a.seconds)

    if a.fold != "off":
        # the folded form (round 5): each path is two launches of both directions, cut in time; the Linear rides in the h half of the gate GEMM
        linp = torch.empty(st.lib.dprnn_lin_pack_floats(), device="cuda")
        st.call("dprnn_lin_pack_f16", p(Wp), 2, p(linp))
        Z2 = torch.empty_like(Z)
        state = torch.empty(max(st.lib.dprnn_rnn_lin_state_floats(n, 2064) for n in (B * S, B * K)), device="cuda")

        def phase(ph, addr, per, fuse):
            tail = (p(packed), p(bsum), p(linp))
            if ph == 1:
                norm = (p(Z), p(stats), p(gamma), p(beta), 1e-8) if fuse else (0, 0, 0, 0, 0.0)
                args = (1, p(X), *norm, per, S * K, p(X2) if fuse else 0, *tail, 0, p(Z2), 0, p(state), *addr, 2064)
            else:
                args = (2, p(X2), 0, 0, 0, 0, 0.0, per, S * K, 0, *tail, p(bp), p(Z2), p(stats2), p(state), *addr, 2064)
            return lambda: st.call("dprnn_rnn_lin_split_fwd", *args)
        # phase 2 needs the state phase 1 of the same path left: run each pair once before timing the second alone
        for nm, addr, per in (("intra", intra, S), ("inter", inter, K)):
            phase(1, addr, per, 1)()
            res[nm + "_p1"] = measure_calls(f"folded {nm}-chunk path, phase 1 (first halves in time; pending norm folded in)", phase(1, addr, per, 1), a.seconds)
            res[nm + "_p2"] = measure_calls(f"folded {nm}-chunk path, phase 2 (second halves; new z and its sums)", phase(2, addr, per, 0), a.seconds)
        f = res["forward"]
        items = [("folded intra paths, phase 1 (12)", 12, res["intra_p1"]), ("folded intra paths, phase 2 (12)", 12, res["intra_p2"]),
                 ("folded inter paths, phase 1 (12)", 12, res["inter_p1"]), ("folded inter paths, phase 2 (12)", 12, res["inter_p2"])]
        ledger(f, idle_w, items)
        print("\n# for comparison, the unfolded launches of the same shapes on this box (not part of this forward):")
        for k in ("intra", "inter", "proj"):
            r = res[k]
            print(f"#   {k:6s} {r['ms_per_call']:.3f} ms, {r['joules_per_call']:.3f} J each")
        return
    items = [("recurrences, first path (2)", 2, res["intra_first"]), ("recurrences, intra-chunk (10)", 10, res["intra"]),
             ("recurrences, inter-chunk (12)", 12, res["inter"]), ("projections (24)", 24, res["proj"])]
    ledger(res["forward"], idle_w, items)


def ledger(f, idle_w, items):
    print(f"\n# forward: {f['joules_per_call']:.1f} J and {f['ms_per_call']:.2f} ms per 64 utterances at {f['mean_watts']:.0f} W "
          f"(mid-run {f['mid_run_watts_sclk']}); idle {idle_w:.0f} W = {idle_w * f['ms_per_call'] / 1e3:.1f} J of it")
    print(f"{'item':34s} {'launches':>8s} {'ms each':>8s} {'W alone':>8s} {'J each':>7s} {'J total':>8s} {'share of the forward':>22s}")
    tot = 0.0
    for name, cnt, r in items:
        e = cnt * r["joules_per_call"]
        tot += e
        print(f"{name:34s} {cnt:8d} {r['ms_per_call']:8.3f} {r['mean_watts']:8.0f} {r['joules_per_call']:7.3f} {e:8.1f} {100 * e / f['joules_per_call']:21.1f} %")
    rest = f["joules_per_call"] - tot
    print(f"{'prologue, mask head, decoder':34s} {'':8s} {'':8s} {'':8s} {'':7s} {rest:8.1f} {100 * rest / f['joules_per_call']:21.1f} %   (remainder)")


if __name__ == "__main__":
    main()
