#!/usr/bin/env python3
"""Config 5 alone (DPRNN-Spe-IRA, full-length utterances from the Libri2Mix-test length distribution, B=1 semantics
through inference.separate_many): sweep of the sequence tile and the stream count on one MI355X."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("HWQ", "16"))
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd import engine, inference, sharding
from tss_with_dprnn_amd.synthetic import full_cfg, make_mixture, make_state

BASE = ["input_size", "feature_size", "hidden_size", "chunk_length", "kernel_size", "hop_length", "n_repeats",
        "bidirectional", "rnn_type", "norm_type", "activation_type", "dropout", "stride"]
SPE = BASE + ["O", "P", "embeddings_size", "num_spks", "fusion_type"]
dev = "cuda:0"
kind = sys.argv[1] if len(sys.argv) > 1 else "ira"
cfg = full_cfg(norm_type="ln")
cls = {"bss": pkg.DPRNNTasNet, "spe": pkg.DPRNNSpeTasNet, "ira": pkg.DPRNNSpeIRATasNet}[kind]
m = cls(**{k: cfg[k] for k in (BASE if kind == "bss" else SPE)}).eval()
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state(kind, cfg, 0).items()})
m = m.to(dev)
rng = np.random.default_rng(0)
lens = np.clip(np.exp(rng.normal(np.log(36720), 0.38, 3000)), 24000, 111920).astype(int)
share = sharding.shard_by_length(lens.tolist(), 8)[0]          # this rank's length-balanced 1/8 of the 3000 test utterances
mine = share[:48]
wavs = [torch.from_numpy(make_mixture(1, int(lens[i]), seed=100 + i)[0]).to(dev) for i in mine]
aux = None if kind == "bss" else [(torch.from_numpy(make_mixture(1, 24000, seed=900 + i)[0]).to(dev), 24000.0) for i in mine]
secs = float(sum(lens[i] for i in mine)) / 8000.0
def timed(fn, reps=2):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


with torch.no_grad():
    # one ragged launch per DPRNN path over all utterances (engine.dprnn_process_ragged; bf16-split gate GEMM, the default)
    if os.environ.get("FL_ONLY") != "share":
        dt = timed(lambda: inference.separate_many(m, wavs, extra_inputs=aux, ragged=True))
        print(f"{kind} ragged batch ({len(mine)} utterances, {secs:.0f} s of audio): {dt:.3f} s = {len(mine) / dt:6.1f} utt/s, {secs / dt:5.0f}x real time", flush=True)
    if os.environ.get("FL_ONLY") == "ragged":
        sys.exit(0)
    if os.environ.get("FL_SHARE", "1") == "1":
        # the rank's whole share (375 utterances), sorted by length so that the utterances of a ragged batch (64 at a time)
        # are of similar length and the inter-chunk launches have short tails
        order = sorted(share, key=lambda i: -lens[i])
        w_all = [torch.from_numpy(make_mixture(1, int(lens[i]), seed=100 + i)[0]).to(dev) for i in order]
        a_all = None if kind == "bss" else [(torch.from_numpy(make_mixture(1, 24000, seed=900 + i)[0]).to(dev), 24000.0) for i in order]
        s_all = float(sum(lens[i] for i in order)) / 8000.0
        dt = timed(lambda: inference.separate_many(m, w_all, extra_inputs=a_all, ragged=True), reps=1)
        if os.environ.get("FL_ONLY") == "share":        # for rocprofv3: the whole share twice more, nothing else
            dt = timed(lambda: inference.separate_many(m, w_all, extra_inputs=a_all, ragged=True), reps=2)
            print(f"{kind} ragged, whole share: {dt:.3f} s = {len(order) / dt:6.1f} utt/s", flush=True)
            sys.exit(0)
        print(f"{kind} ragged, the rank's whole share ({len(order)} utterances sorted by length, {s_all:.0f} s of audio, "
              f"{len(inference.ragged_chunks(list(range(len(order))), [int(w.numel()) for w in w_all], dev))} ragged batch(es)): {dt:.3f} s = "
              f"{len(order) / dt:6.1f} utt/s, {s_all / dt:5.0f}x real time", flush=True)
        # parity at THIS size (a slab of > 2^31 floats): the longest, a middle and the shortest utterances against their batch-of-one forwards
        big = inference.separate_many(m, w_all, extra_inputs=a_all, ragged=True)
        worst_big = 0.0
        for j in (0, 1, len(order) // 2, len(order) - 2, len(order) - 1):
            one = inference.separate_many(m, [w_all[j]], extra_inputs=None if a_all is None else [a_all[j]], ragged=False)[0]
            x0, y0 = (one[0], big[j][0]) if isinstance(one, tuple) else (one, big[j])
            worst_big = max(worst_big, float(((x0.double() - y0.double()).norm() / x0.double().norm()).item()))
        print(f"whole-share ragged batch vs batch-of-one forwards (5 utterances): worst rel-L2 {worst_big:.2e}", flush=True)
        del big
        dt = timed(lambda: inference.separate_many(m, w_all, max_streams=48, extra_inputs=a_all, ragged=False), reps=1)
        print(f"{kind} streams=48, the rank's whole share (one utterance per pipeline): {dt:.3f} s = {len(order) / dt:6.1f} utt/s, "
              f"{s_all / dt:5.0f}x real time", flush=True)
        del w_all, a_all
    # the round-1 path: equal-length batches (here: single utterances) on streams, exact fp32 kernels at these sizes
    for ns in (16, 48):
        dt = timed(lambda: inference.separate_many(m, wavs, max_streams=ns, extra_inputs=aux, ragged=False))
        print(f"{kind} streams={ns:2d} (one utterance per pipeline): {dt:.3f} s = {len(mine) / dt:6.1f} utt/s, {secs / dt:5.0f}x real time", flush=True)
    # parity of the two paths on this very workload
    a_ = inference.separate_many(m, wavs[:6], extra_inputs=None if aux is None else aux[:6], ragged=False)
    b_ = inference.separate_many(m, wavs[:6] + wavs[6:12], extra_inputs=None if aux is None else aux[:12], ragged=True)[:6]
    worst = 0.0
    for x, y in zip(a_, b_):
        x0, y0 = (x[0], y[0]) if isinstance(x, tuple) else (x, y)
        worst = max(worst, float(((x0.double() - y0.double()).norm() / x0.double().norm()).item()))
    print(f"ragged vs one-by-one outputs: worst rel-L2 {worst:.2e}")
