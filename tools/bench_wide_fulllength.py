#!/usr/bin/env python3
"""Full-length utterances (Libri2Mix-test length distribution, 3 .. 14 s) through a DPRNN-TasNet WIDER than the build: ragged batches
(wide.tasnet_forward_ragged via inference.separate_many) against equal-length buckets on streams (ragged=False: one utterance per forward
here, every length differs).  usage: tools/bench_wide_fulllength.py [n_utts] [F] [H]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd import inference
from tss_with_dprnn_amd.synthetic import full_cfg, make_mixture, make_state

BASE = ["input_size", "feature_size", "hidden_size", "chunk_length", "kernel_size", "hop_length", "n_repeats",
        "bidirectional", "rnn_type", "norm_type", "activation_type", "dropout", "stride"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
H = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = "cuda:0"
cfg = full_cfg(feature_size=F, hidden_size=H)
m = pkg.DPRNNTasNet(**{k: cfg[k] for k in BASE}).eval()
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state("bss", cfg, 0).items()})
m = m.to(dev)
rng = np.random.default_rng(0)
lens = np.sort(np.clip(np.exp(rng.normal(np.log(36720), 0.38, n)), 24000, 111920).astype(int))[::-1]
wavs = [torch.from_numpy(make_mixture(1, int(t), seed=100 + i)[0]).to(dev) for i, t in enumerate(lens)]
secs = float(lens.sum()) / 8000.0
for ragged in (True, False, True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = inference.separate_many(m, wavs, ragged=ragged)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"F={F} H={H}, {n} utterances, {secs:.0f} s of audio, ragged={ragged}: {dt:.2f} s = {n / dt:.1f} utt/s = {secs / 3 / dt:.1f} 3-s equivalents/s, "
          f"{secs / dt:.0f} x real time; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    if ragged:
        keep = [o.clone() for o in outs]
    else:
        worst = max(float((a - b).norm() / b.norm()) for a, b in zip(outs, keep))
        print(f"  worst rel-L2 ragged vs per-utterance forwards: {worst:.2e}")
