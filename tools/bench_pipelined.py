#!/usr/bin/env python3
"""Experiment: bench.py's workload (B=64, 3 s) with consecutive forwards issued on alternating caller streams, so the
tail of forward n overlaps the head of forward n+1 (a double-buffered serving loop).  Not the bench.py headline."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd.synthetic import full_cfg, make_mixture, make_state

CTOR = ["input_size", "feature_size", "hidden_size", "chunk_length", "kernel_size", "hop_length", "n_repeats",
        "bidirectional", "rnn_type", "norm_type", "activation_type", "dropout", "stride"]
dev = torch.device("cuda:0")
cfg = full_cfg()
model = pkg.DPRNNTasNet(**{k: cfg[k] for k in CTOR}).eval()
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state("bss", cfg, 0).items()})
model = model.to(dev)
B, K = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 8
xs = [torch.from_numpy(make_mixture(B, 24000, seed=1234 + i)).to(dev) for i in range(2)]
with torch.no_grad():
    ref = [model(x) for x in xs]
    torch.cuda.synchronize()
    for depth in (1, 2, 3):
        streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        outs = [None] * K
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            s = streams[i % depth]
            with torch.cuda.stream(s):
                outs[i] = model(xs[i % 2])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        err = max(float((outs[i] - ref[i % 2]).abs().max()) for i in range(K))
        print(f"forwards in flight {depth}: {K} steps in {dt * 1e3:.1f} ms = {K * B / dt:.1f} utt/s, max |diff| vs sequential {err:.1e}", flush=True)
