#!/usr/bin/env python3
"""Which pair of torch's 32 pooled streams makes a forward slow?  Round 3's first bench showed one forward in 16 taking +5 ms with
no allocator activity and no cgroup throttling (bench.py "host" / "latency.slowest_steps": steps 12, 28, 44 of 50): the two
utterance groups took two FRESH torch.cuda.Stream() objects per forward, i.e. walked torch's round-robin pool of 32 with period
16.  This probe pins the pair: for every position of the cycle it hands the engine that pair as its side streams and times
forwards."""
import os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd import engine
from tss_with_dprnn_amd.synthetic import full_cfg, make_mixture, make_state

CTOR = ["input_size", "feature_size", "hidden_size", "chunk_length", "kernel_size", "hop_length", "n_repeats",
        "bidirectional", "rnn_type", "norm_type", "activation_type", "dropout", "stride"]
cfg = full_cfg()
model = pkg.DPRNNTasNet(**{k: cfg[k] for k in CTOR}).eval()
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state("bss", cfg, 0).items()})
model = model.to("cuda:0")
x = torch.from_numpy(make_mixture(64, 24000, seed=1)).to("cuda:0")
engine.GRAPHS = False
dev = x.device
key = (threading.get_ident(), str(dev))
with torch.no_grad():
    for _ in range(3):
        model(x)
    torch.cuda.synchronize()
    for pos in range(20):
        pair = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]          # the next two of torch's round-robin pool
        engine._SIDE_POOL[key] = pair
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            model(x)
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        print(f"pool position {pos:2d}: streams {pair[0].cuda_stream:#x} {pair[1].cuda_stream:#x}  forward ms {' '.join(f'{t:.1f}' for t in ts)}", flush=True)
