#!/usr/bin/env python3
"""Forward latency by batch size, eager launches against the replayed HIP graph (engine.GRAPHS), default arithmetic (mode 3)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd import engine
from tss_with_dprnn_amd.synthetic import full_cfg, make_mixture, make_state

CTOR = ["input_size", "feature_size", "hidden_size", "chunk_length", "kernel_size", "hop_length", "n_repeats",
        "bidirectional", "rnn_type", "norm_type", "activation_type", "dropout", "stride"]
cfg = full_cfg()
model = pkg.DPRNNTasNet(**{k: cfg[k] for k in CTOR}).eval()
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state("bss", cfg, 0).items()})
model = model.to("cuda:0")
for B in (1, 2, 4, 8, 16, 32, 64):
    x = torch.from_numpy(make_mixture(B, 24000, seed=1)).to("cuda:0")
    res = {}
    for graphs in (False, True):
        engine.GRAPHS = graphs
        engine.drop_graphs()
        with torch.no_grad():
            for _ in range(3):
                model(x)
            torch.cuda.synchronize()
            n = 20 if B <= 16 else 10
            ts = []
            for _ in range(n):
                t0 = time.perf_counter()
                model(x)
                torch.cuda.synchronize()
                ts.append(1e3 * (time.perf_counter() - t0))
        res[graphs] = (float(np.median(ts)), float(np.max(ts)))
    print(f"B={B:2d}: eager {res[False][0]:7.2f} ms (max {res[False][1]:6.2f})   graph replay {res[True][0]:7.2f} ms (max {res[True][1]:6.2f})   "
          f"{B / res[True][0] * 1e3:7.1f} utt/s", flush=True)
