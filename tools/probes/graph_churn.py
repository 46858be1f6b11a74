#!/usr/bin/env python3
"""Captures and replays forwards of many (batch size, mode) keys in one process: does graph eviction / re-capture hold up?
    python tools/probes/graph_churn.py CACHE_MAX SIZES [ROUNDS]      e.g.  4 16,24,32,48,64,96,128"""
import faulthandler, os, sys
faulthandler.enable()
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd import engine
from tss_with_dprnn_amd.synthetic import full_cfg, make_mixture, make_state

CTOR = ["input_size", "feature_size", "hidden_size", "chunk_length", "kernel_size", "hop_length", "n_repeats",
        "bidirectional", "rnn_type", "norm_type", "activation_type", "dropout", "stride"]
engine.GRAPH_CACHE_MAX = int(sys.argv[1])
sizes = [int(v) for v in sys.argv[2].split(",")]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 1
samples = int(os.environ.get("SAMPLES", "24000"))
cfg = full_cfg()
model = pkg.DPRNNTasNet(**{k: cfg[k] for k in CTOR}).eval()
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state("bss", cfg, 0).items()})
model = model.to("cuda:0")
for rd in range(rounds):
    for B in sizes:
        x = torch.from_numpy(make_mixture(B, samples, seed=1)).to("cuda:0")
        for mode in (0, 1, 3):
            engine.GEMM_MODE, engine.SPLIT_MIN_BATCH = mode, 0
            with torch.no_grad():
                outs = [model(x) for _ in range(4)]
            torch.cuda.synchronize()
            assert all(torch.equal(outs[0], o) for o in outs[1:]), (B, mode)
            print(f"round {rd} B={B} mode {mode}: ok, graphs {engine.GRAPH_STATS}, cached {len(engine._GRAPH_CACHE)}, "
                  f"reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB", flush=True)
print("done")
