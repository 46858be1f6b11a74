#!/usr/bin/env python3
"""Does a small third pipeline for the utterances beyond a round boundary beat the balanced split?  Forward time (mode 3) at the
batch sizes where throughput steps down, with engine.GROUP_SIZES forced.  usage: sweep_spill_group.py "21:10,11/10,10,1 33:16,17/16,16,1"."""
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
engine.GEMM_MODE = 3
spec = sys.argv[1] if len(sys.argv) > 1 else "21:10,11/10,10,1 33:16,17/16,16,1 37:18,19/18,18,1 43:21,22/21,21,1"
for item in spec.split():
    B, alts = item.split(":")
    B = int(B)
    x = torch.from_numpy(make_mixture(B, 24000, seed=1)).to("cuda:0")
    cells = []
    for alt in ["auto"] + alts.split("/"):
        engine.GROUP_SIZES = None if alt == "auto" else [int(v) for v in alt.split(",")]
        engine.drop_graphs(release_pool=False)
        with torch.no_grad():
            for _ in range(3):
                model(x)
            torch.cuda.synchronize()
            n = 6
            t0 = time.perf_counter()
            for _ in range(n):
                model(x)
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        cells.append(f"{alt:>10}: {ms:6.2f} ms {B / ms * 1e3:6.1f} utt/s")
    print(f"B={B:3d}  " + "   ".join(cells), flush=True)
engine.GROUP_SIZES = None
