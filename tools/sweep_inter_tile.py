#!/usr/bin/env python3
"""Forward time by batch size with the inter-chunk (and intra-chunk) recurrence tile forced, mode 3: where should split_tile switch from
the 16-sequence resident-weights tile to the 64-sequence one?  usage: sweep_inter_tile.py 24:32:1 [intra codes] [inter codes]"""
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
a, b, stp = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "24:32:1").split(":"))
intras = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
inters = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,2016,2064").split(",")]
engine.GEMM_MODE = 3
for B in range(a, b + 1, stp):
    x = torch.from_numpy(make_mixture(B, 24000, seed=1)).to("cuda:0")
    cells = []
    for ti in intras:
        for te in inters:
            engine.SEQ_TILE["intra"], engine.SEQ_TILE["inter"] = ti, te
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
            cells.append(f"intra {ti or 'auto':>4} inter {te or 'auto':>4}: {ms:6.2f} ms {B / ms * 1e3:6.1f} utt/s")
    print(f"B={B:3d}  " + "   ".join(cells), flush=True)
engine.SEQ_TILE["intra"] = engine.SEQ_TILE["inter"] = 0
