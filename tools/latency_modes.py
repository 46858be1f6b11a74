#!/usr/bin/env python3
"""Forward time of DPRNN-TasNet by batch size under the fp32-class gate-GEMM modes (engine.GEMM_MODE 0 = exact fp32 MFMA,
1 = bf16 split x6, 3 = f16 pieces x3) with the engine's own tile / group choices and the split kernels forced at every
size (engine.SPLIT_MIN_BATCH = 0): where does which mode win?"""
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
def _sizes(spec):
    out = []
    for part in spec.split(","):
        if ":" in part:                      # a:b:step, b inclusive
            a, b, st = (int(v) for v in part.split(":"))
            out += list(range(a, b + 1, st))
        else:
            out.append(int(part))
    return out


sizes = _sizes(sys.argv[1]) if len(sys.argv) > 1 else [1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128]
modes = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (0, 1, 3)
if modes != (0, 1, 3):
    names = {0: "exact fp32", 1: "bf16 split x6", 3: "f16 pieces x3"}
    prev = {m: 0.0 for m in modes}
    for B in sizes:
        x = torch.from_numpy(make_mixture(B, 24000, seed=1)).to("cuda:0")
        cells = []
        for mode in modes:
            engine.GEMM_MODE, engine.SPLIT_MIN_BATCH = mode, 0
            with torch.no_grad():
                for _ in range(3):
                    model(x)
                torch.cuda.synchronize()
                n = 4 if B >= 32 else 8
                t0 = time.perf_counter()
                for _ in range(n):
                    model(x)
                torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            ups = B / ms * 1e3
            cells.append(f"{names[mode]} {ms:8.2f} ms {ups:7.1f} utt/s {'  <-- below B-1 by %.1f %%' % (100 * (1 - ups / prev[mode])) if ups < 0.98 * prev[mode] else ''}")
            prev[mode] = max(prev[mode], ups) if False else ups
        print(f"B={B:3d}: " + "   ".join(cells), flush=True)
        engine.drop_graphs(release_pool=False)
    sys.exit(0)
for B in sizes:
    x = torch.from_numpy(make_mixture(B, 24000, seed=1)).to("cuda:0")
    row = []
    for mode in (0, 1, 3):
        engine.GEMM_MODE, engine.SPLIT_MIN_BATCH = mode, 0
        with torch.no_grad():
            for _ in range(2):
                model(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 4 if B >= 32 else 8
            for _ in range(n):
                model(x)
            torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / n * 1e3)
    print(f"B={B:3d}: exact fp32 {row[0]:8.2f} ms ({B / row[0] * 1e3:6.1f} utt/s)   bf16 split x6 {row[1]:8.2f} ms ({B / row[1] * 1e3:6.1f} utt/s)   "
          f"f16 pieces x3 {row[2]:8.2f} ms ({B / row[2] * 1e3:6.1f} utt/s)   exact / f16 {row[0] / row[2]:.2f}", flush=True)
