#!/usr/bin/env python3
"""Stand-alone time of the Linear 256->128 + GroupNorm sums row GEMM (dprnn_proj_stats_fwd) at the headline size
(32 utterances x 194 x 250 rows), fp32 MFMA form (gate-GEMM mode 0) against the f16-pieces form (mode 3)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd import engine

lib = pkg.load_library()
st = engine.Stage(torch.device("cuda:0"))
B, rows_per = 32, 194 * 250
M = B * rows_per
for kdim in (256, 128):
    H = torch.tanh(torch.randn(M, kdim, device="cuda"))
    W = torch.randn(128, kdim, device="cuda") / kdim ** 0.5
    b = torch.randn(128, device="cuda")
    Z = torch.empty(M, 128, device="cuda")
    stats = torch.zeros(B, 2, device="cuda", dtype=torch.float64)
    gb = M * (kdim + 128) * 4 / 1e9
    for mode in (0, 3):
        lib.dprnn_set_gemm_mode(mode)
        for _ in range(3):
            st.call("dprnn_proj_stats_fwd", H.data_ptr(), W.data_ptr(), b.data_ptr(), Z.data_ptr(), stats.data_ptr(), M, rows_per, kdim)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 20
        for _ in range(n):
            st.call("dprnn_proj_stats_fwd", H.data_ptr(), W.data_ptr(), b.data_ptr(), Z.data_ptr(), stats.data_ptr(), M, rows_per, kdim)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"kdim {kdim} mode {mode}: {ms:.3f} ms per launch, {gb / ms:.2f} TB/s algorithmic ({gb:.2f} GB), "
              f"{M * kdim * 128 * 2 / ms / 1e9:.1f} TFLOP/s", flush=True)
    lib.dprnn_set_gemm_mode(0)
