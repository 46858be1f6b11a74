#!/usr/bin/env python3
"""Stand-alone times of the bottleneck and the mask-head row GEMMs at the headline size (32 utterances x 23999 frames): conv + overlap-add gather,
gate, mask + decode -- fp32 MFMA forms (gate-GEMM mode 0) against the f16-pieces forms (mode 3)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd import engine

lib = pkg.load_library()
st = engine.Stage(torch.device("cuda:0"))
B, T, K, P = 32, 24000, 250, 125
L = T - 1
S = (L + K) // P + 1
g = lambda *s: torch.randn(*s, device="cuda")
X = g(B * S * K, 128) * 0.5
alpha = torch.full((1,), 0.25, device="cuda")
W2d, b2d = g(256, 128) / 11, g(256)
U = torch.empty(B * L, 256, device="cuda")
Wo, bo, Wg, bg = g(128, 128) / 11, g(128), g(128, 128) / 11, g(128)
V = torch.empty(B * L * 2, 128, device="cuda")
Wend = g(64, 128) / 11
wav = g(B, T) * 0.1
We, Wd = g(64, 1, 2), g(64, 1, 2)
taps = torch.empty(B, 2, L, 2, device="cuda")
p = lambda t: t.data_ptr()
sc, sh = g(B, 64).abs() + 0.5, g(B, 64) * 0.1
Wb, bb = g(128, 64) / 8, g(128)
Y = torch.empty(B * L, 128, device="cuda")
calls = {
    "bottleneck": ("dprnn_bottleneck_fwd", (p(wav), p(We), p(sc), p(sh), p(Wb), 64, p(bb), 0, p(Y), B, T, L, 64, 2, 1), B * L * 128 * 4 / 1e9),
    "head_conv_ola": ("dprnn_head_conv_ola_fwd", (p(X), p(alpha), p(W2d), p(b2d), p(U), B, L, K, P, S), (B * S * K * 128 * 2 / K * P * 0 + B * L * (2 * 2 * 128 + 256)) * 4 / 1e9),
    "head_gate": ("dprnn_head_gate_fwd", (p(U), p(Wo), p(bo), p(Wg), p(bg), p(V), B * L * 2), B * L * 2 * 256 * 4 / 1e9),
    "head_mask_decode": ("dprnn_head_mask_decode_fwd", (p(V), p(Wend), p(wav), p(We), p(Wd), p(taps), 0, 0, B, T, L, 64, 2, 1), B * L * 2 * 128 * 4 / 1e9),
}
for name, (fn, args, gb) in calls.items():
    for mode in (0, 3):
        lib.dprnn_set_gemm_mode(mode)
        for _ in range(3):
            st.call(fn, *args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            st.call(fn, *args)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name:18s} mode {mode}: {ms:.3f} ms per launch, {gb / ms:.2f} TB/s algorithmic ({gb:.2f} GB)", flush=True)
    lib.dprnn_set_gemm_mode(0)
