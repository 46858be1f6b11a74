#!/usr/bin/env python3
"""GroupNorm backward passes (gn_bwd_stats: 4 slabs in, 1 out; gn_bwd_dz: 2 in, 1 out) of one 32-utterance micro-batch: ms and TB/s.
Run per library variant (DPRNN_LIB=...) to compare DPRNN_GN_ROWS builds.  usage: python tools/probes/gn_bwd_rows.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tss_with_dprnn_amd as pkg  # noqa: E402,F401
from tss_with_dprnn_amd import engine  # noqa: E402
from tools.bench_serial import timed  # noqa: E402

st = engine.Stage(torch.device("cuda:0"))
p = lambda t: t.data_ptr()
B, S, K = 32, 194, 250
rows = B * S * K
g = torch.Generator(device="cuda").manual_seed(1)
dX = torch.randn(rows, 128, device="cuda", generator=g)
dXd = torch.randn(2, rows, 128, device="cuda", generator=g)
Z = torch.randn(rows, 128, device="cuda", generator=g)
dZ = torch.empty_like(Z)
stats = torch.zeros(B, 2, device="cuda", dtype=torch.float64)
stats[:, 1] = S * K * 128
gamma = torch.ones(128, device="cuda")
red = torch.zeros(2 * B + 256, device="cuda", dtype=torch.float64)
zmax = torch.zeros(1, device="cuda")
unit = rows * 512 / 1e9
t1 = timed(lambda: st.call("dprnn_gn_bwd_stats", p(dX), p(dXd[0]), p(dXd[1]), p(dX), p(Z), p(stats), p(gamma), 1e-5, B, S * K, p(red), red.data_ptr() + 8 * 2 * B), 50)
t2 = timed(lambda: st.call("dprnn_gn_bwd_dz_ex", p(zmax), p(dX), p(Z), p(stats), p(red), p(gamma), 1e-5, B, S * K, p(dZ)), 50)
print(f"lib {os.environ.get('DPRNN_LIB', 'default')}: gn_bwd_stats {t1:.3f} ms = {5 * unit / t1:.2f} TB/s | gn_bwd_dz {t2:.3f} ms = {3 * unit / t2:.2f} TB/s", flush=True)
