#!/usr/bin/env python3
"""Fixed per-launch cost of the folded recurrence launches: time against the number of steps on one memory layout (intra-chunk
addressing, 1000 rows per sequence), 194 workgroups.  usage: python tools/probes/split_fixed_cost.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tss_with_dprnn_amd as pkg  # noqa: E402,F401
from tss_with_dprnn_amd import engine  # noqa: E402
from tools.bench_serial import timed  # noqa: E402

st = engine.Stage(torch.device("cuda:0"))
lib = st.lib
B, S, KL = 32, 194, 1000
nseq = B * S
w = [torch.randn(512, 128, device="cuda") * 0.08, torch.randn(512, 128, device="cuda") * 0.08,
     torch.randn(512, device="cuda") * 0.05, torch.randn(512, device="cuda") * 0.05]
packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device="cuda")
bsum = torch.empty(2 * 512, device="cuda")
st.call("dprnn_rnn_pack_ex", 3, 0, *[t.data_ptr() for t in w], *[t.data_ptr() for t in w], packed.data_ptr(), bsum.data_ptr())
W = torch.randn(128, 256, device="cuda") / 16
b = torch.randn(128, device="cuda")
linp = torch.empty(lib.dprnn_lin_pack_floats(), device="cuda")
st.call("dprnn_lin_pack_f16", W.data_ptr(), 2, linp.data_ptr())
gamma, beta = torch.ones(128, device="cuda"), torch.zeros(128, device="cuda")
rows = nseq * KL
X = torch.randn(rows, 128, device="cuda") * 0.5
X2 = torch.empty_like(X)
Z = torch.randn(rows, 128, device="cuda")
Z2 = torch.empty_like(Z)
H = torch.empty(rows, 256, device="cuda")
stats_in = torch.zeros(B, 2, device="cuda", dtype=torch.float64)
stats_in[:, 1] = S * KL * 128
stats = torch.zeros(B, 2, device="cuda", dtype=torch.float64)
state = torch.empty(lib.dprnn_rnn_lin_state_floats(nseq, 2064), device="cuda")
norm = (Z.data_ptr(), stats_in.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-8)
res = {}
for T in (20, 60, 124, 250, 500, 1000):
    addr = (nseq, T, nseq, 0, KL, 1)
    fns = {
        "unfolded, both directions": lambda: st.call("dprnn_rnn_fwd_hp_ex", 3, 0, X.data_ptr(), *norm, S, S * KL, X2.data_ptr(), packed.data_ptr(), bsum.data_ptr(), H.data_ptr(), *addr, 2, 2064),
        "split phase 1": lambda: st.call("dprnn_rnn_lin_split_fwd", 1, X.data_ptr(), *norm, S, S * KL, X2.data_ptr(), packed.data_ptr(), bsum.data_ptr(), linp.data_ptr(), 0, Z2.data_ptr(), 0, state.data_ptr(), *addr, 2064),
        "split phase 2": lambda: st.call("dprnn_rnn_lin_split_fwd", 2, X2.data_ptr(), 0, 0, 0, 0, 0.0, S, S * KL, 0, packed.data_ptr(), bsum.data_ptr(), linp.data_ptr(), b.data_ptr(), Z2.data_ptr(), stats.data_ptr(), state.data_ptr(), *addr, 2064),
        "serial phase 1 (one direction)": lambda: st.call("dprnn_rnn_lin_fwd", 1, 0, X.data_ptr(), *norm, S, S * KL, X2.data_ptr(), packed.data_ptr(), bsum.data_ptr(), 2, linp.data_ptr(), 0, Z2.data_ptr(), 0, 0, *addr, 2064),
    }
    for k, f in fns.items():
        res.setdefault(k, []).append((T, timed(f, 5)))
for k, v in res.items():
    steps = [(T // 2 if "split" in k else T) for T, _ in v]
    (s0, t0), (s1, t1) = (steps[-2], v[-2][1]), (steps[-1], v[-1][1])
    slope = (t1 - t0) / (s1 - s0)
    print(k + ": " + ", ".join(f"{s} steps {t * 1e3:.0f} us" for s, (_, t) in zip(steps, v)) +
          f" | {slope * 1e3:.2f} us per step, fixed " + ", ".join(f"{(t - slope * s) * 1e3:.0f}" for s, (_, t) in zip(steps, v)) + " us", flush=True)
