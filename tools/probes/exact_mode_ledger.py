#!/usr/bin/env python3
"""Exact-fp32 forward (gate-GEMM mode 0): what its launches cost ALONE on the chip, in the shapes the B = 64 forward uses (four pipelines of
16 utterances: intra-chunk recurrence on 32-sequence tiles, inter-chunk on 64-sequence tiles, projection 256 -> 128 + GroupNorm sums on
the fp32 MFMA), against the forward itself.  Tells whether the projection's share of the kernel time (22 % in the stats of the overlapped
forward) is its own inefficiency or contention.  usage: python tools/probes/exact_mode_ledger.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tss_with_dprnn_amd as pkg  # noqa: E402
from tss_with_dprnn_amd import engine  # noqa: E402
from tss_with_dprnn_amd.synthetic import full_cfg, make_mixture, make_state  # noqa: E402
from tools.bench_serial import timed  # noqa: E402

dev = torch.device("cuda:0")
st = engine.Stage(dev)
lib = st.lib
p = lambda t: t.data_ptr()
g = torch.Generator(device="cuda").manual_seed(1)
for B in (16, 32):
    S, K = 194, 250
    rows = B * S * K
    X = torch.randn(rows, 128, device="cuda", generator=g) * 0.5
    Z = torch.randn(rows, 128, device="cuda", generator=g)
    X2 = torch.empty_like(X)
    H = torch.empty(rows, 256, device="cuda")
    stats = torch.zeros(B, 2, device="cuda", dtype=torch.float64)
    stats[:, 1] = S * K * 128
    stats2 = torch.zeros(B, 2, device="cuda", dtype=torch.float64)
    gamma, beta = torch.ones(128, device="cuda"), torch.zeros(128, device="cuda")
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device="cuda")
    bsum = torch.empty(2 * 512, device="cuda")
    w = [torch.randn(512, 128, device="cuda", generator=g) * 0.08, torch.randn(512, 128, device="cuda", generator=g) * 0.08,
         torch.randn(512, device="cuda", generator=g) * 0.05, torch.randn(512, device="cuda", generator=g) * 0.05]
    st.call("dprnn_rnn_pack_ex", 0, 0, *[p(t) for t in w], *[p(t) for t in w], p(packed), p(bsum))
    Wp = torch.randn(128, 256, device="cuda", generator=g) / 16
    bp = torch.randn(128, device="cuda", generator=g)
    intra, inter = (B * S, K, B * S, 0, K, 1), (B * K, S, K, S * K, 1, K)
    lib.dprnn_set_gemm_mode(0)
    out = {}
    for name, addr, per, tiles in (("intra", intra, S, (16, 32, 64)), ("inter", inter, K, (32, 64))):
        for tile in tiles:
            f = lambda: st.call("dprnn_rnn_fwd_ex", 0, 0, p(X), p(Z), p(stats), p(gamma), p(beta), 1e-8, per, S * K, p(X2), p(packed), p(bsum), p(H), *addr, 2, tile)
            out[f"recurrence {name}, {tile}-sequence tiles"] = timed(f, 5)
    out["projection + sums (fp32 MFMA)"] = timed(lambda: st.call("dprnn_proj_stats_fwd", p(H), p(Wp), p(bp), p(Z), p(stats2), rows, S * K, 256), 10)
    fl_rec, fl_proj = rows * 2 * 2 * 256 * 512, rows * 2 * 256 * 128
    print(f"== one pipeline of {B} utterances alone on the chip (fp32 MFMA peak 157.3 TFLOP/s)")
    for k, ms in out.items():
        fl = fl_proj if k.startswith("proj") else fl_rec
        print(f"{k:44s} {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s = {fl / ms / 1e9 / 157.3:.2f} of peak", flush=True)

# the forward itself
cfg = full_cfg()
ctor = ["input_size", "feature_size", "hidden_size", "chunk_length", "kernel_size", "hop_length", "n_repeats",
        "bidirectional", "rnn_type", "norm_type", "activation_type", "dropout", "stride"]
model = pkg.DPRNNTasNet(**{k: cfg[k] for k in ctor}).eval()
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state("bss", cfg, 0).items()})
model = model.to(dev)
x = torch.from_numpy(make_mixture(64, 24000, seed=1)).to(dev)
engine.GEMM_MODE = 0
with torch.no_grad():
    ms = timed(lambda: model(x), 5)
print(f"== exact-fp32 forward, B = 64: {ms:.1f} ms = {64e3 / ms:.1f} utt/s; 22.45 TFLOP algorithmic = {22.45e3 / ms / 157.3:.2f} of the fp32 MFMA peak")
