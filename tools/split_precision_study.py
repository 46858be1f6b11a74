#!/usr/bin/env python3
"""CPU numerics study for a possible bf16-split gate GEMM (NOT used by the product path).

bf16 MFMAs run at 16x the fp32-MFMA rate on gfx950 and, unlike fp32 MFMAs, hide VALU work.  An fp32 value splits
exactly-ish into bf16 pieces a = a1 + a2 + a3; keeping the 6 products of order <= 2^-16 (a1b1, a1b2, a2b1, a1b3,
a2b2, a3b1) reproduces the fp32 product to ~2^-24, keeping 3 (a1b1, a1b2, a2b1) to ~2^-16.  This script emulates
both inside the LSTM contractions of the numpy oracle (fp32 accumulation, as the MFMA does) and reports the
end-to-end deviation from the plain fp32 oracle, next to the fp32-vs-fp64 noise floor."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dprnn_numpy as onp
from oracle.sisdr import si_sdr
from tss_with_dprnn_amd.synthetic import full_cfg, make_mixture, make_sources, make_state


def bf16(x):
    u = x.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF            # round to nearest even
    return ((u + r) & 0xFFFF0000).view(np.float32)


def split3(x):
    a1 = bf16(x); r = x - a1
    a2 = bf16(r); r = r - a2
    return a1, a2, bf16(r)


def mm_split(x, w, terms):
    """x [.., K] @ w[N, K]^T with bf16-split operands, products exact in fp32, fp32 accumulate."""
    xs, ws = split3(x.astype(np.float32)), split3(w.astype(np.float32))
    pairs = [(0, 0), (0, 1), (1, 0)] if terms == 3 else [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]
    out = np.zeros(x.shape[:-1] + (w.shape[0],), dtype=np.float32)
    for i, j in reversed(pairs):            # small terms first
        out += xs[i] @ ws[j].T
    return out


def run(terms, sd, cfg, mix):
    orig = onp._rnn_dir

    def rnn_dir(sd_, prefix, suf, x, rnn_type, reverse, dtype):
        wih = sd_[f"{prefix}.rnn.weight_ih_l0{suf}"]; whh = sd_[f"{prefix}.rnn.weight_hh_l0{suf}"]
        b = sd_[f"{prefix}.rnn.bias_ih_l0{suf}"] + sd_[f"{prefix}.rnn.bias_hh_l0{suf}"]
        Bn, T, _ = x.shape; H = whh.shape[1]
        h = np.zeros((Bn, H), np.float32); c = np.zeros((Bn, H), np.float32); out = np.zeros((Bn, T, H), np.float32)
        gi = mm_split(x, wih, terms) + b
        for t in (range(T - 1, -1, -1) if reverse else range(T)):
            g = gi[:, t] + mm_split(h, whh, terms)
            i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
            c = onp._sig(f) * c + onp._sig(i) * np.tanh(gg); h = onp._sig(o) * np.tanh(c); out[:, t] = h
        return out
    onp._rnn_dir = rnn_dir
    try:
        return onp.tasnet_forward(sd, cfg, mix)
    finally:
        onp._rnn_dir = orig


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64)))


cfg = full_cfg(n_repeats=6, chunk_length=50, hop_length=25)      # full width and depth, 0.5 s utterances (CPU budget)
sd = make_state("bss", cfg, 0)
tgt, mix = make_sources(2, 4000, seed=5)
ref64 = onp.tasnet_forward(sd, cfg, mix, np.float64)
ref32 = onp.tasnet_forward(sd, cfg, mix, np.float32)
print(f"fp32 oracle vs fp64 oracle         rel-L2 {rel(ref32, ref64):.2e}")
for terms in (6, 3):
    out = run(terms, sd, cfg, mix)
    d = np.abs(si_sdr(out, tgt) - si_sdr(ref32, tgt)).max()
    print(f"bf16 split, {terms} products vs fp64     rel-L2 {rel(out, ref64):.2e}   vs fp32 {rel(out, ref32):.2e}   |dSI-SDR| {d:.2e} dB")
