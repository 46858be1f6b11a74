#!/usr/bin/env python3
"""CPU numerics study of the split gate GEMMs (an emulation; NOT used by the product path).

bf16 MFMAs run at 16x the fp32-MFMA rate on gfx950 and, unlike fp32 MFMAs, hide VALU work.  An fp32 value splits
exactly-ish into bf16 pieces a = a1 + a2 + a3; keeping the 6 products of order <= 2^-16 (a1b1, a1b2, a2b1, a1b3,
a2b2, a3b1) reproduces the fp32 product to ~2^-24, keeping 3 (a1b1, a1b2, a2b1) to ~2^-16.  This script emulates
both inside the LSTM contractions of the numpy oracle (fp32 accumulation, as the MFMA does) and reports the
end-to-end deviation from the plain fp32 oracle, next to the fp32-vs-fp64 noise floor.

f16 pieces (gemm mode 3): f16 has 11 significant bits, so TWO pieces a = a1 + a2 carry 22+ and the three products a1b1, a1b2,
a2b1 leave a2b2 ~ 2^-24 |a||b| behind -- but only 5 exponent bits, so the operands are scaled by powers of two first
(activations x 2^4, weights so that max|W| lands in [2^13, 2^14)) and second pieces may be f16 subnormals.  The second part of
this script shows what each ingredient is worth on a single [4096 x 256] x [256 x 512] contraction: no scaling, the kernel's
scaling, and the same with subnormals flushed (what the hardware would do if the MFMA did not keep them: it does,
tools/probes/mfma_f16_denorm.hip)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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


def f16(x, ftz=False):
    y = x.astype(np.float16)
    if ftz:
        y = np.where(np.abs(y.astype(np.float32)) < 2.0 ** -14, np.float16(0), y)
    return y.astype(np.float32)


def split2_f16(x, ftz=False):
    a1 = f16(x, ftz)
    return a1, f16(x - a1, ftz)


def w_scale(w):
    """the kernel's per-LSTM weight scale: the power of two that puts max|W| in [2^13, 2^14) (lstm_absmax_kernel)"""
    m = float(np.abs(w).max())
    return 2.0 ** (14 - np.frexp(m)[1]) if m > 0 else 1.0


def mm_f16(x, w, sx=16.0, sw=None, ftz=False):
    """x [.., K] @ w[N, K]^T with two f16 pieces of sx * x (clamped to +-65000) and sw * w, three products, fp32 accumulate."""
    sw = w_scale(w) if sw is None else sw
    xs = split2_f16(np.clip(x.astype(np.float32) * np.float32(sx), -65000, 65000), ftz)
    ws = split2_f16(w.astype(np.float32) * np.float32(sw), ftz)
    out = xs[1] @ ws[0].T
    out += xs[0] @ ws[1].T
    out += xs[0] @ ws[0].T
    return out * np.float32(1.0 / (sx * sw))


def fp8(x):
    """round to fp8 e4m3 (4 significant bits, normals from 2^-6, subnormal step 2^-9, largest 448) -- the operand type of
    v_mfma_f32_16x16x128_f8f6f4, which runs at twice the f16 rate"""
    x = np.clip(x.astype(np.float32), -448, 448)
    e = np.floor(np.log2(np.maximum(np.abs(x), 2.0 ** -6)))
    q = (2.0 ** (e - 3)).astype(np.float32)
    return (np.round(x / q) * q).astype(np.float32)


def p2_scale(a, top):
    m = float(np.abs(a).max())
    return 2.0 ** (top - np.frexp(m)[1]) if m > 0 else 1.0


def mm_f16_fp8(x, w, sx=16.0, sw=None):
    """a possible next step, NOT built: the leading product x1 w1 on the f16 MFMA as in mm_f16, the two cross terms x1 w2 + x2 w1 with all
    four operands rounded to fp8 e4m3 (each under its own power-of-two scale, largest value in [2^7, 2^8)) -- 32 instead of 48 matrix
    cycles per 16 x 16 x 32 tile.  The cross terms are 2^-11 of the product and keep 4 bits: error ~2^-15 of the product."""
    sw = w_scale(w) if sw is None else sw
    x1, x2 = split2_f16(np.clip(x.astype(np.float32) * np.float32(sx), -65000, 65000))
    w1, w2 = split2_f16(w.astype(np.float32) * np.float32(sw))
    out = x1 @ w1.T
    for a_, b_ in ((x1, w2), (x2, w1)):
        sa, sb = p2_scale(a_, 8), p2_scale(b_, 8)
        out += (fp8(a_ * np.float32(sa)) @ fp8(b_ * np.float32(sb)).T) * np.float32(1.0 / (sa * sb))
    return out * np.float32(1.0 / (sx * sw))


def mm_split(x, w, terms):
    """x [.., K] @ w[N, K]^T with bf16-split operands, products exact in fp32, fp32 accumulate (terms = "f16": f16 pieces)."""
    if terms == "f16":
        return mm_f16(x, w)
    if terms == "f16+fp8":
        return mm_f16_fp8(x, w)
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
        if terms in ("f16", "f16+fp8"):     # one scale per LSTM (both matrices), as the kernel's pack does
            sw = min(w_scale(wih), w_scale(whh))
            f_ = mm_f16 if terms == "f16" else mm_f16_fp8
            mm = lambda a_, w_, _t: f_(a_, w_, 16.0, sw)
        else:
            mm = mm_split
        gi = mm(x, wih, terms) + b
        for t in (range(T - 1, -1, -1) if reverse else range(T)):
            g = gi[:, t] + mm(h, whh, terms)
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
for terms in (6, 3, "f16", "f16+fp8"):
    out = run(terms, sd, cfg, mix)
    d = np.abs(si_sdr(out, tgt) - si_sdr(ref32, tgt)).max()
    label = {"f16": "f16 pieces, 3 products", "f16+fp8": "f16 x1 + fp8 cross (not built)"}.get(terms, f"bf16 split, {terms} products")
    print(f"{label:30s} vs fp64  rel-L2 {rel(out, ref64):.2e}   vs fp32 {rel(out, ref32):.2e}   |dSI-SDR| {d:.2e} dB")

print()
print("one contraction [4096 x 256] x [256 x 512], error against float64 (rel-L2 / max abs):")
rng = np.random.default_rng(0)
W = rng.uniform(-1 / np.sqrt(128), 1 / np.sqrt(128), (512, 256)).astype(np.float32)
M = 4096
cases = [("x ~ N(0,1), h ~ tanh", np.concatenate([rng.standard_normal((M, 128)), 0.5 * np.tanh(rng.standard_normal((M, 128)))], 1)),
         ("small activations (1e-3)", np.concatenate([1e-3 * rng.standard_normal((M, 128)), 1e-2 * np.tanh(rng.standard_normal((M, 128)))], 1)),
         ("heavy tails (|x| up to 60)", np.concatenate([rng.standard_t(2, (M, 128)).clip(-60, 60), np.tanh(rng.standard_normal((M, 128)))], 1))]
for name, X in cases:
    X = X.astype(np.float32)
    ref = X.astype(np.float64) @ W.astype(np.float64).T
    e = lambda g: f"{np.linalg.norm(g - ref) / np.linalg.norm(ref):.2e} / {np.abs(g - ref).max():.2e}"
    print(f"  {name}")
    print(f"    fp32 matmul                                   {e(X @ W.T)}")
    print(f"    bf16 x6                                       {e(mm_split(X, W, 6))}")
    print(f"    bf16 x3                                       {e(mm_split(X, W, 3))}")
    print(f"    f16 x3, no scaling                            {e(mm_f16(X, W, 1.0, 1.0))}")
    print(f"    f16 x3, kernel scaling (x 2^4, W 2^e)         {e(mm_f16(X, W))}")
    print(f"    f16 x3, kernel scaling, subnormals FLUSHED    {e(mm_f16(X, W, ftz=True))}")
    print(f"    f16 x1 + fp8 cross terms (not built)          {e(mm_f16_fp8(X, W))}")
