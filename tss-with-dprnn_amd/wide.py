"""Configurations WIDER than the kernels' build (round 5; SURVEY.md section 8 f4, "API completeness of the kept surface").

The fused kernels behind ``engine.tasnet_forward`` are built for input_size 64, feature_size = hidden_size 128 (every shipped YAML);
narrower models run on them as zero-padded twins (``padded.py``).  The reference's constructors take any width (dprnn.py:117-120,
234-239): a DPRNNTasNet with feature_size / hidden_size above 128 or input_size above 64 runs here, on the any-width kernels of
``csrc/wide.hip`` (C ABI: ``include/dprnn_mi355x_wide.h``) -- the same channels-last rows with the channel count a run-time
argument, every width (multiples of 16 or not), LSTM / GRU / RNN cells, one or two directions; fp32 operands on the fp32 MFMA, and for LSTM
recurrences and the Linear after them the f16-pieces arithmetic of the built kernels with its ranges by construction
(``csrc/wide_f16.hip``: scales from the measured maxima of a launch's operands, so there is no saturation flag and no repeat).
It is the unfused form of the path: the norms are materialised between the launches and the Linear is a launch of its own, so it
costs about the 10 slab units per path the built kernels started from.

No fallback: a missing library raises (``_lib.load_library``); what the kernels cannot hold (hidden_size above 512, or feature_size
+ hidden_size beyond the LDS of a CU) raises NotImplementedError naming the limit.  Inference only (under autograd the mirror
class raises).  DPRNNSpeTasNet / DPRNNSpeIRATasNet with feature_size / hidden_size above the build run here too when input_size and the
speaker encoder keep the built widths (64; O 128, P 256, embeddings 128 -- every shipped YAML): speaker branch and fusion operands on the
built kernels, the DPRNN stack on the any-width ones.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from ._lib import DprnnError, load_library
from . import engine as E

CELL_ID = {"LSTM": 0, "GRU": 1, "RNN": 2}
# LSTM recurrences in the f16-pieces arithmetic (csrc/wide_f16.hip: three f16 MFMAs per product, 64 sequences per workgroup, scales from the
# measured maxima of the launch's operands) where its LDS image and register budget hold the model; DPRNN_WIDE_F16=0: fp32 MFMA everywhere.
WIDE_F16 = os.environ.get("DPRNN_WIDE_F16", "1") != "0"
H_MAX = 512                 # csrc/wide.hip: MAXUT = 8 unit tiles per wave, 4 waves
LDS_BYTES = 160 * 1024


def _r16(n: int) -> int:
    return (n + 15) // 16 * 16


def _r32(n: int) -> int:
    return (n + 31) // 32 * 32


def f16_tile(H: int) -> int:
    """Sequences per workgroup of the f16-pieces recurrence: 64 with two unit tiles per wave (hidden_size <= 256), 32 with four (<= 512)."""
    return 64 if _r16(H) <= 256 else 32


def f16_recurrence(cell: int, F: int, H: int) -> bool:
    """Whether this recurrence runs in the f16-pieces arithmetic: LSTM, hidden_size <= 512, one image of piece pairs of f16_tile(H) sequences
    within the 160 KB of a CU."""
    return WIDE_F16 and cell == 0 and _r16(H) <= 512 and f16_tile(H) * (2 * (_r32(F) + _r32(H)) + 16) * 2 + 512 <= LDS_BYTES


def check_supported(cfg: dict) -> None:
    bad = []
    F, H = cfg["feature_size"], cfg["hidden_size"]
    if cfg["rnn_type"] not in CELL_ID:
        bad.append(f"rnn_type {cfg['rnn_type']!r}: torch.nn has LSTM, GRU and RNN recurrences")
    if H > H_MAX:
        bad.append(f"hidden_size {H} > {H_MAX} (four waves x eight 16-unit tiles hold a recurrence's state in registers)")
    elif 2 * 16 * (_r16(F) + _r16(H) + 8) * 4 > LDS_BYTES:
        bad.append(f"feature_size {F} + hidden_size {H}: two LDS images of 16 sequences x (x_t | h_t-1) exceed the 160 KB of a CU")
    if cfg["norm_type"] not in ("ln", "gLN"):
        bad.append(f"norm_type {cfg['norm_type']!r} (reference leaves the norms undefined, dprnn.py:72-77)")
    if cfg["activation_type"] not in ("sigmoid", "relu"):
        bad.append(f"activation_type {cfg['activation_type']!r} (dprnn.py:161-164)")
    if cfg["hop_length"] < 1 or cfg["hop_length"] > cfg["chunk_length"]:
        bad.append("hop_length must be in [1, chunk_length]")
    if bad:
        raise NotImplementedError("DPRNN MI355X path (any-width kernels): unsupported configuration: " + "; ".join(bad))


class _Launcher(E.Stage):
    """engine.Stage plus the two calls every stage of this path is made of."""

    def gemm(self, A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], M: int, K: int, O: int, *, sc=None, sh=None,
             rows_per_utt: int = 1, prelu=None, act: int = 0, mul=None, mul_L: int = 1, mul_rep: int = 1, stats=None,
             out: Optional[torch.Tensor] = None, ldw: Optional[int] = None, ubias=None, rscale=None) -> torch.Tensor:
        if out is None:
            out = torch.empty(M, O, device=A.device, dtype=torch.float32)
        ldm = 0 if mul is None else mul.shape[-1]
        self.call("dprnn_wide_gemm", E._p(A), K, E._p(sc), E._p(sh), rows_per_utt, E._p(prelu), E._p(W), K if ldw is None else ldw, E._p(bias),
                  E._p(ubias), E._p(rscale), act, E._p(mul), ldm, mul_L, mul_rep, E._p(out), O, E._p(stats), M, K, O)
        return out

    def norm_affine(self, stats_slot: torch.Tensor, norm, B: int, rows_per_utt: int, C: int):
        gamma, beta, eps = E._norm_params(norm)
        dev = stats_slot.device
        sc = torch.empty(B, C, device=dev, dtype=torch.float32)
        sh = torch.empty_like(sc)
        self.call("dprnn_wide_norm_prep", E._p(stats_slot), E._p(E._param(gamma, "norm.weight", dev)), E._p(E._param(beta, "norm.bias", dev)),
                  eps, E._p(sc), E._p(sh), B, rows_per_utt, C)
        return sc, sh


def _pack_rnn(st: _Launcher, rnn: torch.nn.Module, cell: int, F: int, H: int, dev):
    """One torch RNN module's weights (both directions) in the fragment order of dprnn_wide_rnn_fwd."""
    ndir = 2 if rnn.bidirectional else 1
    per = int(st.lib.dprnn_wide_rnn_pack_floats(cell, F, H))
    gates = 1 if cell == 2 else 4
    Wp = torch.empty(ndir, per, device=dev, dtype=torch.float32)
    bv = torch.empty(ndir, gates * _r16(H), device=dev, dtype=torch.float32)
    for d in range(ndir):
        sfx = "_reverse" if d else ""
        w = [E._param(getattr(rnn, f"{n}_l0{sfx}"), f"rnn.{n}_l0{sfx}", dev) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        st.call("dprnn_wide_rnn_pack", cell, E._p(w[0]), E._p(w[1]), E._p(w[2]), E._p(w[3]), F, H, E._p(Wp[d]), E._p(bv[d]))
    return Wp, bv, ndir


def _recurrence(st: _Launcher, rnn, cell: int, X: torch.Tensor, rows: int, Hbuf: torch.Tensor, F: int, H: int, addr, tiles=None,
                amax_x: Optional[torch.Tensor] = None):
    """SingleRNN.forward over the slab rows X [rows][F] -> (Hout [rows][ndir*H] inside Hbuf, ndir).  addr = (nseq, T, inner_n, outer, inner,
    t_stride) of dprnn_wide_rnn_fwd; tiles (ragged inter-chunk path) = {tile width: (desc, n)} tile descriptors instead."""
    dev = X.device
    ndir = 2 if rnn.bidirectional else 1
    Hout = Hbuf[: rows * ndir * H].view(rows, ndir * H)
    if f16_recurrence(cell, F, H):
        amax = torch.zeros(1 + 2 * ndir, device=dev, dtype=torch.int32)
        if amax_x is not None:      # max |X| left by the launch that wrote X (dprnn_wide_norm_residual_fwd)
            amax[0:1].copy_(amax_x)
        else:
            st.call("dprnn_wide_absmax", E._p(X), rows * F, E._p(amax))
        words = int(st.lib.dprnn_wide_lstm_pack_f16_words(F, H))
        Wp = torch.empty(ndir, words * 4, device=dev, dtype=torch.int32)
        bv = torch.empty(ndir, 4 * _r16(H), device=dev, dtype=torch.float32)
        scal = torch.empty(ndir, 4, device=dev, dtype=torch.float32)
        for d in range(ndir):
            sfx = "_reverse" if d else ""
            w = [E._param(getattr(rnn, f"{n}_l0{sfx}"), f"rnn.{n}_l0{sfx}", dev) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
            st.call("dprnn_wide_absmax", E._p(w[0]), w[0].numel(), E._p(amax[1 + 2 * d:]))
            st.call("dprnn_wide_absmax", E._p(w[1]), w[1].numel(), E._p(amax[2 + 2 * d:]))
            st.call("dprnn_wide_lstm_pack_f16", E._p(w[0]), E._p(w[1]), E._p(w[2]), E._p(w[3]), F, H, d, E._p(amax), E._p(scal), E._p(Wp[d]), E._p(bv[d]))
        if tiles is None:
            st.call("dprnn_wide_lstm_f16_fwd", E._p(X), E._p(Wp), E._p(bv), E._p(scal), E._p(Hout), ndir, F, H, *addr, 0, 0)
        else:
            desc, n = tiles[f16_tile(H)]
            st.call("dprnn_wide_lstm_f16_fwd", E._p(X), E._p(Wp), E._p(bv), E._p(scal), E._p(Hout), ndir, F, H, n, 1, 1, 0, 0, addr[5], E._p(desc), n)
        return Hout, ndir
    Wp, bv, ndir = _pack_rnn(st, rnn, cell, F, H, dev)
    if tiles is None:
        st.call("dprnn_wide_rnn_fwd", cell, E._p(X), E._p(Wp), E._p(bv), E._p(Hout), ndir, F, H, *addr)
    else:
        desc, n = tiles[16]
        st.call("dprnn_wide_rnn_fwd_tiles", cell, E._p(X), E._p(Wp), E._p(bv), E._p(Hout), ndir, F, H, E._p(desc), n, addr[5])
    return Hout, ndir


def _pack_linear_f16(st: _Launcher, lin, K: int, O: int, dev):
    """The Linear after an LSTM recurrence as f16-piece fragments (its rows are h: |a| < 1)."""
    w = E._param(lin.weight, "linear.weight", dev)
    if tuple(w.shape) != (O, K):
        raise DprnnError(f"Linear after the recurrence: weight {tuple(w.shape)}, expected {(O, K)}")
    amax = torch.zeros(1, device=dev, dtype=torch.int32)
    st.call("dprnn_wide_absmax", E._p(w), w.numel(), E._p(amax))
    Wp = torch.empty(int(st.lib.dprnn_wide_linear_pack_f16_words(K, O)) * 4, device=dev, dtype=torch.int32)
    scal = torch.empty(2, device=dev, dtype=torch.float32)
    st.call("dprnn_wide_linear_pack_f16", E._p(w), K, K, O, E._p(amax), 1.0, E._p(Wp), E._p(scal))
    return Wp, scal, E._param(lin.bias, "linear.bias", dev)


def _path(st: _Launcher, single, lin, norm, X, Hbuf, Z, stats_slot, geo, F: int, H: int, cell: int, intra: bool, amax_in=None):
    """One half of DPRNNBlock.forward (dprnn.py:83-92 intra, :94-99 inter): X += norm(Linear(RNN(X))) in place.  amax_in: max |X| as float
    bits when the writer of X left it; returns that word for the X this path leaves (the next path's activation scale)."""
    dev = X.device
    rows = geo.rows
    if intra:       # B*S sequences of K steps, consecutive rows
        addr = (geo.B * geo.S, geo.K, 1, geo.K, 0, 1)
    else:           # B*K sequences of S steps, K rows apart
        addr = (geo.B * geo.K, geo.S, geo.K, geo.S * geo.K, 1, geo.K)
    Hout, ndir = _recurrence(st, single.rnn, cell, X, rows, Hbuf, F, H, addr, amax_x=amax_in)
    if f16_recurrence(cell, F, H):
        Wp, scal, b = _pack_linear_f16(st, lin, ndir * H, F, dev)
        st.call("dprnn_wide_linear_f16", E._p(Hout), ndir * H, E._p(Wp), E._p(scal), E._p(b), E._p(Z), F, E._p(stats_slot), geo.S * geo.K, rows,
                ndir * H, F)
    else:
        w = E._param(lin.weight, "linear.weight", dev)
        if tuple(w.shape) != (F, ndir * H):
            raise DprnnError(f"Linear after the recurrence: weight {tuple(w.shape)}, expected {(F, ndir * H)}")
        st.gemm(Hout, w, E._param(lin.bias, "linear.bias", dev), rows, ndir * H, F, rows_per_utt=geo.S * geo.K, stats=stats_slot, out=Z)
    sc, sh = st.norm_affine(stats_slot, norm, geo.B, geo.S * geo.K, F)
    amax_out = torch.zeros(1, device=dev, dtype=torch.int32) if f16_recurrence(cell, F, H) else None
    st.call("dprnn_wide_norm_residual_fwd", E._p(X), E._p(Z), E._p(sc), E._p(sh), rows, F, geo.S * geo.K, E._p(amax_out))
    return amax_out


def _sep_cfg(sep) -> dict:
    rnn = sep.dprnn_blocks[0].intra_rnn
    return dict(input_size=sep.input_size, feature_size=sep.feature_size, hidden_size=rnn.hidden_size, rnn_type=rnn.rnn.mode.split("_")[0],
                norm_type=sep.norm_type, activation_type=sep.activation_type, hop_length=sep.hop_length, chunk_length=sep.chunk_length)


def _frames_from_rows(st: _Launcher, sep, enc: torch.Tensor, B: int, L: int, stats_slot: torch.Tensor) -> torch.Tensor:
    """sep.bottleneck (dprnn.py:131-136,169) on encoder rows [B*L][N] -> frames [B*L][F]: the global norm folded into the 1x1 conv's loader."""
    dev = enc.device
    N, F = sep.input_size, sep.feature_size
    bott = list(sep.bottleneck)
    conv = bott[-1]
    sc = sh = None
    if len(bott) == 2:
        st.call("dprnn_wide_flat_stats", E._p(enc), E._p(stats_slot), B, L * N)
        sc, sh = st.norm_affine(stats_slot, bott[0], B, L, N)
    w = E._param(conv.weight, "bottleneck.weight", dev)
    if tuple(w.shape[:2]) != (F, N):
        raise DprnnError(f"bottleneck 1x1 conv weight {tuple(w.shape)}, expected ({F}, {N}, 1)")
    return st.gemm(enc, w, E._param(conv.bias, "bottleneck.bias", dev), B * L, N, F, sc=sc, sh=sh, rows_per_utt=L)


def _masks_from_frames(st: _Launcher, sep, Y: torch.Tensor, geo, stats: torch.Tensor, mul_enc: Optional[torch.Tensor] = None,
                       debug: Optional[dict] = None) -> torch.Tensor:
    """DPRNN._dprnn_process (dprnn.py:171-186) on frames [B*L][F] -> mask rows [(b*2 + spk)*L + l][N], times the encoder rows
    [B*L][N] when given (dprnn.py:277)."""
    cfg = _sep_cfg(sep)
    check_supported(cfg)
    dev = Y.device
    N, F, H = cfg["input_size"], cfg["feature_size"], cfg["hidden_size"]
    cell = CELL_ID[cfg["rnn_type"]]
    B, L, K, P, S, rows = geo.B, geo.L, geo.K, geo.P, geo.S, geo.rows
    blocks = list(sep.dprnn_blocks)
    X = torch.empty(rows, F, device=dev, dtype=torch.float32)
    st.call("dprnn_wide_segment_fwd", E._p(Y), E._p(X), B, L, K, P, S, F)
    # the recurrence output and the mask head's conv2d output share one block ([rows][2 H] then [rows][2 F])
    Hbuf = torch.empty(rows * 2 * max(H, F), device=dev, dtype=torch.float32)
    Z = torch.empty(rows, F, device=dev, dtype=torch.float32)
    amax = None
    for i, blk in enumerate(blocks):
        amax = _path(st, blk.intra_rnn, blk.intra_linear, blk.intra_norm, X, Hbuf, Z, stats[1 + 2 * i], geo, F, H, cell, True, amax)
        amax = _path(st, blk.inter_rnn, blk.inter_linear, blk.inter_norm, X, Hbuf, Z, stats[2 + 2 * i], geo, F, H, cell, False, amax)
        if debug is not None:
            debug[f"block{i}"] = X.view(B, S, K, F).clone()
    del Z
    return _head(st, sep, cfg, X, Hbuf, geo, mul_enc)


def _head(st: _Launcher, sep, cfg: dict, X: torch.Tensor, Hbuf: torch.Tensor, geo, mul_enc: Optional[torch.Tensor]) -> torch.Tensor:
    """Mask head (dprnn.py:172-186) on the slab rows X [B*S*K][F]: PReLU + conv2d, overlap-add, tanh x sigmoid gate, end 1x1 conv +
    activation (x the encoder rows when given) -> rows [(b*2 + spk)*L + l][N].  Hbuf: scratch of at least rows x 2F floats."""
    dev = X.device
    N, F = cfg["input_size"], cfg["feature_size"]
    B, L, K, P, S, rows = geo.B, geo.L, geo.K, geo.P, geo.S, geo.rows
    w2 = E._param(sep.conv2d.weight, "conv2d.weight", dev)
    nspk = w2.shape[0] // F
    if nspk != 2 or tuple(w2.shape[:2]) != (2 * F, F):
        raise DprnnError(f"conv2d weight {tuple(w2.shape)}, expected ({2 * F}, {F}, 1, 1)")
    U = Hbuf[: rows * 2 * F].view(rows, 2 * F)
    st.gemm(X, w2, E._param(sep.conv2d.bias, "conv2d.bias", dev), rows, F, 2 * F, prelu=E._param(sep.prelu.weight, "prelu.weight", dev), out=U)
    frames = torch.empty(B * nspk * L, F, device=dev, dtype=torch.float32)
    st.call("dprnn_wide_ola_fwd", E._p(U), E._p(frames), B, L, K, P, S, F, nspk)
    M2 = B * nspk * L
    t_out = st.gemm(frames, E._param(sep.out[0].weight, "out.weight", dev), E._param(sep.out[0].bias, "out.bias", dev), M2, F, F, act=3)
    V = st.gemm(frames, E._param(sep.gate[0].weight, "gate.weight", dev), E._param(sep.gate[0].bias, "gate.bias", dev), M2, F, F, act=2, mul=t_out)
    del frames, t_out
    act = 1 if cfg["activation_type"] == "relu" else 2
    return st.gemm(V, E._param(sep.end_conv1x1.weight, "end_conv1x1.weight", dev), None, M2, F, N, act=act, mul=mul_enc, mul_L=L, mul_rep=nspk)


def _sep_geometry(sep, B: int, L: int):
    K, P = sep.chunk_length, sep.hop_length
    return E.Geometry(B, L, L, K, P, (L + K) // P + 1, 1, 1, sep.input_size)


def separator_masks(sep, frames: torch.Tensor) -> torch.Tensor:
    """DPRNN._dprnn_process (dprnn.py:171-186) at any width: bottleneck output [B, F, L] -> masks [B, 2, N, L]."""
    E._check_frames(frames, sep.feature_size, "bottleneck output")
    E._no_grad_only("_dprnn_process", frames, module=sep)
    dev = frames.device
    B, F, L = frames.shape
    with torch.cuda.device(dev):
        st = _Launcher(dev)
        Y = frames.transpose(1, 2).contiguous().view(B * L, F)          # channels-last rows (layout plumbing)
        stats = torch.zeros(1 + 2 * len(sep.dprnn_blocks), B, 2, device=dev, dtype=torch.float64)
        rows = _masks_from_frames(st, sep, Y, _sep_geometry(sep, B, L), stats)
        return rows.view(B, 2, L, sep.input_size).permute(0, 1, 3, 2).contiguous()


def separator_forward(sep, enc: torch.Tensor) -> torch.Tensor:
    """DPRNN.forward (dprnn.py:166-187) at any width: encoder output [B, N, L] -> masks [B, 2, N, L]."""
    E._check_frames(enc, sep.input_size, "separator input")
    E._no_grad_only("DPRNN.forward", enc, module=sep)
    dev = enc.device
    B, N, L = enc.shape
    with torch.cuda.device(dev):
        st = _Launcher(dev)
        rows_in = enc.transpose(1, 2).contiguous().view(B * L, N)
        stats = torch.zeros(1 + 2 * len(sep.dprnn_blocks), B, 2, device=dev, dtype=torch.float64)
        Y = _frames_from_rows(st, sep, rows_in, B, L, stats[0])
        rows = _masks_from_frames(st, sep, Y, _sep_geometry(sep, B, L), stats)
        return rows.view(B, 2, L, N).permute(0, 1, 3, 2).contiguous()


def _forward_impl(model, wav: torch.Tensor, debug: Optional[dict] = None) -> torch.Tensor:
    """DPRNNTasNet.forward (dprnn.py:271-283): [B, T] -> [B, 2, T_out], every stage a launch of csrc/wide.hip."""
    cfg = model.cfg
    check_supported(cfg)
    dev = wav.device
    N, F = cfg["input_size"], cfg["feature_size"]
    with torch.cuda.device(dev):
        st = _Launcher(dev)
        geo = E.make_geometry(wav.shape[0], wav.shape[1], N, cfg["chunk_length"], cfg["hop_length"], cfg["kernel_size"], cfg["stride"])
        if geo.T_out < 2:
            raise DprnnError("utterance too short")
        B, L = geo.B, geo.L
        sep = model.separation
        enc_w = E._param(model.encoder.conv1d.weight, "encoder.conv1d.weight", dev)
        dec_w = E._param(model.decoder.weight, "decoder.weight", dev)
        stats = torch.zeros(1 + 2 * len(sep.dprnn_blocks), B, 2, device=dev, dtype=torch.float64)
        enc = torch.empty(B * L, N, device=dev, dtype=torch.float32)
        st.call("dprnn_wide_encoder_fwd", E._p(wav), E._p(enc_w), E._p(enc), B, geo.T, L, N, geo.ks, geo.stride)
        Y = _frames_from_rows(st, sep, enc, B, L, stats[0])
        if debug is not None:
            debug["bottleneck"] = Y.view(B, L, F).clone()
        masked = _masks_from_frames(st, sep, Y, geo, stats, mul_enc=enc, debug=debug)          # masks x encoder output (dprnn.py:277)
        del Y
        if debug is not None:
            debug["masked"] = masked.view(B, 2, L, N).clone()
        # decoder (encoder_decoder.py:36-49): per-frame taps, then the transposed convolution's overlap
        wd = dec_w.view(N, geo.ks).t().contiguous()                      # [ks][N]
        taps = st.gemm(masked, wd, None, B * 2 * L, N, geo.ks)
        return E.decode(st, taps, geo, 2)


def single_rnn_forward(rnn, x: torch.Tensor) -> torch.Tensor:
    """SingleRNN.forward (dprnn.py:33-37) at any width: [n, T, F] batch-first -> [n, T, ndir*H], zero initial state."""
    if not isinstance(x, torch.Tensor) or x.dim() != 3 or x.shape[2] != rnn.rnn.input_size:
        raise DprnnError(f"SingleRNN input must be [sequences, steps, {rnn.rnn.input_size}]")
    if not x.is_cuda or x.dtype != torch.float32:
        raise DprnnError("SingleRNN runs on float32 HIP tensors only (no CPU fallback)")
    E._no_grad_only("SingleRNN.forward", x, module=rnn)
    dev = x.device
    n, T, F = x.shape
    H = rnn.hidden_size
    kind = rnn.rnn.mode.split("_")[0]
    check_supported(dict(feature_size=F, hidden_size=H, rnn_type=kind, norm_type="ln", activation_type="relu", hop_length=1, chunk_length=1))
    with torch.cuda.device(dev):
        st = _Launcher(dev)
        xc = x.contiguous()
        Wp, bv, ndir = _pack_rnn(st, rnn.rnn, CELL_ID[kind], F, H, dev)
        out = torch.empty(n, T, ndir * H, device=dev, dtype=torch.float32)
        st.call("dprnn_wide_rnn_fwd", CELL_ID[kind], E._p(xc), E._p(Wp), E._p(bv), E._p(out), ndir, F, H, n, T, 1, T, 0, 1)
        return out


def block_forward(blk, x: torch.Tensor) -> torch.Tensor:
    """DPRNNBlock.forward (dprnn.py:79-99) at any width: [B, F, K, S] -> [B, F, K, S]."""
    F = blk.intra_rnn.rnn.input_size
    if not isinstance(x, torch.Tensor) or x.dim() != 4 or x.shape[1] != F:
        raise DprnnError(f"DPRNNBlock input must be [batch, {F}, chunk_length, chunks]")
    if not x.is_cuda or x.dtype != torch.float32:
        raise DprnnError("DPRNNBlock runs on float32 HIP tensors only (no CPU fallback)")
    if not hasattr(blk, "intra_norm"):
        raise DprnnError("DPRNNBlock without norms (norm_type neither 'ln' nor 'gLN'): the reference's forward fails too")
    E._no_grad_only("DPRNNBlock.forward", x, module=blk)
    dev = x.device
    B, _, K, S = x.shape
    H = blk.intra_rnn.hidden_size
    kind = blk.intra_rnn.rnn.mode.split("_")[0]
    check_supported(dict(feature_size=F, hidden_size=H, rnn_type=kind, norm_type="ln", activation_type="relu", hop_length=1, chunk_length=1))
    with torch.cuda.device(dev):
        st = _Launcher(dev)
        geo = E.Geometry(B, 0, 0, K, 1, S, 1, 1, 0)
        rows = geo.rows
        X = x.permute(0, 3, 2, 1).contiguous().view(rows, F)                     # slab rows (b, s, k)
        Hbuf = torch.empty(rows * 2 * H, device=dev, dtype=torch.float32)
        Z = torch.empty(rows, F, device=dev, dtype=torch.float32)
        stats = torch.zeros(2, B, 2, device=dev, dtype=torch.float64)
        cell = CELL_ID[kind]
        amax = _path(st, blk.intra_rnn, blk.intra_linear, blk.intra_norm, X, Hbuf, Z, stats[0], geo, F, H, cell, True)
        _path(st, blk.inter_rnn, blk.inter_linear, blk.inter_norm, X, Hbuf, Z, stats[1], geo, F, H, cell, False, amax)
        return X.view(B, S, K, F).permute(0, 3, 2, 1).contiguous()


def encoder_forward(enc, wav: torch.Tensor) -> torch.Tensor:
    """Encoder.forward (encoder_decoder.py:25-33) with any filter count / length: [B, T] -> relu(conv1d) [B, N, L]."""
    wav = E.check_input(wav)
    dev = wav.device
    w = E._param(enc.conv1d.weight, "encoder.conv1d.weight", dev)
    N, _, ks = w.shape
    stride = enc.conv1d.stride[0]
    E._no_grad_only("Encoder.forward", wav, module=enc)
    with torch.cuda.device(dev):
        st = _Launcher(dev)
        geo = E.make_geometry(wav.shape[0], wav.shape[1], N, 1, 1, ks, stride)
        rows = torch.empty(geo.B * geo.L, N, device=dev, dtype=torch.float32)
        st.call("dprnn_wide_encoder_fwd", E._p(wav), E._p(w), E._p(rows), geo.B, geo.T, geo.L, N, ks, stride)
        return rows.view(geo.B, geo.L, N).transpose(1, 2).contiguous()


def check_supported_spe(cfg: dict) -> None:
    """The speaker-conditioned classes above the built widths: the DPRNN stack runs at any feature / hidden size; the speaker branch and the
    fusions live on the encoder side of the bottleneck (N channels) and keep the built kernels, so input_size and the speaker encoder's
    widths must be the build's (the shipped YAMLs' values)."""
    check_supported(cfg)
    E.check_supported_spe(dict(cfg, input_size=E.N_BUILT, feature_size=E.F_BUILT, hidden_size=E.F_BUILT, kernel_size=min(cfg["kernel_size"], E.KS_MAX)))
    if cfg["kernel_size"] > E.KS_MAX:
        raise NotImplementedError(f"DPRNN-Spe MI355X path (any-width kernels): the speaker branch's built kernels take kernel_size <= {E.KS_MAX}")
    if cfg["input_size"] != E.N_BUILT:
        raise NotImplementedError(
            f"DPRNN-Spe MI355X path (any-width kernels): a speaker-conditioned model wider than the build needs input_size {E.N_BUILT} (the speaker "
            f"branch and the fusions run on the built kernels); got {cfg['input_size']}")


def _spe_forward_impl(model, wav: torch.Tensor, aux: torch.Tensor, aux_len, ira: bool, debug: Optional[dict] = None):
    """DPRNNSpeTasNet.forward (dprnn_spe.py:314-327) / DPRNNSpeIRATasNet.forward (dprnn_spe_ira.py:179-190) with feature_size / hidden_size
    above the build: speaker branch and fusion operands on the built kernels in their exact-fp32 arithmetic (engine.speaker_embedding,
    engine._fusion_operands, engine._att_row_scale), bottleneck 1x1 conv, DPRNN stack and mask head on the any-width kernels."""
    cfg = model.cfg
    check_supported_spe(cfg)
    if aux.shape[0] != wav.shape[0]:
        raise DprnnError("mixture and reference batch sizes differ")
    dev = wav.device
    N, F = cfg["input_size"], cfg["feature_size"]
    ks, stride = cfg["kernel_size"], cfg["stride"]
    with torch.cuda.device(dev), E.use_gemm_mode(0):
        st = _Launcher(dev)
        st.lib.dprnn_set_gemm_mode(0)
        geo = E.make_geometry(wav.shape[0], wav.shape[1], N, cfg["chunk_length"], cfg["hop_length"], ks, stride)
        geo_a = E.make_geometry(aux.shape[0], aux.shape[1], N, cfg["chunk_length"], cfg["hop_length"], ks, stride)
        if geo.T_out < 2:
            raise DprnnError("utterance too short")
        B, L = geo.B, geo.L
        sep = model.separation
        se = sep.spk_encoder
        enc_w = E._param(model.encoder.conv1d.weight, "encoder.conv1d.weight", dev)
        dec_w = E._param(model.decoder.weight, "decoder.weight", dev)
        nb = 1 + 2 * len(sep.dprnn_blocks)
        stats = torch.zeros((2 if ira else 1) * nb + 2, B, 2, device=dev, dtype=torch.float64)
        aux_len_dev = E._aux_len_tensor(aux_len, B, dev)
        # speaker embedding of the reference utterance (dprnn_spe.py:156-163)
        sc_a, sh_a = E.bottleneck_affine(st, aux, enc_w, se[0], geo_a, stats[-1])
        A0 = E.bottleneck_frames(st, aux, enc_w, sc_a, sh_a, se[1], geo_a, gemm_mode=0, scale_slot=E.SLOT_AUX_BOTTLENECK).view(B * geo_a.L, E.F_BUILT)
        v0 = E.speaker_embedding(st, sep, A0, B, geo_a.L, aux_len_dev, ks)
        if debug is not None:
            debug["emb"] = v0.clone()
        enc = torch.empty(B * L, N, device=dev, dtype=torch.float32)
        st.call("dprnn_wide_encoder_fwd", E._p(wav), E._p(enc_w), E._p(enc), B, geo.T, L, N, geo.ks, geo.stride)
        conv = sep.bottleneck[1]
        wb = E._param(conv.weight, "bottleneck.weight", dev)
        if wb.shape[0] != F or wb.shape[1] < N:
            raise DprnnError(f"bottleneck 1x1 conv weight {tuple(wb.shape)} does not fit {N} -> {F}")
        bb = E._param(conv.bias, "bottleneck.bias", dev)
        att = cfg["fusion_type"] == "att"

        def separate(v, sc, sh, stats_block, first: bool):
            """One separation pass conditioned on embedding v (dprnn_spe.py:165-229, 231-248) -> masked encoder rows [(b, spk, l)][N]."""
            mul, add, row_bias = E._fusion_operands(st, sep, cfg, v)
            if first:
                sc, sh = E.bottleneck_affine(st, wav, enc_w, sep.bottleneck[0], geo, stats[0], mul, add)
            elif mul is not None or add is not None:        # the cached norm output (dprnn_spe_ira.py:96): same sums, new fusion operands
                g, b_, eps = E._norm_params(sep.bottleneck[0])
                st.call("dprnn_bottleneck_prep", E._p(stats[0]), E._p(g), E._p(b_), eps, E._p(mul), E._p(add), E._p(sc), E._p(sh), B, L, N)
            row_scale = E._att_row_scale(st, sep, cfg, wav, enc_w, geo, stats[0], mul) if att else None
            Y = st.gemm(enc, wb, bb, B * L, N, F, sc=sc, sh=sh, rows_per_utt=L, ldw=wb.shape[1], ubias=row_bias, rscale=row_scale)
            if debug is not None and first:
                debug["bottleneck"] = Y.view(B, L, F).clone()
            return _masks_from_frames(st, sep, Y, geo, stats_block, mul_enc=enc, debug=debug if first else None), sc, sh

        masked, sc, sh = separate(v0, None, None, stats[:nb], True)
        v_last = v0
        if ira:
            # second embedding from the first estimate d0 = mask0 * enc (dprnn_spe_ira.py:79-92), then the second pass
            d0 = masked.view(B, 2, L, N)[:, 0].contiguous().view(B * L, N)
            st.call("dprnn_rows_stats", E._p(d0), E._p(stats[-2]), B, L, N)
            g0, b0, _ = E._norm_params(se[0])
            sc_d = torch.empty(B, N, device=dev, dtype=torch.float32)
            sh_d = torch.empty_like(sc_d)
            st.call("dprnn_bottleneck_prep", E._p(stats[-2]), E._p(E._param(g0, "spk_encoder.0.weight", dev)), E._p(E._param(b0, "spk_encoder.0.bias", dev)),
                    1e-5, 0, 0, E._p(sc_d), E._p(sh_d), B, L, N)
            A0d = E._pointwise(st, d0, B * L, N, se[1].weight, se[1].bias, 128, in_scale=sc_d, in_shift=sh_d, aff_stride=N, rows_per_utt=L)
            v1 = E.speaker_embedding(st, sep, A0d, B, L, aux_len_dev, ks)
            v1 = E._linear_small(st, sep.aux_linear.weight, sep.aux_linear.bias, torch.cat((v0, v1), dim=1).contiguous())
            if debug is not None:
                debug["v1"] = v1.clone()
            del masked
            masked, sc, sh = separate(v1, sc, sh, stats[nb: 2 * nb], False)
            v_last = v1
        logits = E._linear_small(st, sep.pred_linear.weight, sep.pred_linear.bias, v_last)
        wd = dec_w.view(N, geo.ks).t().contiguous()
        taps = st.gemm(masked, wd, None, B * 2 * L, N, geo.ks)
        est = E.decode(st, taps, geo, 1)                     # speaker 0 only (dprnn_spe.py:325)
        return est[:, 0, :], logits


def spe_forward(model, wav: torch.Tensor, aux: torch.Tensor, aux_len, ira: bool = False, debug: Optional[dict] = None):
    wav, aux = E.check_input(wav), E.check_input(aux, "aux")
    load_library()
    return _spe_forward_impl(model, wav, aux, aux_len, ira, debug)


def tasnet_forward(model, wav: torch.Tensor, debug: Optional[dict] = None) -> torch.Tensor:
    if torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters()) and wav.requires_grad:
        raise NotImplementedError("the any-width path is inference only")
    wav = E.check_input(wav)
    load_library()
    return _forward_impl(model, wav, debug)


RAGGED_MAX_SLAB_BYTES = 48 << 30          # slab rows x (F + 2 max(F, H) + F) floats of one ragged batch (X, the recurrence output, Z)


class _RaggedSlab:
    """The concatenated slabs [S_u][K][F] of utterances of different lengths and the launches that run over all of them at once."""

    def __init__(self, st: _Launcher, sep, geos):
        import numpy as np
        self.st, self.sep, self.geos = st, sep, geos
        self.cfg = _sep_cfg(sep)
        check_supported(self.cfg)
        self.F, self.H, self.K = self.cfg["feature_size"], self.cfg["hidden_size"], self.cfg["chunk_length"]
        self.cell = CELL_ID[self.cfg["rnn_type"]]
        dev = st.device
        self.rows = [g.rows for g in geos]
        self.starts = np.concatenate([[0], np.cumsum(np.array(self.rows, dtype=np.int64))]).astype(np.int64)
        self.total = int(self.starts[-1])
        self.n_intra = int(sum(g.S for g in geos))
        # inter-chunk tiles: 16 chunk positions of one utterance, S_u steps, rows K apart; longest first (their workgroups run longest)
        self.tiles = {}
        for width in (16, 32, 64):      # 16 positions per tile for the fp32 kernel, 64 / 32 for the f16-pieces one
            desc = []
            for u in np.argsort(-np.array([g.S for g in geos]), kind="stable"):
                for k0 in range(0, self.K, width):
                    desc.append((int(self.starts[u]) + k0, geos[u].S, min(width, self.K - k0), 1))
            self.tiles[width] = (torch.from_numpy(np.asarray(desc, dtype=np.int64)).to(dev), len(desc))
        F, H = self.F, self.H
        self.X = torch.empty(self.total, F, device=dev, dtype=torch.float32)
        self.Hbuf = torch.empty(self.total * 2 * max(H, F), device=dev, dtype=torch.float32)
        self.Z = torch.empty(self.total, F, device=dev, dtype=torch.float32)

    def rows_of(self, t: torch.Tensor, u: int) -> torch.Tensor:
        return t[int(self.starts[u]): int(self.starts[u + 1])]

    def segment(self, u: int, Y: torch.Tensor) -> None:
        g = self.geos[u]
        self.st.call("dprnn_wide_segment_fwd", E._p(Y), E._p(self.rows_of(self.X, u)), 1, g.L, g.K, g.P, g.S, self.F)

    def blocks(self) -> None:
        """All DPRNN blocks over the slab: one recurrence launch per path, the Linear / norm / residual per utterance (its sums are its own)."""
        st, F, H, K, dev = self.st, self.F, self.H, self.K, self.st.device
        U = len(self.geos)
        blocks = list(self.sep.dprnn_blocks)
        stats = torch.zeros(2 * len(blocks), U, 2, device=dev, dtype=torch.float64)
        f16 = f16_recurrence(self.cell, F, H)
        amax = None
        for i, blk in enumerate(blocks):
            for j, (single, lin, norm, intra) in enumerate(((blk.intra_rnn, blk.intra_linear, blk.intra_norm, True),
                                                            (blk.inter_rnn, blk.inter_linear, blk.inter_norm, False))):
                if intra:
                    Hout, ndir = _recurrence(st, single.rnn, self.cell, self.X, self.total, self.Hbuf, F, H, (self.n_intra, K, 1, K, 0, 1), amax_x=amax)
                else:
                    Hout, ndir = _recurrence(st, single.rnn, self.cell, self.X, self.total, self.Hbuf, F, H, (0, 0, 1, 0, 0, K), self.tiles, amax_x=amax)
                amax = torch.zeros(1, device=dev, dtype=torch.int32) if f16 else None
                if f16:
                    Wpl, scl, b = _pack_linear_f16(st, lin, ndir * H, F, dev)
                else:
                    w = E._param(lin.weight, "linear.weight", dev)
                    b = E._param(lin.bias, "linear.bias", dev)
                slot = stats[2 * i + j]
                for u in range(U):
                    n = self.rows[u]
                    Xu, Zu = self.rows_of(self.X, u), self.rows_of(self.Z, u)
                    if f16:
                        st.call("dprnn_wide_linear_f16", E._p(self.rows_of(Hout, u)), ndir * H, E._p(Wpl), E._p(scl), E._p(b), E._p(Zu), F,
                                E._p(slot[u:u + 1]), n, n, ndir * H, F)
                    else:
                        st.gemm(self.rows_of(Hout, u), w, b, n, ndir * H, F, rows_per_utt=n, stats=slot[u:u + 1], out=Zu)
                    sc, sh = st.norm_affine(slot[u:u + 1], norm, 1, n, F)
                    st.call("dprnn_wide_norm_residual_fwd", E._p(Xu), E._p(Zu), E._p(sc), E._p(sh), n, F, n, E._p(amax))

    def head(self, u: int, enc: torch.Tensor) -> torch.Tensor:
        """Mask head of utterance u x its encoder rows -> [(spk, l)][N]."""
        return _head(self.st, self.sep, self.cfg, self.rows_of(self.X, u), self.Hbuf, self.geos[u], enc)


def _ragged_geometries(cfg: dict, wavs):
    geos = [E.make_geometry(1, int(w.numel()), cfg["input_size"], cfg["chunk_length"], cfg["hop_length"], cfg["kernel_size"], cfg["stride"]) for w in wavs]
    for g in geos:
        if g.T_out < 2:
            raise DprnnError("utterance too short")
    return geos


def _ragged_batch(model, wavs: List[torch.Tensor]) -> List[torch.Tensor]:
    """Utterances of DIFFERENT lengths through one recurrence launch per DPRNN path and direction set (batch-of-one semantics each: the
    reference evaluates them one at a time, inferencer.py:54-58).  The slabs [S_u][K][F] are concatenated: the intra-chunk recurrence sees
    sum(S_u) sequences of K steps; the inter-chunk one gets a tile descriptor per (utterance, 16 chunk positions) with its own step count
    S_u (dprnn_wide_rnn_fwd_tiles); everything per-utterance by nature (global norms, Linear + its sums, mask head) is launched per
    utterance on that utterance's rows -- a few per cent of the work."""
    cfg = model.cfg
    check_supported(cfg)
    dev = wavs[0].device
    N = cfg["input_size"]
    sep = model.separation
    with torch.cuda.device(dev):
        st = _Launcher(dev)
        geos = _ragged_geometries(cfg, wavs)
        enc_w = E._param(model.encoder.conv1d.weight, "encoder.conv1d.weight", dev)
        dec_w = E._param(model.decoder.weight, "decoder.weight", dev)
        slab = _RaggedSlab(st, sep, geos)
        stats0 = torch.zeros(len(wavs), 2, device=dev, dtype=torch.float64)
        encs = []
        for u, (w, g) in enumerate(zip(wavs, geos)):
            enc = torch.empty(g.L, N, device=dev, dtype=torch.float32)
            st.call("dprnn_wide_encoder_fwd", E._p(w), E._p(enc_w), E._p(enc), 1, g.T, g.L, N, g.ks, g.stride)
            slab.segment(u, _frames_from_rows(st, sep, enc, 1, g.L, stats0[u:u + 1]))
            encs.append(enc)
        slab.blocks()
        wd = dec_w.view(N, geos[0].ks).t().contiguous()
        outs = []
        for u, g in enumerate(geos):
            taps = st.gemm(slab.head(u, encs[u]), wd, None, 2 * g.L, N, g.ks)
            outs.append(E.decode(st, taps, g, 2)[0])
        return outs


def _spe_ragged_batch(model, wavs, auxs, aux_lens, ira: bool):
    """DPRNNSpeTasNet / DPRNNSpeIRATasNet over mixtures and reference utterances of different lengths -> [(estimate [T_out_u], logits)]:
    speaker branch, fusion operands and bottleneck per utterance (batch-of-one semantics, BatchNorm statistics of the speaker branch
    included, in order: inferencer_spe.py:27-32), the DPRNN stack of every pass as one ragged slab."""
    cfg = model.cfg
    check_supported_spe(cfg)
    dev = wavs[0].device
    N, F = cfg["input_size"], cfg["feature_size"]
    ks, stride = cfg["kernel_size"], cfg["stride"]
    sep = model.separation
    se = sep.spk_encoder
    att = cfg["fusion_type"] == "att"
    U = len(wavs)
    with torch.cuda.device(dev), E.use_gemm_mode(0):
        st = _Launcher(dev)
        st.lib.dprnn_set_gemm_mode(0)
        geos = _ragged_geometries(cfg, wavs)
        enc_w = E._param(model.encoder.conv1d.weight, "encoder.conv1d.weight", dev)
        dec_w = E._param(model.decoder.weight, "decoder.weight", dev)
        conv = sep.bottleneck[1]
        wb = E._param(conv.weight, "bottleneck.weight", dev)
        bb = E._param(conv.bias, "bottleneck.bias", dev)
        stats = torch.zeros(3, U, 2, device=dev, dtype=torch.float64)          # encoder sums of the mixture, of the reference, of d0
        wav2 = [w.reshape(1, -1) for w in wavs]
        encs, v0s, scsh, alens = [], [], [], []
        for u, g in enumerate(geos):
            aux = E.check_input(auxs[u].reshape(1, -1), "aux")
            ga = E.make_geometry(1, aux.shape[1], N, cfg["chunk_length"], cfg["hop_length"], ks, stride)
            alen = E._aux_len_tensor(aux_lens[u], 1, dev)
            sc_a, sh_a = E.bottleneck_affine(st, aux, enc_w, se[0], ga, stats[1, u:u + 1])
            A0 = E.bottleneck_frames(st, aux, enc_w, sc_a, sh_a, se[1], ga, gemm_mode=0, scale_slot=E.SLOT_AUX_BOTTLENECK).view(ga.L, E.F_BUILT)
            v0s.append(E.speaker_embedding(st, sep, A0, 1, ga.L, alen, ks))
            alens.append(alen)
            enc = torch.empty(g.L, N, device=dev, dtype=torch.float32)
            st.call("dprnn_wide_encoder_fwd", E._p(wav2[u]), E._p(enc_w), E._p(enc), 1, g.T, g.L, N, g.ks, g.stride)
            encs.append(enc)
            scsh.append(None)

        def one_pass(vs, first: bool):
            slab = _RaggedSlab(st, sep, geos)
            for u, g in enumerate(geos):
                mul, add, row_bias = E._fusion_operands(st, sep, cfg, vs[u])
                if first:
                    scsh[u] = E.bottleneck_affine(st, wav2[u], enc_w, sep.bottleneck[0], g, stats[0, u:u + 1], mul, add)
                elif mul is not None or add is not None:        # the cached norm output (dprnn_spe_ira.py:96): same sums, new fusion operands
                    gm, bt, eps = E._norm_params(sep.bottleneck[0])
                    st.call("dprnn_bottleneck_prep", E._p(stats[0, u:u + 1]), E._p(gm), E._p(bt), eps, E._p(mul), E._p(add), E._p(scsh[u][0]),
                            E._p(scsh[u][1]), 1, g.L, N)
                row_scale = E._att_row_scale(st, sep, cfg, wav2[u], enc_w, g, stats[0, u:u + 1], mul) if att else None
                Y = st.gemm(encs[u], wb, bb, g.L, N, F, sc=scsh[u][0], sh=scsh[u][1], rows_per_utt=g.L, ldw=wb.shape[1], ubias=row_bias,
                            rscale=row_scale)
                slab.segment(u, Y)
            slab.blocks()
            return [slab.head(u, encs[u]) for u in range(U)]

        masked = one_pass(v0s, True)
        vlast = v0s
        if ira:
            v1s = []
            g0, b0, _ = E._norm_params(se[0])
            g0, b0 = E._param(g0, "spk_encoder.0.weight", dev), E._param(b0, "spk_encoder.0.bias", dev)
            for u, g in enumerate(geos):
                d0 = masked[u][: g.L]                                                # speaker 0's rows: mask_0 x enc (dprnn_spe_ira.py:79-92)
                st.call("dprnn_rows_stats", E._p(d0), E._p(stats[2, u:u + 1]), 1, g.L, N)
                sc_d = torch.empty(1, N, device=dev, dtype=torch.float32)
                sh_d = torch.empty_like(sc_d)
                st.call("dprnn_bottleneck_prep", E._p(stats[2, u:u + 1]), E._p(g0), E._p(b0), 1e-5, 0, 0, E._p(sc_d), E._p(sh_d), 1, g.L, N)
                A0d = E._pointwise(st, d0, g.L, N, se[1].weight, se[1].bias, 128, in_scale=sc_d, in_shift=sh_d, aff_stride=N, rows_per_utt=g.L)
                v1 = E.speaker_embedding(st, sep, A0d, 1, g.L, alens[u], ks)
                v1s.append(E._linear_small(st, sep.aux_linear.weight, sep.aux_linear.bias, torch.cat((v0s[u], v1), dim=1).contiguous()))
            del masked
            masked = one_pass(v1s, False)
            vlast = v1s
        wd = dec_w.view(N, geos[0].ks).t().contiguous()
        outs = []
        for u, g in enumerate(geos):
            logits = E._linear_small(st, sep.pred_linear.weight, sep.pred_linear.bias, vlast[u])
            taps = st.gemm(masked[u], wd, None, 2 * g.L, N, g.ks)
            outs.append((E.decode(st, taps, g, 1)[0, 0, :], logits[0]))
        return outs


def _ragged_chunks(cfg: dict, wavs, passes_held: int = 1):
    """Indices of the utterances of each ragged batch, in the order given: as many as RAGGED_MAX_SLAB_BYTES of slab (X, recurrence output, Z) hold."""
    F, H = cfg["feature_size"], cfg["hidden_size"]
    per_row = 4 * (2 * F + 2 * max(F, H)) * passes_held
    chunks, batch, used = [], [], 0
    for i, w in enumerate(wavs):
        g = E.make_geometry(1, int(w.numel()), cfg["input_size"], cfg["chunk_length"], cfg["hop_length"], cfg["kernel_size"], cfg["stride"])
        need = g.rows * per_row
        if batch and used + need > RAGGED_MAX_SLAB_BYTES:
            chunks.append(batch)
            batch, used = [], 0
        batch.append(i)
        used += need
    if batch:
        chunks.append(batch)
    return chunks


def spe_forward_ragged(model, wavs, auxs, aux_lens, ira: bool = False):
    """The speaker-conditioned classes over utterances of different lengths (inferencer_spe.py:20-54), list of (estimate, logits)."""
    load_library()
    outs: List = [None] * len(wavs)
    for idx in _ragged_chunks(model.cfg, wavs):
        with torch.no_grad():
            res = _spe_ragged_batch(model, [E.check_input(wavs[i].reshape(1, -1)).reshape(-1) for i in idx], [auxs[i] for i in idx],
                                    [aux_lens[i] for i in idx], ira)
        for i, o in zip(idx, res):
            outs[i] = o
    return outs


def tasnet_forward_ragged(model, wavs: List[torch.Tensor]) -> List[torch.Tensor]:
    """Utterances of different lengths, batch-of-one semantics each (inferencer.py:40-60): ragged batches as large as RAGGED_MAX_SLAB_BYTES
    allows, in the order given (pass them sorted by length for short tails)."""
    load_library()
    outs: List[Optional[torch.Tensor]] = [None] * len(wavs)
    for idx in _ragged_chunks(model.cfg, wavs):
        with torch.no_grad():                              # an inference entry point, like engine.tasnet_forward_ragged
            res = _ragged_batch(model, [E.check_input(wavs[i].reshape(1, -1)).reshape(-1) for i in idx])
        for i, o in zip(idx, res):
            outs[i] = o
    return outs
