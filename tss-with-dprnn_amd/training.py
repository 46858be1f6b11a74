"""Training step of DPRNNTasNet on the MI355X path.

The reference trains with stock autograd (src/trainers/trainer.py:98-118): ``out = model(mix)``,
``loss = PITLossWrapper(pairwise_neg_sisdr)(out, target)``, ``loss.backward()``, ``clip_grad_norm_``,
``optimizer.step()``.  Here the model forward under autograd is one ``torch.autograd.Function`` whose forward enqueues
the training variants of the forward kernels (they keep what the backward needs: the slab after every path, the Linear
outputs, the recurrence outputs and the LSTM gate tape) and whose backward enqueues the hand-written gradient kernels of
include/dprnn_mi355x_train.h.  autograd only routes the resulting tensors into ``param.grad``: the reference's
``Trainer`` loop, torch optimizers and ``clip_grad_norm_`` run unchanged on top.  There is no PyTorch fallback for
any of the arithmetic.

Memory (MI355X, 288 GB): the tape is 5 fp32 values per (row, direction, hidden unit) and path -- 3.0 GB per 3 s utterance
for the shipped 6-block model, 4.2 GB with the other saved slabs -- so a whole 64-utterance batch does not fit.
``SAVE_BUDGET_BYTES`` bounds what one pass keeps; a larger batch is cut into micro-batches that are re-run in the
backward (forward recomputation, +1 forward per step), smaller batches keep everything from the first forward.
"""
from __future__ import annotations

import contextlib
import os
import threading
from typing import Dict, List, Optional

import torch

from . import engine
from ._lib import DprnnError, load_library
from .engine import F_BUILT, Stage, _norm_params, _p, _param, _rnn_tensors

SAVE_BUDGET_BYTES = 200 << 30        # cap on the activations one forward may keep for its backward (of 288 GB HBM3E per GPU)
SAVE_BUDGET_FRACTION = 0.70          # ... and at most this share of what is free on the device when the forward starts
# Over-budget batches: True = the first micro-batch is as large as the budget allows (fewest utterances re-run in the
# backward: B = 64 keeps 47 and re-runs 17, 753 ms per step); False = balanced micro-batches (32 + 32: 793 ms).
KEEP_MAX_FIRST = True
PACKT_PER_DIR = 512 * 256
# Utterances are independent, so a (micro-)batch is processed as TRAIN_STREAM_GROUPS complete forward / backward
# pipelines on their own HIP streams (as the inference path does): one group's weight-gradient GEMMs and streaming
# passes fill the CUs another group's serial recurrence launches leave idle.
TRAIN_STREAM_GROUPS = int(os.environ.get("DPRNN_TRAIN_STREAM_GROUPS", "0"))              # 0 = choose from the launch geometry (_auto_groups); n > 0 = up to n groups
MIN_TRAIN_GROUP = 8                  # utterances per group at least (explicit group counts only)
# Gate-GEMM arithmetic of the training FORWARD (recurrence with tape + the Linear after it + mask head): 3 = two f16 pieces / three
# products as in inference (LSTM cells only, others fall back), 0 = exact fp32 MFMA.  DPRNN_TRAIN_GEMM_MODE overrides.
TRAIN_GEMM_MODE = int(os.environ.get("DPRNN_TRAIN_GEMM_MODE", "3"))
# Step GEMM of the LSTM backward (BPTT) AND the LSTM weight gradients dG^T [x | h_prev]: 3 = two f16 pieces / three products with a
# per-launch power-of-two gradient scale (train_recur_f16.hip; the BPTT then writes dG in the quad layout that only
# dprnn_wgrad_lstm_f16 reads), 0 = exact fp32 MFMA for both.  DPRNN_TRAIN_BWD_GEMM_MODE overrides.
TRAIN_BWD_GEMM_MODE = int(os.environ.get("DPRNN_TRAIN_BWD_GEMM_MODE", "3"))
# ... and, with mode 3, the gradient GEMMs of the Linear after each recurrence (dZ @ W, dZ^T H) in f16 pieces, dZ scaled from its own
# maximum (DPRNN_TRAIN_WGRAD_F16=0: fp32 MFMA for those).  Mask-head and bottleneck gradients are always fp32 MFMA.
TRAIN_WGRAD_F16 = os.environ.get("DPRNN_TRAIN_WGRAD_F16", "1") != "0"
# f16-pieces forward AND f16-pieces BPTT of an LSTM: the gate tape holds i, f, g, o as 24-bit fixed point (three words per (row, unit)
# instead of four; csrc/common.h: tape_pack) -- a fifth less of the step's largest HBM stream.  DPRNN_TRAIN_TAPE_PACKED=0: fp32 gates.
TAPE_PACKED = os.environ.get("DPRNN_TRAIN_TAPE_PACKED", "1") != "0"
# The GroupNorm backward's second pass and the Linear's data and weight gradients as ONE launch over the rows (dprnn_lin_bwd_f16: 6 slab units
# per row instead of 9; bidirectional paths with the Linears' gradients in f16 pieces).  DPRNN_TRAIN_LIN_BWD_FUSED=0: the three launches.
LIN_BWD_FUSED = os.environ.get("DPRNN_TRAIN_LIN_BWD_FUSED", "1") != "0"
TAPE_PACKED_FLAG = 0x100             # on the gemm_mode argument of dprnn_rnn_fwd_train_ex / dprnn_rnn_bwd_ex
_SIDE_STREAMS: Dict[torch.device, List[torch.cuda.Stream]] = {}
_TLS = threading.local()


def train_modes():
    """(forward gate-GEMM mode, BPTT + LSTM weight-gradient mode, f16 gradient GEMMs of the Linears) of this thread's training calls: the module settings unless a
    `use_train_modes` block is active on the thread (the saturation fall-back runs one)."""
    o = getattr(_TLS, "modes", None)
    fwd = TRAIN_GEMM_MODE if o is None or o[0] is None else o[0]
    bwd = TRAIN_BWD_GEMM_MODE if o is None or o[1] is None else o[1]
    return (3 if fwd == 3 else 0), (3 if bwd == 3 else 0), (TRAIN_WGRAD_F16 and bwd == 3)


@contextlib.contextmanager
def use_train_modes(fwd: Optional[int] = None, bwd: Optional[int] = None):
    prev = getattr(_TLS, "modes", None)
    _TLS.modes = (fwd if fwd is not None else (prev[0] if prev else None), bwd if bwd is not None else (prev[1] if prev else None))
    try:
        yield
    finally:
        _TLS.modes = prev


def _exact(fn, fwd: bool = True, bwd: bool = True):
    """fn under the exact fp32 kernels (the range-free arithmetic a saturated f16-pieces training call is repeated in)."""
    def run():
        with use_train_modes(0 if fwd else None, 0 if bwd else None):
            return fn()
    return run


def check_supported_train(cfg: dict) -> None:
    engine.check_supported(cfg)          # LSTM, GRU and RNN cells all have forward-with-tape and backward kernels


def save_budget_bytes(dev) -> int:
    """Bytes of activations a forward may keep: SAVE_BUDGET_BYTES, but never more than SAVE_BUDGET_FRACTION of the memory
    that is actually free on `dev` right now (free HBM + what torch's caching allocator holds unused), so that a second
    rank on the same GPU, a smaller part or other resident tensors lead to smaller micro-batches instead of a HIP OOM.
    The fraction is the ratio measured on an otherwise empty MI355X: a 200 GB budget peaks at 205 GiB of 268 GiB."""
    free, _ = torch.cuda.mem_get_info(dev)
    cached = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
    return max(1, min(SAVE_BUDGET_BYTES, int(SAVE_BUDGET_FRACTION * (free + cached))))


def saved_bytes_per_utt(geo: engine.Geometry, n_paths: int, ndirs: List[int]) -> int:
    rows = geo.S * geo.K
    tape = sum(5 * nd for nd in ndirs) * rows * F_BUILT * 4
    slabs = (n_paths + 1) * rows * F_BUILT * 4 + n_paths * rows * F_BUILT * 4 + sum(ndirs) * rows * F_BUILT * 4
    head = geo.L * (2 * F_BUILT + 2 * F_BUILT + 2 * 2 * F_BUILT + 2 * geo.N) * 4
    return tape + slabs + head


def _paths(sep):
    out = []
    for blk in sep.dprnn_blocks:
        out.append((blk.intra_rnn, blk.intra_linear, blk.intra_norm, True))
        out.append((blk.inter_rnn, blk.inter_linear, blk.inter_norm, False))
    return out


def _addr(geo, intra):
    B, S, K = geo.B, geo.S, geo.K
    return (B * S, K, B * S, 0, K, 1) if intra else (B * K, S, K, S * K, 1, K)


def forward_train(model, wav: torch.Tensor, keep: bool = True, fusion: Optional[dict] = None, want_d0: bool = False):
    """DPRNNTasNet.forward (dprnn.py:271-283) keeping the activations the backward needs.  Returns (out, saved).
    fusion (DPRNN-Spe, dprnn_spe.py:165-229): dict(mul, add, row_bias) entering the bottleneck; the output is then the
    [B, 2, T_out] decode of both mask channels, of which DPRNN-Spe uses speaker 0."""
    cfg = model.cfg
    dev = wav.device
    st = Stage(dev)
    lib = st.lib
    geo = engine.make_geometry(wav.shape[0], wav.shape[1], cfg["input_size"], cfg["chunk_length"], cfg["hop_length"],
                               cfg["kernel_size"], cfg["stride"])
    if geo.T_out < 2:
        raise DprnnError("utterance too short")
    sep = model.separation
    paths = _paths(sep)
    npth = len(paths)
    B, L, K, P, S, rows = geo.B, geo.L, geo.K, geo.P, geo.S, geo.rows
    enc_w = _param(model.encoder.conv1d.weight, "encoder.conv1d.weight", dev)
    dec_w = _param(model.decoder.weight, "decoder.weight", dev)
    f32 = dict(device=dev, dtype=torch.float32)
    stats = torch.zeros(1 + npth, B, 2, device=dev, dtype=torch.float64)
    lib.dprnn_set_gemm_mode(0)
    lstm_only = all(rnn.rnn.mode == "LSTM" for rnn, _, _, _ in paths)
    fwd_mode = 3 if (train_modes()[0] == 3 and lstm_only) else 0
    fusion = fusion or {}
    sc, sh = engine.bottleneck_affine(st, wav, enc_w, sep.bottleneck[0], geo, stats[0], fusion.get("mul"), fusion.get("add"))
    Y = engine.bottleneck_frames(st, wav, enc_w, sc, sh, sep.bottleneck[1], geo, fusion.get("row_bias"), fusion.get("row_scale"),
                                 gemm_mode=fwd_mode)
    Y_keep = Y if (keep and fusion.get("row_scale") is not None) else None        # 'att': the backward needs W a' = (Y - b) / rs
    X = [torch.empty(rows, F_BUILT, **f32)]
    Z, H, tapes, tape_flags = [], [], [], []
    st.call("dprnn_segment_fwd", _p(Y), _p(X[0]), B, L, K, P, S)
    del Y
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, **f32)
    bsum = torch.empty(2 * 4 * F_BUILT, **f32)
    # Exact fp32 forward: norm + residual stay a separate streaming pass -- folded into the next recurrence's loader (as the
    # inference path does) the extra vector work inside the fp32-MFMA-bound kernel cost more than the pass it saves once the tape
    # stores are in the same kernel (measured at B = 32: 109.6 ms against 101.4 + 5.7 ms).  f16-pieces forward: vector work rides
    # under the f16 MFMAs, so the pending norm of path p-1 is applied by the loader of path p (which writes x' = X[p] once, for
    # the backward); only the last path's norm is an explicit pass.
    fold = fwd_mode == 3
    lib.dprnn_set_gemm_mode(fwd_mode)       # pack, recurrence and projection of every path; back to the exact mode whatever happens
    try:
        pending = None
        for p, (rnn, lin, norm, intra) in enumerate(paths):
            ndir, fw, bw = _rnn_tensors(rnn, dev)
            cell = engine.CELL_ID[rnn.rnn.mode]
            st.call("dprnn_rnn_pack_ex", fwd_mode, cell, *[_p(t) for t in fw], *[_p(t) for t in bw], _p(packed), _p(bsum))
            addr = _addr(geo, intra)
            Hp = torch.empty(rows, ndir * F_BUILT, **f32)
            tape = torch.empty(lib.dprnn_rnn_tape_floats(cell, addr[0], addr[1], ndir), **f32)
            # the backward of THIS forward runs in the modes of this thread now (ctx.modes carries them)
            tflag = TAPE_PACKED_FLAG if (TAPE_PACKED and fwd_mode == 3 and cell == 0 and train_modes()[1] == 3) else 0
            if pending is None:
                st.call("dprnn_rnn_fwd_train_ex", fwd_mode | tflag, cell, _p(X[p]), 0, 0, 0, 0, 0.0, 1, 1, 0, _p(packed), _p(bsum), _p(Hp), _p(tape), *addr, ndir)
            else:               # X[p] = X[p-1] + norm(Z[p-1]) on the fly
                Zq, sq, gq, bq, eq = pending
                Xn = torch.empty(rows, F_BUILT, **f32)
                st.call("dprnn_rnn_fwd_train_ex", fwd_mode | tflag, cell, _p(X[p - 1]), _p(Zq), _p(sq), _p(gq), _p(bq), eq, S if intra else K, S * K, _p(Xn),
                        _p(packed), _p(bsum), _p(Hp), _p(tape), *addr, ndir)
                X.append(Xn)
                if not keep:
                    X[p - 1] = None
                pending = None
            w = _param(lin.weight, "linear.weight", dev)
            if w.shape != (F_BUILT, ndir * F_BUILT):
                raise DprnnError(f"linear weight shape {tuple(w.shape)} does not match ndir={ndir}")
            Zp = torch.empty(rows, F_BUILT, **f32)
            st.call("dprnn_proj_stats_fwd", _p(Hp), _p(w), _p(_param(lin.bias, "linear.bias", dev)), _p(Zp), _p(stats[1 + p]),
                    rows, S * K, ndir * F_BUILT)
            engine._count_scale(stats[1 + p], norm, st, B)      # zero-padded twin of a narrow model: sums over the REAL channel count
            gamma, beta, eps = _norm_params(norm)
            gq, bq = _param(gamma, "norm.weight", dev), _param(beta, "norm.bias", dev)
            if fold and p + 1 < npth:
                pending = (Zp, stats[1 + p], gq, bq, eps)
            else:
                Xn = torch.empty(rows, F_BUILT, **f32)
                st.call("dprnn_norm_residual_fwd", _p(X[p]), _p(Zp), _p(stats[1 + p]), _p(gq), _p(bq), eps, _p(Xn), B, S * K)
                X.append(Xn)
                if not keep:
                    X[p] = None
            if keep:
                Z.append(Zp); H.append(Hp); tapes.append(tape); tape_flags.append(tflag)
    finally:
        lib.dprnn_set_gemm_mode(0)
    R = B * L * 2
    U = torch.empty(B * L, 2 * F_BUILT, **f32)
    V = torch.empty(R, F_BUILT, **f32)
    TG = torch.empty(R, 2 * F_BUILT, **f32)
    mask = torch.empty(R, geo.N, **f32)
    taps = torch.empty(B, 2, L, geo.ks, **f32)
    d0 = torch.empty(B * L, geo.N, **f32) if want_d0 else None            # mask_0 * enc (DPRNN-Spe-IRA)
    lib.dprnn_set_gemm_mode(fwd_mode)       # the mask head in the forward's arithmetic too (the same kernels as inference)
    try:
        st.call("dprnn_head_conv_ola_fwd", _p(X[npth]), _p(_param(sep.prelu.weight, "prelu.weight", dev)),
                _p(_param(sep.conv2d.weight, "conv2d.weight", dev)), _p(_param(sep.conv2d.bias, "conv2d.bias", dev)),
                _p(U), B, L, K, P, S)
        st.call("dprnn_head_gate_fwd_train", _p(U), _p(_param(sep.out[0].weight, "out.0.weight", dev)),
                _p(_param(sep.out[0].bias, "out.0.bias", dev)), _p(_param(sep.gate[0].weight, "gate.0.weight", dev)),
                _p(_param(sep.gate[0].bias, "gate.0.bias", dev)), _p(V), _p(TG), R)
        st.call("dprnn_head_mask_decode_fwd_train", _p(V), _p(_param(sep.end_conv1x1.weight, "end_conv1x1.weight", dev)),
                _p(wav), _p(enc_w), _p(dec_w), _p(taps), _p(mask), _p(d0), 1 if cfg["activation_type"] == "relu" else 0,
                B, geo.T, L, geo.N, geo.ks, geo.stride)
    finally:
        lib.dprnn_set_gemm_mode(0)
    out = engine.decode(st, taps, geo, 2)
    if not keep:
        return out, None
    saved = dict(geo=geo, wav=wav, stats=stats, sc=sc, sh=sh, X=X, Z=Z, H=H, tapes=tapes, tape_flags=tape_flags, U=U, V=V, TG=TG, mask=mask,
                 fusion=fusion, Y=Y_keep, d0=d0)
    return out, saved


class _Scratch:
    """Zeroed fp64 scratch for the small reductions of one backward pass, handed out in named slots."""

    def __init__(self, dev, n: int = 1 << 16):
        self.buf = torch.zeros(n, device=dev, dtype=torch.float64)
        self.off: Dict[str, int] = {}
        self.top = 0

    def slot(self, name: str, n: int) -> int:
        """Device address of slot `name` (created on first use)."""
        if name not in self.off:
            if self.top + n > self.buf.numel():
                raise DprnnError("training scratch exhausted")
            self.off[name] = self.top
            self.top += n
        return self.buf.data_ptr() + 8 * self.off[name]

    def view(self, name: str, n: int) -> torch.Tensor:
        """Slot `name` (already handed out) as a tensor of n doubles."""
        return self.buf[self.off[name]: self.off[name] + n]


def _count_scale_red(scr: "_Scratch", name: str, norm, st: Stage, n: int) -> None:
    """Zero-padded twin of a narrow model (padded.py): the two per-utterance sums of a global norm's backward (sum of gamma dy, sum of
    gamma dy x_hat -- the padded channels add nothing to either, their gamma is zero) are divided by rows x BUILT channels in the kernel
    that uses them; scaled by built / real channels that is the mean over the real element count, as in the forward
    (engine._count_scale)."""
    f = norm.__dict__.get("_dprnn_count_scale", 1.0)
    if f != 1.0:
        with torch.cuda.stream(st.tstream):
            scr.view(name, 2 * n).mul_(f)


def backward(model, saved: dict, d_out: torch.Tensor, scratch: Optional[_Scratch] = None, extras: Optional[dict] = None
             ) -> Dict[str, torch.Tensor]:
    """Gradients of every parameter of the separator path for the upstream gradient d_out [B, 2, T_out]; keys are
    state_dict names.  With `extras` (DPRNN-Spe) the gradients of the fusion operands are returned in it (d_mul, d_add,
    d_row_bias: float32 [B, N|F]) and the encoder-weight gradient stays in scratch slot 'dWe' for the caller to finish
    after the reference-utterance branch has added its share."""
    cfg = model.cfg
    geo: engine.Geometry = saved["geo"]
    wav = saved["wav"]
    dev = wav.device
    st = Stage(dev)
    lib = st.lib
    sep = model.separation
    paths = _paths(sep)
    npth = len(paths)
    B, L, K, P, S, rows, N, ks = geo.B, geo.L, geo.K, geo.P, geo.S, geo.rows, geo.N, geo.ks
    R = B * L * 2
    f32 = dict(device=dev, dtype=torch.float32)
    d_out = d_out.contiguous().float()
    if d_out.shape != (B, 2, geo.T_out):
        raise DprnnError(f"upstream gradient has shape {tuple(d_out.shape)}, expected {(B, 2, geo.T_out)}")
    enc_w = model.encoder.conv1d.weight
    dec_w = model.decoder.weight
    stats, X, Z, H, tapes = saved["stats"], saved["X"], saved["Z"], saved["H"], saved["tapes"]
    grads: Dict[str, torch.Tensor] = {}

    scr = scratch if scratch is not None else _Scratch(dev)
    fusion = saved.get("fusion") or {}
    if fusion and extras is None:
        raise DprnnError("training path: fusion operands need the DPRNN-Spe backward (training_spe.py)")

    def to_grad(addr, like):
        g = torch.empty_like(like)
        st.call("dprnn_scratch_to_grad", addr, _p(g), g.numel(), 0)
        return g

    # one workspace for every weight-gradient GEMM of this pass: (M, N, rows) of all dprnn_wgrad calls below
    shapes = [(N, F_BUILT, R), (2 * F_BUILT, F_BUILT, R), (2 * F_BUILT, F_BUILT, B * L), (F_BUILT, 2 * F_BUILT, rows),
              (4 * F_BUILT, 2 * F_BUILT, rows), (F_BUILT, N, B * L),
              (4 * F_BUILT, 2 * F_BUILT, K * (-(-(B * S) // 32) * 32)), (4 * F_BUILT, 2 * F_BUILT, S * (-(-(B * K) // 32) * 32))]
    ws = torch.empty(max([lib.dprnn_wgrad_workspace_floats(m_, n_, r_) for m_, n_, r_ in shapes] + [lib.dprnn_lin_bwd_workspace_floats(rows)]), **f32)

    _, bwd_mode, wgrad_f16 = train_modes()

    def wgrad(Pp, ldp, M, Q0, ldq0, n0, dt0, Q1, ldq1, n1, dt1, nrows, t_stride, T, dW, ldm, ldn, db0, db1, f16_max=0):
        # f16_max: device float with the maximum the f16-pieces form scales P from (0 = the fp32 MFMA kernel)
        st.call("dprnn_wgrad_ex", f16_max, Pp, ldp, M, Q0, ldq0, n0, dt0, Q1, ldq1, n1, dt1, nrows, t_stride, T, _p(dW), ldm, ldn,
                _p(db0), _p(db1), 0, _p(ws))

    # ---- decoder, masking, mask activation (encoder_decoder.py:40-42, dprnn.py:277, :161-164)
    dE = torch.empty(R, N, **f32)
    dEnc = torch.empty(B * L, N, **f32)
    st.call("dprnn_mask_decode_bwd", _p(d_out), _p(wav), _p(enc_w), _p(dec_w), _p(saved["mask"]),
            1 if cfg["activation_type"] == "relu" else 0, _p(dE), _p(dEnc), scr.slot("dWd", N * ks),
            _p(extras.get("dM0")) if extras else 0, B, geo.T, L, N, ks, geo.stride)
    grads["decoder.weight"] = to_grad(scr.slot("dWd", N * ks), dec_w)
    # ---- end_conv1x1 (dprnn.py:160,181)
    w_end = sep.end_conv1x1.weight
    dV = torch.empty(R, F_BUILT, **f32)
    st.call("dprnn_dgrad", _p(dE), N, _p(w_end), F_BUILT, N, 0, 0, _p(dV), F_BUILT, F_BUILT, R)
    g = torch.empty_like(w_end)
    wgrad(_p(dE), N, N, _p(saved["V"]), F_BUILT, F_BUILT, 0, 0, 0, 0, 0, R, 1, 1, g, F_BUILT, 1, None, None)
    grads["separation.end_conv1x1.weight"] = g
    del dE
    # ---- tanh(out) * sigmoid(gate) (dprnn.py:157-158,180)
    dAG = torch.empty(R, 2 * F_BUILT, **f32)
    st.call("dprnn_gate_bwd", _p(dV), _p(saved["TG"]), _p(dAG), R)
    del dV
    dU = torch.empty(B * L, 2 * F_BUILT, **f32)          # rows (b,l,spk) of 128 == rows (b,l) of 256
    st.call("dprnn_dgrad", _p(dAG), 2 * F_BUILT, _p(sep.out[0].weight), F_BUILT, F_BUILT, _p(sep.gate[0].weight), F_BUILT,
            _p(dU), F_BUILT, F_BUILT, R)
    gw = torch.empty(2 * F_BUILT, F_BUILT, **f32)
    gb = torch.empty(2 * F_BUILT, **f32)
    wgrad(_p(dAG), 2 * F_BUILT, 2 * F_BUILT, _p(saved["U"]), F_BUILT, F_BUILT, 0, 0, 0, 0, 0, R, 1, 1, gw, F_BUILT, 1, gb, None)
    grads["separation.out.0.weight"] = gw[:F_BUILT].reshape(sep.out[0].weight.shape).clone()
    grads["separation.gate.0.weight"] = gw[F_BUILT:].reshape(sep.gate[0].weight.shape).clone()
    grads["separation.out.0.bias"] = gb[:F_BUILT].clone()
    grads["separation.gate.0.bias"] = gb[F_BUILT:].clone()
    del dAG
    # ---- PReLU -> conv2d -> overlap-add (dprnn.py:173-178,203-217)
    Xlast = X[npth]
    pq = torch.empty(B * L, F_BUILT, **f32)
    st.call("dprnn_ola_prelu_sum", _p(Xlast), _p(sep.prelu.weight), _p(pq), _p(dU), scr.slot("db2d", 2 * F_BUILT), B, L, K, P, S)
    g = torch.empty_like(sep.conv2d.weight)
    wgrad(_p(dU), 2 * F_BUILT, 2 * F_BUILT, _p(pq), F_BUILT, F_BUILT, 0, 0, 0, 0, 0, B * L, 1, 1, g, F_BUILT, 1, None, None)
    grads["separation.conv2d.weight"] = g
    grads["separation.conv2d.bias"] = to_grad(scr.slot("db2d", 2 * F_BUILT), sep.conv2d.bias)
    del pq
    dX = torch.empty(rows, F_BUILT, **f32)
    st.call("dprnn_head_conv_bwd", _p(dU), _p(sep.conv2d.weight), _p(Xlast), _p(sep.prelu.weight), _p(dX), scr.slot("dalpha", 1),
            B, L, K, P, S)
    grads["separation.prelu.weight"] = to_grad(scr.slot("dalpha", 1), sep.prelu.weight)
    del dU

    # ---- DPRNN blocks in reverse (dprnn.py:79-99)
    # the two per-direction dx slabs: a bidirectional BPTT writes every row of both, a unidirectional one leaves the second to the zero_() below
    dXd = torch.empty(2, rows, F_BUILT, **f32)
    dZ = torch.empty(rows, F_BUILT, **f32)
    packedT = torch.empty(lib.dprnn_rnn_pack_bwd_floats(2), **f32)
    zmax = torch.zeros(1, **f32)
    hmax = torch.zeros(1, **f32)
    have_dx = False
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    for p in range(npth - 1, -1, -1):
        rnn, lin, norm, intra = paths[p]
        blk = f"separation.dprnn_blocks.{p // 2}." + ("intra" if intra else "inter")
        ndir, fw, bw = _rnn_tensors(rnn, dev)
        gamma, beta, eps = _norm_params(norm)
        a_red, a_dgb = scr.slot(f"red{p}", 2 * B), scr.slot(f"dgb{p}", 2 * F_BUILT)
        f16_lin = wgrad_f16                                         # dZ @ W and dZ^T H in f16 pieces
        cell = engine.CELL_ID[rnn.rnn.mode]
        bptt_f16 = bwd_mode == 3 and cell == 0
        fused = f16_lin and ndir == 2 and LIN_BWD_FUSED and B <= 1024
        gname, bname = ("gamma", "beta") if hasattr(norm, "gamma") else ("weight", "bias")
        hp = ndir * F_BUILT
        dH = torch.empty(rows, hp, **f32)
        gw = torch.empty_like(lin.weight)
        gb = torch.empty_like(lin.bias)
        Hp = H[p]
        if fused:
            # pass 1 also leaves what bounds |dZ| (the scale of its f16 pieces); pass 2, the Linear's data gradient and its weight gradient
            # are one launch that forms dZ on the fly (dprnn_lin_bwd_f16); max|dH|, the f16 BPTT's scale, is measured where dH is written
            absmax = torch.empty(B, 2, device=dev, dtype=torch.int32)
            st.call("dprnn_gn_bwd_stats_ex", _p(dX), _p(dXd[0]) if have_dx else 0, _p(dXd[1]) if have_dx else 0,
                    _p(dX) if have_dx else 0, _p(Z[p]), _p(stats[1 + p]), _p(gamma), eps, B, S * K, a_red, a_dgb, _p(absmax))
            _count_scale_red(scr, f"red{p}", norm, st, B)
            st.call("dprnn_lin_bwd_f16", _p(dX), _p(Z[p]), _p(stats[1 + p]), a_red, _p(absmax), _p(gamma), eps, B, S * K, _p(Hp),
                    _p(lin.weight), _p(dH), _p(gw), _p(gb), hmax.data_ptr() if bptt_f16 else 0, _p(ws))
        else:
            st.call("dprnn_gn_bwd_stats", _p(dX), _p(dXd[0]) if have_dx else 0, _p(dXd[1]) if have_dx else 0,
                    _p(dX) if have_dx else 0, _p(Z[p]), _p(stats[1 + p]), _p(gamma), eps, B, S * K, a_red, a_dgb)
            _count_scale_red(scr, f"red{p}", norm, st, B)
            st.call("dprnn_gn_bwd_dz_ex", zmax.data_ptr() if f16_lin else 0,      # dZ scaled from its own maximum, which the kernel that writes dZ measures
                    _p(dX), _p(Z[p]), _p(stats[1 + p]), a_red, _p(gamma), eps, B, S * K, _p(dZ))
            # Linear (dprnn.py:88,96); the f16 BPTT scales by max|dH|: measured where dH is written
            st.call("dprnn_dgrad_ex", zmax.data_ptr() if (f16_lin and ndir == 2) else 0, hmax.data_ptr() if bptt_f16 else 0,
                    _p(dZ), F_BUILT, _p(lin.weight), hp, F_BUILT, 0, 0, _p(dH), hp, hp, rows)
            wgrad(_p(dZ), F_BUILT, F_BUILT, _p(Hp), hp, F_BUILT, 0, (Hp.data_ptr() + 4 * F_BUILT) if ndir == 2 else 0, hp,
                  F_BUILT if ndir == 2 else 0, 0, rows, 1, 1, gw, hp, 1, gb, None, f16_max=zmax.data_ptr() if f16_lin else 0)
        grads[f"{blk}_norm.{gname}"] = to_grad(a_dgb, gamma)
        grads[f"{blk}_norm.{bname}"] = to_grad(a_dgb + 8 * F_BUILT, beta)
        grads[f"{blk}_linear.weight"], grads[f"{blk}_linear.bias"] = gw, gb
        # recurrence (dprnn.py:86,95)
        addr = _addr(geo, intra)
        kg = F_BUILT if cell == 2 else 4 * F_BUILT          # gate-gradient columns per direction: LSTM i,f,g,o; GRU r,z,n_x,n_h; RNN a
        bm = 3 if bptt_f16 else 0                               # layout of the pack = kernel of the launch (checked by the library)
        # fp32 kernel: dG [rows][ndir * kg]; f16 BPTT: the quad layout [dir][t][seq / 4][512][4] that dprnn_wgrad_lstm_f16 reads
        dG = torch.empty(lib.dprnn_rnn_bwd_dg_floats(cell, bm, addr[0], addr[1], ndir), **f32)
        st.call("dprnn_rnn_pack_bwd_ex", bm, cell, _p(fw[0]), _p(fw[1]), _p(bw[0]), _p(bw[1]), _p(packedT))
        st.call("dprnn_rnn_bwd_ex", bm | saved["tape_flags"][p], hmax.data_ptr() if bptt_f16 else 0, cell, _p(dH), _p(tapes[p]), _p(packedT), _p(dXd), _p(dG),
                rows, *addr, ndir)
        del dH
        pre = f"{blk}_rnn.rnn."
        for d in range(ndir):
            suf = "_reverse" if d == 1 else ""
            gbase = dG.data_ptr() + 4 * kg * d
            hprev = (Hp.data_ptr() + 4 * F_BUILT * d, hp, F_BUILT, 1 if d == 0 else -1)        # h_{t-1} of this direction
            if cell == 0:         # one GEMM: dG^T [x | h_prev] -> [W_ih | W_hh] (torch rows i,f,g,o)
                gcat = torch.empty(4 * F_BUILT, 2 * F_BUILT, **f32)
                gbias = torch.empty(4 * F_BUILT, **f32)
                if bptt_f16:      # quad-layout dG of the f16 BPTT, f16 pieces under the BPTT's gradient scale (and its saturation check)
                    nsq = -(-addr[0] // 4)
                    st.call("dprnn_wgrad_lstm_f16", packedT.data_ptr() + 4 * lib.dprnn_rnn_bwd_gradmax_offset(ndir),
                            dG.data_ptr() + 4 * d * addr[1] * nsq * 4 * kg, _p(X[p]), hprev[0], hp, d, addr[0], addr[1], addr[2],
                            addr[3], addr[4], addr[5], _p(gcat), _p(gbias), 0, _p(ws))
                else:
                    wgrad(gbase, ndir * kg, 4 * F_BUILT, _p(X[p]), F_BUILT, F_BUILT, 0, *hprev, rows, addr[5], addr[1],
                          gcat, 2 * F_BUILT, 1, gbias, None)
                g_ih, g_hh = gcat[:, :F_BUILT].contiguous(), gcat[:, F_BUILT:].contiguous()
                b_ih, b_hh = gbias, gbias.clone()
            elif cell == 1:       # torch rows r,z,n: W_ih pairs with (r, z, n_x), W_hh with (r, z, n_h)
                g_ih = torch.empty(3 * F_BUILT, F_BUILT, **f32)
                g_hh = torch.empty(3 * F_BUILT, F_BUILT, **f32)
                b_ih = torch.empty(3 * F_BUILT, **f32)
                b_hh = torch.empty(3 * F_BUILT, **f32)
                wgrad(gbase, ndir * kg, 3 * F_BUILT, _p(X[p]), F_BUILT, F_BUILT, 0, 0, 0, 0, 0, rows, 1, 1, g_ih, F_BUILT, 1, b_ih, None)
                wgrad(gbase, ndir * kg, 2 * F_BUILT, hprev[0], hprev[1], F_BUILT, hprev[3], 0, 0, 0, 0, rows, addr[5], addr[1],
                      g_hh, F_BUILT, 1, b_hh, None)
                wgrad(gbase + 4 * 3 * F_BUILT, ndir * kg, F_BUILT, hprev[0], hprev[1], F_BUILT, hprev[3], 0, 0, 0, 0, rows, addr[5],
                      addr[1], g_hh[2 * F_BUILT:], F_BUILT, 1, b_hh[2 * F_BUILT:], None)
            else:
                g_ih = torch.empty(F_BUILT, F_BUILT, **f32)
                g_hh = torch.empty(F_BUILT, F_BUILT, **f32)
                b_ih = torch.empty(F_BUILT, **f32)
                wgrad(gbase, ndir * kg, F_BUILT, _p(X[p]), F_BUILT, F_BUILT, 0, 0, 0, 0, 0, rows, 1, 1, g_ih, F_BUILT, 1, b_ih, None)
                wgrad(gbase, ndir * kg, F_BUILT, hprev[0], hprev[1], F_BUILT, hprev[3], 0, 0, 0, 0, rows, addr[5], addr[1],
                      g_hh, F_BUILT, 1, None, None)
                b_hh = b_ih.clone()
            grads[pre + names[0] + suf], grads[pre + names[1] + suf] = g_ih, g_hh
            grads[pre + names[2] + suf], grads[pre + names[3] + suf] = b_ih, b_hh
        del dG
        if ndir == 1:
            dXd[1].zero_()
        have_dx = True
        # the saved activations of this path are no longer needed
        Z[p] = None; H[p] = None; tapes[p] = None; X[p + 1] = None

    # ---- segmentation (dprnn.py:189-201) and bottleneck (dprnn.py:131-136,169)
    dY = torch.empty(B * L, F_BUILT, **f32)
    st.call("dprnn_segment_bwd", _p(dX), _p(dXd[0]) if have_dx else 0, _p(dXd[1]) if have_dx else 0, _p(dY), B, L, K, P, S)
    del dX, dXd, dZ
    bconv = sep.bottleneck[1]
    wb = bconv.weight
    ldb = wb.shape[1]                      # N, or N + embeddings_size for the 'cat' fusion (dprnn_spe.py:88-89)
    if ldb != N and "row_bias" not in fusion:
        raise DprnnError("training path: bottleneck conv wider than the encoder output without a 'cat' fusion")
    A = torch.empty(B * L, N, **f32)
    st.call("dprnn_encoder_norm_rows", _p(wav), _p(enc_w), _p(saved["sc"]), _p(saved["sh"]), _p(A), B, geo.T, L, N, ks, geo.stride)
    gw = torch.empty_like(wb)
    row_scale = fusion.get("row_scale")
    d_rs = None
    if row_scale is None:
        gb = torch.empty_like(bconv.bias)
        wgrad(_p(dY), F_BUILT, F_BUILT, _p(A), N, N, 0, 0, 0, 0, 0, B * L, 1, 1, gw, ldb, 1, gb, None)     # columns [0, N) of the weight
    else:            # 'att': Y = rs*(W a') + b  ->  the conv sees dG = dY*rs, the bias sees dY
        d_rs = torch.empty(B * L, **f32)
        st.call("dprnn_att_scale_bwd", _p(dY), _p(saved["Y"]), _p(bconv.bias), _p(row_scale), _p(dY), _p(d_rs),
                scr.slot("dbb", F_BUILT), B * L)
        gb = to_grad(scr.slot("dbb", F_BUILT), bconv.bias)
        wgrad(_p(dY), F_BUILT, F_BUILT, _p(A), N, N, 0, 0, 0, 0, 0, B * L, 1, 1, gw, ldb, 1, None, None)
    grads["separation.bottleneck.1.weight"], grads["separation.bottleneck.1.bias"] = gw, gb
    dA = A          # reuse
    st.call("dprnn_dgrad", _p(dY), F_BUILT, _p(wb), ldb, F_BUILT, 0, 0, _p(dA), N, N, B * L)
    gamma, beta, eps = _norm_params(sep.bottleneck[0])
    if extras is not None:
        if fusion.get("row_bias") is not None:           # 'cat': per-utterance bias row W[:, N:] @ emb
            st.call("dprnn_utt_colsum", _p(dY), scr.slot("d_row_bias", B * F_BUILT), B, L, F_BUILT)
            extras["d_row_bias"] = to_grad(scr.slot("d_row_bias", B * F_BUILT), torch.empty(B, F_BUILT, **f32))
        mul, add = fusion.get("mul"), fusion.get("add")
        if mul is not None or add is not None:           # 'add' / 'mul' / 'film': a' = an*mul + add
            sc0 = torch.empty(B, N, **f32)
            sh0 = torch.empty_like(sc0)
            st.call("dprnn_bottleneck_prep", _p(stats[0]), _p(gamma), _p(beta), eps, 0, 0, _p(sc0), _p(sh0), B, L, N)
            dAn = torch.empty(B * L, N, **f32)
            d_scores, avg_w, avg_b, ks_a = None, None, None, 0
            if d_rs is not None:             # 'att': back through the nearest upsampling and the softmax over time
                avg_w, avg_b, ks_a = fusion["avg_w"], fusion["avg_b"], fusion["ks_a"]
                d_scores = torch.empty_like(fusion["scores"])
                st.call("dprnn_att_softmax_bwd", _p(d_rs), _p(fusion["scores"]), _p(d_scores), B, L, ks_a)
            st.call("dprnn_fusion_bwd", _p(wav), _p(enc_w), _p(sc0), _p(sh0), _p(dA), _p(mul), _p(dAn),
                    scr.slot("d_mul", B * N) if mul is not None else 0, scr.slot("d_add", B * N) if add is not None else 0,
                    _p(d_scores), _p(avg_w), _p(avg_b), ks_a, B, geo.T, L, N, ks, geo.stride)
            if mul is not None:
                extras["d_mul"] = to_grad(scr.slot("d_mul", B * N), torch.empty(B, N, **f32))
            if add is not None:
                extras["d_add"] = to_grad(scr.slot("d_add", B * N), torch.empty(B, N, **f32))
            dA = dAn
    a_red0, a_dgb0 = scr.slot("red_bn", 2 * B), scr.slot("dgb_bn", 2 * N)
    st.call("dprnn_encoder_norm_bwd_stats", _p(wav), _p(enc_w), _p(dA), _p(stats[0]), _p(gamma), eps, a_red0, a_dgb0,
            B, geo.T, L, N, ks, geo.stride)
    _count_scale_red(scr, "red_bn", sep.bottleneck[0], st, B)
    gname, bname = ("gamma", "beta") if hasattr(sep.bottleneck[0], "gamma") else ("weight", "bias")
    grads[f"separation.bottleneck.0.{gname}"] = to_grad(a_dgb0, gamma)
    grads[f"separation.bottleneck.0.{bname}"] = to_grad(a_dgb0 + 8 * N, beta)
    st.call("dprnn_encoder_bwd", _p(wav), _p(enc_w), _p(dA), _p(stats[0]), _p(gamma), eps, a_red0, _p(dEnc),
            scr.slot("dWe", N * ks), B, geo.T, L, N, ks, geo.stride)
    if extras is None:
        grads["encoder.conv1d.weight"] = to_grad(scr.slot("dWe", N * ks), enc_w)
    return grads


def _auto_groups(n: int, geo: engine.Geometry) -> int:
    """Stream groups for n utterances, from measurements on MI355X (profiles/r01_train_step.txt).  The recurrence
    launches are grids of 32-sequence workgroups on 2 x CUs slots and every workgroup lives for a whole launch, so a
    launch that fills its last round of slots badly (n = 24, 40, 47: 57-73 % on one of the two paths) loses that share
    of the chip; cut into ~16-utterance pipelines on separate streams the launches of different groups fill each other's
    gaps (n = 48: 542 -> 495 ms per step).  A launch that fills its rounds (n = 32: 76 % / 98 %) is best left whole
    (344 vs 351 ms), and groups below 12 utterances lose more per kernel than the overlap wins (n = 16: 194 vs 254 ms)."""
    slots = 2 * load_library().dprnn_num_cus()

    def fill(nseq):
        wgs = 2 * -(-nseq // 32)
        return wgs / (slots * -(-wgs // slots))
    if min(fill(n * geo.S), fill(n * geo.K)) >= 0.75:
        return 1
    return max(1, min(3, n // 12))


def _group_parts(n: int, geo: engine.Geometry = None):
    if TRAIN_STREAM_GROUPS > 0 or geo is None:
        g = max(1, min(max(1, TRAIN_STREAM_GROUPS), n // MIN_TRAIN_GROUP))
    else:
        g = _auto_groups(n, geo)
    cuts = [round(i * n / g) for i in range(g + 1)]
    return [(cuts[i], cuts[i + 1]) for i in range(g)]


def _run_groups(dev, parts, fn):
    """fn(b0, b1) for every part, each on its own persistent side stream (the same stream in forward and backward, so
    the saved activations never cross streams); the caller's stream waits for all of them."""
    if len(parts) == 1:
        return [fn(*parts[0])]
    pool = _SIDE_STREAMS.setdefault(dev, [])
    while len(pool) < len(parts):
        pool.append(torch.cuda.Stream(device=dev))
    main = torch.cuda.current_stream(dev)
    fork = torch.cuda.Event()
    fork.record(main)
    results = []
    for (b0, b1), side in zip(parts, pool):
        side.wait_event(fork)
        with torch.cuda.stream(side):
            results.append(fn(b0, b1))
    for side in pool[: len(parts)]:
        done = torch.cuda.Event()
        done.record(side)
        main.wait_event(done)
    return results


def _to_main(t: torch.Tensor, dev):
    t.record_stream(torch.cuda.current_stream(dev))       # produced on a side stream, consumed (and freed) on the caller's
    return t


def _forward_groups(model, wav, keep: bool = True):
    dev = wav.device
    cfg = model.cfg
    geo = engine.make_geometry(1, wav.shape[1], cfg["input_size"], cfg["chunk_length"], cfg["hop_length"],
                               cfg["kernel_size"], cfg["stride"])
    parts = _group_parts(wav.shape[0], geo)
    res = _run_groups(dev, parts, lambda b0, b1: forward_train(model, wav[b0:b1].contiguous(), keep=keep))
    outs = [_to_main(r[0], dev) for r in res]
    return (outs[0] if len(outs) == 1 else torch.cat(outs)), (parts, [r[1] for r in res])


def _backward_groups(model, saved, d_out):
    dev = d_out.device
    parts, saved_list = saved
    res = _run_groups(dev, parts, lambda b0, b1: backward(model, saved_list[parts.index((b0, b1))], d_out[b0:b1]))
    saved_list.clear()
    grads = res[0]
    for k in grads:
        _to_main(grads[k], dev)
    for other in res[1:]:
        for k in grads:
            grads[k] += _to_main(other[k], dev)
    return grads


def _accumulate(grads: Dict[str, torch.Tensor], g: Dict[str, torch.Tensor]) -> None:
    """grads[k] += g[k] for every parameter in a few multi-tensor launches instead of one per parameter (170 of them per micro-batch)."""
    keys = list(g)
    torch._foreach_add_([grads[k] for k in keys], [g[k] for k in keys])


class _TasNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, wav, names, *params):
        ctx.model, ctx.names = model, names
        B = wav.shape[0]
        cfg = model.cfg
        geo = engine.make_geometry(1, wav.shape[1], cfg["input_size"], cfg["chunk_length"], cfg["hop_length"],
                                   cfg["kernel_size"], cfg["stride"])
        ndirs = [2 if r.rnn.bidirectional else 1 for r, _, _, _ in _paths(model.separation)]
        per_utt = saved_bytes_per_utt(geo, len(ndirs), ndirs)
        mb_max = max(1, save_budget_bytes(wav.device) // per_utt)
        ctx.done = False
        ctx.wav_all = wav               # 4 bytes per sample: what a saturated backward needs to run again from the waveform
        ctx.fwd_exact = False
        ctx.modes = train_modes()[:2]   # autograd runs backward() on another thread: the modes of THIS forward travel with it

        def run():
            if mb_max >= B:
                out, saved = _forward_groups(model, wav)
                return out, saved, None, B
            # too large to keep: micro-batches.  The first one runs the training forward and keeps its activations; the
            # rest run the SAME training kernels now without keeping anything (forward_train(keep=False): the outputs the
            # loss sees and the gradients the backward recomputes come from one arithmetic, whatever engine.GEMM_MODE
            # says about the inference path) and are re-run with the tape inside the backward, one micro-batch at a time.
            n_mb = -(-B // mb_max)
            mb = mb_max if KEEP_MAX_FIRST else -(-B // n_mb)
            out0, saved = _forward_groups(model, wav[:mb])
            outs = [out0]
            for b0 in range(mb, B, mb):
                outs.append(_forward_groups(model, wav[b0:b0 + mb], keep=False)[0])
            return torch.cat(outs), saved, wav, mb

        def run_exact():                # the f16-pieces forward met a value beyond its range: the exact fp32 kernels, and the
            ctx.fwd_exact = True        # backward recomputes in them too
            return _exact(run, fwd=True, bwd=False)()

        out, ctx.saved, ctx.wav, ctx.mb = engine._guarded(wav.device, run, run_exact, "DPRNNTasNet training forward",
                                                          active=train_modes()[0] == 3)
        return out

    @staticmethod
    def backward(ctx, d_out):
        with torch.cuda.device(d_out.device), use_train_modes(*ctx.modes):
            if ctx.done:
                raise DprnnError("DPRNN MI355X training path: backward() was called a second time on the same forward; the "
                                 "saved activations (tape) are released by the first backward (retain_graph is not supported "
                                 "-- run the forward again)")
            ctx.done = True
            d_out = d_out.contiguous()
            fwd_only = (lambda fn: _exact(fn, fwd=True, bwd=False)) if ctx.fwd_exact else (lambda fn: fn)
            first = fwd_only(lambda: _TasNetFunction._backward(ctx, d_out))
            # a BPTT / weight-gradient launch in f16 pieces met a gradient beyond its scale's range (more than 2^10 above the
            # largest incoming one, or Inf / NaN): everything again from the waveform with the exact fp32 backward kernels
            again = _exact(lambda: _TasNetFunction._recompute(ctx, d_out), fwd=ctx.fwd_exact, bwd=True)
            grads = engine._guarded(d_out.device, first, again, "DPRNNTasNet training backward", active=train_modes()[1] == 3)
            ctx.saved = ctx.wav = ctx.wav_all = None
            return (None, None, None) + tuple(grads[n] for n in ctx.names)

    @staticmethod
    def _backward(ctx, d_out):
        model = ctx.model
        mb = ctx.mb
        grads = _backward_groups(model, ctx.saved, d_out[:mb])
        ctx.saved = None
        if ctx.wav is not None:
            wav = ctx.wav
            for b0 in range(mb, wav.shape[0], mb):
                _, saved = _forward_groups(model, wav[b0:b0 + mb])
                g = _backward_groups(model, saved, d_out[b0:b0 + mb])
                del saved
                _accumulate(grads, g)
        return grads

    @staticmethod
    def _recompute(ctx, d_out):
        model, wav, mb = ctx.model, ctx.wav_all, ctx.mb
        ctx.saved = None
        grads = None
        for b0 in range(0, wav.shape[0], mb):
            _, saved = _forward_groups(model, wav[b0:b0 + mb])
            g = _backward_groups(model, saved, d_out[b0:b0 + mb])
            del saved
            if grads is None:
                grads = g
            else:
                _accumulate(grads, g)
        return grads


def tasnet_forward_autograd(model, wav: torch.Tensor) -> torch.Tensor:
    """model(wav) under autograd: differentiable w.r.t. every parameter (not w.r.t. the waveform)."""
    from . import padded
    wav = engine.check_input(wav)
    if wav.requires_grad:
        raise NotImplementedError("DPRNN MI355X training path: gradients w.r.t. the input waveform are not computed")
    named = [(n, p) for n, p in model.named_parameters()]
    for n, p in named:
        _param(p, n, wav.device)
    if padded.needs_twin(model.cfg):
        # A narrower model trains on its zero-padded twin (padded.py): the kernels run the twin, whose parameters are the narrow ones
        # placed into zeros -- a differentiable function of them, so autograd hands every narrow parameter the block of the twin's
        # gradient that belongs to it and drops the padding's.  (Wider than the build: padded.Twin raises.)
        run, tw = padded.twin_of(model)
        check_supported_train(run.cfg)
        wide = padded.placed_parameters(model, tw)
        with torch.cuda.device(wav.device):
            return _TasNetFunction.apply(run, wav, tuple(n for n, _ in named), *[wide[n] for n, _ in named])
    check_supported_train(model.cfg)
    with torch.cuda.device(wav.device):
        return _TasNetFunction.apply(model, wav, tuple(n for n, _ in named), *[p for _, p in named])


# ------------------------------------------------------------------------------------------------ loss
class _PitSiSdrFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, est, tgt):
        if not (est.is_cuda and tgt.is_cuda):
            raise DprnnError("PITSiSdrLoss runs on the HIP device only")
        est = est.contiguous().float()
        tgt = tgt.contiguous().float()
        B, n, T = est.shape
        if tgt.shape != est.shape:
            raise DprnnError(f"estimate {tuple(est.shape)} / target {tuple(tgt.shape)} shapes differ")
        dev = est.device
        with torch.cuda.device(dev):
            st = Stage(dev)
            mom = torch.empty(B, n, n, 5, device=dev, dtype=torch.float64)
            pw = torch.empty(B, n, n, device=dev, dtype=torch.float64)
            loss_b = torch.empty(B, device=dev, dtype=torch.float64)
            perm = torch.empty(B, n, device=dev, dtype=torch.int32)
            coef = torch.empty(B, n, 4, device=dev, dtype=torch.float64)
            st.call("dprnn_pit_sisdr_fwd", _p(est), _p(tgt), _p(mom), _p(pw), _p(loss_b), _p(perm), _p(coef), B, n, T)
        ctx.save_for_backward(est, tgt, perm, coef)
        ctx.mark_non_differentiable(perm, pw)
        return loss_b.mean().float(), perm, pw

    @staticmethod
    def backward(ctx, g_loss, _g_perm, _g_pw):
        est, tgt, perm, coef = ctx.saved_tensors
        B, n, T = est.shape
        d_est = torch.empty_like(est)
        with torch.cuda.device(est.device):
            st = Stage(est.device)
            up = g_loss.detach().float().contiguous()
            st.call("dprnn_pit_sisdr_bwd", _p(est), _p(tgt), _p(perm), _p(coef), _p(up), _p(d_est), B, n, T)
        return d_est, None


class PITSiSdrLoss(torch.nn.Module):
    """``PITLossWrapper(pairwise_neg_sisdr, pit_from='pw_mtx')`` of asteroid==0.7.0 as the reference uses it
    (src/trainers/trainer.py:39,104): mean over the batch of the minimum over source permutations of the mean negative
    SI-SDR; ``return_est=True`` also returns the estimates reordered to the targets.  Forward and backward are HIP kernels
    (fp64 moments, dprnn_pit_sisdr_fwd / _bwd)."""

    def forward(self, est_targets, targets, return_est: bool = False):
        loss, perm, _ = _PitSiSdrFunction.apply(est_targets, targets)
        if not return_est:
            return loss
        inv = torch.argsort(perm.long(), dim=1)
        reordered = torch.gather(est_targets, 1, inv.unsqueeze(-1).expand(-1, -1, est_targets.shape[-1]))
        return loss, reordered


# ------------------------------------------------------------------------------------------------ optimizer
class FusedAdam(torch.optim.Optimizer):
    """``torch.optim.Adam`` (the optimizer the reference configures, scripts/train/config_bss.yaml:30-33: lr, weight_decay;
    amsgrad off) with ``clip_grad_norm_`` folded in: two HIP launches per step for the whole model
    (dprnn_multi_sumsq, dprnn_adam_multi) instead of ~10 small launches per parameter.

    ``step(clip_norm=...)`` applies the clip coefficient to the gradients on the fly (same update as clipping ``.grad``
    first, trainer.py:115-118).  ``state_dict()`` has torch.optim.Adam's layout (per-parameter ``step``, ``exp_avg``,
    ``exp_avg_sq``), so checkpoints move between the two in both directions."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("FusedAdam: amsgrad is not implemented (the reference configs do not use it)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))

    @torch.no_grad()
    def step(self, closure=None, clip_norm: float = 0.0):
        if closure is not None:
            raise NotImplementedError("FusedAdam.step does not take a closure")
        import numpy as np
        work, dev = [], None
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            if not ps[0].is_cuda:
                raise DprnnError("FusedAdam runs on the HIP device only")
            if dev is not None and ps[0].device != dev:
                raise DprnnError("FusedAdam: all parameters must live on one device")
            dev = ps[0].device
            steps = set()
            rec = np.zeros((len(ps), 5), dtype=np.int64)
            for i, p in enumerate(ps):
                # .grad is read in place through its pointer, so it must be a contiguous float32 tensor itself (a
                # .contiguous() temporary could be freed and its block reused before the kernels run)
                if (p.dtype != torch.float32 or not p.is_contiguous() or p.grad.dtype != torch.float32
                        or not p.grad.is_contiguous()):
                    raise DprnnError("FusedAdam expects contiguous float32 parameters and gradients")
                st_ = self.state[p]
                if len(st_) == 0:
                    st_["step"] = torch.zeros((), dtype=torch.float32)
                    st_["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st_["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st_["step"] += 1
                steps.add(int(st_["step"].item()))
                rec[i] = (p.data_ptr(), p.grad.data_ptr(), st_["exp_avg"].data_ptr(), st_["exp_avg_sq"].data_ptr(), p.numel())
            if len(steps) != 1:
                raise DprnnError("FusedAdam: parameters of one group must share the step count")
            work.append((group, len(ps), rec, steps.pop()))
        if not work:
            return None
        with torch.cuda.device(dev):
            st = Stage(dev)
            tables = [torch.from_numpy(rec).to(dev) for _, _, rec, _ in work]
            # clip_grad_norm_ clips by the GLOBAL norm over every parameter (trainer.py:115-116): the sum of squares is
            # accumulated over all groups before the first update is launched
            sumsq = torch.zeros(1, device=dev, dtype=torch.float64)
            if clip_norm and clip_norm > 0:
                for (group, n, _, _), table in zip(work, tables):
                    st.call("dprnn_multi_sumsq", _p(table), n, _p(sumsq))
            for (group, n, _, step), table in zip(work, tables):
                b1, b2 = group["betas"]
                st.call("dprnn_adam_multi", _p(table), n, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                        float(group["weight_decay"]), step, float(clip_norm or 0.0), _p(sumsq))
        # the kernels wrote the parameters through their pointers: tell torch (version counters), as an in-place torch update would --
        # the zero-padded twin of a narrow model (padded.Twin.sync) and anything else that watches `_version` reads them again
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    torch.autograd.graph.increment_version(p)
        return None
