"""Separator-level and sub-module forwards on the HIP kernels (DPRNN.forward, DPRNNSpe._auxiliary / _fusion, SingleRNN, DPRNNBlock, Encoder,
Decoder, ResBlock, norms, 1x1 convs, Linears): what code that drives the separator's pieces itself calls -- the reference's
DPRNNRawNet (dprnn_rawnet.py:79-105).  Part of the engine (see engine.py); inference only."""
from __future__ import annotations


import torch

from . import engine as E
from ._lib import DprnnError


# ------------------------------------------------------------------- separation-level entry points (materialised frames)
# The TasNet wrappers above never materialise the encoder output.  Code that subclasses the SEPARATOR modules and calls
# their pieces on [B, C, L] tensors -- /root/reference/src/models/dprnn_rawnet.py:79-94 does: self._auxiliary,
# self.bottleneck[i], self._fusion, self._dprnn_process -- gets the same HIP kernels through these functions
# (inference only: the backward kernels are wired to the TasNet wrappers).
def _no_grad_only(what: str, *tensors, module=None) -> None:
    if torch.is_grad_enabled() and (any(t.requires_grad for t in tensors if isinstance(t, torch.Tensor)) or
                                    (module is not None and any(p.requires_grad for p in module.parameters()))):
        raise NotImplementedError(
            f"{what}: the separation-level MI355X entry points are inference-only (call them under torch.no_grad()); "
            "training goes through DPRNNTasNet / DPRNNSpeTasNet / DPRNNSpeIRATasNet.forward, which own the backward kernels")


def _check_frames(x: torch.Tensor, channels: int, what: str) -> torch.Tensor:
    if not isinstance(x, torch.Tensor) or x.dim() != 3 or x.shape[1] != channels:
        raise DprnnError(f"{what} must be a [batch, {channels}, frames] tensor")
    if not x.is_cuda:
        raise DprnnError(f"{what} is on {x.device}: the DPRNN MI355X path runs only on a HIP device (no CPU fallback)")
    if x.dtype != torch.float32:
        raise DprnnError(f"{what} must be float32 (got {x.dtype})")
    return x


def _separator_masks_impl(sep, frames: torch.Tensor) -> torch.Tensor:
    """DPRNN._dprnn_process (dprnn.py:171-186 == dprnn_spe.py:231-248): bottleneck output [B, F, L] -> masks [B, 2, N, L]."""
    cfg = dict(input_size=sep.input_size, feature_size=sep.feature_size, hidden_size=sep.dprnn_blocks[0].intra_rnn.hidden_size,
               rnn_type=sep.dprnn_blocks[0].intra_rnn.rnn.mode, kernel_size=1, norm_type=sep.norm_type,
               activation_type=sep.activation_type, hop_length=sep.hop_length, chunk_length=sep.chunk_length)
    E.check_supported(cfg)
    frames = _check_frames(frames, E.F_BUILT, "bottleneck output")
    _no_grad_only("_dprnn_process", frames, module=sep)
    dev = frames.device
    B, _, L = frames.shape
    N = sep.input_size
    with torch.cuda.device(dev), E._scale_record(sep, dev) as rec:
        st = E.Stage(dev)
        Y = frames.transpose(1, 2).contiguous()                       # channels-last frames [B][L][F] (layout plumbing)
        # the mask kernel also forms decoder taps of mask * encoder(wav); here there is no waveform: a 1-tap all-zero
        # encoder / decoder over a zero signal of L samples keeps its addressing valid, and the taps are discarded
        geo = E.Geometry(B, L, L, sep.chunk_length, sep.hop_length, (L + sep.chunk_length) // sep.hop_length + 1, 1, 1, N)
        zero_w = torch.zeros(N, device=dev, dtype=torch.float32)
        zero_x = torch.zeros(B, L, device=dev, dtype=torch.float32)
        stats = torch.zeros(1 + 2 * len(sep.dprnn_blocks), B, 2, device=dev, dtype=torch.float64)
        masks = torch.empty(B, L, 2, N, device=dev, dtype=torch.float32)         # rows (b, l, speaker)
        E.dprnn_process(st, sep, Y, zero_x, zero_w, zero_w, geo, stats, sep.activation_type == "relu", masks_out=masks)
        E._scale_end(st, rec)
        return masks.permute(0, 2, 3, 1).contiguous()                 # [B, 2, N, L] as dprnn.py:185


def _rows_affine(st: Stage, x_cl, norm, B: int, L: int, N: int):
    """Per-utterance affine (sc, sh) of GroupNorm(1, N) / GlobLN over materialised rows [B*L][N]."""
    dev = x_cl.device
    stats = torch.zeros(B, 2, device=dev, dtype=torch.float64)
    st.call("dprnn_rows_stats", E._p(x_cl), E._p(stats), B, L, N)
    E._count_scale(stats, norm, st, B)
    gamma, beta, eps = E._norm_params(norm)
    sc = torch.empty(B, N, device=dev, dtype=torch.float32)
    sh = torch.empty_like(sc)
    st.call("dprnn_bottleneck_prep", E._p(stats), E._p(E._param(gamma, "norm.weight", dev)), E._p(E._param(beta, "norm.bias", dev)),
            eps, 0, 0, E._p(sc), E._p(sh), B, L, N)
    return sc, sh


def _separator_forward_impl(sep, enc: torch.Tensor) -> torch.Tensor:
    """DPRNN.forward (dprnn.py:166-187) on a materialised encoder output [B, N, L] -> masks [B, 2, N, L]."""
    enc = _check_frames(enc, sep.input_size, "separator input")
    _no_grad_only("DPRNN.forward", enc, module=sep)
    if sep.input_size != E.N_BUILT or len(sep.bottleneck) != 2:
        raise NotImplementedError(f"DPRNN MI355X path: input_size must be {E.N_BUILT} and norm_type 'ln' or 'gLN'")
    dev = enc.device
    B, N, L = enc.shape
    with torch.cuda.device(dev):
        st = E.Stage(dev)
        x_cl = enc.transpose(1, 2).contiguous().view(B * L, N)
        sc, sh = _rows_affine(st, x_cl, sep.bottleneck[0], B, L, N)
        conv = sep.bottleneck[1]
        Y = E._pointwise(st, x_cl, B * L, N, conv.weight, conv.bias, E.F_BUILT, in_scale=sc, in_shift=sh, aff_stride=N,
                       rows_per_utt=L)
    return E.separator_masks(sep, Y.view(B, L, E.F_BUILT).transpose(1, 2))


def _auxiliary_embedding_impl(sep, aux: torch.Tensor, aux_len) -> torch.Tensor:
    """DPRNNSpe._auxiliary (dprnn_spe.py:156-163): encoder output of the reference utterance [B, N, La] -> [B, E]."""
    aux = _check_frames(aux, sep.input_size, "aux")
    _no_grad_only("_auxiliary", aux, module=sep.spk_encoder)
    dev = aux.device
    B, N, La = aux.shape
    se = sep.spk_encoder
    if N != E.N_BUILT or se[1].out_channels != 128 or se[5].in_channels != 256 or se[5].out_channels != 128:
        raise NotImplementedError("DPRNN-Spe MI355X path: speaker encoder must be input_size 64, O=128, P=256, embeddings 128")
    with torch.cuda.device(dev):
        st = E.Stage(dev)
        x_cl = aux.transpose(1, 2).contiguous().view(B * La, N)
        sc, sh = _rows_affine(st, x_cl, se[0], B, La, N)
        A0 = E._pointwise(st, x_cl, B * La, N, se[1].weight, se[1].bias, 128, in_scale=sc, in_shift=sh, aff_stride=N,
                        rows_per_utt=La)
        return E.speaker_embedding(st, sep, A0, B, La, E._aux_len_tensor(aux_len, B, dev), sep.kernel_size)


# ---- sub-module forwards (inference only).  No shipped caller uses them -- the TasNet wrappers fuse everything from the
# waveform -- but the reference's classes have them (dprnn.py:33-37,79-99, encoder_decoder.py:25-33,40-49, dprnn_spe.py:31-42), so
# code that drives a single block / RNN / encoder / decoder / ResBlock on [B, C, ...] tensors gets the same HIP kernels.  Layout
# changes between the reference's channels-first tensors and the kernels' channels-last rows are torch copies (plumbing).
def single_rnn_forward(rnn, x: torch.Tensor) -> torch.Tensor:
    """SingleRNN.forward (dprnn.py:33-37): [n, T, F] batch-first -> [n, T, ndir*H], zero initial state."""
    if rnn.rnn.input_size != E.F_BUILT or rnn.hidden_size != E.F_BUILT:          # off the fused kernels' build: the any-width kernels
        from . import wide
        return wide.single_rnn_forward(rnn, x)
    if not isinstance(x, torch.Tensor) or x.dim() != 3 or x.shape[2] != E.F_BUILT:
        raise DprnnError(f"SingleRNN input must be [sequences, steps, {E.F_BUILT}] with hidden_size {E.F_BUILT}")
    if not x.is_cuda or x.dtype != torch.float32:
        raise DprnnError("SingleRNN runs on float32 HIP tensors only (no CPU fallback)")
    _no_grad_only("SingleRNN.forward", x, module=rnn)
    dev = x.device
    n, T, _ = x.shape
    if rnn.rnn.mode not in E.CELL_ID:
        raise NotImplementedError(f"rnn_type {rnn.rnn.mode!r}")

    def run():
        with torch.cuda.device(dev), E._scale_record(rnn, dev) as rec:
            st = E.Stage(dev)
            E._set_slot(st.lib, E.SLOT_PATH0)
            xc = x.contiguous()
            ndir, fw, bw = E._rnn_tensors(rnn, dev)
            cell = E.CELL_ID[rnn.rnn.mode]
            mode = E.current_mode() if cell == 0 else 0
            if mode == 1 and n < 64:
                mode = 0
            packed = torch.empty(2 * E.PACK_PER_DIR_MAX, device=dev, dtype=torch.float32)
            bsum = torch.empty(2 * 4 * E.F_BUILT, device=dev, dtype=torch.float32)
            st.call("dprnn_rnn_pack_ex", mode, cell, *[E._p(t) for t in fw], *[E._p(t) for t in bw], E._p(packed), E._p(bsum))
            H = torch.empty(n, T, ndir * E.F_BUILT, device=dev, dtype=torch.float32)
            tile = 0 if mode == 0 else E.split_tile(n, ndir, st.lib.dprnn_num_cus(), 1, mode)
            if cell != 0 and tile > 32:
                tile = 32
            st.call("dprnn_rnn_fwd_ex", mode, cell, E._p(xc), 0, 0, 0, 0, 0.0, 1, 1, 0, E._p(packed), E._p(bsum), E._p(H), n, T, n, 0, T, 1, ndir, tile)
            E._scale_end(st, rec)
            return H
    return E._guarded(dev, run, E._in_mode(1, run), "SingleRNN.forward", finite_inputs=[x])


def block_forward(blk, x: torch.Tensor) -> torch.Tensor:
    """DPRNNBlock.forward (dprnn.py:79-99): [B, F, K, S] -> [B, F, K, S]."""
    if blk.intra_rnn.rnn.input_size != E.F_BUILT or blk.intra_rnn.hidden_size != E.F_BUILT:      # off the build: the any-width kernels
        from . import wide
        return wide.block_forward(blk, x)
    if not isinstance(x, torch.Tensor) or x.dim() != 4 or x.shape[1] != E.F_BUILT:
        raise DprnnError(f"DPRNNBlock input must be [batch, {E.F_BUILT}, chunk_length, chunks]")
    if not x.is_cuda or x.dtype != torch.float32:
        raise DprnnError("DPRNNBlock runs on float32 HIP tensors only (no CPU fallback)")
    if not hasattr(blk, "intra_norm"):
        raise DprnnError("DPRNNBlock without norms (norm_type neither 'ln' nor 'gLN'): the reference's forward fails too")
    _no_grad_only("DPRNNBlock.forward", x, module=blk)
    dev = x.device
    B, _, K, S = x.shape

    def run():
        with torch.cuda.device(dev), E._scale_record(blk, dev) as rec:
            st = E.Stage(dev)
            import types
            geo = E.Geometry(B, 0, 0, K, 1, S, 1, 1, E.N_BUILT)
            rows = B * S * K
            g = types.SimpleNamespace(st=st, geo=geo, b0=0, b1=B, pending=None, n_concurrent=1, mode=0,
                                      X=torch.empty(rows * E.F_BUILT, device=dev, dtype=torch.float32), X2=None,
                                      Z=torch.empty(rows * E.F_BUILT, device=dev, dtype=torch.float32),
                                      H=torch.empty(rows * 2 * E.F_BUILT, device=dev, dtype=torch.float32))
            g.X.view(B, S, K, E.F_BUILT).copy_(x.permute(0, 3, 2, 1))          # slab rows (b, s, k)
            mode = E.effective_gemm_mode(B, [blk])
            prev = st.lib.dprnn_get_gemm_mode()
            st.lib.dprnn_set_gemm_mode(mode)
            g.mode = mode
            try:
                stats = torch.zeros(2, B, 2, device=dev, dtype=torch.float64)
                packed = torch.empty(2, 2 * E.PACK_PER_DIR_MAX, device=dev, dtype=torch.float32)
                bsum = torch.empty(2, 2 * 4 * E.F_BUILT, device=dev, dtype=torch.float32)
                for j, (rnn, lin, norm, intra) in enumerate(((blk.intra_rnn, blk.intra_linear, blk.intra_norm, True),
                                                             (blk.inter_rnn, blk.inter_linear, blk.inter_norm, False))):
                    _, fw, bw = E._rnn_tensors(rnn, dev)
                    E._set_slot(st.lib, E.path_slot(0, j == 1))
                    st.call("dprnn_rnn_pack_ex", mode, E.CELL_ID[rnn.rnn.mode], *[E._p(t) for t in fw], *[E._p(t) for t in bw],
                            E._p(packed[j]), E._p(bsum[j]))
                    E._run_path(g, rnn, lin, norm, stats[j], packed[j], bsum[j], intra, None, False, E.path_slot(0, j == 1))
            finally:
                st.lib.dprnn_set_gemm_mode(prev)
            E._scale_end(st, rec)
            return g.X.view(B, S, K, E.F_BUILT).permute(0, 3, 2, 1).contiguous()
    return E._guarded(dev, run, E._in_mode(1, run), "DPRNNBlock.forward", finite_inputs=[x])


def encoder_forward(enc, wav: torch.Tensor) -> torch.Tensor:
    """Encoder.forward (encoder_decoder.py:25-33): [B, T] -> relu(conv1d) [B, N, L]."""
    wav = E.check_input(wav)
    dev = wav.device
    w = E._param(enc.conv1d.weight, "encoder.conv1d.weight", dev)
    N, _, ks = w.shape
    stride = enc.conv1d.stride[0]
    if N != E.N_BUILT or ks > E.KS_MAX:                       # off the fused kernels' build: the any-width encoder kernel
        from . import wide
        return wide.encoder_forward(enc, wav)
    _no_grad_only("Encoder.forward", wav, module=enc)
    with torch.cuda.device(dev):
        st = E.Stage(dev)
        geo = E.make_geometry(wav.shape[0], wav.shape[1], N, 1, 1, ks, stride)
        one = torch.ones(geo.B, N, device=dev, dtype=torch.float32)
        zero = torch.zeros_like(one)
        A = torch.empty(geo.B, geo.L, N, device=dev, dtype=torch.float32)
        st.call("dprnn_encoder_norm_rows", E._p(wav), E._p(w), E._p(one), E._p(zero), E._p(A), geo.B, geo.T, geo.L, N, ks, stride)
        return A.transpose(1, 2).contiguous()


def decoder_forward(dec, x: torch.Tensor) -> torch.Tensor:
    """Decoder.forward (encoder_decoder.py:40-49): [B, N, L] -> [B, T] with the reference's squeeze rule."""
    if not isinstance(x, torch.Tensor) or x.dim() != 3:
        raise DprnnError("Decoder input must be [batch, channels, frames]")   # (the reference also accepts 2-D; its wrappers never pass it)
    if not x.is_cuda or x.dtype != torch.float32:
        raise DprnnError("Decoder runs on float32 HIP tensors only (no CPU fallback)")
    _no_grad_only("Decoder.forward", x, module=dec)
    dev = x.device
    w = E._param(dec.weight, "decoder.weight", dev)
    N, oc, ks = w.shape
    stride = dec.stride[0]
    B, C, L = x.shape
    if C != N or oc != 1:
        raise DprnnError(f"Decoder expects {N} channels and one output channel")
    with torch.cuda.device(dev):
        st = E.Stage(dev)
        rows = x.transpose(1, 2).contiguous().view(B * L, N)
        wt = w[:, 0, :].t().contiguous()                                     # [ks][N]: taps[row][j] = sum_n Wd[n][j] x[row][n]
        taps = torch.zeros(B, 2, L, ks, device=dev, dtype=torch.float32)     # dprnn_decoder_fwd reads [B][2 speakers][L][ks]
        taps[:, 0] = E._linear_small(st, wt, None, rows).view(B, L, ks)
        geo = E.Geometry(B, 0, L, 1, 1, 1, ks, stride, N)
        out = E.decode(st, taps, geo, 1)[:, 0, :]
    from .encoder_decoder import squeeze_like_reference
    return squeeze_like_reference(out)


def resblock_forward(blk, x: torch.Tensor) -> torch.Tensor:
    """ResBlock.forward (dprnn_spe.py:31-42): [B, C_in, T] -> [B, C_out, T // 3]."""
    if not isinstance(x, torch.Tensor) or x.dim() != 3:
        raise DprnnError("ResBlock input must be [batch, channels, frames]")
    if not x.is_cuda or x.dtype != torch.float32:
        raise DprnnError("ResBlock runs on float32 HIP tensors only (no CPU fallback)")
    _no_grad_only("ResBlock.forward", x, module=blk)
    dev = x.device
    B, cin, T = x.shape
    cout = blk.conv1.out_channels
    if blk.conv1.in_channels != cin or cin not in (128, 256) or cout not in (128, 256):
        raise NotImplementedError("ResBlock MI355X path: 128 or 256 channels (the reference's speaker encoder: O=128, P=256)")
    with torch.cuda.device(dev):
        st = E.Stage(dev)
        rows = x.transpose(1, 2).contiguous().view(B * T, cin)
        out, _ = E._resblock(st, blk, rows, B, T, cin, cout)
        return out.transpose(1, 2).contiguous()


# ---- the per-frame glue of the separator-level surfaces as HIP kernels (round 4): a subclass of the reference's DPRNNSpe that
# calls bottleneck[0], _fusion, bottleneck[1], pred_linear itself (src/models/dprnn_rawnet.py:79-105) gets these through the
# modules of device_modules.py -- no MIOpen / rocBLAS kernel behind a surface INTEGRATION.md calls "on the HIP kernels".
def _device_f32(x, what: str):
    if not isinstance(x, torch.Tensor) or not x.is_cuda:
        raise DprnnError(f"{what}: the DPRNN MI355X path runs only on a HIP device (no CPU fallback)")
    if x.dtype != torch.float32:
        raise DprnnError(f"{what} must be float32 (got {x.dtype})")
    return x


def norm_forward(norm, x: torch.Tensor) -> torch.Tensor:
    """nn.GroupNorm(1, C) / norms.GlobLN on [B, C, ...] (dprnn.py:131-136, norms.py:27-31): per-utterance statistics in fp64
    (dprnn_rows_stats), affine from dprnn_bottleneck_prep, applied by dprnn_rows_affine_fwd."""
    x = _device_f32(x, "norm input")
    _no_grad_only(type(norm).__name__ + ".forward", x, module=norm)
    if x.dim() < 3:
        raise DprnnError("norm input must be [batch, channels, ...]")
    B, C = x.shape[:2]
    if C % 4:
        raise NotImplementedError("norm over a channel count that is not a multiple of 4")
    dev = x.device
    with torch.cuda.device(dev):
        st = E.Stage(dev)
        rows = x.reshape(B, C, -1).transpose(1, 2).contiguous()                # channels-last rows [B][R][C] (layout plumbing)
        R = rows.shape[1]
        sc, sh = _rows_affine(st, rows, norm, B, R, C)
        out = torch.empty_like(rows)
        st.call("dprnn_rows_affine_fwd", E._p(rows), E._p(sc), E._p(sh), 0, E._p(out), B * R, C, R)
        return out.transpose(1, 2).reshape(x.shape).contiguous()


def conv1x1_forward(conv, x: torch.Tensor) -> torch.Tensor:
    """nn.Conv1d(C_in, C_out, 1) on [B, C_in, L] as a row GEMM (dprnn_pointwise_fwd).  C_in up to 256 (zero-padded to 64 / 128 /
    256: the 'cat' fusion's 192 channels), C_out a multiple of 128."""
    x = _device_f32(x, "conv input")
    _no_grad_only("Conv1d.forward", x, module=conv)
    if x.dim() != 3 or conv.kernel_size != (1,) or conv.stride != (1,) or conv.groups != 1 or conv.padding != (0,):
        raise NotImplementedError("Conv1d MI355X path: 1x1 convolutions over [batch, channels, frames]")
    B, cin, L = x.shape
    cout = conv.out_channels
    if cin != conv.in_channels or cin > 256 or cout % 128:
        raise NotImplementedError(f"Conv1d MI355X path: in_channels <= 256 and out_channels a multiple of 128 (got {cin} -> {cout})")
    dev = x.device
    kdim = 64 if cin <= 64 else 128 if cin <= 128 else 256
    with torch.cuda.device(dev):
        st = E.Stage(dev)
        w = E._param(conv.weight, "conv.weight", dev).reshape(cout, cin)
        if kdim == cin:
            rows = x.transpose(1, 2).contiguous().view(B * L, cin)
        else:                                                                # zero columns: plumbing, no arithmetic
            rows = torch.zeros(B * L, kdim, device=dev, dtype=torch.float32)
            rows.view(B, L, kdim)[:, :, :cin] = x.transpose(1, 2)
            wp = torch.zeros(cout, kdim, device=dev, dtype=torch.float32)
            wp[:, :cin] = w
            w = wp
        out = torch.empty(B * L, cout, device=dev, dtype=torch.float32)
        bias = E._param(conv.bias, "conv.bias", dev) if conv.bias is not None else None
        for col0 in range(0, cout, 128):
            st.call("dprnn_pointwise_fwd", E._p(rows), kdim, kdim, 0, 0, 0, 0, 0, 0, E._p(w), kdim, E._p(bias), E._p(out), cout, col0, 0, B * L)
        return out.view(B, L, cout).transpose(1, 2).contiguous()


def linear_forward(lin, x: torch.Tensor) -> torch.Tensor:
    """nn.Linear on [..., in_features] through dprnn_linear_small_fwd (the per-utterance Linears: fusion_linear*, aux_linear,
    pred_linear; dprnn_spe.py:88-105,123, dprnn_spe_ira.py:51)."""
    x = _device_f32(x, "linear input")
    _no_grad_only("Linear.forward", x, module=lin)
    if x.shape[-1] != lin.in_features:
        raise DprnnError(f"Linear expects {lin.in_features} input features, got {x.shape[-1]}")
    dev = x.device
    with torch.cuda.device(dev):
        st = E.Stage(dev)
        flat = x.reshape(-1, lin.in_features).contiguous()
        out = E._linear_small(st, lin.weight, lin.bias, flat)
        return out.view(*x.shape[:-1], lin.out_features)


def fusion_forward(sep, emb: torch.Tensor, output: torch.Tensor, L: int) -> torch.Tensor:
    """DPRNNSpe._fusion (dprnn_spe.py:165-229) on device tensors: emb [B, E], output [B, N, L] (already normalised) -> the fused
    [B, C, L].  The per-utterance vectors come from dprnn_linear_small_fwd, the broadcast arithmetic runs in dprnn_rows_affine_fwd,
    the 'att' scores / softmax / nearest upsampling in dprnn_att_row_scale_rows; 'cat' is a copy."""
    output = _device_f32(output, "_fusion input")
    emb = _device_f32(emb, "speaker embedding")
    _no_grad_only("_fusion", output, emb, module=sep)
    ft = sep.fusion_type
    if ft == "cat":
        return torch.cat((output, emb.unsqueeze(-1).expand(-1, -1, L)), dim=1)
    B, N, _ = output.shape
    dev = output.device
    with torch.cuda.device(dev):
        st = E.Stage(dev)
        rows = output.transpose(1, 2).contiguous()                            # [B][L][N]
        out = torch.empty_like(rows)
        lin = lambda m: E._linear_small(st, m.weight, m.bias, emb.contiguous())
        if ft == "film":
            mul, add = lin(sep.fusion_linear_1), lin(sep.fusion_linear_2)      # (both alive until the launch that reads them is enqueued)
            st.call("dprnn_rows_affine_fwd", E._p(rows), E._p(mul), E._p(add), 0, E._p(out), B * L, N, L)
        elif ft == "add":
            add = lin(sep.fusion_linear)
            st.call("dprnn_rows_affine_fwd", E._p(rows), 0, E._p(add), 0, E._p(out), B * L, N, L)
        elif ft == "mul":
            mul = lin(sep.fusion_linear)
            st.call("dprnn_rows_affine_fwd", E._p(rows), E._p(mul), 0, 0, E._p(out), B * L, N, L)
        elif ft == "att":
            a = lin(sep.fusion_linear)
            ks_a = sep.average.kernel_size[0]
            if sep.average.stride[0] != ks_a or sep.average.groups != N:
                raise DprnnError("unexpected averaging conv geometry in the 'att' fusion")
            La = (L - ks_a) // ks_a + 1
            scores = torch.empty(B, La, device=dev, dtype=torch.float32)
            row_scale = torch.empty(B, L, device=dev, dtype=torch.float32)
            st.call("dprnn_att_row_scale_rows", E._p(rows), N, 0, 0, E._p(E._param(sep.average.weight, "average.weight", dev)),
                    E._p(E._param(sep.average.bias, "average.bias", dev)), E._p(a), E._p(scores), E._p(row_scale), B, L, N, ks_a)
            st.call("dprnn_rows_affine_fwd", E._p(rows), E._p(a), 0, E._p(row_scale), E._p(out), B * L, N, L)      # output * a * (att + 1)
        else:
            raise DprnnError(f"unknown fusion_type {ft!r}")
        return out.transpose(1, 2).contiguous()
