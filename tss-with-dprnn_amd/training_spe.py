"""Training step of DPRNNSpeTasNet on the MI355X path (reference src/trainers/trainer_spe.py:25-46):

    est, aux = model(mix, reference, ref_len)
    loss = PIT-SI-SDR(est, target) + ce_gamma * CrossEntropy(aux, spk_idx);  loss.backward()

``DPRNNSpeTasNet.forward`` under autograd is one ``torch.autograd.Function`` with two outputs (estimate, speaker logits).
The separator path re-uses training.forward_train / training.backward (the fusion operands enter the bottleneck); this
module adds the speaker branch: spk_encoder (GroupNorm -> conv -> 3 ResBlocks with BatchNorm1d/PReLU/MaxPool1d(3) -> conv
-> time mean, dprnn_spe.py:115-123,156-163), pred_linear, and the fusion Linears, forward with saved intermediates and
backward through the kernels of include/dprnn_mi355x_train.h (dprnn_resblock_pool_bwd, dprnn_bn_bwd, ...).

Supported under autograd: every fusion_type ('cat', 'add', 'mul', 'film', 'att') with the LSTM cell, for DPRNNSpeTasNet and
for the two-pass DPRNNSpeIRATasNet (dprnn_spe_ira.py:53-115: the first pass's d0 = mask_0*enc feeds the speaker encoder a
second time, aux_linear merges the two embeddings, the second separator pass re-uses the weights; its backward chains the
two separator passes through d0 and sums the shared parameters' gradients).  BatchNorm1d follows the module's mode: batch
statistics (and the running-buffer update) in train(), running statistics in eval().  The whole batch is one pass
(BatchNorm couples the utterances), so the batch must fit the activation budget of training.py.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import engine, training
from ._lib import DprnnError
from .engine import F_BUILT, Stage, _norm_params, _p, _param
from .training import _Scratch

TRAIN_FUSIONS = ("cat", "add", "mul", "film", "att")


def check_supported_spe_train(cfg: dict) -> None:
    engine.check_supported_spe(cfg)
    training.check_supported_train(cfg)
    if cfg["fusion_type"] not in TRAIN_FUSIONS:
        raise NotImplementedError(f"DPRNN-Spe MI355X training path: fusion_type {cfg['fusion_type']!r} has inference kernels "
                                  f"only (backward exists for {TRAIN_FUSIONS})")


def _resblock_train(st: Stage, blk, Xin, B: int, T: int, cin: int, cout: int):
    """engine._resblock keeping what the backward needs."""
    dev = Xin.device
    rows = B * T
    f32 = dict(device=dev, dtype=torch.float32)
    sv = dict(x=Xin, B=B, T=T, cin=cin, cout=cout, blk=blk)
    ys, affs, moms = [], [], []
    src, kin = Xin, cin
    for i, (conv, bn) in enumerate(((blk.conv1, blk.batch_norm1), (blk.conv2, blk.batch_norm2))):
        tr = bn.training
        cs = torch.zeros(cout, 2, device=dev, dtype=torch.float64) if tr else None
        if i == 0:
            y = engine._pointwise(st, src, rows, kin, conv.weight, None, cout, col_stats=cs)
        else:
            y = engine._pointwise(st, src, rows, kin, conv.weight, None, cout, in_scale=affs[0][0], in_shift=affs[0][1],
                                  in_mean=affs[0][2], in_prelu=_param(blk.prelu1.weight, "prelu1.weight", dev), col_stats=cs)
        mean = torch.empty(cout, **f32)
        rstd = torch.empty(cout, **f32)
        st.call("dprnn_bn_moments", _p(cs), _p(bn.running_mean), _p(bn.running_var), rows, float(bn.eps), _p(mean), _p(rstd), cout)
        affs.append(engine._bn_fold(st, bn, cs, rows, dev))
        moms.append((mean, rstd, tr))
        ys.append(y)
        src, kin = y, cout
    res = engine._pointwise(st, Xin, rows, cin, blk.conv_downsample.weight, None, cout) if blk.downsample else Xin
    out = torch.empty(B, T // 3, cout, **f32)
    st.call("dprnn_resblock_pool_fwd", _p(ys[1]), _p(affs[1][0]), _p(affs[1][1]), _p(affs[1][2]), _p(res),
            _p(_param(blk.prelu2.weight, "prelu2.weight", dev)), _p(out), B, T, cout)
    sv.update(y1=ys[0], y2=ys[1], aff1=affs[0], aff2=affs[1], mom1=moms[0], mom2=moms[1], res=res)
    return out, T // 3, sv


def _acc(grads: Dict[str, torch.Tensor], name: str, t: torch.Tensor) -> None:
    """Shared parameters collect one gradient per use (two speaker-encoder runs, two separator passes in IRA)."""
    if name in grads:
        grads[name] = grads[name] + t
    else:
        grads[name] = t


def _spk_tail_forward(st: Stage, sep, A0, B: int, T: int, aux_len_dev, ks: int):
    """spk_encoder[2:] + time mean (dprnn_spe.py:117-122,156-163) from the first conv's output A0 [B*T][128]."""
    se = sep.spk_encoder
    h, t = A0, T
    blocks = []
    for blk, cin, cout in ((se[2], 128, 128), (se[3], 128, 256), (se[4], 256, 256)):
        h, t, sv = _resblock_train(st, blk, h.view(B * t, cin), B, t, cin, cout)
        blocks.append(sv)
    if t < 1:
        raise DprnnError("reference utterance too short for three MaxPool1d(3) stages")
    h3 = h.view(B * t, 256)
    E = engine._pointwise(st, h3, B * t, 256, se[5].weight, se[5].bias, 128)
    emb = torch.empty(B, 128, device=A0.device, dtype=torch.float32)
    st.call("dprnn_time_mean_fwd", _p(E), _p(aux_len_dev), aux_len_dev.numel(), _p(emb), B, t, 128, ks)
    return emb, dict(blocks=blocks, h3=h3, t3=t, aux_len=aux_len_dev)


def _fusion_for(st: Stage, model, wav, emb, geo):
    """Fusion operands of one separator pass for the embedding `emb` (dprnn_spe.py:165-229)."""
    cfg = model.cfg
    sep = model.separation
    dev = wav.device
    N, ks, stride = cfg["input_size"], cfg["kernel_size"], cfg["stride"]
    B = wav.shape[0]
    enc_w = model.encoder.conv1d.weight
    mul, add, row_bias = engine._fusion_operands(st, sep, cfg, emb)
    fusion = dict(mul=mul, add=add, row_bias=row_bias)
    if cfg["fusion_type"] == "att":        # per-frame factor: softmax attention over the averaged norm output (dprnn_spe.py:178-183)
        for prm in sep.average.parameters():
            if prm.requires_grad:
                raise NotImplementedError("DPRNN-Spe MI355X training path: the 'att' averaging conv is frozen in the reference "
                                          "(dprnn_spe.py:102-103); its gradient is not computed")
        stats_m = torch.zeros(B, 2, device=dev, dtype=torch.float64)
        st.call("dprnn_encoder_stats", _p(wav), _p(enc_w), _p(stats_m), B, geo.T, geo.L, N, ks, stride)
        engine._count_scale(stats_m, sep.bottleneck[0], st, B)          # zero-padded twin of a narrow model: the real element count
        g_, b_, eps_ = _norm_params(sep.bottleneck[0])
        sc0 = torch.empty(B, N, device=dev, dtype=torch.float32)
        sh0 = torch.empty_like(sc0)
        st.call("dprnn_bottleneck_prep", _p(stats_m), _p(g_), _p(b_), eps_, 0, 0, _p(sc0), _p(sh0), B, geo.L, N)
        ks_a = sep.average.kernel_size[0]
        if sep.average.stride[0] != ks_a or sep.average.groups != N:
            raise DprnnError("unexpected averaging conv geometry in the 'att' fusion")
        La = (geo.L - ks_a) // ks_a + 1
        scores = torch.empty(B, La, device=dev, dtype=torch.float32)
        row_scale = torch.empty(B, geo.L, device=dev, dtype=torch.float32)
        st.call("dprnn_att_row_scale", _p(wav), _p(enc_w), _p(sc0), _p(sh0), _p(sep.average.weight), _p(sep.average.bias), _p(mul),
                _p(scores), _p(row_scale), B, geo.T, geo.L, N, ks, stride, ks_a)
        fusion.update(row_scale=row_scale, scores=scores, avg_w=sep.average.weight, avg_b=sep.average.bias, ks_a=ks_a)
    return {k: v for k, v in fusion.items() if v is not None}


def spe_forward_train(model, wav, aux, aux_len, ira: bool = False):
    """DPRNNSpeTasNet.forward (dprnn_spe.py:314-327) / DPRNNSpeIRATasNet.forward (dprnn_spe_ira.py:53-115,179-190) keeping
    the activations: returns (est [B,T_out], logits, saved)."""
    cfg = model.cfg
    dev = wav.device
    st = Stage(dev)
    N, ks, stride = cfg["input_size"], cfg["kernel_size"], cfg["stride"]
    geo = engine.make_geometry(wav.shape[0], wav.shape[1], N, cfg["chunk_length"], cfg["hop_length"], ks, stride)
    geo_a = engine.make_geometry(aux.shape[0], aux.shape[1], N, cfg["chunk_length"], cfg["hop_length"], ks, stride)
    B = wav.shape[0]
    sep = model.separation
    ndirs = [2 if r.rnn.bidirectional else 1 for r, _, _, _ in training._paths(sep)]
    per_utt = (2 if ira else 1) * training.saved_bytes_per_utt(engine.make_geometry(1, wav.shape[1], N, cfg["chunk_length"],
                                                                                     cfg["hop_length"], ks, stride), len(ndirs), ndirs)
    budget = training.save_budget_bytes(dev)
    if B * per_utt > budget:
        raise DprnnError(f"DPRNN-Spe training step: batch {B} needs {B * per_utt / 2**30:.0f} GiB of saved activations, above the "
                         f"budget of {budget / 2**30:.0f} GiB (training.save_budget_bytes: SAVE_BUDGET_BYTES capped by the memory "
                         f"free on {dev}); the largest batch that fits is {budget // per_utt}.  BatchNorm couples the utterances "
                         "of the speaker branch, so this path does not cut the batch into micro-batches -- lower the batch size "
                         "(config_tss.yaml ships batch_size 5)")
    se = sep.spk_encoder
    enc_w = _param(model.encoder.conv1d.weight, "encoder.conv1d.weight", dev)
    aux_len_dev = engine._aux_len_tensor(aux_len, B, dev)
    stats_a = torch.zeros(B, 2, device=dev, dtype=torch.float64)
    sc_a, sh_a = engine.bottleneck_affine(st, aux, enc_w, se[0], geo_a, stats_a)
    A0 = engine.bottleneck_frames(st, aux, enc_w, sc_a, sh_a, se[1], geo_a, gemm_mode=training.train_modes()[0]).view(B * geo_a.L, F_BUILT)
    v0, tail0 = _spk_tail_forward(st, sep, A0, B, geo_a.L, aux_len_dev, ks)
    spk0 = dict(aux=aux, geo_a=geo_a, stats_a=stats_a, sc_a=sc_a, sh_a=sh_a, tail=tail0, emb=v0)
    out, saved = training.forward_train(model, wav, keep=True, fusion=_fusion_for(st, model, wav, v0, geo), want_d0=ira)
    saved["spk"] = spk0
    if not ira:
        logits = engine._linear_small(st, sep.pred_linear.weight, sep.pred_linear.bias, v0)
        return out[:, 0, :], logits, saved
    # second embedding from the first estimate d0 = mask_0 * enc (dprnn_spe_ira.py:79-92), GroupNorm folded into the conv loader
    d0 = saved["d0"]
    stats_d = torch.zeros(B, 2, device=dev, dtype=torch.float64)
    st.call("dprnn_rows_stats", _p(d0), _p(stats_d), B, geo.L, N)
    engine._count_scale(stats_d, se[0], st, B)
    g0, b0, eps0 = _norm_params(se[0])
    sc_d = torch.empty(B, N, device=dev, dtype=torch.float32)
    sh_d = torch.empty_like(sc_d)
    st.call("dprnn_bottleneck_prep", _p(stats_d), _p(g0), _p(b0), eps0, 0, 0, _p(sc_d), _p(sh_d), B, geo.L, N)
    A0d = engine._pointwise(st, d0, B * geo.L, N, se[1].weight, se[1].bias, 128, in_scale=sc_d, in_shift=sh_d, aff_stride=N,
                            rows_per_utt=geo.L)
    v1p, tail1 = _spk_tail_forward(st, sep, A0d, B, geo.L, aux_len_dev, ks)
    vcat = torch.cat((v0, v1p), dim=1).contiguous()
    v1 = engine._linear_small(st, sep.aux_linear.weight, sep.aux_linear.bias, vcat)
    out2, saved2 = training.forward_train(model, wav, keep=True, fusion=_fusion_for(st, model, wav, v1, geo))
    logits = engine._linear_small(st, sep.pred_linear.weight, sep.pred_linear.bias, v1)
    ira_saved = dict(pass1=saved, pass2=saved2, d0=d0, stats_d=stats_d, sc_d=sc_d, sh_d=sh_d, tail1=tail1, vcat=vcat, v1=v1, geo=geo)
    return out2[:, 0, :], logits, ira_saved


def _conv_bwd(st, lib, dOut, cout, x, cin, weight, rows):
    """1x1 conv y = x @ W^T (W [cout][cin]): returns (dW, dx)."""
    dev = dOut.device
    f32 = dict(device=dev, dtype=torch.float32)
    gw = torch.empty_like(weight)
    ws = torch.empty(lib.dprnn_wgrad_workspace_floats(cout, cin, rows), **f32)
    q1 = (x.data_ptr() + 4 * 128) if cin == 256 else 0
    st.call("dprnn_wgrad", _p(dOut), cout, cout, _p(x), cin, min(cin, 128), 0, q1, cin, 128 if cin == 256 else 0, 0,
            rows, 1, 1, _p(gw), cin, 1, 0, 0, 0, _p(ws))
    dx = torch.empty(rows, cin, **f32)
    if cin == 256:                          # two 128-column halves
        for c0 in (0, 128):
            st.call("dprnn_dgrad", _p(dOut), cout, weight.data_ptr() + 4 * c0, cin, cout, 0, 0, dx.data_ptr() + 4 * c0, cin, 128, rows)
    else:
        st.call("dprnn_dgrad", _p(dOut), cout, _p(weight), cin, cout, 0, 0, _p(dx), cin, cin, rows)
    return gw, dx


def _resblock_bwd(st, lib, scr: _Scratch, sv: dict, dOut, prefix: str, grads: Dict[str, torch.Tensor], tag: str = ""):
    """Backward of one ResBlock (dprnn_spe.py:31-42): returns the gradient of its input rows [B*T][cin]."""
    blk, B, T, cin, cout = sv["blk"], sv["B"], sv["T"], sv["cin"], sv["cout"]
    dev = dOut.device
    rows = B * T
    f32 = dict(device=dev, dtype=torch.float32)

    def to_grad(addr, like):
        g = torch.empty_like(like)
        st.call("dprnn_scratch_to_grad", addr, _p(g), g.numel(), 0)
        return g

    (s1, t1, mu1), (s2, t2, mu2) = sv["aff1"], sv["aff2"]
    (m1, r1, tr1), (m2, r2, tr2) = sv["mom1"], sv["mom2"]
    # MaxPool1d(3) + PReLU2 + residual add
    dU = torch.empty(rows, cout, **f32)
    a_al2 = scr.slot(tag + prefix + "alpha2", 1)
    st.call("dprnn_resblock_pool_bwd", _p(dOut), _p(sv["y2"]), _p(s2), _p(t2), _p(mu2), _p(sv["res"]), _p(blk.prelu2.weight), _p(dU), a_al2,
            B, T, cout)
    grads[prefix + "prelu2.weight"] = to_grad(a_al2, blk.prelu2.weight)
    # BatchNorm2
    a_red2 = scr.slot(tag + prefix + "bn2", 2 * cout)
    dY2 = torch.empty(rows, cout, **f32)
    st.call("dprnn_bn_bwd", _p(dU), _p(sv["y2"]), _p(m2), _p(r2), _p(blk.batch_norm2.weight), 1 if tr2 else 0, a_red2, _p(dY2), rows, cout)
    grads[prefix + "batch_norm2.bias"] = to_grad(a_red2, blk.batch_norm2.bias)
    grads[prefix + "batch_norm2.weight"] = to_grad(a_red2 + 8 * cout, blk.batch_norm2.weight)
    # conv2 on a1 = prelu1(bn1(y1))
    a1 = torch.empty(rows, cout, **f32)
    st.call("dprnn_affine_prelu_rows", _p(sv["y1"]), _p(s1), _p(t1), _p(mu1), _p(blk.prelu1.weight), _p(a1), rows, cout)
    grads[prefix + "conv2.weight"], da1 = _conv_bwd(st, lib, dY2, cout, a1, cout, blk.conv2.weight, rows)
    del a1, dY2
    # PReLU1 + BatchNorm1
    a_al1 = scr.slot(tag + prefix + "alpha1", 1)
    dB1 = torch.empty(rows, cout, **f32)
    st.call("dprnn_affine_prelu_bwd", _p(da1), _p(sv["y1"]), _p(s1), _p(t1), _p(mu1), _p(blk.prelu1.weight), _p(dB1), a_al1, rows, cout)
    grads[prefix + "prelu1.weight"] = to_grad(a_al1, blk.prelu1.weight)
    a_red1 = scr.slot(tag + prefix + "bn1", 2 * cout)
    dY1 = da1                                       # reuse
    st.call("dprnn_bn_bwd", _p(dB1), _p(sv["y1"]), _p(m1), _p(r1), _p(blk.batch_norm1.weight), 1 if tr1 else 0, a_red1, _p(dY1), rows, cout)
    grads[prefix + "batch_norm1.bias"] = to_grad(a_red1, blk.batch_norm1.bias)
    grads[prefix + "batch_norm1.weight"] = to_grad(a_red1 + 8 * cout, blk.batch_norm1.weight)
    del dB1
    # conv1 and the residual branch meet at the block input
    x = sv["x"]
    grads[prefix + "conv1.weight"], dx = _conv_bwd(st, lib, dY1, cout, x, cin, blk.conv1.weight, rows)
    if blk.downsample:
        grads[prefix + "conv_downsample.weight"], dxr = _conv_bwd(st, lib, dU, cout, x, cin, blk.conv_downsample.weight, rows)
    else:
        dxr = dU
    st.call("dprnn_add_rows", _p(dx), _p(dxr), _p(dx), rows * cin)
    return dx


def _to_grad(st, addr, like):
    g = torch.empty_like(like)
    st.call("dprnn_scratch_to_grad", addr, _p(g), g.numel(), 0)
    return g


def _lin_bwd(st, grads, d_o, x, lin_w, lin_b, wname, bname, d_in, k0=0, dW=None):
    """Gradients of out = W[:, k0:k0+K] x + b for one small per-utterance Linear; d_in (+)= d_o @ W."""
    B, K = x.shape
    gw = dW if dW is not None else torch.empty_like(lin_w)
    gb = torch.empty_like(lin_b) if lin_b is not None else None
    st.call("dprnn_linear_small_bwd", _p(d_o), d_o.shape[1], _p(x), K, _p(lin_w), lin_w.shape[1], k0, K, _p(d_in), d_in.shape[1], 1,
            _p(gw), _p(gb), B, d_o.shape[1])
    if dW is None:
        _acc(grads, wname, gw)
    if gb is not None:
        _acc(grads, bname, gb)


def _fusion_linears_bwd(st, model, grads, extras, gsep, emb, d_emb):
    """d_emb += gradient through the fusion operands of one separator pass; gsep: that pass's gradient dict (the 'cat'
    half of the bottleneck weight is written into its tensor in place)."""
    sep = model.separation
    ft = model.cfg["fusion_type"]
    N = model.cfg["input_size"]
    if ft == "cat":
        _lin_bwd(st, grads, extras["d_row_bias"], emb, sep.bottleneck[1].weight, None, None, None, d_emb, k0=N,
                 dW=gsep["separation.bottleneck.1.weight"])
    elif ft == "add":
        _lin_bwd(st, grads, extras["d_add"], emb, sep.fusion_linear.weight, sep.fusion_linear.bias,
                 "separation.fusion_linear.weight", "separation.fusion_linear.bias", d_emb)
    elif ft in ("mul", "att"):
        _lin_bwd(st, grads, extras["d_mul"], emb, sep.fusion_linear.weight, sep.fusion_linear.bias,
                 "separation.fusion_linear.weight", "separation.fusion_linear.bias", d_emb)
    elif ft == "film":
        _lin_bwd(st, grads, extras["d_mul"], emb, sep.fusion_linear_1.weight, sep.fusion_linear_1.bias,
                 "separation.fusion_linear_1.weight", "separation.fusion_linear_1.bias", d_emb)
        _lin_bwd(st, grads, extras["d_add"], emb, sep.fusion_linear_2.weight, sep.fusion_linear_2.bias,
                 "separation.fusion_linear_2.weight", "separation.fusion_linear_2.bias", d_emb)


def _spk_tail_backward(st, lib, scr: _Scratch, tag: str, sep, tail: dict, d_emb, B: int, ks: int, grads) -> torch.Tensor:
    """Backward of _spk_tail_forward: returns the gradient of A0 [B*T][128]; parameter gradients are added into grads."""
    se = sep.spk_encoder
    dev = d_emb.device
    f32 = dict(device=dev, dtype=torch.float32)
    t3, h3 = tail["t3"], tail["h3"]
    dE = torch.empty(B * t3, F_BUILT, **f32)
    st.call("dprnn_time_mean_bwd", _p(d_emb), _p(tail["aux_len"]), tail["aux_len"].numel(), _p(dE), B, t3, F_BUILT, ks)
    w5 = se[5].weight
    gw5 = torch.empty_like(w5)
    gb5 = torch.empty_like(se[5].bias)
    ws = torch.empty(lib.dprnn_wgrad_workspace_floats(128, 256, B * t3), **f32)
    st.call("dprnn_wgrad", _p(dE), 128, 128, _p(h3), 256, 128, 0, h3.data_ptr() + 4 * 128, 256, 128, 0, B * t3, 1, 1,
            _p(gw5), 256, 1, _p(gb5), 0, 0, _p(ws))
    _acc(grads, "separation.spk_encoder.5.weight", gw5)
    _acc(grads, "separation.spk_encoder.5.bias", gb5)
    dh = torch.empty(B * t3, 256, **f32)
    st.call("dprnn_dgrad", _p(dE), 128, _p(w5), 256, 128, 0, 0, _p(dh), 256, 256, B * t3)
    for i in (4, 3, 2):
        sv = tail["blocks"][i - 2]
        g_blk: Dict[str, torch.Tensor] = {}
        dh = _resblock_bwd(st, lib, scr, sv, dh.view(B, sv["T"] // 3, sv["cout"]), f"separation.spk_encoder.{i}.", g_blk, tag)
        for k, v in g_blk.items():
            _acc(grads, k, v)
    return dh


def _pass_backward(model, saved, d_out, extras):
    scr = _Scratch(d_out.device)
    g = training.backward(model, saved, d_out, scratch=scr, extras=extras)
    return g, scr


def spe_backward(model, saved: dict, d_est: torch.Tensor, d_logits: Optional[torch.Tensor], ira: bool = False
                 ) -> Dict[str, torch.Tensor]:
    cfg = model.cfg
    dev = d_est.device
    st = Stage(dev)
    lib = st.lib
    sep = model.separation
    se = sep.spk_encoder
    last = saved["pass2"] if ira else saved
    first = saved["pass1"] if ira else saved
    geo: engine.Geometry = last["geo"]
    spk = first["spk"]
    B, N, ks = geo.B, geo.N, geo.ks
    f32 = dict(device=dev, dtype=torch.float32)
    enc_w = model.encoder.conv1d.weight
    scr = _Scratch(dev)
    E_ = spk["emb"].shape[1]
    grads: Dict[str, torch.Tensor] = {}

    # the separator pass that produced the estimate: speaker 0 of the two decoded channels (dprnn_spe.py:325)
    d_out = torch.zeros(B, 2, geo.T_out, **f32)
    d_out[:, 0, :] = d_est
    extras: dict = {}
    g_last, scr_last = _pass_backward(model, last, d_out, extras)
    emb_last = saved["v1"] if ira else spk["emb"]
    d_emb_last = torch.zeros(B, E_, **f32)
    if d_logits is not None:
        _lin_bwd(st, grads, d_logits.contiguous().float(), emb_last, sep.pred_linear.weight, sep.pred_linear.bias,
                 "separation.pred_linear.weight", "separation.pred_linear.bias", d_emb_last)
    else:
        grads["separation.pred_linear.weight"] = torch.zeros_like(sep.pred_linear.weight)
        grads["separation.pred_linear.bias"] = torch.zeros_like(sep.pred_linear.bias)
    _fusion_linears_bwd(st, model, grads, extras, g_last, emb_last, d_emb_last)
    for k, v in g_last.items():
        _acc(grads, k, v)
    dWe = _to_grad(st, scr_last.slot("dWe", N * ks), enc_w)

    if ira:
        # aux_linear merges (v0, v1') (dprnn_spe_ira.py:86-92)
        d_vcat = torch.zeros(B, 2 * E_, **f32)
        _lin_bwd(st, grads, d_emb_last, saved["vcat"], sep.aux_linear.weight, sep.aux_linear.bias,
                 "separation.aux_linear.weight", "separation.aux_linear.bias", d_vcat)
        d_v0 = d_vcat[:, :E_].contiguous()
        d_v1p = d_vcat[:, E_:].contiguous()
        # speaker encoder on d0: tail, first conv with the GroupNorm of d0 folded in, GroupNorm backward -> gradient of d0
        dA0d = _spk_tail_backward(st, lib, scr, "d0.", sep, saved["tail1"], d_v1p, B, ks, grads)
        rows_m = B * geo.L
        d0 = saved["d0"]
        An = torch.empty(rows_m, N, **f32)
        st.call("dprnn_rows64_affine", _p(d0), _p(saved["sc_d"]), _p(saved["sh_d"]), _p(An), B, geo.L)
        w1 = se[1].weight
        gw1 = torch.empty_like(w1)
        gb1 = torch.empty_like(se[1].bias)
        ws = torch.empty(lib.dprnn_wgrad_workspace_floats(128, N, rows_m), **f32)
        st.call("dprnn_wgrad", _p(dA0d), 128, 128, _p(An), N, N, 0, 0, 0, 0, 0, rows_m, 1, 1, _p(gw1), N, 1, _p(gb1), 0, 0, _p(ws))
        _acc(grads, "separation.spk_encoder.1.weight", gw1)
        _acc(grads, "separation.spk_encoder.1.bias", gb1)
        dAn = An
        st.call("dprnn_dgrad", _p(dA0d), 128, _p(w1), N, 128, 0, 0, _p(dAn), N, N, rows_m)
        g0, b0, eps0 = _norm_params(se[0])
        a_red, a_dgb = scr.slot("red_d0", 2 * B), scr.slot("dgb_d0", 2 * N)
        dD0 = torch.empty(rows_m, N, **f32)
        st.call("dprnn_rows64_norm_bwd_ex", _p(d0), _p(dAn), _p(saved["stats_d"]), _p(g0), eps0, a_red, a_dgb, _p(dD0), B, geo.L,
                float(se[0].__dict__.get("_dprnn_count_scale", 1.0)))
        _acc(grads, "separation.spk_encoder.0.weight", _to_grad(st, a_dgb, g0))
        _acc(grads, "separation.spk_encoder.0.bias", _to_grad(st, a_dgb + 8 * N, b0))
        # first separator pass: its decoded output is unused; the gradient enters through d0 (dprnn_spe_ira.py:79-80)
        extras1: dict = {"dM0": dD0}
        g1, scr1 = _pass_backward(model, first, torch.zeros(B, 2, geo.T_out, **f32), extras1)
        _fusion_linears_bwd(st, model, grads, extras1, g1, spk["emb"], d_v0)
        for k, v in g1.items():
            _acc(grads, k, v)
        dWe = dWe + _to_grad(st, scr1.slot("dWe", N * ks), enc_w)
        d_emb0 = d_v0
    else:
        d_emb0 = d_emb_last

    # speaker encoder on the reference utterance: tail, conv N -> O, GroupNorm, encoder (dprnn_spe.py:115-122,156-163)
    dA0 = _spk_tail_backward(st, lib, scr, "aux.", sep, spk["tail"], d_emb0, B, ks, grads)
    geo_a = spk["geo_a"]
    rows_a = B * geo_a.L
    aux = spk["aux"]
    An = torch.empty(rows_a, N, **f32)
    st.call("dprnn_encoder_norm_rows", _p(aux), _p(enc_w), _p(spk["sc_a"]), _p(spk["sh_a"]), _p(An), B, geo_a.T, geo_a.L, N, ks, geo_a.stride)
    w1 = se[1].weight
    gw1 = torch.empty_like(w1)
    gb1 = torch.empty_like(se[1].bias)
    ws = torch.empty(lib.dprnn_wgrad_workspace_floats(128, N, rows_a), **f32)
    st.call("dprnn_wgrad", _p(dA0), 128, 128, _p(An), N, N, 0, 0, 0, 0, 0, rows_a, 1, 1, _p(gw1), N, 1, _p(gb1), 0, 0, _p(ws))
    _acc(grads, "separation.spk_encoder.1.weight", gw1)
    _acc(grads, "separation.spk_encoder.1.bias", gb1)
    dAn = An
    st.call("dprnn_dgrad", _p(dA0), 128, _p(w1), N, 128, 0, 0, _p(dAn), N, N, rows_a)
    g0, b0, eps0 = _norm_params(se[0])
    a_red, a_dgb = scr.slot("red_spk", 2 * B), scr.slot("dgb_spk", 2 * N)
    st.call("dprnn_encoder_norm_bwd_stats", _p(aux), _p(enc_w), _p(dAn), _p(spk["stats_a"]), _p(g0), eps0, a_red, a_dgb,
            B, geo_a.T, geo_a.L, N, ks, geo_a.stride)
    training._count_scale_red(scr, "red_spk", se[0], st, B)
    _acc(grads, "separation.spk_encoder.0.weight", _to_grad(st, a_dgb, g0))
    _acc(grads, "separation.spk_encoder.0.bias", _to_grad(st, a_dgb + 8 * N, b0))
    st.call("dprnn_encoder_bwd", _p(aux), _p(enc_w), _p(dAn), _p(spk["stats_a"]), _p(g0), eps0, a_red, 0,
            scr.slot("dWe", N * ks), B, geo_a.T, geo_a.L, N, ks, geo_a.stride)
    grads["encoder.conv1d.weight"] = dWe + _to_grad(st, scr.slot("dWe", N * ks), enc_w)
    return grads


class _SpeFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, wav, aux, aux_len, names, *params):
        ctx.model, ctx.names = model, names
        ctx.ira = bool(getattr(model, "_ira", False))
        ctx.inputs = (wav, aux, aux_len)        # what a saturated backward needs to run again from the waveforms
        ctx.fwd_exact = False
        ctx.modes = training.train_modes()[:2]  # autograd runs backward() on another thread: this forward's modes travel with it
        run = lambda: spe_forward_train(model, wav, aux, aux_len, ctx.ira)

        def run_exact():
            ctx.fwd_exact = True
            return training._exact(run, fwd=True, bwd=False)()
        est, logits, saved = engine._guarded(wav.device, run, run_exact, "DPRNNSpeTasNet training forward", model.separation,
                                             active=training.train_modes()[0] == 3)
        ctx.saved = saved
        return est, logits

    @staticmethod
    def backward(ctx, d_est, d_logits):
        model = ctx.model
        with torch.cuda.device(d_est.device), training.use_train_modes(*ctx.modes):
            d_est = d_est.contiguous().float()
            first = lambda: spe_backward(model, ctx.saved, d_est, d_logits, ctx.ira)
            if ctx.fwd_exact:
                first = training._exact(first, fwd=True, bwd=False)

            def again():                 # f16-pieces backward saturated: forward again (BatchNorm running buffers put back
                ctx.saved = None         # afterwards: the step has already updated them once), backward in the exact kernels
                bufs = engine._bn_buffers(model.separation)
                keep = [b.clone() for b in bufs]
                _, _, saved = spe_forward_train(model, *ctx.inputs, ctx.ira)
                for b, k in zip(bufs, keep):
                    b.copy_(k)
                return spe_backward(model, saved, d_est, d_logits, ctx.ira)
            grads = engine._guarded(d_est.device, first, training._exact(again, fwd=ctx.fwd_exact, bwd=True),
                                    "DPRNNSpeTasNet training backward", active=training.train_modes()[1] == 3)
        ctx.saved = ctx.inputs = None
        return (None, None, None, None, None) + tuple(grads.get(n) for n in ctx.names)


def spe_forward_autograd(model, wav, aux, aux_len):
    """model(mix, reference, ref_len) under autograd: differentiable w.r.t. every parameter."""
    from . import padded
    wav, aux = engine.check_input(wav), engine.check_input(aux, "aux")
    if aux.shape[0] != wav.shape[0]:
        raise DprnnError("mixture and reference batch sizes differ")
    if wav.requires_grad or aux.requires_grad:
        raise NotImplementedError("DPRNN MI355X training path: gradients w.r.t. the waveforms are not computed")
    named = [(n, p) for n, p in model.named_parameters()]
    for n, p in named:
        _param(p, n, wav.device)
    if padded.needs_twin(model.cfg):
        # a narrower model trains on its zero-padded twin (padded.py, training.tasnet_forward_autograd); the BatchNorm running statistics
        # the twin's train-mode forward moved go back into the caller's buffers (their real channels), as after an inference forward
        run, tw = padded.twin_of(model)
        check_supported_spe_train(run.cfg)
        wide = padded.placed_parameters(model, tw)
        with torch.cuda.device(wav.device):
            out = _SpeFunction.apply(run, wav, aux, aux_len, tuple(n for n, _ in named), *[wide[n] for n, _ in named])
        if any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training for m in model.modules()):
            tw.write_back_buffers(model)
        return out
    check_supported_spe_train(model.cfg)
    with torch.cuda.device(wav.device):
        return _SpeFunction.apply(model, wav, aux, aux_len, tuple(n for n, _ in named), *[p for _, p in named])
