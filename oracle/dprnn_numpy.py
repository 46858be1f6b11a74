"""Explicit-formula numpy restatement of the DPRNN separator forward path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- never imported by the product.

Every function cites the reference lines it restates (paths relative to
/root/reference).  ``sd`` is a state_dict as ``{key: np.ndarray}`` with the
reference's key names; ``cfg`` the constructor kwargs (oracle.weights.full_cfg).
``dtype`` np.float32 reproduces the reference arithmetic; np.float64 gives a
ground truth for error budgeting.  Pinned by tests/test_oracle_golden.py against
vectors generated from the reference classes themselves.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from .weights import full_cfg


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


# ----------------------------------------------------------------------------- encoder / decoder
def encoder(sd, cfg, wav, dtype=np.float32):
    """encoder_decoder.py:25-33: relu(conv1d(x[:,None,:], W, stride)), no bias."""
    w = sd["encoder.conv1d.weight"].astype(dtype)[:, 0, :]           # [N, ks]
    ks, st = cfg["kernel_size"], cfg["stride"]
    x = wav.astype(dtype)
    B, T = x.shape
    L = (T - ks) // st + 1
    out = np.zeros((B, w.shape[0], L), dtype=dtype)
    for j in range(ks):
        out += w[None, :, j, None] * x[:, None, j: j + (L - 1) * st + 1: st]
    return np.maximum(out, 0)


def decoder(sd, cfg, z, dtype=np.float32):
    """encoder_decoder.py:40-49: ConvTranspose1d(N->1, ks, stride, no bias) + squeeze rule."""
    w = sd["decoder.weight"].astype(dtype)[:, 0, :]                   # [N, ks]
    ks, st = cfg["kernel_size"], cfg["stride"]
    B, N, L = z.shape
    T = (L - 1) * st + ks
    y = np.zeros((B, T), dtype=dtype)
    for j in range(ks):
        y[:, j: j + (L - 1) * st + 1: st] += np.einsum("bnl,n->bl", z, w[:, j])
    # squeeze rule (:44-47): [B,1,T] -> if fully squeezed is 1-D (B == 1) keep [1,T] else squeeze()
    if B == 1 and T == 1:
        return y.reshape(())
    return y                      # [1,T] (B == 1), [B,1] (T == 1) and [B,T] all keep both dims


# ----------------------------------------------------------------------------- norms / 1x1 convs
def channel_norm(sd, prefix, x, norm_type, dtype=np.float32):
    """nn.GroupNorm(1, C) (dprnn.py:76-77,134; eps 1e-5) or norms.GlobLN (norms.py:6-31; eps 1e-8).
    Statistics over all non-batch dims, biased variance, per-channel affine on dim 1."""
    x = x.astype(dtype)
    axes = tuple(range(1, x.ndim))
    mean = x.mean(axis=axes, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=axes, keepdims=True)
    if norm_type == "gLN":
        eps, g, b = 1e-8, sd[prefix + ".gamma"], sd[prefix + ".beta"]
    else:
        eps, g, b = 1e-5, sd[prefix + ".weight"], sd[prefix + ".bias"]
    shp = (1, -1) + (1,) * (x.ndim - 2)
    return ((x - mean) / np.sqrt(var + dtype(eps))) * g.astype(dtype).reshape(shp) + b.astype(dtype).reshape(shp)


def conv1x1(w, b, x, dtype=np.float32):
    """nn.Conv1d / nn.Conv2d with kernel 1 on channel dim 1."""
    w2 = w.astype(dtype).reshape(w.shape[0], w.shape[1])
    y = np.einsum("oc,bc...->bo...", w2, x.astype(dtype))
    if b is not None:
        y = y + b.astype(dtype).reshape((1, -1) + (1,) * (x.ndim - 2))
    return y


def prelu(a, x):
    return np.where(x >= 0, x, a.reshape(()).astype(x.dtype) * x)


# ----------------------------------------------------------------------------- segmentation / OLA
def n_chunks(L, K, P):
    """unfold with padding K both sides: (L + 2K - K)//P + 1  (dprnn.py:192-198)."""
    return (L + K) // P + 1


def segmentation(x, K, P):
    """dprnn.py:189-201: seg[b,n,k,s] = xpad[b,n,s*P+k], xpad zero-padded by K on both sides."""
    B, N, L = x.shape
    S = n_chunks(L, K, P)
    xp = np.zeros((B, N, L + 2 * K), dtype=x.dtype)
    xp[:, :, K: K + L] = x
    idx = np.arange(S)[None, :] * P + np.arange(K)[:, None]          # [K, S]
    return xp[:, :, idx]                                             # [B, N, K, S]


def overlap_add(x, L, K, P):
    """dprnn.py:203-217: fold = plain sum of overlapping chunk entries, then crop the padding."""
    B, N, K_, S = x.shape
    yp = np.zeros((B, N, L + 2 * K), dtype=x.dtype)
    for s in range(S):
        yp[:, :, s * P: s * P + K] += x[:, :, :, s]
    return yp[:, :, K: K + L]


# ----------------------------------------------------------------------------- recurrent layers
def _rnn_dir(sd, prefix, suf, x, rnn_type, reverse, dtype):
    """One direction of torch.nn.{LSTM,GRU,RNN} (1 layer, batch_first, zero initial state).
    LSTM gate order i,f,g,o; GRU r,z,n with n = tanh(W_in x + b_in + r*(W_hn h + b_hn)); RNN tanh."""
    wih = sd[f"{prefix}.rnn.weight_ih_l0{suf}"].astype(dtype)
    whh = sd[f"{prefix}.rnn.weight_hh_l0{suf}"].astype(dtype)
    bih = sd[f"{prefix}.rnn.bias_ih_l0{suf}"].astype(dtype)
    bhh = sd[f"{prefix}.rnn.bias_hh_l0{suf}"].astype(dtype)
    Bn, T, _ = x.shape
    H = whh.shape[1]
    h = np.zeros((Bn, H), dtype=dtype)
    c = np.zeros((Bn, H), dtype=dtype)
    out = np.zeros((Bn, T, H), dtype=dtype)
    gi_all = x @ wih.T + bih                                         # [Bn, T, G*H]
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        gi = gi_all[:, t]
        gh = h @ whh.T + bhh
        if rnn_type == "LSTM":
            g = gi + gh
            i, f, gg, o = (g[:, 0:H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:4 * H])
            c = _sig(f) * c + _sig(i) * np.tanh(gg)
            h = _sig(o) * np.tanh(c)
        elif rnn_type == "GRU":
            r = _sig(gi[:, 0:H] + gh[:, 0:H])
            z = _sig(gi[:, H:2 * H] + gh[:, H:2 * H])
            n = np.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
            h = (1 - z) * n + z * h
        else:
            h = np.tanh(gi + gh)
        out[:, t] = h
    return out


def single_rnn(sd, prefix, x, rnn_type, bidirectional, dtype=np.float32):
    """dprnn.py:7-37 SingleRNN: outputs only; bidirectional -> cat(fwd, bwd) on the last dim."""
    x = x.astype(dtype)
    f = _rnn_dir(sd, prefix, "", x, rnn_type, False, dtype)
    if not bidirectional:
        return f
    b = _rnn_dir(sd, prefix, "_reverse", x, rnn_type, True, dtype)
    return np.concatenate([f, b], axis=-1)


def linear(sd, prefix, x, dtype=np.float32):
    return x.astype(dtype) @ sd[prefix + ".weight"].astype(dtype).T + sd[prefix + ".bias"].astype(dtype)


def dprnn_block(sd, prefix, cfg, x, dtype=np.float32, taps: Optional[dict] = None):
    """dprnn.py:79-99 DPRNNBlock.forward on [B, N, K, S]."""
    B, N, K, S = x.shape
    rt, nt = cfg["rnn_type"], cfg["norm_type"]
    # intra: transpose(1,-1).reshape(B*S, K, N)                          (:84)
    seq = np.transpose(x, (0, 3, 2, 1)).reshape(B * S, K, N)
    seq = single_rnn(sd, prefix + ".intra_rnn", seq, rt, True, dtype)               # (:86)
    seq = linear(sd, prefix + ".intra_linear", seq, dtype)                           # (:88)
    y = np.transpose(seq.reshape(B, S, K, N), (0, 3, 2, 1))                          # (:89)
    y = channel_norm(sd, prefix + ".intra_norm", y, nt, dtype)                       # (:90)
    mid = x + y                                                                       # (:92)
    if taps is not None:
        taps[prefix + ".mid"] = mid
    # inter: transpose(1,2).transpose(2,-1).reshape(B*K, S, N)           (:94)
    seq = np.transpose(mid, (0, 2, 3, 1)).reshape(B * K, S, N)
    seq = single_rnn(sd, prefix + ".inter_rnn", seq, rt, bool(cfg["bidirectional"]), dtype)   # (:95)
    seq = linear(sd, prefix + ".inter_linear", seq, dtype)                           # (:96)
    y = np.transpose(seq.reshape(B, K, S, N), (0, 3, 1, 2))                          # (:97)
    y = channel_norm(sd, prefix + ".inter_norm", y, nt, dtype)                       # (:98)
    return mid + y                                                                    # (:99)


def dprnn_process(sd, cfg, x, L, dtype=np.float32, taps: Optional[dict] = None):
    """dprnn.py:171-186 == dprnn_spe.py:231-248: segmentation .. activation -> masks [B,2,N,L]."""
    sep = "separation"
    K, P, F, N = cfg["chunk_length"], cfg["hop_length"], cfg["feature_size"], cfg["input_size"]
    B = x.shape[0]
    seg = segmentation(x, K, P)
    S = seg.shape[-1]
    if taps is not None:
        taps["segmented"] = seg
    h = seg
    for i in range(cfg["n_repeats"]):
        h = dprnn_block(sd, f"{sep}.dprnn_blocks.{i}", cfg, h, dtype, taps)
        if taps is not None:
            taps[f"block{i}"] = h
    h = prelu(sd[f"{sep}.prelu.weight"], h)                                           # (:173)
    h = conv1x1(sd[f"{sep}.conv2d.weight"], sd[f"{sep}.conv2d.bias"], h, dtype)       # (:174)
    h = h.reshape(B * 2, F, K, S)                                                     # (:176)
    h = overlap_add(h, L, K, P)                                                       # (:178)
    if taps is not None:
        taps["ola"] = h
    o = np.tanh(conv1x1(sd[f"{sep}.out.0.weight"], sd[f"{sep}.out.0.bias"], h, dtype))
    g = _sig(conv1x1(sd[f"{sep}.gate.0.weight"], sd[f"{sep}.gate.0.bias"], h, dtype))
    h = o * g                                                                         # (:180)
    h = conv1x1(sd[f"{sep}.end_conv1x1.weight"], None, h, dtype)                      # (:181)
    if cfg["activation_type"] == "sigmoid":                                           # (:161-164)
        h = _sig(h)
    elif cfg["activation_type"] == "relu":
        h = np.maximum(h, 0)
    else:
        raise ValueError("activation_type must be 'sigmoid' or 'relu'")
    return h.reshape(B, 2, N, L)                                                      # (:184)


def separator(sd, cfg, enc, dtype=np.float32, taps: Optional[dict] = None):
    """dprnn.py:166-187 DPRNN.forward."""
    sep = "separation"
    B, _, L = enc.shape
    h = channel_norm(sd, f"{sep}.bottleneck.0", enc, cfg["norm_type"], dtype)
    h = conv1x1(sd[f"{sep}.bottleneck.1.weight"], sd[f"{sep}.bottleneck.1.bias"], h, dtype)
    if taps is not None:
        taps["bottleneck"] = h
    return dprnn_process(sd, cfg, h, L, dtype, taps)


def tasnet_forward(sd: Dict[str, np.ndarray], cfg: dict, wav: np.ndarray, dtype=np.float32,
                   taps: Optional[dict] = None) -> np.ndarray:
    """dprnn.py:271-283 DPRNNTasNet.forward: [B, T] -> [B, 2, T]."""
    cfg = full_cfg(**cfg)
    enc = encoder(sd, cfg, wav, dtype)
    if taps is not None:
        taps["encoder"] = enc
    masks = separator(sd, cfg, enc, dtype, taps)
    if taps is not None:
        taps["masks"] = masks
    out = masks * enc[:, None]
    return np.stack([decoder(sd, cfg, out[:, i], dtype) for i in range(2)], axis=1)


# ----------------------------------------------------------------------------- DPRNN-Spe / IRA
def batch_norm(sd, prefix, x, training: bool, dtype=np.float32):
    """nn.BatchNorm1d on [B, C, T] (eps 1e-5): running stats in eval, batch stats (biased var) in train."""
    x = x.astype(dtype)
    if training:
        mean = x.mean(axis=(0, 2))
        var = ((x - mean[None, :, None]) ** 2).mean(axis=(0, 2))
    else:
        mean = sd[prefix + ".running_mean"].astype(dtype)
        var = sd[prefix + ".running_var"].astype(dtype)
    w, b = sd[prefix + ".weight"].astype(dtype), sd[prefix + ".bias"].astype(dtype)
    return (x - mean[None, :, None]) / np.sqrt(var[None, :, None] + dtype(1e-5)) * w[None, :, None] + b[None, :, None]


def res_block(sd, prefix, x, training: bool, dtype=np.float32):
    """dprnn_spe.py:31-42 ResBlock.forward (MaxPool1d(3): floor, stride 3)."""
    y = conv1x1(sd[prefix + ".conv1.weight"], None, x, dtype)
    y = batch_norm(sd, prefix + ".batch_norm1", y, training, dtype)
    y = prelu(sd[prefix + ".prelu1.weight"], y)
    y = conv1x1(sd[prefix + ".conv2.weight"], None, y, dtype)
    y = batch_norm(sd, prefix + ".batch_norm2", y, training, dtype)
    if (prefix + ".conv_downsample.weight") in sd:
        y = y + conv1x1(sd[prefix + ".conv_downsample.weight"], None, x, dtype)
    else:
        y = y + x
    y = prelu(sd[prefix + ".prelu2.weight"], y)
    T3 = y.shape[-1] // 3
    return y[:, :, : T3 * 3].reshape(y.shape[0], y.shape[1], T3, 3).max(axis=-1)


def auxiliary(sd, cfg, aux_enc, aux_len, training: bool, dtype=np.float32):
    """dprnn_spe.py:156-163 (spk_encoder :115-122): embedding = sum_t / aux_T."""
    se = "separation.spk_encoder"
    h = channel_norm(sd, se + ".0", aux_enc, "ln", dtype)           # always nn.GroupNorm (:116)
    h = conv1x1(sd[se + ".1.weight"], sd[se + ".1.bias"], h, dtype)
    for j in (2, 3, 4):
        h = res_block(sd, f"{se}.{j}", h, training, dtype)
    h = conv1x1(sd[se + ".5.weight"], sd[se + ".5.bias"], h, dtype)
    ks = cfg["kernel_size"]
    aux_T = np.floor_divide(np.asarray(aux_len, dtype=np.float64) - ks, ks // 2) + 1
    aux_T = np.floor_divide(np.floor_divide(np.floor_divide(aux_T, 3), 3), 3)
    return (h.sum(-1) / aux_T.reshape(-1, 1).astype(np.float32)).astype(dtype)


def fusion(sd, cfg, emb, x, dtype=np.float32):
    """dprnn_spe.py:165-229: cat / add / mul / film / att."""
    sep, ft = "separation", cfg["fusion_type"]
    B, N, L = x.shape
    lin = lambda p: (emb @ sd[p + ".weight"].astype(dtype).T + sd[p + ".bias"].astype(dtype))[:, :, None]
    if ft == "cat":
        return np.concatenate([x, np.repeat(emb[:, :, None], L, axis=2)], axis=1)
    if ft == "add":
        return x + lin(f"{sep}.fusion_linear")
    if ft == "mul":
        return x * lin(f"{sep}.fusion_linear")
    if ft == "film":
        return x * lin(f"{sep}.fusion_linear_1") + lin(f"{sep}.fusion_linear_2")
    if ft == "att":
        ks = cfg["kernel_size"]
        w = sd[f"{sep}.average.weight"].astype(dtype)[:, 0, :]      # depth-wise, stride ks (:98)
        b = sd[f"{sep}.average.bias"].astype(dtype)
        La = (L - ks) // ks + 1
        avg = sum(w[None, :, j, None] * x[:, :, j: j + (La - 1) * ks + 1: ks] for j in range(ks)) + b[None, :, None]
        a = lin(f"{sep}.fusion_linear")                              # [B, N, 1]
        att = (avg * a).sum(1, keepdims=True)                        # (:218)
        att = np.exp(att - att.max(-1, keepdims=True))
        att = att / att.sum(-1, keepdims=True)                       # softmax over time (:219)
        att = att * a + a                                            # (:220-222)
        idx = np.floor(np.arange(L) * (La / L)).astype(np.int64)    # nn.Upsample(size=L, 'nearest') (:180)
        idx = np.minimum(idx, La - 1)
        return x * att[:, :, idx]
    raise ValueError(ft)


def spe_forward(sd, cfg, wav, aux_wav, aux_len, training_bn=False, dtype=np.float32, taps=None):
    """dprnn_spe.py:314-327 DPRNNSpeTasNet.forward (+ DPRNNSpe.forward :125-154)."""
    cfg = full_cfg(**cfg)
    sep = "separation"
    enc = encoder(sd, cfg, wav, dtype)
    aenc = encoder(sd, cfg, aux_wav, dtype)
    B, _, L = enc.shape
    emb = auxiliary(sd, cfg, aenc, aux_len, training_bn, dtype)
    if taps is not None:
        taps["emb"] = emb
    h = channel_norm(sd, f"{sep}.bottleneck.0", enc, cfg["norm_type"], dtype)
    h = fusion(sd, cfg, emb, h, dtype)
    h = conv1x1(sd[f"{sep}.bottleneck.1.weight"], sd[f"{sep}.bottleneck.1.bias"], h, dtype)
    masks = dprnn_process(sd, cfg, h, L, dtype, taps)
    logits = emb @ sd[f"{sep}.pred_linear.weight"].astype(dtype).T + sd[f"{sep}.pred_linear.bias"].astype(dtype)
    out = masks * enc[:, None]
    return decoder(sd, cfg, out[:, 0], dtype), logits


def ira_forward(sd, cfg, wav, aux_wav, aux_len, training_bn=False, dtype=np.float32, taps=None):
    """dprnn_spe_ira.py:179-190 wrapper + DPRNNSpeIRA.forward :53-115 (two passes)."""
    cfg = full_cfg(**cfg)
    sep = "separation"
    enc = encoder(sd, cfg, wav, dtype)
    aenc = encoder(sd, cfg, aux_wav, dtype)
    B, _, L = enc.shape
    v0 = auxiliary(sd, cfg, aenc, aux_len, training_bn, dtype)
    xn = channel_norm(sd, f"{sep}.bottleneck.0", enc, cfg["norm_type"], dtype)
    bw, bb = sd[f"{sep}.bottleneck.1.weight"], sd[f"{sep}.bottleneck.1.bias"]
    masks = dprnn_process(sd, cfg, conv1x1(bw, bb, fusion(sd, cfg, v0, xn, dtype), dtype), L, dtype)
    d0 = (masks * enc[:, None])[:, 0]
    v1 = auxiliary(sd, cfg, d0, aux_len, training_bn, dtype)          # note: aux_len of the *reference* (:84)
    v1 = np.concatenate([v0, v1], axis=1) @ sd[f"{sep}.aux_linear.weight"].astype(dtype).T \
        + sd[f"{sep}.aux_linear.bias"].astype(dtype)
    if taps is not None:
        taps["v0"], taps["v1"] = v0, v1
    masks = dprnn_process(sd, cfg, conv1x1(bw, bb, fusion(sd, cfg, v1, xn, dtype), dtype), L, dtype)
    d1 = (masks * enc[:, None])[:, 0]
    logits = v1 @ sd[f"{sep}.pred_linear.weight"].astype(dtype).T + sd[f"{sep}.pred_linear.bias"].astype(dtype)
    return decoder(sd, cfg, d1, dtype), logits
