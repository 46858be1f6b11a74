"""The same forward path on stock torch CPU ops (TEST INFRASTRUCTURE, see oracle/__init__.py).

This is the algorithm of oracle/dprnn_numpy.py expressed with the ATen CPU
operators the reference itself dispatches to (torch.lstm -> mkldnn_rnn_layer,
group_norm, conv1d, fold/unfold: SURVEY.md section 2), as plain functions over a
state_dict -- no nn.Module, no reference code.  It exists because the numpy
restatement is too slow for full-size (3 s @ 8 kHz, batch >= 8) cases and
because it is the "reference's own CPU path" that bench.py times as
``cpu_baseline`` (kind "port").  Never used by the product path.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .weights import full_cfg


def _t(sd, key):
    v = sd[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


def to_torch_state(sd) -> Dict[str, torch.Tensor]:
    return {k: _t(sd, k) for k in sd}


def encoder(sd, cfg, wav):                      # encoder_decoder.py:25-33
    return F.relu(F.conv1d(wav.unsqueeze(1), _t(sd, "encoder.conv1d.weight"), None, stride=cfg["stride"]))


def decoder(sd, cfg, z):                        # encoder_decoder.py:40-49
    y = F.conv_transpose1d(z, _t(sd, "decoder.weight"), None, stride=cfg["stride"])
    return y.squeeze(1) if torch.squeeze(y).dim() == 1 else torch.squeeze(y)


def channel_norm(sd, prefix, x, norm_type):     # dprnn.py:76-77,134 / norms.py:6-31
    if norm_type == "gLN":
        dims = list(range(1, x.dim()))
        mean = x.mean(dim=dims, keepdim=True)
        var = torch.var(x, dim=dims, keepdim=True, unbiased=False)
        v = (x - mean) / torch.sqrt(var + 1e-8)
        return (_t(sd, prefix + ".gamma") * v.transpose(1, -1) + _t(sd, prefix + ".beta")).transpose(1, -1)
    return F.group_norm(x, 1, _t(sd, prefix + ".weight"), _t(sd, prefix + ".bias"), 1e-5)


def single_rnn(sd, prefix, x, rnn_type, bidirectional):      # dprnn.py:7-37
    sufs = [""] + (["_reverse"] if bidirectional else [])
    flat = []
    for s in sufs:
        flat += [_t(sd, f"{prefix}.rnn.weight_ih_l0{s}"), _t(sd, f"{prefix}.rnn.weight_hh_l0{s}"),
                 _t(sd, f"{prefix}.rnn.bias_ih_l0{s}"), _t(sd, f"{prefix}.rnn.bias_hh_l0{s}")]
    H = flat[1].shape[1]
    h0 = x.new_zeros(len(sufs), x.shape[0], H)
    if rnn_type == "LSTM":
        return torch.lstm(x, (h0, h0.clone()), flat, True, 1, 0.0, False, bidirectional, True)[0]
    if rnn_type == "GRU":
        return torch.gru(x, h0, flat, True, 1, 0.0, False, bidirectional, True)[0]
    return torch.rnn_tanh(x, h0, flat, True, 1, 0.0, False, bidirectional, True)[0]


def dprnn_block(sd, p, cfg, x):                 # dprnn.py:79-99
    B, N, K, S = x.shape
    rt, nt = cfg["rnn_type"], cfg["norm_type"]
    seq = x.transpose(1, -1).reshape(B * S, K, N)
    seq = single_rnn(sd, p + ".intra_rnn", seq, rt, True)
    seq = F.linear(seq, _t(sd, p + ".intra_linear.weight"), _t(sd, p + ".intra_linear.bias"))
    y = channel_norm(sd, p + ".intra_norm", seq.reshape(B, S, K, N).transpose(1, -1), nt)
    mid = x + y
    seq = mid.transpose(1, 2).transpose(2, -1).reshape(B * K, S, N)
    seq = single_rnn(sd, p + ".inter_rnn", seq, rt, bool(cfg["bidirectional"]))
    seq = F.linear(seq, _t(sd, p + ".inter_linear.weight"), _t(sd, p + ".inter_linear.bias"))
    y = seq.reshape(B, K, S, N).transpose(1, -1).transpose(2, -1).contiguous()
    return mid + channel_norm(sd, p + ".inter_norm", y, nt)


def dprnn_process(sd, cfg, x, L, taps: Optional[dict] = None):    # dprnn.py:171-186
    sep = "separation"
    K, P, Fs, N = cfg["chunk_length"], cfg["hop_length"], cfg["feature_size"], cfg["input_size"]
    B = x.shape[0]
    h = F.unfold(x.unsqueeze(-1), kernel_size=(K, 1), padding=(K, 0), stride=(P, 1))
    S = h.shape[-1]
    h = h.reshape(B, Fs, K, S)
    for i in range(cfg["n_repeats"]):
        h = dprnn_block(sd, f"{sep}.dprnn_blocks.{i}", cfg, h)
        if taps is not None:
            taps[f"block{i}"] = h
    h = F.prelu(h, _t(sd, f"{sep}.prelu.weight"))
    h = F.conv2d(h, _t(sd, f"{sep}.conv2d.weight"), _t(sd, f"{sep}.conv2d.bias"))
    h = h.reshape(B * 2, Fs * K, S)
    h = F.fold(h, (L, 1), kernel_size=(K, 1), padding=(K, 0), stride=(P, 1)).reshape(B * 2, Fs, -1)
    o = torch.tanh(F.conv1d(h, _t(sd, f"{sep}.out.0.weight"), _t(sd, f"{sep}.out.0.bias")))
    g = torch.sigmoid(F.conv1d(h, _t(sd, f"{sep}.gate.0.weight"), _t(sd, f"{sep}.gate.0.bias")))
    h = F.conv1d(o * g, _t(sd, f"{sep}.end_conv1x1.weight"))
    h = torch.sigmoid(h) if cfg["activation_type"] == "sigmoid" else F.relu(h)
    return h.reshape(B, 2, N, L)


def tasnet_forward_autograd(sd, cfg, wav: torch.Tensor) -> torch.Tensor:
    """tasnet_forward with autograd enabled: sd values that require grad receive .grad from a backward() through the
    stock ATen operators -- the CPU restatement of the reference's training step (trainer.py:102-106)."""
    return tasnet_forward.__wrapped__(sd, cfg, wav)


@torch.no_grad()
def tasnet_forward(sd, cfg, wav: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """dprnn.py:271-283: [B,T] -> [B,2,T]."""
    cfg = full_cfg(**cfg)
    sep = "separation"
    enc = encoder(sd, cfg, wav)
    L = enc.shape[-1]
    h = channel_norm(sd, f"{sep}.bottleneck.0", enc, cfg["norm_type"])
    h = F.conv1d(h, _t(sd, f"{sep}.bottleneck.1.weight"), _t(sd, f"{sep}.bottleneck.1.bias"))
    masks = dprnn_process(sd, cfg, h, L, taps)
    out = masks * enc.unsqueeze(1)
    return torch.stack([decoder(sd, cfg, out[:, i]) for i in range(2)], dim=1)


def res_block(sd, p, x, training):              # dprnn_spe.py:31-42
    def bn(q, y):
        return F.batch_norm(y, None if training else _t(sd, q + ".running_mean"),
                            None if training else _t(sd, q + ".running_var"),
                            _t(sd, q + ".weight"), _t(sd, q + ".bias"), training, 0.1, 1e-5)
    y = F.prelu(bn(p + ".batch_norm1", F.conv1d(x, _t(sd, p + ".conv1.weight"))), _t(sd, p + ".prelu1.weight"))
    y = bn(p + ".batch_norm2", F.conv1d(y, _t(sd, p + ".conv2.weight")))
    y = y + (F.conv1d(x, _t(sd, p + ".conv_downsample.weight")) if (p + ".conv_downsample.weight") in sd else x)
    return F.max_pool1d(F.prelu(y, _t(sd, p + ".prelu2.weight")), 3)


def auxiliary(sd, cfg, aenc, aux_len, training):   # dprnn_spe.py:156-163
    se = "separation.spk_encoder"
    h = F.group_norm(aenc, 1, _t(sd, se + ".0.weight"), _t(sd, se + ".0.bias"), 1e-5)
    h = F.conv1d(h, _t(sd, se + ".1.weight"), _t(sd, se + ".1.bias"))
    for j in (2, 3, 4):
        h = res_block(sd, f"{se}.{j}", h, training)
    h = F.conv1d(h, _t(sd, se + ".5.weight"), _t(sd, se + ".5.bias"))
    ks = cfg["kernel_size"]
    aux_T = (aux_len - ks) // (ks // 2) + 1
    aux_T = ((aux_T // 3) // 3) // 3
    return h.sum(-1) / aux_T.view(-1, 1).float()


def fusion_cat(emb, x):                         # dprnn_spe.py:186-192
    return torch.cat([x, emb.unsqueeze(-1).repeat(1, 1, x.shape[-1])], 1)


def spe_forward_autograd(sd, cfg, wav, aux_wav, aux_len, training_bn=False):
    """spe_forward ('cat' fusion) with autograd enabled: CPU restatement of the reference's DPRNN-Spe training step
    (trainer_spe.py:36-45)."""
    return spe_forward.__wrapped__(sd, cfg, wav, aux_wav, aux_len, training_bn)


@torch.no_grad()
def spe_forward(sd, cfg, wav, aux_wav, aux_len, training_bn=False):
    """dprnn_spe.py:314-327, 'cat' fusion only (other fusions: oracle.dprnn_numpy)."""
    cfg = full_cfg(**cfg)
    assert cfg["fusion_type"] == "cat"
    sep = "separation"
    enc, aenc = encoder(sd, cfg, wav), encoder(sd, cfg, aux_wav)
    L = enc.shape[-1]
    emb = auxiliary(sd, cfg, aenc, aux_len, training_bn)
    h = fusion_cat(emb, channel_norm(sd, f"{sep}.bottleneck.0", enc, cfg["norm_type"]))
    h = F.conv1d(h, _t(sd, f"{sep}.bottleneck.1.weight"), _t(sd, f"{sep}.bottleneck.1.bias"))
    masks = dprnn_process(sd, cfg, h, L)
    logits = F.linear(emb, _t(sd, f"{sep}.pred_linear.weight"), _t(sd, f"{sep}.pred_linear.bias"))
    return decoder(sd, cfg, (masks * enc.unsqueeze(1))[:, 0]), logits


@torch.no_grad()
def ira_forward(sd, cfg, wav, aux_wav, aux_len, training_bn=False):
    """dprnn_spe_ira.py:53-115,179-190, 'cat' fusion."""
    cfg = full_cfg(**cfg)
    assert cfg["fusion_type"] == "cat"
    sep = "separation"
    enc, aenc = encoder(sd, cfg, wav), encoder(sd, cfg, aux_wav)
    L = enc.shape[-1]
    bw, bb = _t(sd, f"{sep}.bottleneck.1.weight"), _t(sd, f"{sep}.bottleneck.1.bias")
    v0 = auxiliary(sd, cfg, aenc, aux_len, training_bn)
    xn = channel_norm(sd, f"{sep}.bottleneck.0", enc, cfg["norm_type"])
    masks = dprnn_process(sd, cfg, F.conv1d(fusion_cat(v0, xn), bw, bb), L)
    d0 = (masks * enc.unsqueeze(1))[:, 0]
    v1 = auxiliary(sd, cfg, d0, aux_len, training_bn)
    v1 = F.linear(torch.cat((v0, v1), 1), _t(sd, f"{sep}.aux_linear.weight"), _t(sd, f"{sep}.aux_linear.bias"))
    masks = dprnn_process(sd, cfg, F.conv1d(fusion_cat(v1, xn), bw, bb), L)
    d1 = (masks * enc.unsqueeze(1))[:, 0]
    logits = F.linear(v1, _t(sd, f"{sep}.pred_linear.weight"), _t(sd, f"{sep}.pred_linear.bias"))
    return decoder(sd, cfg, d1), logits
