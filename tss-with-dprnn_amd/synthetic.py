"""Version-independent synthetic weight / input recipe (numpy only; no model arithmetic).

Used by bench.py, tools/ and __graft_entry__ to build the synthetic workload of SURVEY.md 8(d), and re-exported by
oracle/weights.py for the tests, so that product code imports nothing from oracle/.

``state_spec`` restates the state_dict contract of the reference model classes
(key names and shapes):
  DPRNNTasNet        /root/reference/src/models/dprnn.py:237-269 (+ DPRNN :117-164,
                     DPRNNBlock :50-77, SingleRNN :17-28, Encoder
                     encoder_decoder.py:14-23, Decoder :35-38)
  DPRNNSpeTasNet     dprnn_spe.py:273-312 (DPRNNSpe :65-123, ResBlock :16-29)
  DPRNNSpeIRATasNet  dprnn_spe_ira.py:138-177 (aux_linear :51)

``make_state`` fills those tensors from ``numpy.random.default_rng`` streams keyed
by (seed, crc32(key)), so the bytes do not depend on the torch version, on
parameter ordering or on which side (reference model / oracle / HIP module) asks.
Ranges mimic torch's default initialisers but norm affine parameters, PReLU
slopes and BatchNorm running statistics are deliberately non-trivial so that a
kernel that ignores them cannot pass.
"""
from __future__ import annotations

import zlib
from typing import Dict, List, Tuple

import numpy as np

SHIPPED_CFG = dict(  # scripts/train/config_bss.yaml:18-28
    input_size=64, feature_size=128, hidden_size=128, chunk_length=250,
    kernel_size=2, hop_length=125, n_repeats=6, bidirectional=True,
    rnn_type="LSTM", norm_type="ln", activation_type="sigmoid", dropout=0,
    stride=None,
)
SHIPPED_SPE_EXTRA = dict(  # scripts/train/config_tss.yaml:19-34
    O=128, P=256, embeddings_size=128, num_spks=251, fusion_type="cat",
)


def full_cfg(**over) -> dict:
    cfg = dict(SHIPPED_CFG)
    cfg.update(SHIPPED_SPE_EXTRA)
    cfg.update(over)
    if cfg.get("hop_length") is None:
        cfg["hop_length"] = cfg["chunk_length"] // 2        # dprnn.py:128
    if cfg.get("stride") is None:
        cfg["stride"] = cfg["kernel_size"] // 2              # dprnn.py:243
    return cfg


def _norm_keys(prefix: str, c: int, norm_type: str) -> List[Tuple[str, tuple]]:
    if norm_type == "gLN":                                    # norms.py:20-21
        return [(prefix + ".gamma", (c,)), (prefix + ".beta", (c,))]
    return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,))]


def _rnn_keys(prefix: str, in_size: int, hid: int, bidir: bool, rnn_type: str):
    gates = {"LSTM": 4, "GRU": 3, "RNN": 1}[rnn_type]
    out = []
    for suf in ([""] + (["_reverse"] if bidir else [])):
        out += [
            (f"{prefix}.rnn.weight_ih_l0{suf}", (gates * hid, in_size)),
            (f"{prefix}.rnn.weight_hh_l0{suf}", (gates * hid, hid)),
            (f"{prefix}.rnn.bias_ih_l0{suf}", (gates * hid,)),
            (f"{prefix}.rnn.bias_hh_l0{suf}", (gates * hid,)),
        ]
    return out


def _resblock_keys(prefix: str, cin: int, cout: int):
    out = [
        (prefix + ".conv1.weight", (cout, cin, 1)),
        (prefix + ".conv2.weight", (cout, cout, 1)),
    ]
    for bn in ("batch_norm1", "batch_norm2"):
        out += [
            (f"{prefix}.{bn}.weight", (cout,)), (f"{prefix}.{bn}.bias", (cout,)),
            (f"{prefix}.{bn}.running_mean", (cout,)), (f"{prefix}.{bn}.running_var", (cout,)),
            (f"{prefix}.{bn}.num_batches_tracked", ()),
        ]
    out += [(prefix + ".prelu1.weight", (1,)), (prefix + ".prelu2.weight", (1,))]
    if cin != cout:
        out += [(prefix + ".conv_downsample.weight", (cout, cin, 1))]
    return out


def state_spec(kind: str, cfg: dict) -> List[Tuple[str, tuple]]:
    """(key, shape) list, in the reference's registration order.

    kind: 'bss' (DPRNNTasNet) | 'spe' (DPRNNSpeTasNet) | 'ira' (DPRNNSpeIRATasNet).
    """
    cfg = full_cfg(**cfg)
    N, F, H = cfg["input_size"], cfg["feature_size"], cfg["hidden_size"]
    ks, nt, rt = cfg["kernel_size"], cfg["norm_type"], cfg["rnn_type"]
    bidir = bool(cfg["bidirectional"])
    E = cfg["embeddings_size"]
    sep = "separation"
    spec: List[Tuple[str, tuple]] = [("encoder.conv1d.weight", (N, 1, ks))]
    # DPRNN.__init__ registration order: bottleneck, dprnn_blocks, prelu, conv2d, out, gate, end_conv1x1
    fus = cfg["fusion_type"]
    bott_in = N + E if (kind != "bss" and fus == "cat") else N
    bott = _norm_keys(f"{sep}.bottleneck.0", N, nt) + [
        (f"{sep}.bottleneck.1.weight", (F, bott_in, 1)), (f"{sep}.bottleneck.1.bias", (F,))]
    blocks = []
    for i in range(cfg["n_repeats"]):
        p = f"{sep}.dprnn_blocks.{i}"
        blocks += _rnn_keys(p + ".intra_rnn", F, H, True, rt)
        blocks += [(p + ".intra_linear.weight", (F, 2 * H)), (p + ".intra_linear.bias", (F,))]
        blocks += _rnn_keys(p + ".inter_rnn", F, H, bidir, rt)
        blocks += [(p + ".inter_linear.weight", (F, (2 if bidir else 1) * H)),
                   (p + ".inter_linear.bias", (F,))]
        blocks += _norm_keys(p + ".intra_norm", F, nt) + _norm_keys(p + ".inter_norm", F, nt)
    head = [
        (f"{sep}.prelu.weight", (1,)),
        (f"{sep}.conv2d.weight", (2 * F, F, 1, 1)), (f"{sep}.conv2d.bias", (2 * F,)),
        (f"{sep}.out.0.weight", (F, F, 1)), (f"{sep}.out.0.bias", (F,)),
        (f"{sep}.gate.0.weight", (F, F, 1)), (f"{sep}.gate.0.bias", (F,)),
        (f"{sep}.end_conv1x1.weight", (N, F, 1)),
    ]
    if kind == "bss":
        spec += bott + blocks + head
    else:
        # DPRNNSpe.__init__: super().__init__ registers bottleneck(first!)/blocks/head, then
        # fusion layers, then re-assigns self.bottleneck (position in the OrderedDict is kept),
        # then spk_encoder, pred_linear  (dprnn_spe.py:72-123)
        spec += bott + blocks + head
        if fus in ("add", "mul"):
            spec += [(f"{sep}.fusion_linear.weight", (N, E)), (f"{sep}.fusion_linear.bias", (N,))]
        if fus == "film":
            for j in (1, 2):
                spec += [(f"{sep}.fusion_linear_{j}.weight", (N, E)), (f"{sep}.fusion_linear_{j}.bias", (N,))]
        if fus == "att":
            spec += [(f"{sep}.fusion_linear.weight", (N, E)), (f"{sep}.fusion_linear.bias", (N,)),
                     (f"{sep}.average.weight", (N, 1, ks)), (f"{sep}.average.bias", (N,))]
        O, P = cfg["O"], cfg["P"]
        se = f"{sep}.spk_encoder"
        spec += [(f"{se}.0.weight", (N,)), (f"{se}.0.bias", (N,)),
                 (f"{se}.1.weight", (O, N, 1)), (f"{se}.1.bias", (O,))]
        spec += _resblock_keys(f"{se}.2", O, O) + _resblock_keys(f"{se}.3", O, P) + _resblock_keys(f"{se}.4", P, P)
        spec += [(f"{se}.5.weight", (E, P, 1)), (f"{se}.5.bias", (E,))]
        spec += [(f"{sep}.pred_linear.weight", (cfg["num_spks"], E)), (f"{sep}.pred_linear.bias", (cfg["num_spks"],))]
        if kind == "ira":
            spec += [(f"{sep}.aux_linear.weight", (E, 2 * E)), (f"{sep}.aux_linear.bias", (E,))]
    spec += [("decoder.weight", (N, 1, ks))]
    return spec


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(key.encode())])


def make_tensor(key: str, shape: tuple, seed: int) -> np.ndarray:
    """One tensor of the recipe (float32; int64 for num_batches_tracked)."""
    g = _rng(seed, key)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.array(0, dtype=np.int64)
    if leaf == "running_mean":
        return (0.05 * g.standard_normal(shape)).astype(np.float32)
    if leaf == "running_var":
        return (0.6 + 0.8 * g.random(shape)).astype(np.float32)
    if ".average." in key:                                    # dprnn_spe.py:99-100 (frozen averaging conv)
        if leaf == "weight":
            return np.full(shape, 1.0 / shape[-1], dtype=np.float32)
        return np.zeros(shape, dtype=np.float32)
    if "prelu" in key:
        return (0.1 + 0.3 * g.random(shape)).astype(np.float32)
    is_norm = (leaf in ("gamma", "beta")) or (len(shape) == 1 and leaf in ("weight", "bias") and (
        "norm" in key or "bottleneck.0" in key or "spk_encoder.0" in key))
    if is_norm:
        if leaf in ("weight", "gamma"):
            return (1.0 + 0.2 * (g.random(shape) - 0.5)).astype(np.float32)
        return (0.2 * (g.random(shape) - 0.5)).astype(np.float32)
    if ".rnn." in key:      # U(-1/sqrt(hidden), ..) needs the hidden size: generated in make_state
        raise RuntimeError("rnn tensors are generated in make_state")
    if len(shape) >= 2:                                        # conv / linear weight: U(-1/sqrt(fan_in), ..)
        fan_in = int(np.prod(shape[1:]))
        a = 1.0 / np.sqrt(fan_in)
        return ((2 * g.random(shape) - 1) * a).astype(np.float32)
    raise RuntimeError(f"bias tensors are generated in make_state: {key}")


def make_state(kind: str, cfg: dict, seed: int = 0, style: str = "init", **recipe) -> Dict[str, np.ndarray]:
    """style "init": ranges of torch's default initialisers (above).  style "trained": the same tensors reshaped towards
    the statistics of a trained separator, where an arithmetic that leans on benign magnitudes would show it --
      * every LSTM weight row times its own factor, log-uniform in [1, 30] (heavy-tailed rows: a few units dominate a gate);
      * forget-gate biases (rows H .. 2H of bias_ih: gate order i, f, g, o) + U(1, 3) (the usual remember-by-default offset);
      * norm gains log-uniform in [0.2, 8] (intra / inter GroupNorm, bottleneck norm), shifts U(-0.5, 0.5): the residual stream
        grows over the 12 residual additions instead of staying O(1)."""
    if style == "trained":
        return _trained_like(kind, cfg, seed, **recipe)
    if style != "init":
        raise ValueError(f"unknown weight style {style!r}")
    cfg = full_cfg(**cfg)
    spec = state_spec(kind, cfg)
    shapes = dict(spec)
    out: Dict[str, np.ndarray] = {}
    H = cfg["hidden_size"]
    for key, shape in spec:
        leaf = key.rsplit(".", 1)[-1]
        if ".rnn." in key:
            a = 1.0 / np.sqrt(H)
            out[key] = ((2 * _rng(seed, key).random(shape) - 1) * a).astype(np.float32)
            continue
        is_plain_bias = (leaf == "bias" and len(shape) == 1 and not (
            "norm" in key or "bottleneck.0" in key or "spk_encoder.0" in key or ".average." in key))
        if is_plain_bias:
            wshape = shapes[key[: -len("bias")] + "weight"]
            a = 1.0 / np.sqrt(int(np.prod(wshape[1:])))
            out[key] = ((2 * _rng(seed, key).random(shape) - 1) * a).astype(np.float32)
            continue
        out[key] = make_tensor(key, shape, seed)
    return out


TRAINED = dict(r_ih=30.0, r_hh=30.0, g_lo=0.2, g_hi=8.0, f_lo=1.0, f_hi=3.0)


def _trained_like(kind: str, cfg: dict, seed: int, **recipe) -> Dict[str, np.ndarray]:
    rc = dict(TRAINED, **recipe)
    cfg = full_cfg(**cfg)
    H = cfg["hidden_size"]
    out = make_state(kind, cfg, seed, style="init")
    for key in list(out):
        g = _rng(seed + 7919, key)
        v = out[key]
        leaf = key.rsplit(".", 1)[-1]
        if ".rnn.weight_" in key:
            rows = np.exp(g.uniform(0.0, np.log(rc["r_ih"] if "weight_ih" in key else rc["r_hh"]), size=(v.shape[0], 1))).astype(np.float32)
            out[key] = (v * rows).astype(np.float32)
        elif ".rnn.bias_ih" in key and cfg["rnn_type"] == "LSTM":
            v = v.copy()
            v[H: 2 * H] += g.uniform(rc["f_lo"], rc["f_hi"], size=H).astype(np.float32)
            out[key] = v
        elif ("_norm." in key or "bottleneck.0." in key) and leaf in ("weight", "gamma"):
            out[key] = np.exp(g.uniform(np.log(rc["g_lo"]), np.log(rc["g_hi"]), size=v.shape)).astype(np.float32)
        elif ("_norm." in key or "bottleneck.0." in key) and leaf in ("bias", "beta"):
            out[key] = g.uniform(-0.5, 0.5, size=v.shape).astype(np.float32)
    return out


def make_mixture(batch: int, length: int, seed: int = 1234, scale: float = 0.1) -> np.ndarray:
    """Synthetic waveform batch: scale * N(0,1), float32, numpy-seeded."""
    g = np.random.default_rng([seed, batch, length])
    return (scale * g.standard_normal((batch, length))).astype(np.float32)


def make_sources(batch: int, length: int, seed: int = 1236, scale: float = 0.07):
    """Two synthetic sources and their mixture (for SI-SDR parity: SURVEY.md 8d)."""
    s = np.stack([make_mixture(batch, length, seed + i, scale) for i in range(2)], axis=1)
    return s, s.sum(axis=1).astype(np.float32)
