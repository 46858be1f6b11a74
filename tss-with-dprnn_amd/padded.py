"""Narrow configurations on the kernels' built widths (round 4; SURVEY.md section 8 f4: "API completeness of the kept surface").

The HIP kernels are built for input_size 64, feature_size = hidden_size 128 (speaker encoder: O 128, P 256, embeddings 128) --
every shipped YAML.  The reference's constructors take any width (dprnn.py:117-120, dprnn_spe.py:44-50).  A NARROWER model is
computed exactly by the built kernels on a zero-padded twin:

* a hidden unit whose weights and biases are all zero stays at c = 0, h = 0 for ever (LSTM: i = f = o = 1/2, g = 0; GRU: n = 0;
  Elman: tanh(0)), a channel whose norm gain and shift are zero stays 0 after every norm, a zero encoder filter gives relu(0) = 0,
  and zero terms add nothing to any dot product -- so the real channels of every tensor hold exactly what the narrow model computes
  and the padded ones hold zeros;
* the only thing that sees the padding is the ELEMENT COUNT of the global norms (GroupNorm(1, C) / gLN divide by rows x C): the
  engine multiplies the (sum, sum of squares) pair of every such norm by built / real channels, which turns S / (rows x built) into
  S / (rows x real) for the mean and the second moment alike (``count_scale`` below; powers of two for the usual widths).

``twin_of(model)`` builds (once) a model of the same class at the built widths, fills it from the narrow model's ``state_dict``
by placing every tensor's blocks -- gate blocks of the RNN weights, direction blocks of the Linears after a recurrence, speaker blocks of
the mask head's conv2d, the [x | embedding] column blocks of the 'cat' bottleneck and of aux_linear -- at the built offsets, and
refreshes it in place whenever a parameter of the narrow model changed (``_version`` counters: the addresses the captured graphs
hold stay valid).  BatchNorm buffers a train-mode forward updates are written back.  TRAINING (round 5): the training
kernels run the twin as well -- the same element-count correction on the two sums of every global norm's backward
(training._count_scale_red) -- and ``placed_parameters`` makes the twin's parameters differentiable functions of the narrow ones, so
``loss.backward()`` leaves each narrow parameter the real block of its twin's gradient (DPRNNTasNet, DPRNNSpeTasNet and
DPRNNSpeIRATasNet: tests/golden/grad_*_narrow_*.npz hold the reference's gradients).  WIDER than built: inference runs on
the any-width kernels (wide.py, csrc/wide.hip: all three model classes); training at such widths raises here.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

N_BUILT, F_BUILT = 64, 128
SPE_BUILT = dict(O=128, P=256, embeddings_size=128)


def needs_twin(cfg: dict) -> bool:
    if cfg["input_size"] != N_BUILT or cfg["feature_size"] != F_BUILT or cfg["hidden_size"] != F_BUILT:
        return True
    return any(k in cfg and cfg[k] != v for k, v in SPE_BUILT.items())


def fits(cfg: dict) -> bool:
    ok = 1 <= cfg["input_size"] <= N_BUILT and 1 <= cfg["feature_size"] <= F_BUILT and 1 <= cfg["hidden_size"] <= F_BUILT
    return ok and all(k not in cfg or 1 <= cfg[k] <= v for k, v in SPE_BUILT.items())


def _blocks(name: str, shape, cfg: dict):
    """How tensor `name` of the narrow model splits into blocks along each dimension: per dim a list of (real size) blocks; block b of
    dim d lands at offset b * (built size of the dim / number of blocks)... expressed as [(narrow_start, size, built_start)]."""
    N, F, H = cfg["input_size"], cfg["feature_size"], cfg["hidden_size"]
    E = cfg.get("embeddings_size", 0)
    leaf = name.rsplit(".", 1)[-1]
    dims = [[(0, s, 0)] for s in shape]                       # default: the leading corner
    if ".rnn." in name:                                      # [G*H, F | H] weights and [G*H] biases: G gate blocks of H rows
        G = shape[0] // H
        dims[0] = [(g * H, H, g * F_BUILT) for g in range(G)]
    elif name.endswith(("intra_linear.weight", "inter_linear.weight")):      # [F, ndir * H]: direction blocks of the columns
        nd = shape[1] // H
        dims[1] = [(d * H, H, d * F_BUILT) for d in range(nd)]
    elif ".conv2d." in name:                                 # [2F, F, 1, 1] / [2F]: speaker blocks of the rows
        dims[0] = [(k * F, F, k * F_BUILT) for k in range(shape[0] // F)]
    elif name.endswith("bottleneck.1.weight") and len(shape) == 3 and shape[1] == N + E and E:      # 'cat': [F, N + E, 1]
        dims[1] = [(0, N, 0), (N, E, N_BUILT)]
    elif name.endswith("aux_linear.weight"):                 # [E, 2E]: (v0 | v1) column blocks
        dims[1] = [(0, E, 0), (E, E, SPE_BUILT["embeddings_size"])]
    del leaf
    return dims


def _place(dst: torch.Tensor, src: torch.Tensor, dims, copy: bool = True) -> None:
    import itertools
    for combo in itertools.product(*dims):
        s_idx = tuple(slice(a, a + n) for a, n, _ in combo)
        d_idx = tuple(slice(b, b + n) for _, n, b in combo)
        if copy:
            dst[d_idx].copy_(src[s_idx])
        else:
            dst[d_idx] = src[s_idx]          # index assignment: recorded by autograd when src requires grad


def placed_parameters(model, tw: "Twin") -> Dict[str, torch.Tensor]:
    """The twin's parameters as differentiable functions of the narrow model's (training): every narrow parameter placed into zeros of
    the built shape with autograd watching, so the gradient of a twin parameter flows back to the blocks that are real."""
    built = {k: tuple(v.shape) for k, v in tw.model.named_parameters()}
    out = {}
    for k, v in model.named_parameters():
        if built[k] == tuple(v.shape):
            out[k] = v
            continue
        d = v.new_zeros(built[k])
        _place(d, v, _blocks(k, tuple(v.shape), tw.narrow_cfg), copy=False)
        out[k] = d
    return out


class Twin:
    """The zero-padded model of a narrow one, kept in step with it."""

    def __init__(self, model):
        cfg = dict(model.cfg)
        if not fits(cfg):
            raise NotImplementedError(
                "DPRNN MI355X path: widths above the kernels' build (input_size 64, feature_size / hidden_size 128; speaker encoder "
                f"O 128, P 256, embeddings 128) are not supported by the fused / training kernels: got input_size {cfg['input_size']}, feature_size "
                f"{cfg['feature_size']}, hidden_size {cfg['hidden_size']}.  (INFERENCE at such widths runs on the any-width kernels, wide.py -- "
                "call the model under torch.no_grad(); the speaker-conditioned classes need input_size 64 and the built speaker encoder "
                "there.  Training stops at the built widths.)")
        self.narrow_cfg = cfg
        wide = dict(cfg, input_size=N_BUILT, feature_size=F_BUILT, hidden_size=F_BUILT)
        for k, v in SPE_BUILT.items():
            if k in wide:
                wide[k] = v
        import inspect
        keys = [k for k in inspect.signature(type(model).__init__).parameters if k in wide]
        dev = next(model.parameters()).device
        self.model = type(model)(**{k: wide[k] for k in keys}).to(dev)
        self.model.train(model.training)
        self.model.requires_grad_(False)
        # every global norm of the twin divides by rows x BUILT channels: the engine multiplies its (sum, sum of squares) by this
        for (_, a), (_, b) in zip(self.model.named_modules(), model.named_modules()):
            ca = getattr(a, "num_channels", getattr(a, "channel_size", None))
            cb = getattr(b, "num_channels", getattr(b, "channel_size", None))
            if ca is not None and cb is not None and not isinstance(a, torch.nn.modules.batchnorm._BatchNorm):
                a.__dict__["_dprnn_count_scale"] = float(ca) / float(cb)
        self.versions: Tuple = ()
        self.sync(model)

    def _state(self, model) -> Dict[str, torch.Tensor]:
        return dict(model.state_dict(keep_vars=True))

    def sync(self, model) -> None:
        src = self._state(model)
        versions = tuple((k, v.data_ptr(), v._version) for k, v in src.items())
        modes = tuple(m.training for m in model.modules())
        if (versions, modes) == self.versions:
            return
        dst = self._state(self.model)
        with torch.no_grad():
            for k, v in src.items():
                d = dst[k]
                if d.device != v.device:
                    raise RuntimeError(f"{k}: the narrow model moved to {v.device}; its twin lives on {d.device}")
                if d.dim() == 0:
                    d.copy_(v)
                    continue
                d.zero_()
                _place(d, v.detach(), _blocks(k, tuple(v.shape), self.narrow_cfg))
        for a, b in zip(self.model.modules(), model.modules()):
            a.training = b.training
        self.versions = (versions, modes)

    def write_back_buffers(self, model) -> None:
        """BatchNorm running statistics a train-mode forward of the twin updated -> the narrow model (its real channels)."""
        bufs = dict(model.named_buffers())
        if not bufs:
            return
        twin_bufs = dict(self.model.named_buffers())
        with torch.no_grad():
            for k, b in bufs.items():
                t = twin_bufs[k]
                if b.dim() == 0:
                    b.copy_(t)
                else:
                    b.copy_(t[tuple(slice(0, n) for n in b.shape)])
        self.versions = ()                                   # the narrow model changed: re-read it next time (cheap)


def twin_of(model):
    """The model the engine runs: `model` itself at the built widths, else its zero-padded twin (built on first use, refreshed when
    a parameter changed)."""
    if not needs_twin(model.cfg):
        return model, None
    tw = model.__dict__.get("_dprnn_twin")
    dev = next(model.parameters()).device
    if tw is None or next(tw.model.parameters()).device != dev:
        tw = model.__dict__["_dprnn_twin"] = Twin(model)
    else:
        tw.sync(model)
    return tw.model, tw
