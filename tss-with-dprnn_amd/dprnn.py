"""DPRNN-TasNet on the MI355X path: API mirror of the reference's src/models/dprnn.py.

Class names, constructor signatures (positional order included, dprnn.py:250-262), sub-module
attribute names and therefore state_dict keys are the reference's, so ``hydra.utils.instantiate``
of the shipped YAML (scripts/train/config_bss.yaml:16-28) and ``load_state_dict(ckpt['model'])``
(src/trainers/trainer.py:72, src/inferencers/inferencer.py:38) work unchanged.  The sub-modules
are parameter holders; ``DPRNNTasNet.forward`` enqueues the hand-written HIP kernels
(engine.tasnet_forward) and nothing else -- there is no PyTorch/CPU fallback.
"""
import math
import numbers
import warnings

import torch
from torch import nn

from . import device_modules as dm
from . import engine, norms
from ._lib import DprnnError
from .encoder_decoder import Decoder, Encoder

_GATES = {"LSTM": 4, "GRU": 3, "RNN": 1}


class _RNNParams(nn.Module):
    """Holds the tensors of a 1-layer batch_first torch.nn.{LSTM,GRU,RNN} under the same names
    (weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0 [+ _reverse]) and the same default
    init U(-1/sqrt(H), 1/sqrt(H)); gate order i,f,g,o for LSTM."""

    def __init__(self, rnn_type, input_size, hidden_size, bidirectional, dropout=0):
        super().__init__()
        if rnn_type not in _GATES:
            raise AttributeError(f"module 'torch.nn' has no attribute '{rnn_type}'")   # getattr(nn, rnn_type), dprnn.py:23
        # dropout reaches a num_layers=1 torch RNN in the reference (dprnn.py:23-28), where torch validates the value,
        # warns, and never applies it (dropout acts between stacked layers only): same checks, same warning, same no-op
        if not isinstance(dropout, numbers.Number) or not 0 <= dropout <= 1 or isinstance(dropout, bool):
            raise ValueError("dropout should be a number in range [0, 1] representing the probability of an element being "
                             "zeroed")
        if dropout > 0:
            warnings.warn("dropout option adds dropout after all but last recurrent layer, so non-zero dropout expects "
                          f"num_layers greater than 1, but got dropout={dropout} and num_layers=1")
        self.dropout = float(dropout)
        self.mode, self.input_size, self.hidden_size = rnn_type, input_size, hidden_size
        self.bidirectional = bool(bidirectional)
        g = _GATES[rnn_type] * hidden_size
        for suf in ([""] + (["_reverse"] if self.bidirectional else [])):
            self.register_parameter("weight_ih_l0" + suf, nn.Parameter(torch.empty(g, input_size)))
            self.register_parameter("weight_hh_l0" + suf, nn.Parameter(torch.empty(g, hidden_size)))
            self.register_parameter("bias_ih_l0" + suf, nn.Parameter(torch.empty(g)))
            self.register_parameter("bias_hh_l0" + suf, nn.Parameter(torch.empty(g)))
        self.reset_parameters()

    def reset_parameters(self):
        a = 1.0 / math.sqrt(self.hidden_size) if self.hidden_size > 0 else 0
        for p in self.parameters():
            nn.init.uniform_(p, -a, a)

    def flatten_parameters(self):       # SingleRNN.forward calls it (dprnn.py:34); nothing to flatten here
        return None

    def forward(self, *a, **k):
        raise DprnnError("the recurrence runs in dprnn_bilstm_fwd; call the TasNet wrapper")


class SingleRNN(nn.Module):
    """Single RNN layer (dprnn.py:7-37): parameters under ``.rnn``."""

    def __init__(self, rnn_type, input_size, hidden_size, dropout=0, bidirectional=False):
        super().__init__()
        self.hidden_size = hidden_size
        self.bidirectional = bidirectional
        self.rnn = _RNNParams(rnn_type, input_size, hidden_size, bidirectional, dropout)

    def output_size(self):
        return self.hidden_size * (2 if self.bidirectional else 1)

    def forward(self, input):
        """dprnn.py:33-37: [sequences, steps, input_size] -> [sequences, steps, output_size()] (inference only; HIP recurrence)."""
        return engine.single_rnn_forward(self, input)


def _make_norm(norm_type, channels):
    if norm_type == "gLN":
        return norms.GlobLN(channels)
    if norm_type == "ln":
        return dm.GroupNorm(1, channels)     # nn.GroupNorm whose forward runs on the HIP kernels (device_modules.py)
    return None           # the reference silently leaves the attribute undefined (dprnn.py:72-77)


class DPRNNBlock(nn.Module):
    """Dual-Path RNN block (dprnn.py:39-99): intra BiLSTM -> Linear -> norm -> +, inter likewise."""

    def __init__(self, rnn_type, feature_size, hidden_size, norm_type='gLN', dropout=0, bidirectional=True):
        super().__init__()
        self.intra_rnn = SingleRNN(rnn_type, feature_size, hidden_size, dropout=dropout, bidirectional=True)
        self.intra_linear = nn.Linear(self.intra_rnn.output_size(), feature_size)
        self.inter_rnn = SingleRNN(rnn_type, feature_size, hidden_size, dropout=dropout, bidirectional=bidirectional)
        self.inter_linear = nn.Linear(self.inter_rnn.output_size(), feature_size)
        n1, n2 = _make_norm(norm_type, feature_size), _make_norm(norm_type, feature_size)
        if n1 is not None:
            self.intra_norm, self.inter_norm = n1, n2

    def forward(self, input):
        """dprnn.py:79-99: [B, F, K, S] -> [B, F, K, S] (inference only; the TasNet wrappers run the fused form of this)."""
        return engine.block_forward(self, input)


class DPRNN(nn.Module):
    """Dual-Path RNN separator (dprnn.py:101-217) -- parameter holder with the reference's layout."""

    def __init__(self, input_size, feature_size=128, hidden_size=128, chunk_length=200, hop_length=None,
                 n_repeats=6, bidirectional=True, rnn_type='LSTM', norm_type='gLN',
                 activation_type='sigmoid', dropout=0):
        super().__init__()
        self.input_size = input_size
        self.feature_size = feature_size
        self.chunk_length = chunk_length
        self.hop_length = hop_length if hop_length is not None else chunk_length // 2   # dprnn.py:128
        self.norm_type = norm_type
        self.activation_type = activation_type

        linear_norm = _make_norm(norm_type, input_size)
        start_conv1d = dm.Conv1d(input_size, feature_size, 1)
        self.bottleneck = nn.Sequential(*([linear_norm] if linear_norm is not None else []), start_conv1d)
        self.dprnn_blocks = nn.Sequential(*[
            DPRNNBlock(rnn_type=rnn_type, feature_size=feature_size, hidden_size=hidden_size, norm_type=norm_type,
                       dropout=dropout, bidirectional=bidirectional) for _ in range(n_repeats)])
        self.prelu = nn.PReLU()
        self.conv2d = nn.Conv2d(feature_size, feature_size * 2, kernel_size=1)
        self.out = nn.Sequential(nn.Conv1d(feature_size, feature_size, 1), nn.Tanh())
        self.gate = nn.Sequential(nn.Conv1d(feature_size, feature_size, 1), nn.Sigmoid())
        self.end_conv1x1 = nn.Conv1d(feature_size, input_size, 1, bias=False)
        if activation_type == 'sigmoid':
            self.activation = nn.Sigmoid()
        if activation_type == 'relu':
            self.activation = nn.ReLU()

    # Separation-level entry points.  The TasNet wrappers never come here (they start from the waveform and never
    # materialise the encoder output); these exist for code that holds [B, C, L] tensors and calls the separator or its
    # pieces itself, as the reference's DPRNNRawNet subclass does (src/models/dprnn_rawnet.py:79-94).  Inference only.
    def forward(self, input):
        """dprnn.py:166-187: encoder output [B, N, L] -> masks [B, 2, N, L] (norm, 1x1 conv and the whole DPRNN stack in
        HIP kernels)."""
        return engine.separator_forward(self, input)

    def _dprnn_process(self, output, B, L):
        """dprnn.py:171-186: bottleneck output [B, F, L] -> masks [B, 2, N, L]."""
        if tuple(output.shape[::2]) != (B, L):
            raise DprnnError(f"_dprnn_process: got {tuple(output.shape)} for B={B}, L={L}")
        return engine.separator_masks(self, output)


class DPRNNTasNet(nn.Module):
    """DPRNN-TasNet (dprnn.py:219-283): forward(input[B, T]) -> [B, 2, T]."""

    def __init__(self, input_size, feature_size=128, hidden_size=128, chunk_length=200, kernel_size=2,
                 hop_length=None, n_repeats=6, bidirectional=True, rnn_type='LSTM', norm_type='ln',
                 activation_type='sigmoid', dropout=0, stride=None):
        super().__init__()
        self.stride = stride if stride is not None else kernel_size // 2                # dprnn.py:243
        self.encoder = Encoder(kernel_size, input_size, stride=self.stride, bias=False)
        self.separation = DPRNN(input_size, feature_size, hidden_size, chunk_length, hop_length, n_repeats,
                                bidirectional, rnn_type, norm_type, activation_type, dropout)
        self.decoder = Decoder(in_channels=input_size, out_channels=1, kernel_size=kernel_size,
                               stride=self.stride, bias=False)
        self.cfg = dict(input_size=input_size, feature_size=feature_size, hidden_size=hidden_size,
                        chunk_length=chunk_length, kernel_size=kernel_size, hop_length=self.separation.hop_length,
                        n_repeats=n_repeats, bidirectional=bool(bidirectional), rnn_type=rnn_type,
                        norm_type=norm_type, activation_type=activation_type, dropout=dropout, stride=self.stride)

    def forward(self, input, debug=None):
        # As in torch, what decides whether activations are kept for a backward is grad mode, not model.training: eval() without
        # torch.no_grad() keeps the tape (about 4 GB per 3-s utterance) -- evaluate under no_grad / inference_mode, as the
        # reference's inferencer does (src/inferencers/inferencer.py:53).
        if debug is None and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from . import training          # differentiable forward: autograd.Function over the HIP kernels
            return training.tasnet_forward_autograd(self, input)
        return engine.tasnet_forward(self, input, debug)
