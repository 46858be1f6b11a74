"""DPRNN-Spe on the MI355X path: API mirror of the reference's src/models/dprnn_spe.py.

Same class names, constructor signatures, registration order and state_dict keys (incl. the BatchNorm
buffers) as the reference; the modules hold parameters, ``DPRNNSpeTasNet.forward`` enqueues the HIP kernels
(engine.spe_forward).  BatchNorm1d follows ``module.training`` exactly like torch: the reference's
Inferencer never calls ``.eval()`` (src/inferencers/inferencer_spe.py:20-32), so batch statistics -- and the
in-place running-statistics update -- are supported too.  ``DPRNNTasNet`` is re-exported because
scripts/test/config_bss.yaml:12 targets ``src.models.dprnn_spe.DPRNNTasNet``.
"""
import torch
from torch import nn

from . import device_modules as dm
from . import engine, norms
from ._lib import DprnnError
from .dprnn import DPRNN, DPRNNTasNet  # noqa: F401  (DPRNNTasNet: see module docstring)
from .encoder_decoder import Decoder, Encoder


class ResBlock(nn.Module):
    """ResNet block of the speaker encoder (dprnn_spe.py:8-42): 1x1 conv, BN, PReLU, 1x1 conv, BN, (+x | +conv(x)),
    PReLU, MaxPool1d(3).  Evaluated by dprnn_pointwise_fwd / dprnn_bn_prep / dprnn_resblock_pool_fwd."""

    def __init__(self, in_dims, out_dims):
        super().__init__()
        self.conv1 = nn.Conv1d(in_dims, out_dims, kernel_size=1, bias=False)
        self.conv2 = nn.Conv1d(out_dims, out_dims, kernel_size=1, bias=False)
        self.batch_norm1 = nn.BatchNorm1d(out_dims)
        self.batch_norm2 = nn.BatchNorm1d(out_dims)
        self.prelu1 = nn.PReLU()
        self.prelu2 = nn.PReLU()
        self.maxpool = nn.MaxPool1d(3)
        self.downsample = in_dims != out_dims
        if self.downsample:
            self.conv_downsample = nn.Conv1d(in_dims, out_dims, kernel_size=1, bias=False)

    def forward(self, x):
        """dprnn_spe.py:31-42: [B, C_in, T] -> [B, C_out, T // 3] (inference only; BatchNorm follows module.training)."""
        return engine.resblock_forward(self, x)


class DPRNNSpe(DPRNN):
    """DPRNN with speaker-embedding fusion (dprnn_spe.py:44-248) -- parameter holder."""

    def __init__(self, input_size, feature_size=128, hidden_size=128, chunk_length=200, hop_length=None,
                 n_repeats=6, bidirectional=True, rnn_type='LSTM', norm_type='gLN', activation_type='sigmoid',
                 dropout=0, O=128, P=256, embeddings_size=128, num_spks=251, kernel_size=2, fusion_type='cat'):
        super().__init__(input_size, feature_size, hidden_size, chunk_length, hop_length, n_repeats, bidirectional,
                         rnn_type, norm_type, activation_type, dropout=dropout)
        self.kernel_size = kernel_size
        self.fusion_type = fusion_type
        start_conv1d = None
        if fusion_type == 'cat':
            start_conv1d = dm.Conv1d(input_size + embeddings_size, feature_size, 1)
        if fusion_type in ('add', 'mul'):
            self.fusion_linear = dm.Linear(embeddings_size, input_size)
            start_conv1d = dm.Conv1d(input_size, feature_size, 1)
        if fusion_type == 'film':
            self.fusion_linear_1 = dm.Linear(embeddings_size, input_size)
            self.fusion_linear_2 = dm.Linear(embeddings_size, input_size)
            start_conv1d = dm.Conv1d(input_size, feature_size, 1)
        if fusion_type == 'att':
            self.fusion_linear = dm.Linear(embeddings_size, input_size)
            self.average = nn.Conv1d(input_size, input_size, kernel_size, kernel_size, groups=input_size)
            self.average.weight = nn.Parameter(torch.ones(input_size, 1, kernel_size) / kernel_size)
            self.average.bias = nn.Parameter(torch.zeros(input_size))
            for p in self.average.parameters():
                p.requires_grad = False
            start_conv1d = dm.Conv1d(input_size, feature_size, 1)
        if start_conv1d is None:
            raise UnboundLocalError("local variable 'start_conv1d' referenced before assignment")   # as the reference does
        linear_norm = norms.GlobLN(input_size) if norm_type == 'gLN' else dm.GroupNorm(1, input_size)   # dprnn_spe.py:108-111
        self.bottleneck = nn.Sequential(linear_norm, start_conv1d)
        self.spk_encoder = nn.Sequential(
            nn.GroupNorm(1, input_size),
            nn.Conv1d(input_size, O, 1),
            ResBlock(O, O),
            ResBlock(O, P),
            ResBlock(P, P),
            nn.Conv1d(P, embeddings_size, 1),
        )
        self.pred_linear = dm.Linear(embeddings_size, num_spks)

    # Separation-level entry points for code that holds encoder outputs [B, N, L] and drives the pieces itself -- the
    # reference's DPRNNRawNet subclass overrides _auxiliary and calls bottleneck[0], _fusion, bottleneck[1], _dprnn_process and
    # pred_linear (src/models/dprnn_rawnet.py:79-105).  Every piece runs on the library's HIP kernels: the speaker encoder and the
    # DPRNN stack (99 % of the FLOPs) as in the fused path, the per-frame glue through device_modules (norm: dprnn_rows_stats +
    # dprnn_bottleneck_prep + dprnn_rows_affine_fwd; 1x1 conv: dprnn_pointwise_fwd; Linears: dprnn_linear_small_fwd) and
    # engine.fusion_forward (dprnn_rows_affine_fwd, dprnn_att_row_scale_rows); what torch does here is layout (transposes, the
    # 'cat' copy).  Inference only.  DPRNNSpeTasNet / DPRNNSpeIRATasNet do NOT use these: their forward is fused from the waveform.
    def forward(self, input, aux, aux_len):
        """dprnn_spe.py:125-154: (mixture encoding [B,N,L], reference encoding [B,N,La], aux_len) -> (masks [B,2,N,L],
        speaker logits [B,num_spks])."""
        B, _, L = input.size()
        emb = self._auxiliary(aux, aux_len)
        output = self.bottleneck[0](input)
        output = self._fusion(emb, output, L)
        output = self.bottleneck[1](output)
        return self._dprnn_process(output, B, L), self.pred_linear(emb)

    def _auxiliary(self, aux, aux_len):
        """dprnn_spe.py:156-163: speaker encoder + time mean, [B,N,La] -> [B,E] (HIP kernels)."""
        return engine.auxiliary_embedding(self, aux, aux_len)

    def _fusion(self, aux, output, L):
        """dprnn_spe.py:165-229 on device tensors: aux [B,E], output [B,N,L] (already normalised) -> fused [B,C,L]."""
        return engine.fusion_forward(self, aux, output, L)


def _wants_grad(module):
    return torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters())


class DPRNNSpeTasNet(nn.Module):
    """DPRNN-Spe-TasNet (dprnn_spe.py:250-327): forward(input[B,T], aux[B,Ta], aux_len) -> (est[B,T], logits)."""

    _separator_cls = DPRNNSpe

    def __init__(self, input_size, feature_size=128, hidden_size=128, chunk_length=200, kernel_size=2,
                 hop_length=None, n_repeats=6, bidirectional=True, rnn_type='LSTM', norm_type='gLN',
                 activation_type='sigmoid', dropout=0, stride=None, O=128, P=256, embeddings_size=128,
                 num_spks=251, fusion_type='cat'):
        super().__init__()
        self.stride = stride if stride is not None else kernel_size // 2
        self.encoder = Encoder(kernel_size, input_size, stride=self.stride, bias=False)
        self.separation = self._separator_cls(input_size, feature_size, hidden_size, chunk_length, hop_length,
                                              n_repeats, bidirectional, rnn_type, norm_type, activation_type, dropout,
                                              O, P, embeddings_size, num_spks, kernel_size, fusion_type)
        self.decoder = Decoder(in_channels=input_size, out_channels=1, kernel_size=kernel_size, stride=self.stride,
                               bias=False)
        self.cfg = dict(input_size=input_size, feature_size=feature_size, hidden_size=hidden_size,
                        chunk_length=chunk_length, kernel_size=kernel_size, hop_length=self.separation.hop_length,
                        n_repeats=n_repeats, bidirectional=bool(bidirectional), rnn_type=rnn_type, norm_type=norm_type,
                        activation_type=activation_type, dropout=dropout, stride=self.stride, O=O, P=P,
                        embeddings_size=embeddings_size, num_spks=num_spks, fusion_type=fusion_type)

    _ira = False

    def forward(self, input, aux, aux_len, debug=None):
        if debug is None and _wants_grad(self):
            from . import training_spe              # differentiable forward: autograd.Function over the HIP kernels
            return training_spe.spe_forward_autograd(self, input, aux, aux_len)
        return engine.spe_forward(self, input, aux, aux_len, ira=self._ira, debug=debug)
