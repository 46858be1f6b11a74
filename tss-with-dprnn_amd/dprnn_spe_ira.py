"""DPRNN-Spe-IRA on the MI355X path: API mirror of the reference's src/models/dprnn_spe_ira.py
(two separation passes; the second speaker embedding is Linear(cat(v0, spk_enc(mask0 * enc))),
dprnn_spe_ira.py:53-115)."""
import torch
from torch import nn

from ._lib import DprnnError
from . import device_modules as dm
from .dprnn_spe import DPRNNSpe, DPRNNSpeTasNet


class DPRNNSpeIRA(DPRNNSpe):
    """Parameter holder: DPRNNSpe + ``aux_linear`` (dprnn_spe_ira.py:51)."""

    def __init__(self, input_size, feature_size=128, hidden_size=128, chunk_length=200, hop_length=None,
                 n_repeats=6, bidirectional=True, rnn_type='LSTM', norm_type='gLN', activation_type='sigmoid',
                 dropout=0, O=128, P=256, embeddings_size=128, num_spks=251, kernel_size=2, fusion_type='cat'):
        super().__init__(input_size, feature_size, hidden_size, chunk_length, hop_length, n_repeats, bidirectional,
                         rnn_type, norm_type, activation_type, dropout, O, P, embeddings_size, num_spks, kernel_size,
                         fusion_type)
        self.aux_linear = dm.Linear(2 * embeddings_size, embeddings_size)

    def forward(self, input, aux, aux_len):
        """dprnn_spe_ira.py:53-115 on encoder outputs: (mixture encoding [B,N,L], reference encoding [B,N,La], aux_len) ->
        (d1 = mask_0 * input of the second pass [B,N,L], speaker logits [B,num_spks]).  Built from the same device pieces as
        DPRNNSpe.forward (all of them HIP kernels; mask * encoding and the embedding concatenation are torch element-wise /
        copy ops); inference only.  Note the reference normalises the embedding of d0 with the REFERENCE utterance's aux_len (:84): kept."""
        B, _, L = input.size()
        v0 = self._auxiliary(aux, aux_len)
        output_norm = self.bottleneck[0](input)                              # cached for the second pass (:62,:96)
        masks = self._dprnn_process(self.bottleneck[1](self._fusion(v0, output_norm, L)), B, L)
        d0 = (masks * input.unsqueeze(1))[:, 0, :, :].contiguous()
        v1 = self.aux_linear(torch.cat((v0, self._auxiliary(d0, aux_len)), dim=1))
        masks = self._dprnn_process(self.bottleneck[1](self._fusion(v1, output_norm, L)), B, L)
        d1 = (masks * input.unsqueeze(1))[:, 0, :, :]
        return d1, self.pred_linear(v1)


class DPRNNSpeIRATasNet(DPRNNSpeTasNet):
    """DPRNN-Spe-IRA-TasNet (dprnn_spe_ira.py:117-190): same signature as DPRNNSpeTasNet."""

    _separator_cls = DPRNNSpeIRA
    _ira = True
