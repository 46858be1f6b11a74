"""DPRNN-Spe-IRA on the MI355X path: API mirror of the reference's src/models/dprnn_spe_ira.py
(two separation passes; the second speaker embedding is Linear(cat(v0, spk_enc(mask0 * enc))),
dprnn_spe_ira.py:53-115)."""
from torch import nn

from ._lib import DprnnError
from .dprnn_spe import DPRNNSpe, DPRNNSpeTasNet


class DPRNNSpeIRA(DPRNNSpe):
    """Parameter holder: DPRNNSpe + ``aux_linear`` (dprnn_spe_ira.py:51)."""

    def __init__(self, input_size, feature_size=128, hidden_size=128, chunk_length=200, hop_length=None,
                 n_repeats=6, bidirectional=True, rnn_type='LSTM', norm_type='gLN', activation_type='sigmoid',
                 dropout=0, O=128, P=256, embeddings_size=128, num_spks=251, kernel_size=2, fusion_type='cat'):
        super().__init__(input_size, feature_size, hidden_size, chunk_length, hop_length, n_repeats, bidirectional,
                         rnn_type, norm_type, activation_type, dropout, O, P, embeddings_size, num_spks, kernel_size,
                         fusion_type)
        self.aux_linear = nn.Linear(2 * embeddings_size, embeddings_size)

    def forward(self, input, aux, aux_len):
        raise DprnnError("DPRNNSpeIRA consumes raw waveforms on the MI355X path; call DPRNNSpeIRATasNet")


class DPRNNSpeIRATasNet(DPRNNSpeTasNet):
    """DPRNN-Spe-IRA-TasNet (dprnn_spe_ira.py:117-190): same signature as DPRNNSpeTasNet."""

    _separator_cls = DPRNNSpeIRA
    _ira = True
