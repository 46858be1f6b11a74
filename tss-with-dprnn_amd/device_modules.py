"""torch.nn modules of the reference's separators whose ``forward`` runs on the HIP kernels.

The TasNet wrappers never call these forwards (their whole forward is fused from the waveform: engine.tasnet_forward /
spe_forward); they exist for code that holds encoder outputs and drives the separator's pieces itself, as the reference's
``DPRNNRawNet(DPRNNSpe)`` does with ``self.bottleneck[0]``, ``self._fusion``, ``self.bottleneck[1]``, ``self.pred_linear``
(src/models/dprnn_rawnet.py:79-105).  Each class IS the torch module it mirrors (same constructor, parameters, ``state_dict``
keys), so checkpoints and optimizers see no difference; only ``forward`` is replaced: device tensors, inference only (training
goes through the TasNet wrappers, which own the backward kernels), no CPU fallback."""
from torch import nn

from . import engine


class GroupNorm(nn.GroupNorm):
    """nn.GroupNorm(1, C) of dprnn.py:76-77,131-136: dprnn_rows_stats + dprnn_bottleneck_prep + dprnn_rows_affine_fwd."""

    def forward(self, x):
        if self.num_groups != 1:
            raise NotImplementedError("GroupNorm MI355X path: one group (the reference's global norm)")
        return engine.norm_forward(self, x)


class Conv1d(nn.Conv1d):
    """1x1 nn.Conv1d (dprnn.py:135-136, dprnn_spe.py:90-105): dprnn_pointwise_fwd."""

    def forward(self, x):
        return engine.conv1x1_forward(self, x)


class Linear(nn.Linear):
    """Per-utterance nn.Linear (dprnn_spe.py:88-105,123; dprnn_spe_ira.py:51): dprnn_linear_small_fwd."""

    def forward(self, x):
        return engine.linear_forward(self, x)
