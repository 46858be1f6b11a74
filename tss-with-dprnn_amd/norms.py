"""Parameter holders for the global layer norm of the reference (src/models/norms.py:16-31).

``GlobLN`` keeps the reference's attribute names (``gamma``, ``beta``) so checkpoints load with
strict=True.  The arithmetic (z-norm over all non-batch dims, eps 1e-8, biased variance,
norms.py:6-14) runs inside the HIP kernels: dprnn_bottleneck_prep / dprnn_proj_stats_fwd +
dprnn_norm_residual_fwd with eps=1e-8.
"""
import torch
from torch import nn

from ._lib import DprnnError

GLN_EPS = 1e-8
LN_EPS = 1e-5     # nn.GroupNorm default, used by norm_type='ln' (dprnn.py:76-77)


class GlobLN(nn.Module):
    """Global Layer Normalization (globLN): parameter holder for the fused kernels (+ a device-tensor forward for callers
    that apply the module themselves)."""

    def __init__(self, channel_size):
        super().__init__()
        self.channel_size = channel_size
        self.gamma = nn.Parameter(torch.ones(channel_size), requires_grad=True)
        self.beta = nn.Parameter(torch.zeros(channel_size), requires_grad=True)

    def forward(self, x, EPS: float = GLN_EPS):
        """norms.py:6-31 on a device tensor [B, C, ...]: glue for code that calls ``bottleneck[0](x)`` itself
        (src/models/dprnn_rawnet.py:85); the TasNet wrappers never call it -- their norms are folded into the HIP kernels
        (dprnn_bottleneck_prep, dprnn_proj_stats_fwd + the consumers' loaders).  Here: dprnn_rows_stats +
        dprnn_bottleneck_prep + dprnn_rows_affine_fwd."""
        if EPS != GLN_EPS:
            raise NotImplementedError("GlobLN MI355X path: the reference's eps (1e-8)")
        from . import engine
        return engine.norm_forward(self, x)
