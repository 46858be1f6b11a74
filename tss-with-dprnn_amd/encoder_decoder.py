"""Conv-TasNet encoder / decoder parameter holders (reference src/models/encoder_decoder.py).

Same constructor signatures, attribute names and state_dict keys as the reference
(``encoder.conv1d.weight`` (N,1,ks), ``decoder.weight`` (N,1,ks)).  On the MI355X path the
encoder output is never materialised: every kernel that needs enc[b,n,l] =
relu(sum_j W[n,j] x[b, l*stride+j]) (encoder_decoder.py:25-33) recomputes it from the waveform, and
the transposed-conv decoder (encoder_decoder.py:40-49) is fused behind the mask head
(dprnn_head_mask_decode_fwd + dprnn_decoder_fwd).
"""
from torch import nn

from ._lib import DprnnError


class Encoder(nn.Module):
    """Conv-TasNet Encoder: Conv1d(1 -> output_size, kernel_size, stride) + ReLU."""

    def __init__(self, kernel_size, output_size, stride, bias=False):
        super().__init__()
        if bias:
            raise NotImplementedError("Encoder(bias=True): the reference wrappers always pass bias=False (dprnn.py:244-249)")
        self.conv1d = nn.Conv1d(in_channels=1, out_channels=output_size, kernel_size=kernel_size,
                                stride=stride, groups=1, bias=bias)

    def forward(self, input):
        """encoder_decoder.py:25-33: [B, T] -> [B, N, L] (inference only; the TasNet wrappers never materialise this)."""
        from . import engine
        return engine.encoder_forward(self, input)


class Decoder(nn.ConvTranspose1d):
    """Conv-TasNet Decoder: ConvTranspose1d(N -> 1); subclassing keeps the key ``decoder.weight``."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.bias is not None:
            raise NotImplementedError("Decoder(bias=True): the reference wrappers always pass bias=False (dprnn.py:263-269)")

    def forward(self, input):
        """encoder_decoder.py:40-49: [B, N, L] -> [B, T] (inference only; the TasNet wrappers fuse this behind the mask head)."""
        from . import engine
        return engine.decoder_forward(self, input)


def squeeze_like_reference(y):
    """The reference's squeeze rule (encoder_decoder.py:44-48) applied to one decoded speaker [B, T]:
    [B,1,T] squeezed fully; if that is 1-D (B == 1 or T == 1) only dim 1 is squeezed instead."""
    B, T = y.shape
    if B == 1 and T == 1:
        return y.reshape(())
    return y
