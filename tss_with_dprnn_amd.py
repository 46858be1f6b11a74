"""Import shim: ``import tss_with_dprnn_amd`` -> the package directory ``tss-with-dprnn_amd/``.

The package directory carries the project's name (with hyphens), which Python cannot import
directly; this module turns itself into that package by pointing ``__path__`` at it, so
``tss_with_dprnn_amd.dprnn`` resolves to ``tss-with-dprnn_amd/dprnn.py`` etc.
"""
import os as _os

# The host driver of this pool supports dmabuf IPC only: RCCL (torch.distributed backend "nccl") and device tensors shared between
# the ranks of one node need HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment BEFORE the HIP runtime comes up (without it:
# `hipIpcGetMemHandle: invalid argument`).  Importing this package is early enough in every launch form (python, torchrun,
# a trainer script): nothing below touches the GPU.  An explicit setting in the environment wins.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tss-with-dprnn_amd")]
__version__ = "0.1.0"

from tss_with_dprnn_amd._lib import build_library, load_library, library_path  # noqa: E402,F401
from tss_with_dprnn_amd.dprnn import DPRNN, DPRNNBlock, DPRNNTasNet, SingleRNN  # noqa: E402,F401
from tss_with_dprnn_amd.encoder_decoder import Decoder, Encoder  # noqa: E402,F401
from tss_with_dprnn_amd.dprnn_spe import DPRNNSpe, DPRNNSpeTasNet, ResBlock  # noqa: E402,F401
from tss_with_dprnn_amd.dprnn_spe_ira import DPRNNSpeIRA, DPRNNSpeIRATasNet  # noqa: E402,F401
