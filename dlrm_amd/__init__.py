"""dlrm_amd: MI355X (gfx950) DLRM hot path -- host side above the C ABI.

The compute lives in hand-written HIP kernels inside libdlrm_amd.so
(dlrm_amd/csrc, C ABI in include/dlrm_amd.h).  PyTorch is used only for device
memory, streams and torch.distributed.
"""
from . import _capi  # noqa: F401

__all__ = ["_capi"]
