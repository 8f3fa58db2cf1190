"""rustlight_amd — MI355X-native drop-in for rustlight's `path` integrator (host-side Python glue).

The product is the C-ABI library built from rustlight_amd/csrc (HIP kernels for gfx950 + C++ host
code); this package only binds it (ctypes), provides synthetic scene fixtures and the
torch.distributed plumbing for the multi-GPU framebuffer reduce.
"""
from . import scenes  # noqa: F401

__all__ = ["scenes"]
