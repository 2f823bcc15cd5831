"""imageexperiments_amd -- MI355X (gfx950) drop-in for the CompressionLib tile-encode path of
mnesbit/ImageExperiments: host dictionary/quant/entropy code in C++, the quantized matching
pursuit as hand-written HIP, behind the C ABI of include/mpcodec.h.

Python here is plumbing only (ctypes binding, torch for device memory/streams/torch.distributed).
"""
from .api import (CompressionContext, MpcError, create_compression_context, library_path, load_library)  # noqa: F401

__all__ = ["CompressionContext", "MpcError", "create_compression_context", "library_path", "load_library"]
