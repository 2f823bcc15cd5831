"""imageexperiments_amd -- MI355X (gfx950) drop-in for the CompressionLib tile-encode path of
mnesbit/ImageExperiments: host dictionary/quant/entropy code in C++, the quantized matching
pursuit as hand-written HIP, behind the C ABI of include/mpcodec.h.

Python here is plumbing only (ctypes binding, torch for device memory/streams/torch.distributed).
"""
from .api import (CompressionContext, MpcError, PatchStatistics, format_double, assemble_streams, assemble_symbol_streams, calculate_psnr, create_compression_context,  # noqa: F401
                  decode_image, huffman_decode, huffman_encode, library_path, load_library, read_compressed,
                  run_length_decode, run_length_encode, write_compressed)

__all__ = ["CompressionContext", "MpcError", "PatchStatistics", "format_double", "assemble_streams", "assemble_symbol_streams", "calculate_psnr", "create_compression_context",
           "decode_image", "huffman_decode", "huffman_encode", "library_path", "load_library", "read_compressed",
           "run_length_decode", "run_length_encode", "write_compressed"]
