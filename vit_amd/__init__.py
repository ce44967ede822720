"""vit_amd — batched Git-object SHA-1 for AMD MI355X (gfx950).

The product is the C ABI in include/vit_hash.h (vit_amd/libvit_hash.so, built
from vit_amd/csrc by `python -m vit_amd.build`) and the C++ object store and
`vit` front end layered on it.  This package only exposes that ABI to Python
for the tests and the benchmark.
"""
from .hasher import (Hasher, VitError, DeviceBuffer, KERNEL_LDS, KERNEL_LANE, header_len, object_blocks,
                     pack_arena, bin_order, synth_bytes)

__all__ = ["Hasher", "VitError", "DeviceBuffer", "KERNEL_LDS", "KERNEL_LANE", "header_len",
           "object_blocks", "pack_arena", "bin_order", "synth_bytes"]
