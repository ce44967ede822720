"""ctypes declarations for include/vit_hash.h and include/vit_hash_bench.h (libvit_hash.so).

The library is loaded from the package directory only (built in-tree by
vit_amd.build).  A missing library is an error, never a fallback.

torch, when installed, is imported BEFORE the library is loaded: PyTorch-ROCm
bundles its own libamdhip64.so.7, and a process must hold exactly one HIP
runtime, so whichever is loaded first has to serve both.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libvit_hash.so")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p

# name -> (restype, argtypes); the single source of truth for the symbol test.
SIGNATURES = {
    "vit_hasher_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]),
    "vit_hasher_destroy": (None, [vp]),
    "vit_last_error": (C.c_char_p, [vp]),
    "vit_hasher_device_count": (C.c_int, [vp]),
    "vit_hasher_set_option": (C.c_int, [vp, C.c_char_p, C.c_uint64]),
    "vit_hasher_get_stat": (C.c_int, [vp, C.c_char_p, u64p]),
    "vit_header_len": (C.c_uint32, [C.c_char_p, C.c_uint64]),
    "vit_write_header": (C.c_uint32, [C.c_char_p, C.c_uint64, vp]),
    "vit_object_blocks": (C.c_uint64, [C.c_char_p, C.c_uint64]),
    "vit_hash_objects": (C.c_int, [vp, C.c_char_p, vp, vp, C.c_uint64, vp]),
    "vit_hash_blobs": (C.c_int, [vp, vp, vp, C.c_uint64, vp]),
    "vit_sha1_buffers": (C.c_int, [vp, vp, vp, C.c_uint64, vp]),
    "vit_host_object_id": (C.c_int, [C.c_char_p, vp, C.c_uint64, vp]),
    "vit_host_object_ids": (C.c_int, [C.c_char_p, vp, vp, C.c_uint64, vp]),
    "vit_arena_alloc": (C.c_int, [vp, C.c_uint64, C.POINTER(vp)]),
    "vit_arena_free": (C.c_int, [vp, vp]),
    "vit_arena_pack": (C.c_int, [C.c_char_p, vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp, u64p, C.c_int]),
    "vit_arena_pack_aligned": (C.c_int, [C.c_char_p, vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp, u64p, C.c_int,
                                         C.c_uint32]),
    "vit_bin_order": (C.c_int, [vp, C.c_uint64, vp]),
    "vit_bin_order_device": (C.c_int, [vp, C.c_int, vp, C.c_uint64, vp, vp]),
    "vit_deflate_arena_device": (C.c_int, [vp, C.c_int, vp, C.c_uint64, vp, vp, vp, C.c_uint64, vp, vp, vp, vp]),
    "vit_deflate_bound": (C.c_uint32, [C.c_uint64]),
    "vit_deflate_objects": (C.c_int, [vp, C.c_char_p, vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp]),
    "vit_deflate_out_bytes": (C.c_uint64, [vp, C.c_char_p, vp, C.c_uint64]),
    "vit_shard_plan": (C.c_int, [vp, C.c_uint64, C.c_int, vp]),
    "vit_host_split_plan": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, vp]),
    "vit_numa_plan": (C.c_int, [C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vit_numa_node_cpus": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "vit_launch_shape": (C.c_int, [C.c_int, C.c_uint64, u32p, u32p]),
    "vit_hash_arena": (C.c_int, [vp, C.c_int, vp, C.c_uint64, vp, vp, C.c_uint64, vp]),
    "vit_hash_arena_device": (C.c_int, [vp, C.c_int, vp, C.c_uint64, vp, vp, vp, C.c_uint64, vp, vp]),
    "vit_hash_arena_device_static": (C.c_int, [vp, C.c_int, vp, C.c_uint64, vp, vp, C.c_uint64, vp, vp]),
    "vit_device_status": (C.c_int, [vp, C.c_int, u32p]),
    "vit_device_alloc": (C.c_int, [vp, C.c_int, C.c_uint64, C.POINTER(vp)]),
    "vit_device_free": (C.c_int, [vp, C.c_int, vp]),
    "vit_device_upload": (C.c_int, [vp, C.c_int, vp, vp, C.c_uint64]),
    "vit_device_download": (C.c_int, [vp, C.c_int, vp, vp, C.c_uint64]),
    "vit_device_sync": (C.c_int, [vp, C.c_int]),
    "vit_synth_arena_device": (C.c_int, [vp, C.c_int, C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                        C.c_uint64, vp, vp, vp, vp, C.c_uint64, vp]),
    "vit_synth_bytes": (None, [C.c_uint64, C.c_uint64, vp, C.c_uint64]),
    "vit_synth_fill_host": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, vp, vp, vp, C.c_int]),
    "vit_time_hash_device": (C.c_int, [vp, C.c_int, vp, C.c_uint64, vp, vp, vp, C.c_uint64, vp,
                                      C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "vit_valu_peak": (C.c_int, [vp, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_float)]),
}

_lib = None


def load():
    """dlopen libvit_hash.so and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -m vit_amd.build` (needs hipcc). "
            "There is no CPU fallback." % LIB_PATH)
    try:
        import torch  # noqa: F401  (see module docstring)
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
