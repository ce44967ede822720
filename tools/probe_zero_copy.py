#!/usr/bin/env python3
"""Experiment: let the kernel read a packed arena straight from pinned host memory over PCIe
(zero-copy) instead of copying it to device memory first."""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_amd  # noqa: E402
from vit_amd import _abi  # noqa: E402

h = vit_amd.Hasher((0,))
lib = _abi.load()
n, L = 262144, 4096
nb = vit_amd.object_blocks("blob", L)
arena_bytes = n * nb * 64
p = C.c_void_p()
h._check(lib.vit_arena_alloc(h.handle, arena_bytes, C.byref(p)))
arena = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(arena_bytes,))
host = np.empty(n * L, dtype=np.uint8)
for i in range(n):
    lib.vit_synth_bytes(7, i, host.ctypes.data + i * L, L)
ptrs = (np.arange(n, dtype=np.uint64) * np.uint64(L) + np.uint64(host.ctypes.data)).astype(np.uint64)
lens = np.full(n, L, dtype=np.uint64)
off = np.zeros(n, dtype=np.uint64); nbs = np.zeros(n, dtype=np.uint32); used = C.c_uint64(0)
h._check(lib.vit_arena_pack(b"blob", ptrs.ctypes.data, lens.ctypes.data, n, p, arena_bytes, off.ctypes.data, nbs.ctypes.data, C.byref(used), 16))
ref = h.hash_arena(arena, off, nbs)                       # copy + kernel
t = []
for _ in range(5):
    t0 = time.perf_counter(); h.hash_arena(arena, off, nbs); t.append(time.perf_counter() - t0)
print(json.dumps({"mode": "copy then hash", "ms": round(min(t) * 1e3, 2), "GBps": round(n * L / min(t) / 1e9, 1)}))
d_off = h.alloc(8 * n).upload(off); d_nb = h.alloc(4 * n).upload(nbs); d_dig = h.alloc(20 * n)
batch = dict(arena=type("P", (), {"ptr": p.value})(), offsets=d_off, nblocks=d_nb, digests=d_dig, n=n, arena_bytes=arena_bytes, dev=0)
for k in (0, 1, 2):
    h.set_option("kernel", k)
    ms = h.time_hash_device(batch, warmup=1, iters=5)
    got = d_dig.download(np.uint8, 20 * n).reshape(-1, 20)
    print(json.dumps({"mode": "zero-copy kernel %d" % k, "ms": round(min(ms), 2), "GBps": round(n * L / min(ms) / 1e6, 1),
                      "same_ids": bool((got == ref).all())}))
