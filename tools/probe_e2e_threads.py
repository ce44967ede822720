#!/usr/bin/env python3
"""Uniform 4 KiB blobs from host memory through vit_hash_blobs (pack + H2D + kernel + D2H) under different
packing-thread budgets and chunk sizes: does steadier, slower packing leave the copy engine more of the
host's DRAM than sixteen threads packing in bursts?  Prints GB/s of payload per setting (median of 5)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import vit_amd  # noqa: E402
from vit_amd import _abi  # noqa: E402

lib = _abi.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
L = 4096
sizes = np.full(n, L, dtype=np.uint64)
buf, offs = bench.fill_host(lib, bench.SEED, sizes)
ptrs = (offs[:n] + np.uint64(buf.ctypes.data)).astype(np.uint64)
out = {}
with vit_amd.Hasher((0,)) as h:
    for chunk_mb in (32, 64, 128):
        for threads in (4, 6, 8, 12, 16):
            h.set_option("chunk_bytes", chunk_mb << 20)
            h.set_option("pack_threads", threads)
            _, (total_ns, pack_ns, *_rest) = bench.time_host_batch(h, ptrs, sizes, n, runs=5)
            out["chunk %3d MiB, %2d threads" % (chunk_mb, threads)] = {
                "GBps": round(n * L / total_ns, 2), "total_ms": round(total_ns / 1e6, 2), "pack_ms": round(pack_ns / 1e6, 2)}
    # the host lane's share of a uniform batch, were small objects allowed to move (default: 16 KiB and up only)
    h.set_option("chunk_bytes", 64 << 20)
    h.set_option("pack_threads", 16)
    for min_bytes in (16 << 10, 1):
        h.set_option("host_min_bytes", min_bytes)
        _, (total_ns, pack_ns, on_gpu, on_host, *_r) = bench.time_host_batch(h, ptrs, sizes, n, runs=5)
        out["host_min_bytes %d" % min_bytes] = {"GBps": round(n * L / total_ns, 2), "total_ms": round(total_ns / 1e6, 2),
                                                "objects_gpu": on_gpu, "objects_host": on_host}
print(json.dumps(out, indent=1))
