#!/usr/bin/env python3
"""Latency of one vit_hash_blobs call (pack + H2D + kernel + D2H + collect) for small batches,
the shape of the per-depth tree batches and of hash-object.  Time is taken inside the library
(last_total_ns), so Python's argument marshalling is not in it.  Measurement tool."""
import json
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_amd  # noqa: E402
from vit_amd import _abi  # noqa: E402

h = vit_amd.Hasher((0,))
lib = _abi.load()
out = []
for n, L in [(1, 4096), (10, 4096), (100, 4096), (1000, 4096), (10000, 4096), (100000, 4096), (1000, 300), (1, 40)]:
    host = np.empty(max(n * L, 1), dtype=np.uint8)
    for i in range(n):
        lib.vit_synth_bytes(7, i, host.ctypes.data + i * L, L)
    blobs = [host[i * L:(i + 1) * L] for i in range(n)]
    keep, ptrs, lens = vit_amd.hasher._as_ptr_arrays(blobs)
    dig = np.zeros((n, 20), dtype=np.uint8)
    t = []
    for k in range(30):
        h._check(lib.vit_hash_blobs(h.handle, ptrs, lens.ctypes.data, n, dig.ctypes.data))
        if k >= 5:
            t.append(h.stat("last_total_ns") / 1e3)
    out.append({"n": n, "len": L, "median_us": round(statistics.median(t), 1), "min_us": round(min(t), 1),
                "us_per_object": round(statistics.median(t) / n, 3)})
    print(json.dumps(out[-1]), flush=True)
