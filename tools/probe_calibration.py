#!/usr/bin/env python3
"""A fresh hasher on a 400,000-object mixed slice from host memory, four calls in a row: the measured rates, what
each call moved to the host threads and when its two sides finished.  Shows what the first call (calibration,
freshly pinned staging memory) costs against the ones that follow."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, vit_amd
from vit_amd import _abi
lib = _abi.load()
m = 400000
sizes = bench.mixed_sizes(m)
buf, offs = bench.fill_host(lib, 0x5EED0004, sizes)
ptrs = (offs[:m] + np.uint64(buf.ctypes.data)).astype(np.uint64)
dig = np.zeros((m, 20), dtype=np.uint8)
with vit_amd.Hasher((0,)) as h:
    for k in range(4):
        h._check(lib.vit_hash_blobs(h.handle, ptrs.ctypes.data, sizes.ctypes.data, m, dig.ctypes.data))
        print(json.dumps({k2: h.stat(k2) for k2 in ("pack_rate", "core_rate", "core_rate_multi", "copy_rate", "lane_rate", "calibrate_ns", "last_total_ns", "last_gpu_done_ns", "last_host_done_ns", "last_host_objects", "last_pack_ns")}), flush=True)
