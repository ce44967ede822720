#!/usr/bin/env python3
"""Does the start alignment of the object slots change what the kernel fetches?

Hashes the headline batch (1 M x 4 KiB blobs, 65 blocks each) with slots packed
back to back (64-byte aligned, stride 4160) or rounded up to 128 bytes (stride
4224).  Run it under `rocprofv3 --pmc FETCH_SIZE` to compare the bytes fetched;
alone it prints the kernel times.  Measurement tool.
"""
import argparse
import json
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--align", type=int, default=64)
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--len", type=int, default=4096)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--kernel", type=int, default=0)
a = ap.parse_args()
h = vit_amd.Hasher((0,))
h.set_option("kernel", a.kernel)
b = h.synth_arena("blob", 0x5EED0002, 0, a.n, a.len, slot_align=a.align)
ms = h.time_hash_device(b, warmup=3, iters=a.iters)
d = b["digests"].download(np.uint8, 20 * a.n).reshape(-1, 20)
print(json.dumps({"align": a.align, "arena_bytes": b["arena_bytes"], "median_ms": round(statistics.median(ms), 4),
                  "min_ms": round(min(ms), 4), "xor": np.bitwise_xor.reduce(d, axis=0).tobytes().hex()}))
