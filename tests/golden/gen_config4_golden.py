#!/usr/bin/env python3
"""Generate tests/golden/config4.json: the XOR of all 10,000,000 blob ids of BASELINE configs[3], computed on the CPU
with libcrypto's SHA1() (the function the reference calls, src/commit.cpp:94) through the oracle's restatement of the
reference's hashing loop (oracle/cpu_baseline.c), slice by slice.  Independent of the GPU path, which tests/test_gpu_configs.py
checks against it.  About 150 GB of SHA-1: several minutes on 8 cores, 16 GB of memory.

    python tests/golden/gen_config4_golden.py
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as O  # noqa: E402

SEED, N, SLICES, CLIP = 0x5EED0004, 10_000_000, 10, 32 << 20


def slice_sizes(k, n):
    """Sizes of slice k: lognormal, median 4.5 KB, sigma 1.55, scaled to a 15 KB mean, clipped at 32 MiB;
    one numpy generator stream per slice (seed + 1000 k), as tools/bench_skew.py draws them."""
    rng = np.random.default_rng(SEED + 1000 * k)
    sizes = np.exp(rng.normal(np.log(4500.0), 1.55, n))
    return np.minimum(sizes * (15000.0 / sizes.mean()), CLIP).astype(np.uint64)


def main():
    threads = os.cpu_count() or 1
    per = N // SLICES
    acc = np.zeros(20, dtype=np.uint8)
    payload = 0
    per_slice = []
    t0 = time.time()
    for k in range(SLICES):
        sizes = slice_sizes(k, per)
        offs = np.zeros(per + 1, dtype=np.uint64)
        np.cumsum(sizes, out=offs[1:])
        buf = np.empty(int(offs[-1]) + 1, dtype=np.uint8)
        O.synth_fill_var(SEED, k * per, sizes, offs[:per], buf, threads=threads)
        sec, dig, how = O.bench_hash_loop_mt(buf, offs[:per], sizes, threads, reps=1, per_thread_ctx=True)
        assert how == 2, "libcrypto not found"
        x = np.bitwise_xor.reduce(dig, axis=0)
        per_slice.append(x.tobytes().hex())
        acc ^= x
        payload += int(offs[-1])
        print("slice %d: %.1f GB, %.1f s so far, xor %s" % (k, int(offs[-1]) / 1e9, time.time() - t0, per_slice[-1]), flush=True)
        del buf
    out = {"about": "XOR of the 20-byte ids of BASELINE configs[3] (10,000,000 blobs, lognormal sizes up to 32 MiB), from libcrypto "
                    "SHA1 on the CPU (tests/golden/gen_config4_golden.py); per_slice = the same per slice of 1,000,000",
           "seed": SEED, "n": N, "slices": SLICES, "clip": CLIP, "payload_bytes": payload,
           "xor_of_ids": acc.tobytes().hex(), "per_slice": per_slice}
    json.dump(out, open(os.path.join(HERE, "config4.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
