#!/usr/bin/env python3
"""Skewed-size workloads (BASELINE configs[3] and [4]) on one GPU.

  mixed   one GPU's share of config 4: N blobs with the Linux-kernel-like lognormal size
          distribution, generated and hashed resident in HBM, with and without length binning.
  skew    config 5: a few multi-GiB blobs plus 1,000,000 tiny ones (0..256 B) from host memory
          through vit_hash_blobs: the giants go to host threads, the rest to the GPU, concurrently.

Measurement tool; ids are spot-checked against hashlib.  Prints one JSON line.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vit_amd  # noqa: E402


def git_id(b):
    return hashlib.sha1(b"blob %d\0" % len(b) + b).digest()


def mixed(a):
    h = vit_amd.Hasher((0,))
    rng = np.random.default_rng(0x5EED0004)
    sizes = np.exp(rng.normal(np.log(4500.0), 1.55, a.n))
    sizes = np.minimum(sizes * (15000.0 / sizes.mean()), a.clip).astype(np.uint64)
    b = h.synth_arena("blob", 0x5EED0004, 0, a.n, 0, lens=sizes)
    nb = b["nblocks"].download(np.uint32, 4 * a.n)
    out = {"workload": "mixed", "n": a.n, "payload_bytes": int(sizes.sum()), "arena_bytes": b["arena_bytes"],
           "max_blocks": int(nb.max()), "clip_bytes": a.clip}
    ms0 = h.time_hash_device(b, warmup=1, iters=a.iters)
    d0 = b["digests"].download(np.uint8, 20 * a.n).reshape(-1, 20).copy()
    t0 = time.perf_counter()
    order_host, differs = vit_amd.bin_order(nb)
    out["bin_order_host_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    order = h.alloc(4 * a.n).upload(order_host)
    b["digests"].upload(np.zeros(20 * a.n, dtype=np.uint8))
    ms1 = h.time_hash_device(b, order=order, warmup=1, iters=a.iters)
    d1 = b["digests"].download(np.uint8, 20 * a.n).reshape(-1, 20)
    out["match_binned_vs_unbinned"] = bool((d0 == d1).all())
    for i in (0, 1, a.n // 2, a.n - 1, int(np.argmax(sizes))):
        assert d1[i].tobytes() == git_id(vit_amd.synth_bytes(0x5EED0004, i, int(sizes[i]))), i
    pay = float(sizes.sum())
    out["unbinned"] = {"ms": round(min(ms0), 3), "GBps": round(pay / min(ms0) / 1e6, 1)}
    out["binned"] = {"ms": round(min(ms1), 3), "GBps": round(pay / min(ms1) / 1e6, 1)}
    out["longest_lane_ms_estimate"] = round(int(nb.max()) * 1.3e-3, 3)
    print(json.dumps(out), flush=True)


def skew(a):
    from oracle import pyoracle as O
    h = vit_amd.Hasher((0,))
    giants = [O.synth_fill_many(0x5EED0005, 10_000_000 + i, 1, a.giant_bytes, threads=1) for i in range(a.giants)]
    rng = np.random.default_rng(0x5EED0005)
    tiny_len = rng.integers(0, 257, a.tiny)
    pool = O.synth_fill_many(0x5EED0005, 0, 1, 256 * 4096 + 512, threads=1)
    tiny = [pool[(i % 4096) * 256 + (i // 4096 % 200):][:int(L)] for i, L in enumerate(tiny_len)]
    blobs = giants + tiny
    out = {"workload": "skew", "giants": a.giants, "giant_bytes": a.giant_bytes, "tiny": a.tiny,
           "payload_bytes": int(a.giants * a.giant_bytes + tiny_len.sum())}
    keep, ptrs, lens = vit_amd.hasher._as_ptr_arrays(blobs)
    dig = np.zeros((len(blobs), 20), dtype=np.uint8)
    for policy in ("auto", "never_host"):
        if policy == "never_host":
            if a.giant_bytes > (64 << 20):
                continue                                   # a lane would need minutes for a multi-GiB object
            h.set_option("host_threshold_bytes", 0)
        g0, h0, b0 = h.stat("objects_gpu"), h.stat("objects_host"), h.stat("bytes_host")
        t0 = time.perf_counter()
        h._check(h.lib.vit_hash_blobs(h.handle, ptrs, lens.ctypes.data, len(blobs), dig.ctypes.data))
        dt = time.perf_counter() - t0
        out[policy] = {"seconds": round(dt, 3), "objects_gpu": h.stat("objects_gpu") - g0,
                       "objects_host": h.stat("objects_host") - h0, "bytes_host": h.stat("bytes_host") - b0,
                       "host_threshold": h.stat("last_host_threshold"),
                       "objects_per_s": round(len(blobs) / dt, 1)}
        for i in list(range(a.giants)) + [a.giants, a.giants + 1, len(blobs) - 1, a.giants + a.tiny // 2]:
            hh = hashlib.sha1(b"blob %d\0" % len(blobs[i]))
            hh.update(memoryview(blobs[i]))
            assert dig[i].tobytes() == hh.digest(), i
    out["host_sha_accelerated"] = h.stat("host_sha_accelerated")
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    m = sub.add_parser("mixed")
    m.add_argument("--n", type=int, default=1_250_000)
    m.add_argument("--clip", type=int, default=32 << 20)
    m.add_argument("--iters", type=int, default=3)
    s = sub.add_parser("skew")
    s.add_argument("--giants", type=int, default=4)
    s.add_argument("--giant-bytes", type=int, default=2 << 30)
    s.add_argument("--tiny", type=int, default=1_000_000)
    a = ap.parse_args()
    mixed(a) if a.cmd == "mixed" else skew(a)


if __name__ == "__main__":
    main()
