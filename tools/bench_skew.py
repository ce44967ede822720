#!/usr/bin/env python3
"""Skewed-size workloads (BASELINE configs[3] and [4]) on one GPU.

  mixed   one GPU's share of config 4: N blobs with the Linux-kernel-like lognormal size
          distribution, generated and hashed resident in HBM, with and without length binning.
  skew    config 5: a few multi-GiB blobs plus 1,000,000 tiny ones (0..256 B) from host memory
          through vit_hash_blobs: the giants go to host threads, the rest to the GPU, concurrently.

  c4      all of config 4 through one GPU: 10,000,000 blobs of that distribution (about 150 GB),
          streamed in slices from host memory through vit_hash_blobs (pack + H2D + kernel + D2H,
          long objects on host threads).  Hashing only: nothing is deflated or stored.

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
    h.set_option("kernel", a.kernel)
    rng = np.random.default_rng(0x5EED0004)
    sizes = np.exp(rng.normal(np.log(4500.0), 1.55, a.n))
    sizes = np.minimum(sizes * (15000.0 / sizes.mean()), a.clip).astype(np.uint64)
    b = h.synth_arena("blob", 0x5EED0004, 0, a.n, 0, lens=sizes)
    nb = b["nblocks"].download(np.uint32, 4 * a.n)
    out = {"workload": "mixed", "kernel": a.kernel, "n": a.n, "payload_bytes": int(sizes.sum()), "arena_bytes": b["arena_bytes"],
           "max_blocks": int(nb.max()), "clip_bytes": a.clip}
    ms0 = h.time_hash_device(b, warmup=1, iters=a.iters)
    d0 = b["digests"].download(np.uint8, 20 * a.n).reshape(-1, 20).copy()
    t0 = time.perf_counter()
    order_host, differs = vit_amd.bin_order(nb)
    out["bin_order_host_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    order = h.alloc(4 * a.n).upload(order_host)
    h.bin_order_device(b); h.sync()                        # first call sizes the sort's scratch memory
    t0 = time.perf_counter()
    d_order = h.bin_order_device(b); h.sync()
    out["bin_order_device_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    out["device_order_equals_host_order"] = bool((d_order.download(np.uint32, 4 * a.n) == order_host).all())
    b["digests"].upload(np.zeros(20 * a.n, dtype=np.uint8))
    h.set_option("sched_waves", 0)                         # hardware dispatch: one workgroup per 256 objects
    ms1 = h.time_hash_device(b, order=order, warmup=1, iters=a.iters)
    d1 = b["digests"].download(np.uint8, 20 * a.n).reshape(-1, 20).copy()
    out["scheduled"] = {}
    h.set_option("sched_tail_percent", 0)                  # forced: the sweep measures the launch, not the library's choice
    for waves in a.sched_waves:                            # software-scheduled: resident waves pulling groups down the order
        for share in a.share_below:
            h.set_option("sched_waves", waves)
            h.set_option("sched_share_below", share)
            b["digests"].upload(np.zeros(20 * a.n, dtype=np.uint8))
            msw = h.time_hash_device(b, order=order, warmup=1, iters=a.iters)
            dw = b["digests"].download(np.uint8, 20 * a.n).reshape(-1, 20)
            out["scheduled"]["%d_waves_share_below_%d" % (waves, share)] = {
                "ms": round(min(msw), 3), "GBps": round(float(sizes.sum()) / min(msw) / 1e6, 1), "match": bool((dw == d1).all())}
    # the library's own choice (both launches queued, the waves decide by the longest object against a SIMD's mean share)
    h.set_option("sched_waves", 1); h.set_option("sched_share_below", 128); h.set_option("sched_tail_percent", 75)
    b["digests"].upload(np.zeros(20 * a.n, dtype=np.uint8))
    msa = h.time_hash_device(b, order=order, warmup=1, iters=a.iters)
    da = b["digests"].download(np.uint8, 20 * a.n).reshape(-1, 20)
    out["auto"] = {"ms": round(min(msa), 3), "GBps": round(float(sizes.sum()) / min(msa) / 1e6, 1), "match": bool((da == d1).all())}
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


def c4(a):
    from concurrent.futures import ThreadPoolExecutor
    from vit_amd import _abi
    lib = _abi.load()
    h = vit_amd.Hasher((0,))
    seed = 0x5EED0004
    per = (a.n + a.slices - 1) // a.slices
    threads = min(os.cpu_count() or 1, 16)
    out = {"workload": "c4", "n": a.n, "slices": a.slices, "fill_threads": threads}
    summary = np.zeros(20, dtype=np.uint8)
    hash_s = fill_s = 0.0
    payload = checked = 0
    per_slice = []
    sides = []
    g0, h0, b0 = h.stat("objects_gpu"), h.stat("objects_host"), h.stat("bytes_host")
    for k in range(a.slices):
        first, n = k * per, min(per, a.n - k * per)
        if n <= 0:
            break
        rng = np.random.default_rng(seed + 1000 * k)              # sizes: one stream per slice
        sizes = np.exp(rng.normal(np.log(4500.0), 1.55, n))
        sizes = np.minimum(sizes * (15000.0 / sizes.mean()), a.clip).astype(np.uint64)
        if a.uniform_bytes is not None:
            sizes[:] = a.uniform_bytes
        offs = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(sizes, out=offs[1:])
        buf = np.empty(int(offs[-1]) + 1, dtype=np.uint8)
        base = buf.ctypes.data
        t0 = time.perf_counter()

        def fill(r):
            for i in range(r[0], r[1]):
                lib.vit_synth_bytes(seed, first + i, base + int(offs[i]), int(sizes[i]))
        step = (n + 8 * threads - 1) // (8 * threads)
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(fill, [(i, min(i + step, n)) for i in range(0, n, step)]))
        fill_s += time.perf_counter() - t0
        ptrs = (offs[:n] + np.uint64(base)).astype(np.uint64)
        dig = np.zeros((n, 20), dtype=np.uint8)
        if a.trace:                                    # one slice twice, the second with the library's chunk timeline
            lib.vit_hash_blobs(h.handle, ptrs.ctypes.data, sizes.ctypes.data, n, dig.ctypes.data)
            h.set_option("trace", 1)
            h._check(lib.vit_hash_blobs(h.handle, ptrs.ctypes.data, sizes.ctypes.data, n, dig.ctypes.data))
            return
        if a.sweep:                                    # one slice, several pipeline settings, then stop
            res = []
            for mult, thr in [(8192, None), (8192, None), (4096, None), (16384, None), (8192, None), (4096, None),
                              (16384, None), (2048, None), (8192, 256 << 10), (8192, 32 << 10)]:
                chunk_mb = 64
                h.set_option("chunk_object_multiple", mult)
                if thr is None:
                    h.set_option("host_auto", 1)
                else:
                    h.set_option("host_threshold_bytes", thr)
                t0 = time.perf_counter()
                h._check(lib.vit_hash_blobs(h.handle, ptrs.ctypes.data, sizes.ctypes.data, n, dig.ctypes.data))
                dt = time.perf_counter() - t0
                res.append({"mult": mult, "host_threshold": thr, "s": round(dt, 3),
                            "GBps": round(int(offs[-1]) / dt / 1e9, 1), "pack_ms": round(h.stat("last_pack_ns") / 1e6, 1),
                            "thr_used": h.stat("last_host_threshold")})
                print(json.dumps(res[-1]), flush=True)
            return
        t0 = time.perf_counter()
        h._check(lib.vit_hash_blobs(h.handle, ptrs.ctypes.data, sizes.ctypes.data, n, dig.ctypes.data))
        dt = time.perf_counter() - t0
        hash_s += dt
        payload += int(offs[-1])
        per_slice.append(round(dt, 3))
        sides.append((round(h.stat("last_gpu_done_ns") / 1e6, 1), round(h.stat("last_host_done_ns") / 1e6, 1),
                      h.stat("last_host_objects")))
        summary ^= np.bitwise_xor.reduce(dig, axis=0)
        pick = set(np.random.default_rng(k).integers(0, n, 200).tolist()) | {0, n - 1, int(np.argmax(sizes))}
        for i in pick:
            hh = hashlib.sha1(b"blob %d\0" % int(sizes[i]))
            hh.update(memoryview(buf[int(offs[i]):int(offs[i + 1])]))
            assert dig[i].tobytes() == hh.digest(), (k, i)
        checked += len(pick)
        del buf
    out.update({"payload_bytes": payload, "hash_seconds": round(hash_s, 3), "fill_seconds": round(fill_s, 1),
                "GBps": round(payload / hash_s / 1e9, 2), "objects_per_s": round(a.n / hash_s, 1),
                "per_slice_seconds": per_slice, "per_slice_gpu_done_ms_host_done_ms_host_objects": sides,
                "objects_gpu": h.stat("objects_gpu") - g0,
                "objects_host": h.stat("objects_host") - h0, "bytes_host": h.stat("bytes_host") - b0,
                "last_host_threshold": h.stat("last_host_threshold"), "ids_checked_vs_hashlib": checked,
                "xor_of_ids": summary.tobytes().hex()})
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    m = sub.add_parser("mixed")
    m.add_argument("--n", type=int, default=1_250_000)
    m.add_argument("--clip", type=int, default=32 << 20)
    m.add_argument("--iters", type=int, default=3)
    m.add_argument("--kernel", type=int, default=0)
    m.add_argument("--sched-waves", type=int, nargs="*", default=[1, 3, 5])
    m.add_argument("--share-below", type=int, nargs="*", default=[0, 32, 128, 512, 100000])
    s = sub.add_parser("skew")
    s.add_argument("--giants", type=int, default=4)
    s.add_argument("--giant-bytes", type=int, default=2 << 30)
    s.add_argument("--tiny", type=int, default=1_000_000)
    c = sub.add_parser("c4")
    c.add_argument("--n", type=int, default=10_000_000)
    c.add_argument("--slices", type=int, default=10)
    c.add_argument("--clip", type=int, default=32 << 20)
    c.add_argument("--sweep", action="store_true", help="first slice only, under several chunk sizes / host thresholds")
    c.add_argument("--uniform-bytes", type=int, default=None, help="every blob this long instead of the distribution")
    c.add_argument("--trace", action="store_true", help="first slice only, print the per-chunk timeline")
    a = ap.parse_args()
    {"mixed": mixed, "skew": skew, "c4": c4}[a.cmd](a)


if __name__ == "__main__":
    main()
