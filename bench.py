#!/usr/bin/env python3
"""bench.py — throughput of the batched Git-blob hasher on MI355X.

Workload (BASELINE.json configs[1]): 1,000,000 x 4 KiB random blobs per GPU,
already framed and resident in HBM.  One "step" = one pass of the hot path over
that batch = one launch of the SHA-1 kernel through the C ABI
(vit_hash_arena_device).  N > 1: one process per GPU (torch.distributed.run),
every rank hashes its own 1M blobs (weak scaling, no data-path collective — each
digest depends on its own blob only); the only communication is the timing
barrier and the max-over-ranks of the elapsed time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

SEED = 0x5EED0002           # config 2 of the synthetic-input plan
N_BLOBS = 1_000_000
BLOB_LEN = 4096
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E peak (spec)
VALU_PEAK_TOPS = 39.3216    # 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz: issue rate of the 3-input / shift integer ops
OPS_PER_BLOCK = 597          # vector ops per 64-byte block: schedule 192 + rounds 400 + 5 chaining adds


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blobs", type=int, default=N_BLOBS, help="blobs per GPU (default: the BASELINE config)")
    ap.add_argument("--blob-len", type=int, default=BLOB_LEN)
    ap.add_argument("--kernel", choices=["lds", "lane"], default="lds")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work budget for the baseline leg")
    ap.add_argument("--dist-backend", default=None,
                    help="torch.distributed backend for the timing barrier (default: nccl = RCCL; gloo lets two "
                         "ranks rehearse on one GPU)")
    ap.add_argument("--end-to-end", action="store_true",
                    help="also time pack + H2D + kernel + D2H from host memory (reported beside value, never as value); "
                         "off by default so that a profile of the default command holds same-size launches only")
    return ap.parse_args()


def cpu_baseline(args, gpu_digests_of_sample):
    """The reference's hashing loop (src/commit.cpp:90-94) on host cores, on a
    bounded sample of the same blobs.  Checker/measurement only."""
    from oracle import pyoracle as O
    n_sample = min(args.blobs, 65536)
    cores = os.cpu_count() or 1
    cores = min(cores, 16) if os.environ.get("OMP_NUM_THREADS") is None else min(cores, int(os.environ["OMP_NUM_THREADS"]))
    blobs = O.synth_fill_many(SEED, 0, n_sample, args.blob_len, threads=cores)
    offsets = (np.arange(n_sample, dtype=np.uint64) * np.uint64(args.blob_len))
    lens = np.full(n_sample, args.blob_len, dtype=np.uint64)
    # calibrate, then repeat the sample until ~cpu_seconds/2 of single-thread work is done
    sec, dig, used = O.bench_hash_loop(blobs, offsets, lens, threads=1)
    reps = max(1, int((args.cpu_seconds * 0.6) / max(sec, 1e-6)))
    t1 = []
    for _ in range(reps):
        s, dig, used = O.bench_hash_loop(blobs, offsets, lens, threads=1)
        t1.append(s)
    sec1 = statistics.median(t1)
    tn = []
    for _ in range(max(3, int((args.cpu_seconds * 0.3) / max(sec / cores, 1e-6)) // 4)):
        s, dign, _ = O.bench_hash_loop(blobs, offsets, lens, threads=cores)
        tn.append(s)
    secn = statistics.median(tn)
    parity = bool((dig == gpu_digests_of_sample[:n_sample]).all())
    payload = n_sample * args.blob_len
    out = {
        "value": round(payload / sec1 / 1e9, 4), "unit": "GB/s", "cores": 1,
        "kind": "port",
        "sample": "%d of the same %d-byte blobs (%.1f MB), header+concat+one-shot %s per blob as reference "
                  "src/commit.cpp:90-94, median of %d runs" % (n_sample, args.blob_len, payload / 1e6,
                                                              "libcrypto SHA1()" if used else "oracle SHA-1", reps),
        "objects_per_s": round(n_sample / sec1, 1),
        "all_cores": {"value": round(payload / secn / 1e9, 4), "unit": "GB/s", "cores": cores,
                      "objects_per_s": round(n_sample / secn, 1)},
        "gpu_matches_cpu_on_sample": parity,
    }
    # The reference's full per-object path (hash + zlib + loose-object write), from its own commit.cpp.
    ref = O.ref_lib()
    if ref is not None:
        import tempfile
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as td:
            os.makedirs(os.path.join(td, ".git", "objects"))
            os.chdir(td)
            try:
                m = min(n_sample, 2000)
                t0 = time.perf_counter()
                for i in range(m):
                    O.ref_write_blob(blobs[i * args.blob_len:(i + 1) * args.blob_len].tobytes())
                dt = time.perf_counter() - t0
            finally:
                os.chdir(cwd)
        out["reference_write_blob"] = {"objects_per_s": round(m / dt, 1), "value": round(m * args.blob_len / dt / 1e9, 5),
                                       "unit": "GB/s", "cores": 1, "kind": "reference",
                                       "sample": "%d blobs through the reference's writeBlob (SHA1 + zlib + file write)" % m}
    return out


def main():
    args = parse()
    import torch
    import vit_amd
    from vit_amd import dist as vdist
    rank, local_rank, world = vdist.env()
    dist = vdist.init(args.dist_backend)   # RCCL ("nccl") process group when WORLD_SIZE > 1; carries timing only

    dev_id = local_rank if torch.cuda.device_count() > local_rank else 0
    h = vit_amd.Hasher((dev_id,))
    h.set_option("kernel", vit_amd.KERNEL_LDS if args.kernel == "lds" else vit_amd.KERNEL_LANE)

    n, L = args.blobs, args.blob_len
    # rank r: blobs [r*n, (r+1)*n); slots start on 128-byte boundaries (the L2 line), see vit_arena_pack_aligned
    batch = h.synth_arena("blob", SEED, vdist.weak_first_index(rank, n), n, L, slot_align=128)
    hdr = vit_amd.header_len("blob", L)
    nb = vit_amd.object_blocks("blob", L)

    def barrier():
        h.sync(0)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # warm-up (untimed), then K timed steps bracketed by barrier + sync
    h.time_hash_device(batch, warmup=args.warmup, iters=0)
    barrier()
    t0 = time.perf_counter()
    ms = h.time_hash_device(batch, warmup=0, iters=args.steps)     # per-launch HIP-event durations
    h.sync(0)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = vdist.max_over_ranks(dist, elapsed)
    status = vdist.sum_over_ranks(dist, h.device_status(0))
    if dist is not None:
        dist.barrier()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    steps = args.steps
    payload_total = world * n * L * steps
    value = payload_total / elapsed / 1e9
    kern_ms = statistics.mean(ms)
    alg_bytes = n * (hdr + L + 20)                                # per launch: framed message read + digest written
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    valu_tops = n * nb * OPS_PER_BLOCK / (kern_ms * 1e-3) / 1e12
    traffic = None
    tpath = os.path.join(HERE, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("bytes_per_launch")
        except Exception:
            traffic = None

    dig = batch["digests"].download(np.uint8, 20 * n).reshape(-1, 20)
    line = {
        "metric": "GB/s + objects/s of Git-blob SHA-1 (write-tree, 1M×4KiB files) at 1–8 GPUs",
        "value": round(value, 2), "unit": "GB/s",
        "objects_per_s": round(world * n * steps / elapsed, 1),
        "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic (counter-based splitmix64 bytes, seed 0x5EED0002, generated on device)",
        "config": {"workload": "%d x %d B random blobs per GPU, Git blob framing, resident in HBM (BASELINE configs[1])"
                               % (n, L),
                   "blobs_per_gpu": n, "blob_bytes": L, "blocks_per_blob": nb, "kernel": args.kernel,
                   "parallelism": "file-sharded x%d, no collective" % world},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "kernel": "sha1_lds_kernel<2, 4>" if args.kernel == "lds" else "sha1_lane_kernel",
                     "kernel_ms_avg": round(kern_ms, 4), "algorithmic_bytes_per_launch": alg_bytes,
                     "note": "the kernel is bound by integer VALU issue, not HBM: see valu",
                     "valu": {"achieved": round(valu_tops, 2), "peak": VALU_PEAK_TOPS, "unit": "T lane-ops/s",
                              "frac": round(valu_tops / VALU_PEAK_TOPS, 4), "ops_per_block": OPS_PER_BLOCK}},
        "device_status": status,
    }
    if not args.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline(args, dig)
    if args.end_to_end and world == 1:
        from vit_amd import _abi
        m = min(n, 262144)
        host = np.empty(m * L, dtype=np.uint8)             # the same blobs, in host memory (the library's generator)
        fill = _abi.load().vit_synth_bytes
        for i in range(m):
            fill(SEED, i, host.ctypes.data + i * L, L)
        blobs = [host[i * L:(i + 1) * L] for i in range(m)]
        h.hash_blobs(blobs)                               # first call sizes the pinned and device staging buffers
        runs = []
        for _ in range(3):
            got = h.hash_blobs(blobs)
            runs.append((h.stat("last_total_ns"), h.stat("last_pack_ns")))
        total_ns, pack_ns = sorted(runs)[1]
        # time inside vit_hash_blobs (pack + H2D + kernel + D2H), without the Python argument marshalling
        line["end_to_end_host_memory"] = {"value": round(m * L / total_ns, 3), "unit": "GB/s", "blobs": m,
                                          "total_ms": round(total_ns / 1e6, 2), "pack_ms": round(pack_ns / 1e6, 2),
                                          "pack_threads": min(os.cpu_count() or 1, 16),
                                          "matches": bool((got == dig[:m]).all())}
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
