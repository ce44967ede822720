#!/usr/bin/env python3
"""bench.py's host-memory legs as rank 0 of an N-GPU run would execute them, with the N device queues all on
GPU 0: rehearses the code path of `bench.py --gpus N` (one hasher, N queues, one thread budget) where only one
GPU exists.  The rates say nothing about N GPUs (one PCIe link carries every queue); crashes and wrong ids do."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import vit_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--queues", type=int, nargs="+", default=[1, 2, 4, 8])
ap.add_argument("--objects", type=int, default=500_000)
ap.add_argument("--weak", action="store_true", help="N x objects for N queues, as mixed_slice_weak")
a = ap.parse_args()
args = argparse.Namespace(blobs=65536, blob_len=4096, e2e_objects=a.objects)
h = vit_amd.Hasher((0,))
batch = h.synth_arena("blob", bench.SEED, 0, args.blobs, args.blob_len, slot_align=128)
resident = h.hash_device(batch)
for q in a.queues:
    out = bench.end_to_end(args, q, resident)
    m = out["mixed_slice_weak"] if a.weak and q > 1 else out["mixed_slice"]
    print(json.dumps({"queues": q, "uniform_4k_matches": out["uniform_4k"]["matches_resident_digests"],
                      "mixed_GBps": m["value"], "objects": m["objects"], "total_ms": m["total_ms"], "plan_ms": m["plan_ms"],
                      "gpu_side_done_ms": m["gpu_side_done_ms"], "host_side_done_ms": m["host_side_done_ms"], "objects_gpu": m["objects_gpu"],
                      "objects_host": m["objects_host"], "host_threads": m["host_threads"],
                      "copy_rate": m["measured_rates_bytes_per_s"]["copy_rate"], "copy_rate_total": m["copy_rate_total"],
                      "placement": m["placement"], "matches": m["matches"]}),
          flush=True)
