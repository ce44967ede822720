#!/usr/bin/env python3
"""Does the kernel time step with the number of objects per lane of the chip (65 536 lanes)?
Batches of 15, 15.26 (1 M) and 16 objects per lane, timed interleaved in one process."""
import json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_amd  # noqa: E402
h = vit_amd.Hasher((0,))
ns = [983040, 1_000_000, 1_015_808, 1_048_576]
bs = [h.synth_arena("blob", 0x5EED0002, 0, n, 4096, slot_align=128) for n in ns]
for b in bs:
    h.time_hash_device(b, warmup=3, iters=0)
t = {n: [] for n in ns}
for _ in range(12):
    for n, b in zip(ns, bs):
        t[n] += h.time_hash_device(b, warmup=0, iters=3)
base = statistics.median(t[ns[0]])
for n in ns:
    m = statistics.median(t[n])
    print(json.dumps({"objects": n, "per_lane": round(n / 65536, 3), "median_ms": round(m, 4),
                      "time_vs_15_per_lane": round(m / base, 4), "objects_vs_15_per_lane": round(n / ns[0], 4)}))
