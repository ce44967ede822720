#!/usr/bin/env python3
"""Per-launch duration of the SHA-1 kernel on 1,000,000 resident 4 KiB blobs straight after a short warm-up,
in a second run directly behind the first, and after two idle seconds: how many launches the chip needs to reach
its clocks (what bench.py sizes its untimed warm-up by).  Run from the repository root on the GPU box."""
import sys, os
sys.path.insert(0, os.getcwd())
import vit_amd
h = vit_amd.Hasher((0,))
b = h.synth_arena("blob", 0x5EED0002, 0, 1_000_000, 4096, slot_align=128)
h.time_hash_device(b, warmup=5, iters=0)
ms = h.time_hash_device(b, warmup=0, iters=40)
print("after 5 warm-up:", " ".join("%.3f" % m for m in ms))
ms = h.time_hash_device(b, warmup=0, iters=20)
print("straight after:", " ".join("%.3f" % m for m in ms))
import time; time.sleep(2)
ms = h.time_hash_device(b, warmup=0, iters=20)
print("after 2 s idle:", " ".join("%.3f" % m for m in ms))
