#!/usr/bin/env python3
"""Config 2 (1 M x 4 KiB, uniform) under the software-scheduled launch against the hardware-dispatched one:
what the scheduler costs on a batch that does not need it.  Interleaved, medians."""
import json
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_amd  # noqa: E402

h = vit_amd.Hasher((0,))
n = 1_000_000
b = h.synth_arena("blob", 0x5EED0002, 0, n, 4096, slot_align=128)
order = h.alloc(4 * n).upload(np.arange(n, dtype=np.uint32))
h.set_option("sched_min_groups", 1)
h.set_option("sched_tail_percent", 0)
res = {}
for rep in range(3):
    for waves in (0, 1, 2, 3, 4, 5):
        h.set_option("sched_waves", waves)
        ms = h.time_hash_device(b, order=order, warmup=2, iters=20)
        res.setdefault(waves, []).append(statistics.median(ms))
h.set_option("sched_waves", 0)
plain = statistics.median(h.time_hash_device(b, warmup=2, iters=20))
print(json.dumps({"no_order_hw_dispatch_ms": round(plain, 4),
                  "identity_order_ms_by_sched_waves": {str(k): round(statistics.median(v), 4) for k, v in res.items()}}))
