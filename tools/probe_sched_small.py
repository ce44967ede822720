#!/usr/bin/env python3
"""Binned batches of SHORT objects under the software-scheduled launch against the hardware dispatcher:
1 M objects of 0..256 B (config 5's tiny files) and of 0..8 KiB, resident in HBM.  The scheduler exists for
long tails; this measures what it does to batches without one."""
import json
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_amd  # noqa: E402

h = vit_amd.Hasher((0,))
h.set_option("sched_min_groups", 1)
h.set_option("sched_tail_percent", 0)          # forced: this measures the launch, not the choice
out = {}
for label, hi in (("0..256B", 257), ("0..8KiB", 8193)):
    n = 1_000_000
    lens = np.random.default_rng(5).integers(0, hi, n).astype(np.uint64)
    b = h.synth_arena("blob", 0x5EED0055, 0, n, 0, lens=lens)
    nb = b["nblocks"].download(np.uint32, 4 * n)
    order_host, _ = vit_amd.bin_order(nb)
    order = h.alloc(4 * n).upload(order_host)
    res = {}
    for waves, share in ((0, 0), (1, 128), (2, 128), (2, 1 << 20)):
        h.set_option("sched_waves", waves)
        h.set_option("sched_share_below", share)
        ms = h.time_hash_device(b, order=order, warmup=2, iters=10)
        res["waves %d share_below %d" % (waves, share)] = round(statistics.median(ms), 4)
    h.set_option("sched_waves", 1); h.set_option("sched_share_below", 128); h.set_option("sched_tail_percent", 75)
    h.set_option("sched_min_groups", 4096)
    res["library's choice"] = round(statistics.median(h.time_hash_device(b, order=order, warmup=2, iters=10)), 4)
    h.set_option("sched_tail_percent", 0); h.set_option("sched_min_groups", 1)
    out[label] = res
    del b, order
print(json.dumps(out))
