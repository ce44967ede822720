#!/usr/bin/env python3
"""Interleaved A/B of the SHA-1 kernel variants on config 2 in ONE process
(rounds of every variant back to back, median and min per variant), checking
that all variants produce identical digests.  Development tool."""
import statistics
import sys

import numpy as np

import vit_amd

NAMES = {0: "lds<2>x4", 1: "lane", 2: "lds<1>x1", 3: "lds<4>x1", 4: "lds<2>x2", 5: "lds<2>x1"}


def main():
    variants = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2, 3]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    h = vit_amd.Hasher((0,))
    b = h.synth_arena("blob", 0x5EED0002, 0, n, L, slot_align=128)
    ref = None
    times = {v: [] for v in variants}
    for v in variants:                                   # correctness + warm-up
        h.set_option("kernel", v)
        d = h.hash_device(b).copy()
        if ref is None:
            ref = d
        assert (d == ref).all(), "variant %d differs" % v
    for _ in range(rounds):
        for v in variants:
            h.set_option("kernel", v)
            times[v] += h.time_hash_device(b, warmup=0, iters=3)
    pay = n * L
    for v in variants:
        t = times[v]
        print("%-7s median %.4f ms  min %.4f ms  -> %.1f GB/s (median)  %.1f GB/s (best)"
              % (NAMES.get(v, str(v)), statistics.median(t), min(t), pay / statistics.median(t) / 1e6, pay / min(t) / 1e6),
              flush=True)


if __name__ == "__main__":
    main()
