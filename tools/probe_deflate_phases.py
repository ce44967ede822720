#!/usr/bin/env python3
"""Where a workgroup of the workgroup-per-object deflate kernels spends its cycles.  While the "wave_trace_ptr" option
names a buffer of 128 bytes, the wave of every workgroup that runs the serial parts adds its shader-clock cycles by
phase to its first eight counters (load + checksum, chains, waiting for the searches, decisions, symbols to scratch,
trees by lanes, block content; last: objects done), and one searching wave of every workgroup adds what it did to the
next six (cycles while a tile's positions are handed out and after, rounds of either, cycles at the tiles' barriers,
chain entries looked at after the tile was handed out).  Printed per object.
    python tools/probe_deflate_phases.py [n] [len] [random|text]"""
import json, os, sys, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
kind = sys.argv[3] if len(sys.argv) > 3 else "random"
h = vit_amd.Hasher((0,))
rng = np.random.default_rng(12)
if kind == "text":
    vocab = [bytes(rng.integers(97, 123, int(k)).astype(np.uint8)) for k in rng.integers(2, 10, 3000)]
    vocab += [b"    ", b"\n", b"{", b"}", b"(", b")", b";", b"return ", b"const ", b"int ", b"if (", b"for ("]
    pick = np.minimum(rng.zipf(1.3, 3_000_000), len(vocab)) - 1
    text = b" ".join(vocab[int(k)] for k in pick)
    starts = rng.integers(0, len(text) - L, n)
    blobs = [text[int(s):int(s) + L] for s in starts]
else:
    blobs = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for _ in range(n)]
trace = h.alloc(128)
names = ["load+checksum", "chains", "waiting_for_searches", "decisions", "symbols_out", "trees_by_lanes", "block_content"]
got, dt = h.deflate_objects("blob", blobs)
trace.upload(np.zeros(16, dtype=np.uint64))
h.set_option("wave_trace_ptr", trace.ptr)
got, dt = h.deflate_objects("blob", blobs)
h.set_option("wave_trace_ptr", 0)
t = trace.download(np.uint64, 128)
bad = sum(1 for i in range(0, n, max(1, n // 500)) if got[i] != zlib.compress(b"blob %d\0" % L + blobs[i], 6))
objs = max(int(t[7]), 1)
print(json.dumps({"objects": n, "bytes_each": L, "content": kind, "seconds_in_call": round(dt, 4), "objects_per_s": round(n / dt, 1),
                  "cycles_per_object_by_phase": {k: int(t[i] // objs) for i, k in enumerate(names)},
                  "cycles_per_object": int(t[:7].sum() // objs), "one_searching_wave_per_object": dict(zip(["cycles_handing_out", "cycles_after", "rounds_handing_out", "rounds_after", "cycles_at_tile_barriers", "entries_after"],
                                                        [round(float(x) / objs, 1) for x in t[8:14]])), "objects_traced": int(t[7]), "mismatches_in_sample": bad}))
