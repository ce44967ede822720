#!/usr/bin/env python3
"""Where a wave of the wave-per-object deflate kernel spends its cycles: the kernels add their shader-clock cycles by
phase, as thread 0 of every workgroup sees them, to a buffer the "wave_trace_ptr" option names (load + checksum, chains,
searches, decisions, symbols to scratch, the group's blocks by lanes, stored bytes; last: objects done).
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
