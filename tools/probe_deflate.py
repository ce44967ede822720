#!/usr/bin/env python3
"""The deflate kernel (vit_deflate_arena_device) on device-resident synthetic blobs: every stream compared with
zlib.compress(header + content) from the zlib library, and the time per batch.
    python tools/probe_deflate.py [n] [len]        len 0 = mixed sizes up to 60,000 bytes"""
import json, os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 0
check = int(sys.argv[3]) if len(sys.argv) > 3 else min(n, 5000)
h = vit_amd.Hasher((0,))
h.set_option("deflate_max_bytes", 65536)       # mixed sizes: the objects beyond what a workgroup takes in LDS too (the lane kernel)
if os.environ.get("VIT_DEFLATE_LANES"):
    h.set_option("deflate_lanes", int(os.environ["VIT_DEFLATE_LANES"]))
rng = np.random.default_rng(8)
lens = np.full(n, L, dtype=np.uint64) if L else np.minimum(np.exp(rng.normal(np.log(3000.0), 1.3, n)), 60000).astype(np.uint64)
b = h.synth_arena("blob", 0x5EED00AA, 0, n, 0, lens=lens)
framed = [int(l) + vit_amd.header_len("blob", int(l)) for l in lens]
order = None
if not L:                                      # mixed sizes: objects by length, so a wave's 64 end together
    nbk = b["nblocks"].download(np.uint32, 4 * n)
    order = h.alloc(4 * n).upload(vit_amd.bin_order(nbk)[0])
out, dt = h.deflate_device(b, framed, order)   # first call allocates the workspace
out, dt = h.deflate_device(b, framed, order)
bad = 0
for i in rng.choice(n, size=check, replace=False) if check < n else range(n):
    data = vit_amd.synth_bytes(0x5EED00AA, int(i), int(lens[i]))
    want = zlib.compress(b"blob %d\0" % len(data) + data, 6)
    if out[int(i)] != want:
        bad += 1
        if bad < 4:
            print("MISMATCH object", int(i), "len", int(lens[i]), "got", len(out[int(i)]), "want", len(want), file=sys.stderr)
print(json.dumps({"objects": n, "payload_bytes": int(lens.sum()), "seconds": round(dt, 4), "objects_per_s": round(n / dt, 1),
                  "GBps": round(float(lens.sum()) / dt / 1e9, 3), "checked": int(check), "mismatches": bad,
                  "device_status": h.device_status(0)}))
sys.exit(1 if bad else 0)
