#!/usr/bin/env python3
"""vit_deflate_objects on text-like blobs (words drawn with a Zipf law: matches everywhere, unlike the random bytes of the
other probes): seconds per batch, streams against zlib.compress, and one host core's zlib on a sample of the same blobs.
    python tools/probe_deflate_text.py [n] [len]"""
import json, os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rng = np.random.default_rng(12)
vocab = [bytes(rng.integers(97, 123, int(k)).astype(np.uint8)) for k in rng.integers(2, 10, 3000)]
vocab += [b"    ", b"\n", b"{", b"}", b"(", b")", b";", b"return ", b"const ", b"int ", b"if (", b"for ("]
pick = np.minimum(rng.zipf(1.3, 16_000_000), len(vocab)) - 1
text = b" ".join(vocab[int(k)] for k in pick[:12_000_000])
starts = rng.integers(0, len(text) - L, n)
blobs = [text[int(s):int(s) + L] for s in starts]
h = vit_amd.Hasher((0,))
h.set_option("deflate_max_bytes", 65536)
if os.environ.get("VIT_DEFLATE_CHUNK"):
    h.set_option("deflate_chunk_bytes", int(os.environ["VIT_DEFLATE_CHUNK"]))
if os.environ.get("VIT_DEFLATE_LANES"):
    h.set_option("deflate_lanes", int(os.environ["VIT_DEFLATE_LANES"]))
got, dt = h.deflate_objects("blob", blobs)      # first call allocates
got, dt = h.deflate_objects("blob", blobs)
bad = 0
t0 = time.perf_counter()
m = min(n, 4000)
for i in range(m):
    if got[i] != zlib.compress(b"blob %d\0" % L + blobs[i], 6):
        bad += 1
host = time.perf_counter() - t0
ratio = sum(len(z) for z in got[:m]) / (m * L)
print(json.dumps({"objects": n, "bytes_each": L, "content": "text-like (Zipf words)", "compressed_to": round(ratio, 3),
                  "seconds_in_call": round(dt, 4), "objects_per_s": round(n / dt, 1), "GBps": round(n * L / dt / 1e9, 3),
                  "checked": m, "mismatches": bad, "host_zlib_one_core_objects_per_s": round(m / host, 1),
                  "device_status": h.device_status()}))
sys.exit(1 if bad else 0)
