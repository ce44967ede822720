#!/usr/bin/env python3
"""Soak of the deflate kernel: for `seconds`, batches of objects of every kind of content (random bytes, text-like, one
byte repeated, a four-letter alphabet, a repeated 700-byte base, near copies at a distance) and of sizes from 0 to the
device's 65,536 framed bytes go through vit_deflate_objects, and every stream is compared with zlib.compress.
    python tools/soak_deflate.py [seconds] [seed]"""
import json, os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_amd  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
vocab = [bytes(rng.integers(97, 123, int(k)).astype(np.uint8)) for k in rng.integers(2, 10, 2000)] + [b"    ", b"\n", b"{", b"}", b"return "]
pick = np.minimum(rng.zipf(1.3, 1_500_000), len(vocab)) - 1
text = b" ".join(vocab[int(k)] for k in pick)
noise = rng.integers(0, 256, 8 << 20, dtype=np.uint8).tobytes()


def make(kind, L):
    if kind == 0:
        s = int(rng.integers(0, len(noise) - L - 1)); return noise[s:s + L]
    if kind == 1:
        s = int(rng.integers(0, len(text) - L - 1)); return text[s:s + L]
    if kind == 2:
        return bytes([int(rng.integers(0, 256))]) * L
    if kind == 3:
        return (rng.integers(0, 4, L, dtype=np.uint8) + 97).tobytes()
    if kind == 4:
        s = int(rng.integers(0, len(noise) - 701)); return (noise[s:s + 700] * (L // 700 + 1))[:L]
    s = int(rng.integers(0, len(text) - 8000)); base = text[s:s + 5000]          # near copies 5,000 bytes apart
    out = bytearray((base * (L // 5000 + 1))[:L])
    for k in rng.integers(0, max(L, 1), L // 97).tolist():
        out[k] ^= 1
    return bytes(out)


h = vit_amd.Hasher((0,))
h.set_option("deflate_max_bytes", 65536)
t0 = time.time(); last = t0
batches = objects = nbytes = bad = 0
while time.time() - t0 < seconds:
    n = int(rng.integers(1, 6000))
    top = int(rng.choice([300, 5000, 17000, 65526]))
    blobs = [make(int(rng.integers(0, 6)), int(rng.integers(0, top))) for _ in range(n)]
    got, _ = h.deflate_objects("blob", blobs)
    for d, z in zip(blobs, got):
        if z != zlib.compress(b"blob %d\0" % len(d) + d, 6):
            bad += 1
    batches += 1; objects += n; nbytes += sum(len(d) for d in blobs)
    if time.time() - last > 20:
        last = time.time()
        print("... %d batches, %d objects, %d mismatches" % (batches, objects, bad), file=sys.stderr, flush=True)
print(json.dumps({"seconds": seconds, "seed": seed, "batches": batches, "objects": objects, "GB": round(nbytes / 1e9, 2),
                  "mismatches": bad, "device_status": h.device_status()}))
sys.exit(1 if bad else 0)
