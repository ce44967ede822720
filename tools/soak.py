#!/usr/bin/env python3
"""Soak: random batches through vit_hash_blobs / vit_sha1_buffers under random pipeline settings
(chunk sizes, host-lane thresholds, thread budget, hardware or software scheduling of binned chunks, one or two
queues on GPU 0, both kernels), device-resident batches through the binning entry point that allocates nothing, every id compared with
hashlib, for a given number of seconds.  Prints one JSON line; exits non-zero at the first mismatch."""
import hashlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_amd  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
pool = np.frombuffer(vit_amd.synth_bytes(99, seed, 64 << 20), dtype=np.uint8)
hashers = [vit_amd.Hasher((0,)), vit_amd.Hasher((0, 0))]
t_end = time.time() + seconds
t_say = time.time() + 30
batches = objects = nbytes = 0
while time.time() < t_end:
    h = hashers[int(rng.integers(0, 2))]
    h.set_option("kernel", int(rng.choice([0, 1, 5])))
    h.set_option("chunk_bytes", int(rng.choice([1 << 20, 4 << 20, 64 << 20])))
    h.set_option("chunk_object_multiple", int(rng.choice([0, 64, 8192])))
    h.set_option("sched_waves", int(rng.choice([0, 1, 2])))            # software-scheduled launch of binned chunks
    h.set_option("sched_min_groups", int(rng.choice([1, 16, 4096])))
    h.set_option("sched_tail_percent", int(rng.choice([0, 1, 75])))
    h.set_option("sched_share_below", int(rng.choice([0, 128, 1 << 20])))
    h.set_option("pack_threads", int(rng.choice([3, 8, 16])))
    h.set_option("host_min_bytes", int(rng.choice([1, 4096, 16384])))   # how short an object the balance may move
    mode = int(rng.integers(0, 3))
    if mode == 0:
        h.set_option("host_auto", 1)
    else:
        h.set_option("host_threshold_bytes", int(rng.choice([0, 4096, 1 << 20])))
    n = int(rng.choice([1, 7, 64, 65, 300, 5000, 40000]))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        lens = rng.integers(0, 200, size=n)
    elif kind == 1:
        lens = rng.integers(0, 9000, size=n)
    elif kind == 2:
        lens = np.minimum(np.exp(rng.normal(np.log(3000.0), 1.8, n)), 6 << 20).astype(np.int64)
    else:
        lens = np.full(n, int(rng.choice([0, 55, 56, 64, 4096, 70000])))
    if int(lens.sum()) > (1 << 30):
        lens = lens // 8
    starts = rng.integers(0, len(pool) - int(lens.max()) - 1, size=n)
    bufs = [pool[int(s):int(s) + int(L)] for s, L in zip(starts, lens)]
    raw = bool(rng.integers(0, 2))
    if n <= 5000 and int(rng.integers(0, 4)) == 0:         # now and then: a pre-packed arena, 64- or 128-byte slots
        raw = False
        arena, off, nb, used = vit_amd.pack_arena("blob", bufs, slot_align=int(rng.choice([64, 128])), threads=3)
        got = h.hash_arena(arena, off, nb, arena_bytes=used)
    else:
        got = h.sha1_buffers(bufs) if raw else h.hash_blobs(bufs)
    check = range(n) if n <= 5000 else rng.choice(n, size=3000, replace=False)
    for i in check:
        b = bufs[int(i)].tobytes()
        want = hashlib.sha1(b if raw else b"blob %d\0" % len(b) + b).digest()
        if got[int(i)].tobytes() != want:
            print(json.dumps({"mismatch": True, "batch": batches, "object": int(i), "len": len(b), "raw": raw, "n": n}))
            sys.exit(1)
    batches += 1; objects += n; nbytes += int(lens.sum())
    if time.time() > t_say:                                # a long silent run looks hung to whoever watches it
        print("... %d batches, %d objects, no mismatch" % (batches, objects), file=sys.stderr, flush=True)
        t_say = time.time() + 30
    if batches % 7 == 0:                                   # a device-resident batch: plain launch against the static-scratch launch
        m = int(rng.choice([1, 65, 3000, 200000]))
        dl = np.minimum(np.exp(rng.normal(np.log(900.0), 1.7, m)), 150000).astype(np.uint64)
        b = h.synth_arena("blob", 1234 + batches, 0, m, 0, lens=dl)
        h.set_option("sched_waves", 0)
        want = h.hash_device(b).copy()
        h.set_option("sched_waves", int(rng.choice([0, 1, 2])))
        if not (h.hash_device_static(b) == want).all() or h.device_status(0) != 0:
            print(json.dumps({"mismatch": True, "batch": batches, "static_entry": True, "n": m}))
            sys.exit(1)
        j = int(rng.integers(0, m))
        data = vit_amd.synth_bytes(1234 + batches, j, int(dl[j]))
        if want[j].tobytes() != hashlib.sha1(b"blob %d\0" % len(data) + data).digest():
            print(json.dumps({"mismatch": True, "batch": batches, "device_resident": True, "n": m, "object": j}))
            sys.exit(1)
        del b
print(json.dumps({"seconds": seconds, "seed": seed, "batches": batches, "objects": objects, "GB": round(nbytes / 1e9, 2),
                  "mismatches": 0}))
