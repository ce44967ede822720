#!/usr/bin/env python3
"""Per-wave timeline of the skewed resident batch: when each wave ran and on which SIMD.

    python tools/wave_timeline.py [n] [deal]     deal: the longest wave of every SIMD first, then the shortest waves
"""
import json, os, sys, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000

h = vit_amd.Hasher((0,))
# VIT_SCHED_WAVES=0: hardware dispatch; 1..5: the software-scheduled launch with that many waves per SIMD
h.set_option("sched_tail_percent", 0)          # the environment picks the launch, not the lengths
for opt, env in (("sched_waves", "VIT_SCHED_WAVES"), ("sched_share_below", "VIT_SHARE_BELOW")):
    if os.environ.get(env) is not None:
        h.set_option(opt, int(os.environ[env]))
rng = np.random.default_rng(0x5EED0004)
sizes = np.exp(rng.normal(np.log(4500.0), 1.55, n))
sizes = np.minimum(sizes * (15000.0 / sizes.mean()), 262144).astype(np.uint64)
if len(sys.argv) > 2 and sys.argv[2] == "uniform":
    sizes[:] = 4096
b = h.synth_arena("blob", 0x5EED0004, 0, n, 0, lens=sizes)
nb = b["nblocks"].download(np.uint32, 4 * n)
order_host, _ = vit_amd.bin_order(nb)
if len(sys.argv) > 2 and sys.argv[2] == "deal":
    # the longest wave of each SIMD first, then the shortest waves into the other four slots of every SIMD
    head = 1024 * 64
    tail = min(4096 * 64, max(n - head, 0)) // 256 * 256
    order_host = np.concatenate([order_host[:head], order_host[n - tail:], order_host[head:n - tail]]).astype(np.uint32)
    assert len(np.unique(order_host)) == n
order = h.alloc(4 * n).upload(order_host)
waves = (n + 255) // 256 * 4
tr = h.alloc(32 * waves)
tr.upload(np.zeros(4 * waves, dtype=np.uint64))
h.time_hash_device(b, order=order, warmup=2, iters=0)
h.set_option("wave_trace_ptr", tr.ptr)
ms = h.time_hash_device(b, order=order, warmup=0, iters=1)
h.set_option("wave_trace_ptr", 0)
rec = tr.download(np.uint64, 32 * waves).reshape(-1, 4)
rec = rec[rec[:, 1] > 0]
t0 = rec[:, 0].min()
start = (rec[:, 0] - t0) / 100.0      # us (100 MHz clock)
end = (rec[:, 1] - t0) / 100.0
hw = rec[:, 2] & 0xFFFFFFFF
xcc = (rec[:, 2] >> 32) & 0xF
simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 64 + cu * 4 + simd
far = (rec[:, 3] >> 40) & 1
mx = rec[:, 3] & 0xFFFFFFFF
print('far waves', int(far.sum()))
print(json.dumps({"kernel_ms": round(ms[0], 3), "waves": int(len(rec)), "span_us": round(float(end.max()), 1),
                  "distinct_simds": int(len(np.unique(key)))}))
# when does each SIMD go idle for good, and how much work (wave-blocks) did it get
last = collections.defaultdict(float); work = collections.defaultdict(int); cnt = collections.defaultdict(int)
for k, e, m in zip(key, end, mx):
    last[k] = max(last[k], e); work[k] += int(m); cnt[k] += 1
ends = np.array(sorted(last.values())); works = np.array([work[k] for k in last])
print("SIMD finish time us: min %.0f p10 %.0f median %.0f p90 %.0f max %.0f" % (ends[0], np.percentile(ends, 10), np.median(ends), np.percentile(ends, 90), ends[-1]))
print("wave-blocks per SIMD: min %d median %d max %d  (sum %d)" % (works.min(), np.median(works), works.max(), works.sum()))
cnts = np.array([cnt[k] for k in last])
print("groups per SIMD: min %d median %d max %d" % (cnts.min(), np.median(cnts), cnts.max()))
# how many groups a SIMD runs at once, sampled at a few instants
for t in (50, 200, 400, 800, 1000):
    live = collections.Counter(key[(start <= t) & (end > t)].tolist())
    if live:
        v = np.array(list(live.values()))
        print("at %4d us: %d SIMDs busy, groups in flight per busy SIMD: min %d median %d max %d" % (t, len(v), v.min(), np.median(v), v.max()))
# the waves that end last
idx = np.argsort(-end)[:12]
for i in idx:
    print("late wave: blocks %5d start %7.0f end %7.0f dur %7.0f simd-key %d (that SIMD got %d waves, %d wave-blocks)" % (
        mx[i], start[i], end[i], end[i] - start[i], key[i], cnt[key[i]], work[key[i]]))
# duration per block by length class
for lo, hi in ((1, 8), (8, 64), (64, 256), (256, 1024), (1024, 2048), (2048, 5000)):
    m = (mx >= lo) & (mx < hi)
    if m.any():
        print("waves of %4d..%4d blocks: %6d waves, us per block: median %.2f  p90 %.2f; start median %.0f us" % (
            lo, hi, m.sum(), np.median((end[m] - start[m]) / mx[m]), np.percentile((end[m] - start[m]) / mx[m], 90), np.median(start[m])))
