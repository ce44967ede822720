"""Coarse guards on speed, collected after every parity test: a noisy box may turn one of these red, and
`pytest -x` must by then have run everything that checks ids.  Not benchmarks (bench.py is); each bound is
a factor that a past regression exceeded."""
import statistics

import numpy as np
import pytest

from test_gpu_parity import skewed_sizes_beyond_4gib, tail_cases

pytestmark = pytest.mark.gpu


def test_choice_between_the_launches_costs_little(hasher):
    """Lengths known to the device only: both launches are queued and the waves of the one not chosen leave
    at once.  The choice must land on the faster launch, give or take the cost of the launch that steps aside
    (a batch of short objects ran 3.3 times slower under the software scheduler)."""
    import vit_amd
    for name, lens in tail_cases().items():
        n = len(lens)
        b = hasher.synth_arena("blob", 0x5EED0090, 0, n, 0, lens=lens)
        nb = b["nblocks"].download(np.uint32, 4 * n)
        order_host, _ = vit_amd.bin_order(nb)
        order = hasher.alloc(4 * n).upload(order_host)
        hasher.set_option("sched_waves", 0)
        t_plain = statistics.median(hasher.time_hash_device(b, order=order, warmup=1, iters=5))
        hasher.set_option("sched_waves", 1)
        hasher.set_option("sched_tail_percent", 0)
        t_forced = statistics.median(hasher.time_hash_device(b, order=order, warmup=1, iters=5))
        hasher.set_option("sched_tail_percent", 75)
        t_auto = statistics.median(hasher.time_hash_device(b, order=order, warmup=1, iters=5))
        assert hasher.device_status() == 0
        print("%s: plain %.3f ms, forced %.3f ms, chosen %.3f ms" % (name, t_plain, t_forced, t_auto))
        assert t_auto < 1.25 * min(t_plain, t_forced) + 0.03, (name, t_plain, t_forced, t_auto)


def test_binned_skewed_arena_beyond_4gib_keeps_its_rate(hasher):
    """A skewed batch in an arena beyond 4 GiB, hashed in binned order (every wave's objects lie all over the
    arena), must stay within 2.2x of the time the same number of blocks takes on uniform objects.  (Sending
    such waves to a plain per-lane loop once cost 2.6x.)"""
    import vit_amd
    sizes = skewed_sizes_beyond_4gib()
    n = len(sizes)
    b = hasher.synth_arena("blob", 0x5EED0044, 0, n, 0, lens=sizes)
    nb = b["nblocks"].download(np.uint32, 4 * n)
    order_host, _ = vit_amd.bin_order(nb)
    order = hasher.alloc(4 * n).upload(order_host)
    skew_ms = min(hasher.time_hash_device(b, order=order, warmup=1, iters=3))
    del b
    u = hasher.synth_arena("blob", 1, 0, 200_000, 4096)
    uni_ms = min(hasher.time_hash_device(u, warmup=1, iters=3))
    per_block = uni_ms / (200_000 * 65)
    floor_ms = float(nb.astype(np.uint64).sum()) * per_block
    print("skewed %.3f ms, floor %.3f ms, ratio %.2f" % (skew_ms, floor_ms, skew_ms / floor_ms))
    assert skew_ms < 2.2 * floor_ms, (skew_ms, floor_ms)


def test_small_objects_are_deflated_by_workgroups_not_by_lanes(hasher):
    """65,536 text-like objects of 4 KiB through vit_deflate_objects: the kernel that takes one object per workgroup
    with its tables in LDS, chunks overlapped, gets through 1.7 M of them per second, the one-object-per-lane kernel
    it replaced through 0.6 M.  Falling back to the slow path (objects sent to the lane kernel, one chunk in flight,
    walk loads that the compiler serialises) shows as a rate below 0.9 M."""
    import zlib
    rng = np.random.default_rng(12)
    vocab = [bytes(rng.integers(97, 123, int(k)).astype(np.uint8)) for k in rng.integers(2, 10, 3000)]
    vocab += [b"    ", b"\n", b"{", b"}", b"(", b")", b";", b"return ", b"const ", b"int ", b"if (", b"for ("]
    pick = np.minimum(rng.zipf(1.3, 2_000_000), len(vocab)) - 1
    text = b" ".join(vocab[int(k)] for k in pick)
    n, L = 65536, 4096
    starts = rng.integers(0, len(text) - L, n)
    blobs = [text[int(s):int(s) + L] for s in starts]
    hasher.deflate_objects("blob", blobs)                   # the first call allocates
    got, dt = min((hasher.deflate_objects("blob", blobs) for _ in range(3)), key=lambda r: r[1])
    for i in range(0, n, 257):
        assert got[i] == zlib.compress(b"blob %d\0" % L + blobs[i], 6), i
    print("%.2f M objects/s" % (n / dt / 1e6))
    assert n / dt > 0.9e6, dt
