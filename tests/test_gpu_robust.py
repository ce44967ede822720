"""Behaviour of the host pipeline around the kernels: failure in mid-batch, the chunked pre-packed
arena, the calibrated host/GPU balance, the shared thread budget.  Through the C ABI; ids bit-exact."""
import hashlib
import os
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def git_ids(blobs, type_=b"blob"):
    return np.array([np.frombuffer(hashlib.sha1(type_ + b" %d\0" % len(b) + bytes(b)).digest(), dtype=np.uint8)
                     for b in blobs]).reshape(-1, 20)


def test_failure_in_mid_batch_leaves_nothing_behind(hasher, oracle):
    """A call that fails after some chunks are queued must not deliver those chunks' digests into the
    NEXT call's (smaller) output buffer, and the hasher must keep working."""
    import vit_amd
    lens = [(i * 37) % 5000 for i in range(4000)]
    blobs = [oracle.synth_bytes(61, i, L) for i, L in enumerate(lens)]
    hasher.set_option("chunk_bytes", 1 << 20)
    try:
        for at in (2, 3, 5):
            hasher.set_option("debug_fail_at_chunk", at)
            with pytest.raises(vit_amd.VitError) as e:
                hasher.hash_blobs(blobs)
            assert e.value.code == -2 and "injected" in str(e.value)
            hasher.set_option("debug_fail_at_chunk", 0)
            # a much smaller batch next: stale chunks would write far past its 3 x 20 bytes
            small = [b"a", b"", b"hello\n"]
            guard = np.full((64, 20), 0xEE, dtype=np.uint8)
            keep, ptrs, ln = vit_amd.hasher._as_ptr_arrays(small)
            hasher._check(hasher.lib.vit_hash_blobs(hasher.handle, ptrs, ln.ctypes.data, 3, guard.ctypes.data))
            assert (guard[:3] == git_ids(small)).all()
            assert (guard[3:] == 0xEE).all()
            # the pre-packed entry point shares the staging sets
            arena, off, nb, used = vit_amd.pack_arena("blob", blobs[:300])
            assert (hasher.hash_arena(arena, off, nb) == git_ids(blobs[:300])).all()
        assert (hasher.hash_blobs(blobs) == git_ids(blobs)).all()
    finally:
        hasher.set_option("debug_fail_at_chunk", 0)
        hasher.set_option("chunk_bytes", 64 << 20)


def test_failure_with_two_queues_and_host_lane(oracle):
    import vit_amd
    lens = [(i * 53) % 9000 for i in range(3000)] + [1 << 20, 3 << 20]
    blobs = [oracle.synth_bytes(62, i, L) for i, L in enumerate(lens)]
    with vit_amd.Hasher((0, 0)) as h2:
        h2.set_option("chunk_bytes", 1 << 20)
        h2.set_option("host_threshold_bytes", 1 << 19)
        h2.set_option("debug_fail_at_chunk", 2)
        with pytest.raises(vit_amd.VitError):
            h2.hash_blobs(blobs)
        h2.set_option("debug_fail_at_chunk", 0)
        assert (h2.hash_blobs(blobs[:5]) == git_ids(blobs[:5])).all()
        assert (h2.hash_blobs(blobs) == git_ids(blobs)).all()


def test_prepacked_arena_goes_over_in_pieces(hasher, oracle):
    """vit_hash_arena on an arena several times chunk_bytes: pieces alternate between the two staging
    sets (copy of one beside the kernel of the other); a shuffled offset table takes the one-piece path."""
    import vit_amd
    rng = np.random.default_rng(8)
    lens = rng.integers(0, 30000, size=1500).tolist() + [300_000, 0, 1 << 20, 17]
    blobs = [oracle.synth_bytes(63, i, L) for i, L in enumerate(lens)]
    want = git_ids(blobs)
    arena, off, nb, used = vit_amd.pack_arena("blob", blobs)
    assert used > (20 << 20)
    hasher.set_option("chunk_bytes", 1 << 20)
    try:
        l0 = hasher.stat("launches")
        got = hasher.hash_arena(arena, off, nb, arena_bytes=used)
        assert hasher.stat("launches") - l0 >= 15
        assert (got == want).all()
        perm = rng.permutation(len(lens))
        l0 = hasher.stat("launches")
        got = hasher.hash_arena(arena, off[perm], nb[perm], arena_bytes=used)
        assert hasher.stat("launches") - l0 == 1
        assert (got == want[perm]).all()
    finally:
        hasher.set_option("chunk_bytes", 64 << 20)
    assert hasher.device_status(0) == 0


def test_uniform_batch_of_large_objects_is_shared_not_dumped_on_the_host(hasher):
    """Auto policy on 3 000 x 256 KiB: equal lengths must not all follow the first one to the host
    threads; both sides get work, and the split follows the measured rates when they are perturbed."""
    import vit_amd
    n, L = 3000, 256 << 10
    buf = np.frombuffer(vit_amd.synth_bytes(64, 0, L + n), dtype=np.uint8)
    blobs = [buf[i:i + L] for i in range(n)]                       # overlapping views: n distinct contents, little memory
    hasher.set_option("host_auto", 1)
    g0, h0 = hasher.stat("objects_gpu"), hasher.stat("objects_host")
    got = hasher.hash_blobs(blobs)
    on_gpu, on_host = hasher.stat("objects_gpu") - g0, hasher.stat("objects_host") - h0
    assert on_gpu + on_host == n and on_gpu > 0 and on_host > 0, (on_gpu, on_host)
    assert hasher.stat("last_host_objects") == on_host
    for i in (0, 1, n // 2, n - 1):
        assert got[i].tobytes() == hashlib.sha1(b"blob %d\0" % L + blobs[i].tobytes()).digest()
    rates = {k: hasher.stat(k) for k in ("lane_rate", "core_rate", "core_rate_multi", "copy_rate", "pack_rate")}
    print("calibrated rates (bytes/s):", rates, "in", hasher.stat("calibrate_ns") / 1e6, "ms; split gpu/host", on_gpu, on_host)
    assert 10e6 < rates["lane_rate"] < 200e6 and 0.2e9 < rates["core_rate"] < 10e9
    assert 5e9 < rates["copy_rate"] < 100e9 and 1e9 < rates["pack_rate"] < 100e9
    try:
        # hold the rates where they are (the packing rate otherwise follows what each large batch achieves)
        for k, v in rates.items():
            hasher.set_option(k, v)
        assert (hasher.hash_blobs(blobs) == got).all()
        base = hasher.stat("last_host_objects")
        assert 0 < base < n
        # a host core believed twice as fast takes more objects, half as fast fewer; ids unchanged
        moved = {}
        for scale in (0.5, 2.0):
            hasher.set_option("core_rate", int(rates["core_rate"] * scale))
            hasher.set_option("core_rate_multi", int(rates["core_rate_multi"] * scale))
            assert (hasher.hash_blobs(blobs) == got).all()
            moved[scale] = hasher.stat("last_host_objects")
        assert moved[0.5] < base < moved[2.0], (moved, base)
        hasher.set_option("core_rate", rates["core_rate"])
        hasher.set_option("core_rate_multi", rates["core_rate_multi"])
        # a lane believed twice as fast keeps more objects on the GPU (or as many)
        hasher.set_option("lane_rate", int(rates["lane_rate"] * 2))
        hasher.hash_blobs(blobs[:200])
        fast_lane = hasher.stat("last_host_objects")
        hasher.set_option("lane_rate", int(rates["lane_rate"] / 2))
        hasher.hash_blobs(blobs[:200])
        assert fast_lane <= hasher.stat("last_host_objects")
    finally:
        for k in rates:
            hasher.set_option(k, 0)


def test_balance_learns_from_when_the_two_sides_finish(hasher):
    """The host cores believed 2.5 times as fast as they are: the first call leaves the host threads working long
    after the device queues are done; the factor the hasher keeps on the host side's estimate then grows call by
    call until the two sides finish together again.  Ids never change."""
    import vit_amd
    rng = np.random.default_rng(21)
    n = 60_000
    lens = np.minimum(np.exp(rng.normal(np.log(30000.0), 1.0, n)), 4 << 20).astype(np.uint64)     # 3 GB, half of it >= 16 KiB objects
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    buf = np.frombuffer(vit_amd.synth_bytes(77, 0, 8 << 20), dtype=np.uint8)
    buf = np.tile(buf, int(offs[-1]) // len(buf) + 1)
    blobs = [buf[int(offs[i]):int(offs[i + 1])] for i in range(n)]
    want = hasher.hash_blobs(blobs)                                   # calibrates; staging memory pinned
    hasher.hash_blobs(blobs)
    rates = {k: hasher.stat(k) for k in ("lane_rate", "core_rate", "core_rate_multi", "copy_rate", "pack_rate")}
    try:
        for k, v in rates.items():
            hasher.set_option(k, int(v * 2.5) if k.startswith("core") else v)
        hasher.set_option("host_bias_percent", 100)
        ratios = []
        for _ in range(7):
            assert (hasher.hash_blobs(blobs) == want).all()
            ratios.append(hasher.stat("last_host_done_ns") / hasher.stat("last_gpu_done_ns"))
        print("host done / gpu done per call:", [round(r, 2) for r in ratios], "bias", hasher.stat("host_bias_percent"))
        assert ratios[0] > 1.3 and 0.7 < sorted(ratios[-3:])[1] < 1.3 and hasher.stat("host_bias_percent") > 125, ratios
    finally:
        for k in rates:
            hasher.set_option(k, 0)
        hasher.set_option("host_bias_percent", 100)


def _threads_now():
    return len(os.listdir("/proc/self/task"))


def test_four_queues_share_one_thread_budget(oracle):
    """Four device queues on one GPU: the pool is sized once for the hasher, the per-queue threads
    count against the budget, and the process never runs more threads than budget + what it had."""
    import vit_amd
    rng = np.random.default_rng(9)
    lens = rng.integers(0, 20000, size=20000).tolist() + [2 << 20] * 6
    blobs = [oracle.synth_bytes(65, i, L) for i, L in enumerate(lens)]
    want = git_ids(blobs)
    keep, ptrs, ln = vit_amd.hasher._as_ptr_arrays(blobs)
    with vit_amd.Hasher((0, 0, 0, 0)) as h4:
        budget = 16
        h4.set_option("pack_threads", budget)
        h4.set_option("chunk_bytes", 4 << 20)
        assert h4.stat("pool_threads") == budget - 4
        out = np.zeros((len(blobs), 20), dtype=np.uint8)
        h4._check(h4.lib.vit_hash_blobs(h4.handle, ptrs, ln.ctypes.data, len(blobs), out.ctypes.data))   # sizes staging
        base = _threads_now()                           # main, HIP runtime threads, the pool
        peak, stop = [base], threading.Event()

        def watch():
            while not stop.is_set():
                peak[0] = max(peak[0], _threads_now())
                time.sleep(0.0005)
        w = threading.Thread(target=watch)
        w.start()
        for _ in range(3):
            h4._check(h4.lib.vit_hash_blobs(h4.handle, ptrs, ln.ctypes.data, len(blobs), out.ctypes.data))
        stop.set()
        w.join()
        assert (out == want).all()
        assert h4.stat("objects_host") > 0
        # during a call only the four queue threads come on top (plus the watcher itself)
        assert peak[0] <= base + 4 + 1, (base, peak[0])
        assert h4.stat("pool_peak_busy") <= budget - 4
