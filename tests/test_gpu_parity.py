"""Parity of the HIP path with the oracle and with the reference's golden ids.
Every call goes through the C ABI (libvit_hash.so).  Bar: bit-exact digests."""
import concurrent.futures
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KERNELS = [0, 1, 5]   # VIT_KERNEL_LDS (4 waves per workgroup), VIT_KERNEL_LANE, the LDS kernel with 1 wave per workgroup


def ids_of(oracle, type_, blobs):
    return np.array([np.frombuffer(oracle.object_id(type_, b), dtype=np.uint8) for b in blobs]).reshape(-1, 20)


def assert_same(got, want, labels):
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(bad) == 0, "digests differ at %s" % [labels[i] for i in bad[:12]]


def tail_cases():
    """A batch of short objects, which must not pay the software scheduler's price, and one with a long tail,
    which must get its benefit (also used by tests/test_zzz_gpu_perf.py)."""
    rng = np.random.default_rng(77)
    return {"short": rng.integers(0, 257, 600_000).astype(np.uint64),
            "tail": np.concatenate([rng.integers(0, 2000, 400_000), [3_000_000, 2_500_000]]).astype(np.uint64)}


def skewed_sizes_beyond_4gib(n=340_000):
    rng = np.random.default_rng(5)
    sizes = np.exp(rng.normal(np.log(4500.0), 1.55, n))
    return np.minimum(sizes * (15000.0 / sizes.mean()), 131072).astype(np.uint64)


@pytest.mark.parametrize("kernel", KERNELS)
def test_golden_blob_ids_from_the_reference(hasher, oracle, golden_objects, kernel):
    g = golden_objects["blobs"]
    hasher.set_option("kernel", kernel)
    lens = [L for L, _ in g["ids"]]
    blobs = [oracle.synth_bytes(g["seed"], L, L) for L in lens]
    got = hasher.hash_blobs(blobs)
    want = np.array([np.frombuffer(bytes.fromhex(x), dtype=np.uint8) for _, x in g["ids"]])
    assert_same(got, want, lens)


@pytest.mark.parametrize("kernel", KERNELS)
def test_golden_typed_objects(hasher, oracle, golden_objects, kernel):
    g = golden_objects["typed"]
    hasher.set_option("kernel", kernel)
    for t in ("blob", "tree", "commit"):
        rows = [(L, x) for tt, L, x in g["ids"] if tt == t]
        got = hasher.hash_objects(t, [oracle.synth_bytes(g["seed"], L, L) for L, _ in rows])
        want = np.array([np.frombuffer(bytes.fromhex(x), dtype=np.uint8) for _, x in rows])
        assert_same(got, want, [(t, L) for L, _ in rows])


@pytest.mark.parametrize("kernel", KERNELS)
def test_known_answers(hasher, golden_objects, kernel):
    k = golden_objects["kats"]
    hasher.set_option("kernel", kernel)
    got = hasher.hash_blobs([b"", k["git_hello_blob"][0].encode()])
    assert got[0].tobytes().hex() == k["git_empty_blob"]
    assert got[1].tobytes().hex() == k["git_hello_blob"][1]
    assert hasher.hash_objects("tree", [b""])[0].tobytes().hex() == k["git_empty_tree"]


def test_empty_batch_and_empty_objects(hasher, oracle):
    assert hasher.hash_blobs([]).shape == (0, 20)
    got = hasher.hash_blobs([b""] * 200)
    assert (got == np.frombuffer(oracle.blob_id(b""), dtype=np.uint8)).all()


@pytest.mark.parametrize("kernel", KERNELS)
def test_ragged_batch_every_wave_position(hasher, oracle, kernel):
    """Lengths chosen so that waves mix 1..70-block objects, batch sizes that are
    not multiples of 64, and binning reorders everything."""
    hasher.set_option("kernel", kernel)
    rng = np.random.default_rng(1234 + kernel)
    for n in (1, 63, 64, 65, 1000, 4099):
        lens = rng.integers(0, 4500, size=n).tolist()
        blobs = [oracle.synth_bytes(42, i, L) for i, L in enumerate(lens)]
        got = hasher.hash_blobs(blobs)
        assert_same(got, ids_of(oracle, "blob", blobs), lens)


@pytest.mark.parametrize("kernel", KERNELS)
def test_skewed_batch_with_large_objects(hasher, oracle, kernel):
    """A few multi-MiB objects among tiny ones, forced onto the GPU lanes
    (host_threshold 0), then with the host lane taking the large ones."""
    hasher.set_option("kernel", kernel)
    lens = [3 << 20, 5, 0, (2 << 20) + 17, 200, 64, 1 << 20] + [i % 300 for i in range(500)]
    blobs = [oracle.synth_bytes(43, i, L) for i, L in enumerate(lens)]
    want = ids_of(oracle, "blob", blobs)
    hasher.set_option("host_threshold_bytes", 0)
    g0, h0 = hasher.stat("objects_gpu"), hasher.stat("objects_host")
    assert_same(hasher.hash_blobs(blobs), want, lens)
    assert hasher.stat("objects_gpu") - g0 == len(lens) and hasher.stat("objects_host") == h0
    hasher.set_option("host_threshold_bytes", 1 << 20)
    g0, h0 = hasher.stat("objects_gpu"), hasher.stat("objects_host")
    assert_same(hasher.hash_blobs(blobs), want, lens)
    assert hasher.stat("objects_host") - h0 == 2 and hasher.stat("objects_gpu") - g0 == len(lens) - 2
    hasher.set_option("host_auto", 1)
    # auto policy: a lane would need tens of ms for each MiB-sized object, a host core under 2 ms, so
    # all three move to the host; the threshold reported is the shortest moved length less one
    g0, h0 = hasher.stat("objects_gpu"), hasher.stat("objects_host")
    assert_same(hasher.hash_blobs(blobs), want, lens)
    assert hasher.stat("last_host_threshold") == (1 << 20) - 1
    assert hasher.stat("objects_host") - h0 == 3


def test_slots_aligned_to_128_bytes_hash_the_same(hasher, oracle):
    """A resident arena may start its slots on 128-byte boundaries (the gap after an
    odd block count is never read as data): same ids from a packed host arena and
    from the device generator."""
    import vit_amd
    lens = [0, 1, 55, 56, 119, 120, 200, 4096, 4097, 9000] + [(i * 53) % 3000 for i in range(700)]
    blobs = [oracle.synth_bytes(48, i, L) for i, L in enumerate(lens)]
    want = ids_of(oracle, "blob", blobs)
    arena, off, nb, used = vit_amd.pack_arena("blob", blobs, slot_align=128)
    assert (off % 128 == 0).all()
    assert_same(hasher.hash_arena(arena, off, nb, arena_bytes=used), want, lens)
    n = 5000
    b64 = hasher.synth_arena("blob", 77, 3, n, 4096)
    b128 = hasher.synth_arena("blob", 77, 3, n, 4096, slot_align=128)
    assert b128["arena_bytes"] == n * 4224 and b64["arena_bytes"] == n * 4160
    d64, d128 = hasher.hash_device(b64), hasher.hash_device(b128)
    assert (d64 == d128).all()
    assert d128[17].tobytes() == oracle.object_id("blob", oracle.synth_bytes(77, 3 + 17, 4096))
    assert hasher.device_status(0) == 0


def test_pipeline_settings_do_not_change_ids(hasher, oracle, capfd):
    """Chunk growth on or off, host lane on auto or forced low, trace printing: same ids;
    the trace prints one line per staged chunk."""
    rng = np.random.default_rng(99)
    lens = rng.integers(0, 3000, size=2500).tolist() + [70_000, 300_000, 1 << 20, 65_536, 65_537]
    blobs = [oracle.synth_bytes(49, i, L) for i, L in enumerate(lens)]
    want = ids_of(oracle, "blob", blobs)
    try:
        for mult, thr, trace in ((0, None, 0), (8192, None, 1), (64, 0, 1), (8192, 65_536, 0)):
            hasher.set_option("chunk_bytes", 1 << 20)
            hasher.set_option("chunk_object_multiple", mult)
            if thr is None:
                hasher.set_option("host_auto", 1)
            else:
                hasher.set_option("host_threshold_bytes", thr)
            hasher.set_option("trace", trace)
            capfd.readouterr()
            l0 = hasher.stat("launches")
            assert_same(hasher.hash_blobs(blobs), want, lens)
            err = capfd.readouterr().err
            chunks = hasher.stat("launches") - l0
            assert chunks >= 2
            assert err.count("set ") == (chunks if trace else 0)
            assert ("[vit trace] queue 0" in err) == bool(trace)
    finally:
        hasher.set_option("trace", 0)
        hasher.set_option("chunk_bytes", 64 << 20)
        hasher.set_option("chunk_object_multiple", 8192)
        hasher.set_option("host_auto", 1)


def test_wave_timeline_diagnostic_records_every_wave(hasher, oracle):
    """With a trace buffer set, every wave of the launch writes when and where it ran; ids unchanged."""
    n, L = 3000, 700
    b = hasher.synth_arena("blob", 91, 0, n, L)
    want = hasher.hash_device(b).copy()
    waves = (n + 255) // 256 * 4
    tr = hasher.alloc(32 * waves)
    tr.upload(np.zeros(4 * waves, dtype=np.uint64))
    try:
        hasher.set_option("wave_trace_ptr", tr.ptr)
        got = hasher.hash_device(b)
    finally:
        hasher.set_option("wave_trace_ptr", 0)
    assert (got == want).all()
    assert got[5].tobytes() == oracle.object_id("blob", oracle.synth_bytes(91, 5, L))
    rec = tr.download(np.uint64, 32 * waves).reshape(-1, 4)
    live = (n + 63) // 64
    assert (rec[:live, 1] > rec[:live, 0]).all() and (rec[:live, 0] > 0).all()
    assert (rec[:live, 3] == hasher_blocks(L)).all()
    assert (rec[live:, 3] == 0).all()                      # a wave without objects records 0 blocks


def hasher_blocks(L):
    import vit_amd
    return vit_amd.object_blocks("blob", L)


def test_device_binning_gives_the_host_order(hasher):
    """vit_bin_order_device = vit_bin_order: longest first, ties in index order; and the ids under it."""
    import vit_amd
    rng = np.random.default_rng(77)
    for n in (1, 63, 1000, 70_001):
        lens = rng.integers(0, 9000, size=n).astype(np.uint64)
        lens[rng.integers(0, n, size=max(n // 50, 1))] = 200_000
        b = hasher.synth_arena("blob", 0x5EED0077, 0, n, 0, lens=lens)
        nb = b["nblocks"].download(np.uint32, 4 * n)
        want, _ = vit_amd.bin_order(nb)
        d_order = hasher.bin_order_device(b)
        got = d_order.download(np.uint32, 4 * n)
        assert (got == want).all(), n
        plain = hasher.hash_device(b).copy()
        assert (hasher.hash_device(b, order=d_order) == plain).all()
    assert hasher.device_status() == 0


@pytest.mark.parametrize("waves", [1, 2])
def test_software_scheduled_launch_gives_the_same_ids(hasher, waves):
    """The persistent-wave launch (groups pulled from both ends of the binned order) forced onto batches of
    every shape: fewer groups than resident waves, ragged last group, a long tail, launches back to back
    (each must leave the scheduler state clean for the next), and the host pipeline's binned chunks."""
    import vit_amd
    rng = np.random.default_rng(500 + waves)
    hasher.set_option("sched_waves", waves)
    hasher.set_option("sched_min_groups", 1)
    hasher.set_option("sched_tail_percent", 0)
    for n in (1, 63, 65, 1000, 70_001, 400_000):
        lens = np.minimum(np.exp(rng.normal(np.log(1500.0), 1.5, n)), 300_000).astype(np.uint64)
        b = hasher.synth_arena("blob", 0x5EED0088, 0, n, 0, lens=lens)
        nb = b["nblocks"].download(np.uint32, 4 * n)
        order_host, _ = vit_amd.bin_order(nb)
        order = hasher.alloc(4 * n).upload(order_host)
        hasher.set_option("sched_waves", 0)
        want = hasher.hash_device(b, order=order).copy()
        hasher.set_option("sched_waves", waves)
        for _ in range(3):
            b["digests"].upload(np.zeros(20 * n, dtype=np.uint8))
            got = hasher.hash_device(b, order=order)
            assert (got == want).all(), (n, int(np.nonzero((got != want).any(axis=1))[0][0]))
        hasher.time_hash_device(b, order=order, warmup=0, iters=8)           # eight launches queued back to back
        assert (b["digests"].download(np.uint8, 20 * n).reshape(-1, 20) == want).all()
        for i in (0, n // 2, n - 1, int(np.argmax(lens))):
            data = vit_amd.synth_bytes(0x5EED0088, i, int(lens[i]))
            assert want[i].tobytes() == hashlib.sha1(b"blob %d\0" % len(data) + data).digest()
        assert hasher.device_status() == 0
    # through the host pipeline: several binned chunks in flight on different streams, each with a state of its own
    lens = rng.integers(0, 20_000, size=30_000).tolist() + [300_000, 1 << 20]
    buf = np.frombuffer(vit_amd.synth_bytes(0x5EED0089, 0, (1 << 20) + len(lens)), dtype=np.uint8)
    blobs = [buf[i:i + L] for i, L in enumerate(lens)]
    hasher.set_option("chunk_bytes", 4 << 20)
    hasher.set_option("chunk_object_multiple", 0)
    hasher.set_option("host_threshold_bytes", 0)
    got = hasher.hash_blobs(blobs)
    hasher.set_option("sched_waves", 0)
    assert (hasher.hash_blobs(blobs) == got).all()
    for i in (0, 7, len(lens) - 1):
        assert got[i].tobytes() == hashlib.sha1(b"blob %d\0" % lens[i] + blobs[i].tobytes()).digest()


def test_scheduled_launch_is_chosen_by_the_tail(hasher):
    """Default options, lengths known to the device only: the scheduled launch and the plain one are both
    queued and their waves decide (longest object against a SIMD's mean share of the work).  Same ids either
    way, and from the entry point that allocates nothing too.  (What the choice costs in time is checked in
    tests/test_zzz_gpu_perf.py, which runs last.)"""
    import vit_amd
    for name, lens in tail_cases().items():
        n = len(lens)
        b = hasher.synth_arena("blob", 0x5EED0090, 0, n, 0, lens=lens)
        nb = b["nblocks"].download(np.uint32, 4 * n)
        order_host, _ = vit_amd.bin_order(nb)
        order = hasher.alloc(4 * n).upload(order_host)
        hasher.set_option("sched_waves", 0)
        want = hasher.hash_device(b, order=order).copy()
        hasher.set_option("sched_waves", 1)
        hasher.set_option("sched_tail_percent", 0)
        b["digests"].upload(np.zeros(20 * n, dtype=np.uint8))
        assert (hasher.hash_device(b, order=order) == want).all(), name
        hasher.set_option("sched_tail_percent", 75)
        b["digests"].upload(np.zeros(20 * n, dtype=np.uint8))
        assert (hasher.hash_device(b, order=order) == want).all(), name
        b["digests"].upload(np.zeros(20 * n, dtype=np.uint8))
        assert (hasher.hash_device_static(b) == want).all(), name
        for i in (0, n - 1, int(np.argmax(lens))):
            data = vit_amd.synth_bytes(0x5EED0090, i, int(lens[i]))
            assert want[i].tobytes() == hashlib.sha1(b"blob %d\0" % len(data) + data).digest()
        assert hasher.device_status() == 0


def test_plain_sha1_of_buffers(hasher, oracle, golden_objects):
    """vit_sha1_buffers: SHA1(data, len) with no Git framing, the reference's call (src/commit.cpp:93-94)
    n at a time: FIPS 180 known answers, ragged buffers against hashlib, framed strings give object ids,
    and a long buffer takes the host lane."""
    for text, want in golden_objects["kats"]["sha1"]:
        assert hasher.sha1_buffers([text.encode()])[0].tobytes().hex() == want
    rng = np.random.default_rng(3)
    lens = [0, 1, 55, 56, 63, 64, 65, 119, 120, 4096, 70_000, 3 << 20] + rng.integers(0, 5000, size=600).tolist()
    bufs = [oracle.synth_bytes(50, i, L) for i, L in enumerate(lens)]
    h0 = hasher.stat("objects_host")
    got = hasher.sha1_buffers(bufs)
    assert hasher.stat("objects_host") - h0 >= 1                       # the 3 MiB buffer
    for i, b in enumerate(bufs):
        assert got[i].tobytes() == hashlib.sha1(b).digest(), lens[i]
    framed = [b"blob %d\0" % len(b) + b for b in bufs[:50]]
    assert (hasher.sha1_buffers(framed) == hasher.hash_blobs(bufs[:50])).all()
    hasher.set_option("host_threshold_bytes", 0)                       # every buffer on a GPU lane
    assert (hasher.sha1_buffers(bufs) == got).all()


def test_chunked_pipeline_matches_single_chunk(hasher, oracle):
    """Force many small staging chunks so both pinned buffers cycle several times."""
    lens = [(i * 37) % 5000 for i in range(3000)]
    blobs = [oracle.synth_bytes(44, i, L) for i, L in enumerate(lens)]
    want = ids_of(oracle, "blob", blobs)
    hasher.set_option("chunk_bytes", 1 << 20)
    l0 = hasher.stat("launches")
    got = hasher.hash_blobs(blobs)
    assert hasher.stat("launches") - l0 >= 5
    hasher.set_option("chunk_bytes", 256 << 20)
    assert_same(got, want, lens)


def test_prepacked_arena_and_input_checks(hasher, oracle):
    import vit_amd
    lens = [0, 10, 55, 56, 4096, 100000]
    blobs = [oracle.synth_bytes(45, i, L) for i, L in enumerate(lens)]
    arena, off, nb, used = vit_amd.pack_arena("blob", blobs)
    assert_same(hasher.hash_arena(arena, off, nb), ids_of(oracle, "blob", blobs), lens)
    # out-of-range and misaligned input is rejected on the host, nothing is launched
    bad_off = off.copy(); bad_off[3] = used
    with pytest.raises(vit_amd.VitError) as e:
        hasher.hash_arena(arena, bad_off, nb)
    assert e.value.code == -4
    bad_off = off.copy(); bad_off[2] += 4
    with pytest.raises(vit_amd.VitError):
        hasher.hash_arena(arena, bad_off, nb)
    bad_nb = nb.copy(); bad_nb[-1] += 1000
    with pytest.raises(vit_amd.VitError):
        hasher.hash_arena(arena, off, bad_nb)


@pytest.mark.parametrize("kernel", KERNELS)
def test_device_kernel_flags_bad_input_instead_of_faulting(hasher, oracle, kernel):
    """Device-resident batches cannot be checked on the host; the kernel must skip
    objects that would read outside the arena and raise a status bit."""
    hasher.set_option("kernel", kernel)
    b = hasher.synth_arena("blob", 7, 0, 300, 1000)
    good = hasher.hash_device(b)
    assert hasher.device_status() == 0
    off = b["offsets"].download(np.uint64, 8 * 300)
    bad = off.copy()
    bad[17] = np.uint64(b["arena_bytes"])            # starts at the end: out of bounds
    bad[40] += np.uint64(8)                          # misaligned
    b["offsets"].upload(bad)
    got = hasher.hash_device(b)
    assert hasher.device_status() == (1 | 2)
    assert hasher.device_status() == 0               # reading clears it
    keep = np.ones(300, dtype=bool); keep[[17, 40]] = False
    assert (got[keep] == good[keep]).all()
    b["offsets"].upload(off)
    order = hasher.alloc(4 * 300)
    o = np.arange(300, dtype=np.uint32)[::-1].copy(); o[5] = 1000
    order.upload(o)
    hasher.hash_device(b, order=order)
    assert hasher.device_status() == 4


@pytest.mark.parametrize("kernel", KERNELS)
def test_generator_and_device_path_on_ragged_lengths(hasher, oracle, kernel):
    """Device generator == host packer byte for byte, and the device-resident
    launch (with and without binning order) gives the oracle's ids."""
    import vit_amd
    hasher.set_option("kernel", kernel)
    rng = np.random.default_rng(5)
    lens = np.concatenate([rng.integers(0, 300, 700), rng.integers(3000, 70000, 30), [0, 9, 10, 99, 100]]).astype(np.uint64)
    n = len(lens)
    b = hasher.synth_arena("blob", 0x5EED0003, 100, n, 0, lens=lens)
    blobs = [oracle.synth_bytes(0x5EED0003, 100 + i, int(L)) for i, L in enumerate(lens)]
    arena, off, nb, used = vit_amd.pack_arena("blob", blobs)
    assert used == b["arena_bytes"]
    assert (b["arena"].download(np.uint8, used) == arena[:used]).all()
    assert (b["nblocks"].download(np.uint32, 4 * n) == nb).all()
    want = ids_of(oracle, "blob", blobs)
    assert_same(hasher.hash_device(b), want, lens.tolist())
    order_host, differs = vit_amd.bin_order(nb)
    assert differs
    order = hasher.alloc(4 * n).upload(order_host)
    b["digests"].upload(np.zeros(20 * n, dtype=np.uint8))
    assert_same(hasher.hash_device(b, order=order), want, lens.tolist())
    assert hasher.device_status() == 0


def test_two_queues_on_one_gpu_shard_and_agree(oracle):
    """The multi-device path (one worker per queue, byte-balanced shards) exercised
    on a single GPU by naming it twice."""
    import vit_amd
    lens = [(i * 53) % 9000 for i in range(5000)] + [2 << 20, 3 << 20]
    blobs = [oracle.synth_bytes(46, i, L) for i, L in enumerate(lens)]
    want = ids_of(oracle, "blob", blobs)
    with vit_amd.Hasher((0, 0)) as h2:
        assert h2.device_count == 2
        h2.set_option("host_threshold_bytes", 0)
        assert_same(h2.hash_blobs(blobs), want, lens)
        assert h2.stat("launches") >= 2


def test_waves_spanning_more_than_4gib_take_the_per_lane_path(hasher):
    """The LDS-staged kernel addresses a wave's 64 objects as base + 32-bit offset.
    In an arena beyond 4 GiB an order that puts the first and the last objects in
    one wave breaks that; such waves must fall back to 64-bit per-lane loads and
    still give the same ids."""
    import vit_amd
    n, L = 1_100_000, 4096                         # 4.58 GB arena
    hasher.set_option("kernel", 0)
    b = hasher.synth_arena("blob", 0x5EED0012, 0, n, L)
    assert b["arena_bytes"] > (1 << 32) + (1 << 28)
    near = hasher.hash_device(b).copy()            # identity order: every wave is compact
    order_host = np.empty(n, dtype=np.uint32)
    order_host[0::2] = np.arange(0, (n + 1) // 2, dtype=np.uint32)          # 0, n-1, 1, n-2, ...
    order_host[1::2] = np.arange(n - 1, (n + 1) // 2 - 1, -1, dtype=np.uint32)
    assert len(np.unique(order_host)) == n        # a permutation; only the middle waves stay within 4 GiB
    order = hasher.alloc(4 * n).upload(order_host)
    b["digests"].upload(np.zeros(20 * n, dtype=np.uint8))
    far = hasher.hash_device(b, order=order)
    assert hasher.device_status() == 0
    assert (far == near).all()
    for i in (0, 1, n // 2, n - 2, n - 1):
        want = hashlib.sha1(b"blob 4096\0" + vit_amd.synth_bytes(0x5EED0012, i, L)).digest()
        assert far[i].tobytes() == want


def test_binned_skewed_arena_beyond_4gib_gives_the_right_ids(hasher):
    """A skewed batch in an arena beyond 4 GiB, hashed in binned order: every wave's objects lie all over the
    arena, so most waves address their objects with 64 bits per lane.  (Its speed is checked in
    tests/test_zzz_gpu_perf.py, which runs last.)"""
    import vit_amd
    sizes = skewed_sizes_beyond_4gib()
    n = len(sizes)
    b = hasher.synth_arena("blob", 0x5EED0044, 0, n, 0, lens=sizes)
    assert b["arena_bytes"] > (1 << 32)
    nb = b["nblocks"].download(np.uint32, 4 * n)
    order_host, differs = vit_amd.bin_order(nb)
    assert differs
    order = hasher.alloc(4 * n).upload(order_host)
    got = hasher.hash_device(b, order=order)
    assert hasher.device_status() == 0
    for i in [0, 7, n - 1, int(np.argmax(sizes))] + np.random.default_rng(8).integers(0, n, 60).tolist():
        want = hashlib.sha1(b"blob %d\0" % int(sizes[i]) + vit_amd.synth_bytes(0x5EED0044, i, int(sizes[i]))).digest()
        assert got[i].tobytes() == want


def test_config2_full_size_1m_x_4k(hasher):
    """BASELINE configs[1] at full size: 1,000,000 x 4 KiB blobs generated on the
    device; every digest checked against hashlib (OpenSSL) on the host copy of the
    same bytes, on 16 threads; both kernels must agree bit for bit."""
    import vit_amd
    n, L = 1_000_000, 4096
    hasher.set_option("kernel", 0)
    b = hasher.synth_arena("blob", 0x5EED0002, 0, n, L)
    d_lds = hasher.hash_device(b).copy()
    hasher.set_option("kernel", 1)
    d_lane = hasher.hash_device(b).copy()
    hasher.set_option("kernel", 0)
    assert hasher.device_status() == 0
    assert (d_lds == d_lane).all()
    slot = vit_amd.object_blocks("blob", L) * 64
    hl = vit_amd.header_len("blob", L)
    # generator spot check against the host restatement of the byte stream
    # (the arena stores each big-endian message word as a native dword: byteswap gives the message bytes)
    for i in (0, 1, 999_999):
        raw = b["arena"].download(np.uint8, slot, offset=i * slot).view("<u4").byteswap().tobytes()
        assert raw[:hl] == b"blob 4096\0" and raw[hl:hl + L] == vit_amd.synth_bytes(0x5EED0002, i, L)

    words = b["arena"].download(np.uint8, n * slot).view("<u4")  # one 4.16 GB copy, main thread
    words.byteswap(inplace=True)                                 # back to message byte order
    mv = memoryview(words.view(np.uint8))

    def check(lo, hi):
        for i in range(lo, hi):
            base = i * slot
            if hashlib.sha1(mv[base:base + hl + L]).digest() != d_lds[i].tobytes():
                return i
        return -1
    step = 12500
    with concurrent.futures.ThreadPoolExecutor(16) as ex:
        bad = [r for r in ex.map(lambda lo: check(lo, min(lo + step, n)), range(0, n, step)) if r >= 0]
    assert not bad, "first differing blob: %d" % bad[0]
    # order-independent summary of all ids, for the record
    acc = np.bitwise_xor.reduce(d_lds, axis=0)
    assert acc.shape == (20,)
