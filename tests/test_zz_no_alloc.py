"""vit_hash_arena_device_static, the entry point that allocates nothing (its order array, binning
counters and status word live in a file-scope device array), against vit_hash_arena_device on the same
device-resident batches: bit-exact digests, on two calls in a row, the second with a smaller input."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_static_entry_point_matches_the_allocating_one_twice_in_a_row():
    import vit_amd
    rng = np.random.default_rng(21)
    with vit_amd.Hasher((0,)) as h:
        cases = []
        # first call: 300 000 skewed objects; second: 1 000 objects of other lengths (smaller input,
        # the scratch still holds the first call's order and counters); third: uniform lengths
        big = np.minimum(np.exp(rng.normal(np.log(1500.0), 1.6, 300_000)), 200_000).astype(np.uint64)
        cases.append((0x5EED0070, big))
        cases.append((0x5EED0071, rng.integers(0, 9000, size=1000).astype(np.uint64)))
        cases.append((0x5EED0072, np.full(70_001, 4096, dtype=np.uint64)))
        cases.append((0x5EED0073, np.array([0, 1, 55, 56, 119, 120], dtype=np.uint64)))
        for seed, lens in cases:
            n = len(lens)
            b = h.synth_arena("blob", seed, 0, n, 0, lens=lens)
            want = h.hash_device(b).copy()                                   # the existing entry point
            b["digests"].upload(np.zeros(20 * n, dtype=np.uint8))
            got = h.hash_device_static(b)
            bad = np.nonzero((got != want).any(axis=1))[0]
            assert len(bad) == 0, "n=%d: %d digests differ, first at %d (len %d)" % (n, len(bad), bad[0], lens[bad[0]])
            assert h.device_status(0) == 0
            for i in (0, n // 2, n - 1):
                data = vit_amd.synth_bytes(seed, i, int(lens[i]))
                assert got[i].tobytes() == hashlib.sha1(b"blob %d\0" % len(data) + data).digest()
        # bad input is flagged in the scratch's status word, read and cleared by vit_device_status
        lens = np.full(500, 1000, dtype=np.uint64)
        b = h.synth_arena("blob", 7, 0, 500, 0, lens=lens)
        good = h.hash_device_static(b).copy()
        off = b["offsets"].download(np.uint64, 8 * 500)
        bad_off = off.copy()
        bad_off[17] = np.uint64(b["arena_bytes"])
        b["offsets"].upload(bad_off)
        got = h.hash_device_static(b)
        assert h.device_status(0) == 1 and h.device_status(0) == 0
        keep = np.ones(500, dtype=bool)
        keep[17] = False
        assert (got[keep] == good[keep]).all()
        # beyond the stated maximum: refused, nothing launched
        b["offsets"].upload(off)
        with pytest.raises(vit_amd.VitError) as e:
            h._check(h.lib.vit_hash_arena_device_static(h.handle, 0, b["arena"].ptr, b["arena_bytes"], b["offsets"].ptr,
                                                        b["nblocks"].ptr, (1 << 21) + 1, b["digests"].ptr, None))
        assert e.value.code == -4 and "at most" in str(e.value)


def test_static_calls_from_two_hashers_on_one_gpu_take_turns():
    """Every call on a GPU works in that GPU's one scratch array.  Two hashers on the same GPU, driven by two
    threads at once (the object store itself keeps two hashers per device), must still get their own digests:
    the library orders the calls with an event per GPU and a lock around wait-launch-record."""
    import threading
    import vit_amd
    rng = np.random.default_rng(33)
    with vit_amd.Hasher((0,)) as h1, vit_amd.Hasher((0, 0)) as h2:
        work = []
        for k, (h, dev) in enumerate(((h1, 0), (h2, 1))):
            lens = np.minimum(np.exp(rng.normal(np.log(900.0 * (k + 1)), 1.5, 150_000 + 50_000 * k)), 100_000).astype(np.uint64)
            b = h.synth_arena("blob", 0x5EED0080 + k, 0, len(lens), 0, lens=lens, dev=dev)
            want = h.hash_device(b).copy()
            work.append((h, b, want, lens))
        errors = []

        def run(h, b, want, lens):
            try:
                for _ in range(6):
                    b["digests"].upload(np.zeros(20 * b["n"], dtype=np.uint8))
                    got = h.hash_device_static(b)
                    if not (got == want).all():
                        errors.append("digests differ: %d of %d" % (int((got != want).any(axis=1).sum()), b["n"]))
            except Exception as e:                                             # a worker must not die silently
                errors.append(repr(e))
        threads = [threading.Thread(target=run, args=w) for w in work]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        assert h1.device_status(0) == 0 and h2.device_status(1) == 0
