"""Property tests (hypothesis).  CPU: the packed arena of ANY batch hashes, block by
block, to the oracle's ids; binning and sharding keep their invariants.  GPU:
random batches through the C ABI equal the oracle."""
import hashlib
import struct

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

lengths = st.lists(st.one_of(st.integers(0, 300), st.integers(3000, 9000), st.sampled_from([54, 55, 56, 63, 64, 119, 120])),
                   min_size=0, max_size=40)
types = st.sampled_from(["blob", "tree", "commit"])


def sha1_of_padded_blocks(slot):
    """SHA-1 state update over pre-padded blocks whose message words are stored as
    native little-endian dwords, without any further padding: what the device
    kernel does with an arena slot."""
    h = [0x67452301, 0xEFCDAB89, 0x98BADCFE, 0x10325476, 0xC3D2E1F0]
    rol = lambda x, n: ((x << n) | (x >> (32 - n))) & 0xFFFFFFFF
    for off in range(0, len(slot), 64):
        w = list(struct.unpack("<16I", slot[off:off + 64]))
        for t in range(16, 80):
            w.append(rol(w[t - 3] ^ w[t - 8] ^ w[t - 14] ^ w[t - 16], 1))
        a, b, c, d, e = h
        for t in range(80):
            if t < 20:
                f, k = (b & c) | (~b & d), 0x5A827999
            elif t < 40:
                f, k = b ^ c ^ d, 0x6ED9EBA1
            elif t < 60:
                f, k = (b & c) | (b & d) | (c & d), 0x8F1BBCDC
            else:
                f, k = b ^ c ^ d, 0xCA62C1D6
            a, b, c, d, e = (rol(a, 5) + f + e + k + w[t]) & 0xFFFFFFFF, a, rol(b, 30), c, d
        h = [(x + y) & 0xFFFFFFFF for x, y in zip(h, (a, b, c, d, e))]
    return struct.pack(">5I", *h)


@settings(max_examples=40, deadline=None)
@given(lengths, types, st.integers(0, 2 ** 32))
def test_packed_slots_hash_to_object_ids(oracle, lens, type_, seed):
    import vit_amd
    blobs = [oracle.synth_bytes(seed, i, L) for i, L in enumerate(lens)]
    arena, off, nb, used = vit_amd.pack_arena(type_, blobs)
    for i, b in enumerate(blobs):
        slot = arena[int(off[i]):int(off[i]) + 64 * int(nb[i])].tobytes()
        assert sha1_of_padded_blocks(slot) == oracle.object_id(type_, b) == \
            hashlib.sha1(("%s %d" % (type_, len(b))).encode() + b"\0" + b).digest()


@settings(max_examples=60, deadline=None)
@given(st.lists(st.integers(1, 70000), min_size=0, max_size=500))
def test_bin_order_is_a_stable_descending_permutation(nblocks):
    import vit_amd
    nb = np.array(nblocks, dtype=np.uint32)
    order, differs = vit_amd.bin_order(nb)
    assert sorted(order.tolist()) == list(range(len(nb)))
    want = sorted(range(len(nb)), key=lambda i: (-int(nb[i]), i))
    assert order.tolist() == want
    assert differs == (len(set(nblocks)) > 1)


@settings(max_examples=60, deadline=None)
@given(st.lists(st.integers(1, 10 ** 6), min_size=1, max_size=400), st.integers(1, 8))
def test_shard_plan_covers_everything_and_bounds_the_imbalance(nblocks, q):
    from vit_amd import _abi
    nb = np.array(nblocks, dtype=np.uint32)
    plan = np.full(len(nb), 99, dtype=np.uint32)
    assert _abi.load().vit_shard_plan(nb.ctypes.data, len(nb), q, plan.ctypes.data) == 0
    assert plan.max() < q
    load = np.bincount(plan, weights=nb.astype(np.float64), minlength=q)
    # greedy bound: no queue exceeds a fair share by more than the largest object (+ rounding)
    assert load.max() <= load.sum() / q + nb.max() + q


@pytest.mark.gpu
@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(lengths, types, st.integers(0, 2 ** 32), st.sampled_from([0, 1]))
def test_random_batches_on_the_gpu_equal_the_oracle(hasher, oracle, lens, type_, seed, kernel):
    hasher.set_option("kernel", kernel)
    blobs = [oracle.synth_bytes(seed, i, L) for i, L in enumerate(lens)]
    got = hasher.hash_objects(type_, blobs)
    for i, b in enumerate(blobs):
        assert got[i].tobytes() == oracle.object_id(type_, b), (type_, len(b))
    hasher.set_option("kernel", 0)
