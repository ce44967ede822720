"""CPU-side checks of the C ABI: the library loads, exports every symbol the
header declares, and its host-only parts (framing, packing, binning, sharding,
the host-lane SHA-1) agree with the oracle.  No GPU compute here."""
import ctypes as C
import hashlib
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from vit_amd import _abi
    return _abi.load()


HEADERS = ("vit_hash.h", "vit_hash_bench.h")     # the boundary; measurement, test and diagnostic entry points


def declared_functions(headers=HEADERS):
    names = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(vit_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: the header alone must compile as C99, warnings as errors."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc here")
    for header in HEADERS:
        src = tmp_path / "use.c"
        src.write_text('#include "%s"\nint main(void) { vit_hasher *h = 0; (void)h; return VIT_OK; }\n' % header)
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                            "-fsyntax-only", str(src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_library_exports_every_declared_symbol(lib):
    from vit_amd import _abi
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libvit_hash.so does not export %s" % n
        assert n in _abi.SIGNATURES, "no ctypes prototype for %s" % n
    assert sorted(_abi.SIGNATURES) == names
    # what a caller of the reference's object-store functions binds is the first header alone, and it holds no
    # generator, timer or test hook
    product = declared_functions(HEADERS[:1])
    assert 25 <= len(product) <= 32
    assert not [n for n in product if re.search(r"synth|time_|valu|launch_shape|split_plan|numa|device_(alloc|free|upload|download|sync)", n)]
    assert "debug_" not in open(os.path.join(ROOT, "include", HEADERS[0])).read().replace("debug_*", "")


def test_header_and_block_count(lib, oracle):
    for t in ("blob", "tree", "commit"):
        for L in (0, 1, 9, 10, 99, 100, 4095, 4096, 99999, 100000, (1 << 32) + 5, 10 ** 18):
            hdr = oracle.header(t, L)
            assert lib.vit_header_len(t.encode(), L) == len(hdr)
            buf = C.create_string_buffer(32)
            assert lib.vit_write_header(t.encode(), L, buf) == len(hdr)
            assert buf.raw[:len(hdr)] == hdr
            assert lib.vit_object_blocks(t.encode(), L) == (len(hdr) + L + 9 + 63) // 64
    assert lib.vit_object_blocks(b"blob", 4096) == 65          # the BASELINE config: 65 blocks per 4 KiB blob
    assert lib.vit_header_len(b"", 1) == 0 and lib.vit_header_len(b"has space", 1) == 0


def test_pack_layout_is_what_sha1_compresses(oracle):
    """A packed slot = framed message + FIPS 180-4 padding, each big-endian message word
    stored as a native little-endian dword (the kernel swaps no bytes)."""
    import vit_amd
    lens = list(range(0, 450)) + [4095, 4096, 4097, 65530]
    blobs = [oracle.synth_bytes(21, i, L) for i, L in enumerate(lens)]
    room = np.zeros(sum(len(b) + 128 for b in blobs) + 128, dtype=np.uint8)
    skip = (-room.ctypes.data) % 64
    # an aligned arena takes the packer's streaming-store path, one off by 16 bytes its plain path
    for t, given in (("blob", None), ("tree", None), ("commit", None), ("blob", room[skip + 16:])):
        arena, off, nb, used = vit_amd.pack_arena(t, blobs, arena=given, threads=3)
        assert (arena.ctypes.data % 64 == 0) == (given is None)
        assert used == int(nb.astype(np.uint64).sum()) * 64
        assert (off % 64 == 0).all()
        assert (off[1:] == off[:-1] + nb[:-1].astype(np.uint64) * 64).all()
        for i, b in enumerate(blobs):
            slot = arena[int(off[i]):int(off[i]) + int(nb[i]) * 64].view("<u4").byteswap().tobytes()
            framed = oracle.header(t, len(b)) + b
            assert slot[:len(framed)] == framed
            assert slot[len(framed)] == 0x80
            assert not any(slot[len(framed) + 1:-8])
            assert int.from_bytes(slot[-8:], "big") == 8 * len(framed)
            assert int(nb[i]) == (len(framed) + 9 + 63) // 64


def test_pack_aligned_starts_slots_on_the_boundary(oracle):
    """slot_align = 128: same slots, each starting on a 128-byte boundary, gaps untouched."""
    import vit_amd
    lens = [0, 1, 55, 56, 119, 120, 200, 4096, 4097, 64 * 7 - 15]
    blobs = [oracle.synth_bytes(22, i, L) for i, L in enumerate(lens)]
    a64, off64, nb64, used64 = vit_amd.pack_arena("blob", blobs)
    room = np.full(used64 * 2 + 256, 0xEE, dtype=np.uint8)
    skip = (-room.ctypes.data) % 128
    a, off, nb, used = vit_amd.pack_arena("blob", blobs, arena=room[skip:], slot_align=128)
    assert (nb == nb64).all() and (off % 128 == 0).all()
    assert used == int(off[-1]) + int(nb[-1]) * 64
    for i in range(len(blobs)):
        want = a64[int(off64[i]):int(off64[i]) + int(nb64[i]) * 64]
        assert (a[int(off[i]):int(off[i]) + int(nb[i]) * 64] == want).all()
        end = int(off[i]) + int(nb[i]) * 64
        nxt = int(off[i + 1]) if i + 1 < len(blobs) else end
        assert nxt - end in (0, 64) and (a[end:nxt] == 0xEE).all()
    with pytest.raises(vit_amd.VitError):
        vit_amd.pack_arena("blob", blobs, slot_align=96)


def test_pack_rejects_small_arena_and_bad_args(lib):
    import vit_amd
    blobs = [b"x" * 100]
    with pytest.raises(vit_amd.VitError) as e:
        vit_amd.pack_arena("blob", blobs, arena=np.zeros(64, dtype=np.uint8))
    assert e.value.code == -4
    with pytest.raises(vit_amd.VitError):
        vit_amd.pack_arena("two words", blobs)
    arena, off, nb, used = vit_amd.pack_arena("blob", [])
    assert used == 0 and len(off) == 0


def test_bin_order_longest_first_and_stable():
    import vit_amd
    rng = np.random.default_rng(1)
    nb = rng.integers(1, 5000, size=10000).astype(np.uint32)
    nb[::97] = 1 << 22                                     # a few very long ones (third radix digit)
    order, differs = vit_amd.bin_order(nb)
    assert differs
    assert sorted(order.tolist()) == list(range(len(nb)))
    s = nb[order]
    assert (s[:-1] >= s[1:]).all()
    same = s[:-1] == s[1:]
    assert (order[:-1][same] < order[1:][same]).all()      # ties keep index order
    order, differs = vit_amd.bin_order(np.full(1000, 65, dtype=np.uint32))
    assert not differs and (order == np.arange(1000)).all()


def test_shard_plan_balances_bytes_not_counts(lib):
    rng = np.random.default_rng(2)
    # adversarial skew: 4 huge objects + a million tiny ones
    nb = np.concatenate([np.full(4, 33554433, dtype=np.uint32), rng.integers(1, 6, size=200000).astype(np.uint32)])
    rng.shuffle(nb)
    for q in (1, 2, 4, 8):
        plan = np.empty(len(nb), dtype=np.uint32)
        assert lib.vit_shard_plan(nb.ctypes.data, len(nb), q, plan.ctypes.data) == 0
        assert plan.max() < q
        load = np.bincount(plan, weights=nb.astype(np.float64), minlength=q)
        if q <= 4:                                        # 4 giants fit 1, 2 or 4 queues evenly
            assert load.max() <= 1.02 * load.sum() / q
        huge = plan[nb == 33554433]
        assert len(set(huge.tolist())) == min(q, 4)        # the giants land on different queues
    # uniform sizes: contiguous, equal shares
    nb = np.full(100000, 65, dtype=np.uint32)
    plan = np.empty(len(nb), dtype=np.uint32)
    assert lib.vit_shard_plan(nb.ctypes.data, len(nb), 8, plan.ctypes.data) == 0
    counts = np.bincount(plan, minlength=8)
    assert counts.max() - counts.min() <= 16
    assert (np.diff(plan.astype(np.int64)) >= 0).all()


RATES = dict(lane=47e6, core=2.5e9, copy=56e9, pack=9e9)      # what one MI355X box measured


def split(lib, lens, queues=1, threads=16, **over):
    r = dict(RATES, **over)
    lens = np.ascontiguousarray(lens, dtype=np.uint64)
    out = np.zeros(max(len(lens), 1), dtype=np.uint8)
    rc = lib.vit_host_split_plan(lens.ctypes.data, len(lens), int(r["lane"]), int(r["core"]), int(r["copy"]),
                                 int(r.get("copy_total", 0)), int(r["pack"]), queues, threads, out.ctypes.data)
    assert rc == 0
    return out[:len(lens)].astype(bool)


def test_host_split_plan_balances_the_two_sides(lib):
    """The policy behind host_auto, as a pure function of lengths and rates (no device needed)."""
    rng = np.random.default_rng(11)
    # small objects never move, whatever the rates
    assert not split(lib, rng.integers(0, 16384, 5000)).any()
    assert not split(lib, []).any()
    # config 5: four 2 GiB giants among a million tiny objects -> exactly the giants
    lens = np.concatenate([np.full(4, 2 << 30), rng.integers(0, 257, 100000)]).astype(np.uint64)
    on = split(lib, lens)
    assert on[:4].all() and not on[4:].any()
    # one object beyond 4 GiB among small ones: a lane would need 90 s
    on = split(lib, [6, (1 << 32) + 5, 0])
    assert on.tolist() == [False, True, False]
    # a uniform batch of large objects is SHARED, not dumped on the host: equal lengths are split by index
    lens = np.full(3000, 256 << 10, dtype=np.uint64)
    on = split(lib, lens)
    k = int(on.sum())
    assert 0 < k < 3000 and on[:k].all() and not on[k:].any()
    # the host takes more when its cores are faster, the GPU keeps more when its lanes are
    assert split(lib, lens, core=1.25e9).sum() < k < split(lib, lens, core=5e9).sum()
    assert split(lib, lens, lane=94e6).sum() <= k
    # more queues drain the GPU side faster: fewer objects move; more threads: more do
    assert split(lib, lens, queues=4, threads=16).sum() < k
    # ... unless the queues share what one of them gets alone (one PCIe link, the host's memory): then as one queue
    assert split(lib, lens, queues=4, threads=16, copy_total=RATES["copy"]).sum() == k
    assert split(lib, lens, queues=4, threads=16, copy_total=2 * RATES["copy"]).sum() == split(lib, lens, queues=2, threads=16).sum()
    assert split(lib, lens, queues=1, copy_total=10 * RATES["copy"]).sum() == k       # never more than the queues' own rates added up
    assert split(lib, lens, threads=4).sum() < k < split(lib, lens, threads=64).sum()
    # skewed (config 4): what moves is a prefix of the (length, index) order: nothing shorter than a moved object stays
    sizes = np.exp(rng.normal(np.log(4500.0), 1.55, 200000))
    sizes = np.minimum(sizes * (15000.0 / sizes.mean()), 32 << 20).astype(np.uint64)
    on = split(lib, sizes)
    assert 0 < on.sum() < 0.1 * len(sizes)
    assert sizes[on].min() >= 16384 and sizes[on].min() >= sizes[~on].max()
    moved_bytes, all_bytes = float(sizes[on].sum()), float(sizes.sum())
    assert 0.2 < moved_bytes / all_bytes < 0.7                       # both sides carry a real share of the bytes
    # the model behind it: neither side is left far behind the other
    gpu_t = max((all_bytes - moved_bytes) / min(0.8 * RATES["copy"], RATES["pack"] * 16), float(sizes[~on].max()) / RATES["lane"])
    host_t = (moved_bytes / RATES["core"] + (all_bytes - moved_bytes) / RATES["pack"]) / 16
    assert 0.7 < host_t / gpu_t < 1.3, (host_t, gpu_t)
    # bad arguments
    one = np.zeros(1, dtype=np.uint8)
    assert lib.vit_host_split_plan(lens.ctypes.data, 1, 0, 1, 1, 0, 1, 1, 1, one.ctypes.data) == -1
    assert lib.vit_host_split_plan(lens.ctypes.data, 1, 1, 1, 1, 0, 1, 0, 1, one.ctypes.data) == -1
    assert lib.vit_host_split_plan(None, 1, 1, 1, 1, 0, 1, 1, 1, one.ctypes.data) == -1


def fake_sysfs(root, gpus, nodes):
    """gpus: {pci id: node}; nodes: {node: cpulist text}."""
    for pci, node in gpus.items():
        d = root / "bus" / "pci" / "devices" / pci
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % node)
    for node, cpus in nodes.items():
        d = root / "devices" / "system" / "node" / ("node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")


def numa_plan(lib, root, pci_ids, workers):
    ids = (C.c_char_p * len(pci_ids))(*[p.encode() if p is not None else None for p in pci_ids])
    qn = (C.c_int * len(pci_ids))()
    wn = (C.c_int * max(workers, 1))()
    assert lib.vit_numa_plan(str(root).encode(), ids, len(pci_ids), workers, qn, wn) == 0
    return list(qn), list(wn)[:workers]


def test_numa_plan_on_a_made_up_sysfs_tree(lib, tmp_path):
    """Which node a GPU hangs off, which CPUs a node has, and where the pool's workers are pinned: read from sysfs
    files, so a made-up tree stands in for a two-socket, eight-GPU machine."""
    gpus = {"0000:05:00.0": 0, "0000:15:00.0": 0, "0000:65:00.0": 0, "0000:75:00.0": 0,
            "0000:85:00.0": 1, "0000:95:00.0": 1, "0000:e5:00.0": 1, "0000:f5:00.0": 1, "0000:aa:00.0": -1}
    fake_sysfs(tmp_path, gpus, {0: "0-63,128-191", 1: "64-127,192-255", 2: "300"})
    cpus = (C.c_int * 256)()
    assert lib.vit_numa_node_cpus(str(tmp_path).encode(), 0, cpus, 256) == 128
    assert list(cpus)[:128] == list(range(64)) + list(range(128, 192))
    assert lib.vit_numa_node_cpus(str(tmp_path).encode(), 1, cpus, 4) == 128 and list(cpus)[:4] == [64, 65, 66, 67]
    assert lib.vit_numa_node_cpus(str(tmp_path).encode(), 7, cpus, 4) == 0            # no such node
    eight = [p for p in gpus if gpus[p] >= 0]
    # eight GPUs, four per socket: workers split evenly and handed out in turn
    qn, wn = numa_plan(lib, tmp_path, eight, 120)
    assert qn == [0, 0, 0, 0, 1, 1, 1, 1] and wn.count(0) == 60 and wn.count(1) == 60 and wn[:4] == [0, 1, 0, 1]
    # the hex digits of a bus id may come in upper case
    assert numa_plan(lib, tmp_path, ["0000:E5:00.0"], 3) == ([1], [1, 1, 1])
    # three queues on one socket, one on the other: workers in that proportion
    qn, wn = numa_plan(lib, tmp_path, eight[:3] + eight[4:5], 12)
    assert qn == [0, 0, 0, 1] and wn.count(0) == 9 and wn.count(1) == 3
    # every node with a queue gets a worker while workers last
    qn, wn = numa_plan(lib, tmp_path, eight[:7] + eight[7:], 2)
    assert sorted(wn) == [0, 1]
    # one GPU: its node's workers, as long as the node has the CPUs
    assert numa_plan(lib, tmp_path, eight[:1], 15) == ([0], [0] * 15)
    # more workers than the node has CPUs, an unknown node, an id that is not there: nobody is pinned
    assert numa_plan(lib, tmp_path, eight[:1], 200)[1] == [-1] * 200
    assert numa_plan(lib, tmp_path, eight[:2] + ["0000:aa:00.0"], 8) == ([0, 0, -1], [-1] * 8)
    assert numa_plan(lib, tmp_path, ["0000:00:99.0", eight[0]], 4) == ([-1, 0], [-1] * 4)
    assert numa_plan(lib, tmp_path, [None], 2) == ([-1], [-1, -1])
    assert numa_plan(lib, tmp_path, eight, 0) == ([0, 0, 0, 0, 1, 1, 1, 1], [])
    ids = (C.c_char_p * 1)(b"x")
    assert lib.vit_numa_plan(None, ids, 0, 1, (C.c_int * 1)(), (C.c_int * 1)()) == -1
    # odd cpulist texts
    fake_sysfs(tmp_path / "odd", {}, {0: "3", 1: "", 2: "5-4", 3: "1,1,2-3,garbage", 4: "0-2,"})
    for node, want in ((0, [3]), (1, []), (2, []), (3, [1, 2, 3]), (4, [0, 1, 2])):
        n = lib.vit_numa_node_cpus(str(tmp_path / "odd").encode(), node, cpus, 256)
        assert list(cpus)[:n] == want, node


def test_launch_shape_does_not_wrap_at_the_top_of_the_object_range(lib):
    """The ABI admits n up to 2^32 - 1; the launch arithmetic must cover every object there
    (n + 63 and n + 255 wrap in 32 bits for the last values of n)."""
    blocks, threads = C.c_uint32(0), C.c_uint32(0)
    for kernel in range(6):
        for n in (1, 63, 64, 65, 255, 256, 257, 1_000_000, 0xFFFFFF00, 0xFFFFFFC0, 0xFFFFFFC1, 0xFFFFFFFF):
            assert lib.vit_launch_shape(kernel, n, C.byref(blocks), C.byref(threads)) == 0
            assert threads.value in (64, 128, 256)
            lanes = blocks.value * threads.value
            assert lanes >= n and lanes - n < threads.value, (kernel, n, blocks.value, threads.value)
    assert lib.vit_launch_shape(0, 1 << 32, C.byref(blocks), C.byref(threads)) == -1
    assert lib.vit_launch_shape(9, 5, C.byref(blocks), C.byref(threads)) == -1


def test_host_lane_sha1_matches_reference_golden(lib, oracle, golden_objects):
    g = golden_objects["blobs"]
    for L, want in g["ids"]:
        data = oracle.synth_bytes(g["seed"], L, L)
        out = C.create_string_buffer(20)
        arr = np.frombuffer(data, dtype=np.uint8) if L else np.zeros(1, dtype=np.uint8)
        assert lib.vit_host_object_id(b"blob", arr.ctypes.data, L, out) == 0
        assert out.raw.hex() == want, L
    out = C.create_string_buffer(20)
    assert lib.vit_host_object_id(b"tree", None, 0, out) == 0
    assert out.raw.hex() == golden_objects["kats"]["git_empty_tree"]
    assert lib.vit_host_object_id(b"blob", None, 5, out) == -1


def test_interleaved_host_lane_matches_reference_golden(lib, oracle, golden_objects):
    """vit_host_object_ids (several objects' SHA-1 streams side by side on one core) on the 256 golden blob lengths,
    in input order and in groups whose members differ in length by orders of magnitude, and on typed objects."""
    g = golden_objects["blobs"]
    ids = g["ids"]
    for order in (list(range(len(ids))), list(range(len(ids) - 1, -1, -1)),
                  np.random.default_rng(4).permutation(len(ids)).tolist()):
        bufs = [np.frombuffer(oracle.synth_bytes(g["seed"], ids[i][0], ids[i][0]) or b"\0", dtype=np.uint8) for i in order]
        lens = np.array([ids[i][0] for i in order], dtype=np.uint64)
        ptrs = (C.c_void_p * len(order))(*[b.ctypes.data if L else None for b, L in zip(bufs, lens)])
        out = np.zeros((len(order), 20), dtype=np.uint8)
        assert lib.vit_host_object_ids(b"blob", ptrs, lens.ctypes.data, len(order), out.ctypes.data) == 0
        for k, i in enumerate(order):
            assert out[k].tobytes().hex() == ids[i][1], (ids[i][0], k)
    for n in (0, 1, 2, 3, 5):                                     # counts around the group size; another type
        data = [oracle.synth_bytes(9, i, 1000 * i + i) for i in range(n)]
        bufs = [np.frombuffer(d or b"\0", dtype=np.uint8) for d in data]
        lens = np.array([len(d) for d in data], dtype=np.uint64)
        ptrs = (C.c_void_p * max(n, 1))(*[b.ctypes.data if len(d) else None for b, d in zip(bufs, data)])
        out = np.zeros((max(n, 1), 20), dtype=np.uint8)
        assert lib.vit_host_object_ids(b"commit", ptrs, lens.ctypes.data, n, out.ctypes.data) == 0
        for i in range(n):
            assert out[i].tobytes() == oracle.object_id("commit", data[i])
    one = (C.c_void_p * 1)(None)
    five = np.array([5], dtype=np.uint64)
    assert lib.vit_host_object_ids(b"blob", one, five.ctypes.data, 1, out.ctypes.data) == -1


def test_library_synth_bytes_is_the_shared_generator(oracle):
    import vit_amd
    for L in (0, 1, 8, 9, 4096, 4099):
        assert vit_amd.synth_bytes(0x5EED0002, 999999, L) == oracle.synth_bytes(0x5EED0002, 999999, L)


def test_create_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import vit_amd
    with pytest.raises(vit_amd.VitError) as e:
        vit_amd.Hasher((0,))
    assert e.value.code == -5
    assert "no CPU path" in str(e.value)


def test_product_does_not_use_the_oracle():
    """Nothing under vit_amd/ (the shipped path) may import, link or call oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "vit_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                text = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"\boracle\b|vit_oracle|liboracle|libvitref", text):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
