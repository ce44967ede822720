"""The deflate kernel through the C ABI against the zlib library (zlib.compress at the default level, the
reference's compress() call at src/commit.cpp:100): every stream byte for byte."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _framed(type_, data):
    return type_.encode() + b" %d\0" % len(data) + data


def test_device_streams_equal_zlibs_for_every_kind_of_content(hasher):
    """vit_deflate_objects: random, text-like, constant and repetitive contents, empty objects, lengths around the
    block and table edges; objects above deflate_max_bytes come back with length 0 (the caller's to do)."""
    rng = np.random.default_rng(31)
    words = [b"the ", b"quick ", b"brown ", b"fox ", b"int ", b"return ", b"void ", b"#include ", b"\n", b"    ",
             b"for (", b"i++) {", b"}", b"const ", b"0x", b"==", b"abcabcabc"]
    blobs = []
    for L in [0, 1, 2, 3, 55, 56, 57, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 16000, 16383 - 10, 16384, 20000, 40000, 65000,
              65536 - 11, 65536 - 10, 65536, 70000] + rng.integers(0, 9000, 400).tolist():
        kind = int(rng.integers(0, 5))
        if kind == 0:
            d = rng.integers(0, 256, L, dtype=np.uint8).tobytes()
        elif kind == 1:
            d = b"".join(words[int(k)] for k in rng.integers(0, len(words), L // 3 + 1))[:L]
        elif kind == 2:
            d = bytes([int(rng.integers(0, 256))]) * L
        elif kind == 3:
            d = (rng.integers(0, 4, L, dtype=np.uint8) + 97).tobytes()
        else:
            base = rng.integers(0, 256, 700, dtype=np.uint8).tobytes()
            d = (base * (L // 700 + 1))[:L]
        blobs.append(d)
    hasher.set_option("deflate_max_bytes", 65536)
    try:
        for type_ in ("blob", "tree"):
            got, _ = hasher.deflate_objects(type_, blobs)
            for d, z in zip(blobs, got):
                framed = _framed(type_, d)
                if len(framed) > 65536:
                    assert z == b"", len(d)
                else:
                    assert z == zlib.compress(framed, 6), (type_, len(d))
        hasher.set_option("deflate_max_bytes", 4096)          # the cut between the device's objects and the caller's
        got, _ = hasher.deflate_objects("blob", blobs)
        for d, z in zip(blobs, got):
            framed = _framed("blob", d)
            assert z == (zlib.compress(framed, 6) if len(framed) <= 4096 else b""), len(d)
    finally:
        hasher.set_option("deflate_max_bytes", 16382)
    assert hasher.device_status() == 0


def _content_for_framed_length(target, type_="blob"):
    """Length of the content whose framed object (type, length in decimal, NUL, content) has exactly `target` bytes;
    None where no length gives it (one more digit in the header skips a value)."""
    for digits in range(1, 7):
        L = target - (len(type_) + 1 + digits + 1)
        if L >= 0 and len(str(L)) == digits:
            return L
    return None


def test_objects_at_the_edges_of_the_size_classes_and_of_the_tiles(hasher):
    """Framed lengths at and around what separates the workgroup kernels' four size classes (1536, 4608, 9216, 16382),
    the one-per-lane kernel behind them (16383) and the tiles the positions are searched in (512, 1024, 2048 and
    their multiples), the last 15 positions (searched by zlib's own loop) and the shortest objects, with contents
    that make the longest chains (one byte repeated, a period of three, a four-letter alphabet), none at all
    (random) and ordinary ones (words)."""
    rng = np.random.default_rng(77)
    words = [b"the ", b"quick ", b"brown ", b"fox ", b"return ", b"void ", b"\n", b"    ", b"for (", b"}", b"const "]
    edges = [1536, 4608, 9216, 16382]
    framed_lengths = sorted({e + k for e in edges for k in (-17, -16, -15, -2, -1, 0, 1, 2, 15, 16, 17)} |
                            {t * m + k for t in (512, 1024, 2048) for m in (1, 2, 3, 7) for k in (-16, -1, 0, 1, 16)} |
                            set(range(7, 40)) | {16383, 16384, 16400})
    blobs = []
    for target in framed_lengths:
        L = _content_for_framed_length(target)
        if L is None:
            continue
        blobs += [rng.integers(0, 256, L, dtype=np.uint8).tobytes(),
                  b"".join(words[int(k)] for k in rng.integers(0, len(words), L // 2 + 1))[:L],
                  bytes([int(rng.integers(0, 256))]) * L,
                  (b"abc" * (L // 3 + 1))[:L],
                  (rng.integers(0, 4, L, dtype=np.uint8) + 97).tobytes()]
    hasher.set_option("deflate_max_bytes", 65536)
    try:
        got, _ = hasher.deflate_objects("blob", blobs)
    finally:
        hasher.set_option("deflate_max_bytes", 16382)
    for d, z in zip(blobs, got):
        assert z == zlib.compress(_framed("blob", d), 6), (len(d), d[:12])
    got, _ = hasher.deflate_objects("blob", blobs)               # the default limit: what is beyond one block is the caller's
    for d, z in zip(blobs, got):
        framed = _framed("blob", d)
        assert z == (zlib.compress(framed, 6) if len(framed) <= 16382 else b""), len(d)
    assert hasher.device_status() == 0


def test_device_resident_arena_streams_and_the_files_they_make(hasher):
    """vit_deflate_arena_device on 100,000 synthetic 4 KiB blobs resident in HBM, in order of length and without:
    sampled streams equal zlib's and inflate to header + content."""
    import vit_amd
    n = 100_000
    lens = np.random.default_rng(5).integers(0, 8200, n).astype(np.uint64)
    b = hasher.synth_arena("blob", 0x5EED00AB, 0, n, 0, lens=lens)
    framed = [int(l) + vit_amd.header_len("blob", int(l)) for l in lens]
    nb = b["nblocks"].download(np.uint32, 4 * n)
    order = hasher.alloc(4 * n).upload(vit_amd.bin_order(nb)[0])
    plain, _ = hasher.deflate_device(b, framed)
    binned, _ = hasher.deflate_device(b, framed, order)
    assert plain == binned
    for i in np.random.default_rng(6).integers(0, n, 1500).tolist() + [0, n - 1, int(np.argmax(lens)), int(np.argmin(lens))]:
        data = vit_amd.synth_bytes(0x5EED00AB, i, int(lens[i]))
        assert plain[i] == zlib.compress(_framed("blob", data), 6), i
        assert zlib.decompress(plain[i]) == _framed("blob", data)
    assert hasher.device_status() == 0


def test_launches_on_two_streams_take_turns_with_the_queues_workspace(hasher):
    """Two vit_deflate_arena_device calls on different streams, neither waited for: they share the queue's tables, so
    the library orders them; both give the streams they give alone."""
    import torch
    import vit_amd
    n = 40_000
    batches, framed = [], []
    for seed in (0x5EED00AC, 0x5EED00AD):
        lens = np.random.default_rng(seed & 0xFF).integers(1000, 6000, n).astype(np.uint64)
        batches.append(hasher.synth_arena("blob", seed, 0, n, 0, lens=lens))
        framed.append([int(l) + vit_amd.header_len("blob", int(l)) for l in lens])
    alone = [hasher.deflate_device(b, f)[0] for b, f in zip(batches, framed)]
    streams = [torch.cuda.Stream(device=0), torch.cuda.Stream(device=0)]
    hasher.sync(0)
    launched = [hasher.deflate_device_launch(b, f, stream=s.cuda_stream) for b, f, s in zip(batches, framed, streams)]
    torch.cuda.synchronize(0)
    hasher.sync(0)
    for k in range(2):
        assert hasher.deflate_device_collect(launched[k]) == alone[k], k
    assert hasher.device_status() == 0


def test_text_like_objects_at_the_size_of_a_batch(hasher):
    """8,192 text-like objects (words drawn with a Zipf law: a hash chain to walk at nearly every position, matches of every
    length, lanes of a wave far apart in their objects) of 200 to 16,000 bytes: every stream is zlib's."""
    rng = np.random.default_rng(77)
    vocab = [bytes(rng.integers(97, 123, int(k)).astype(np.uint8)) for k in rng.integers(2, 10, 2000)]
    vocab += [b"    ", b"\n", b"{", b"}", b"(", b")", b";", b"return ", b"const ", b"int ", b"if (", b"for ("]
    pick = np.minimum(rng.zipf(1.3, 1_200_000), len(vocab)) - 1
    text = b" ".join(vocab[int(k)] for k in pick)
    n = 8192
    lens = rng.integers(200, 16000, n)
    starts = rng.integers(0, len(text) - 16000, n)
    blobs = [text[int(s):int(s) + int(L)] for s, L in zip(starts, lens)]
    got, _ = hasher.deflate_objects("blob", blobs)
    for d, z in zip(blobs, got):
        assert z == zlib.compress(_framed("blob", d), 6), len(d)
    assert hasher.device_status() == 0
