"""Python view of the C ABI (include/vit_hash.h), used by the tests and bench.py.

Every method is a direct call into libvit_hash.so; nothing is computed here.
Digests come back as numpy uint8 arrays of shape (n, 20), in input order —
the raw bytes the reference turns into hex with hashToHexString
(reference src/commit.cpp:52-59,95).
"""
import ctypes as C

import numpy as np

from . import _abi

KERNEL_LDS = 0
KERNEL_LANE = 1


class VitError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("vit error %d: %s" % (code, text))
        self.code = code


def header_len(type_, length):
    return _abi.load().vit_header_len(type_.encode(), length)


def object_blocks(type_, length):
    return _abi.load().vit_object_blocks(type_.encode(), length)


def synth_bytes(seed, index, length):
    out = np.empty(max(length, 1), dtype=np.uint8)
    _abi.load().vit_synth_bytes(seed, index, out.ctypes.data, length)
    return out[:length].tobytes()


def _as_ptr_arrays(blobs):
    """list of bytes-like -> (keepalive, uint8* array, uint64 lengths array)."""
    n = len(blobs)
    ptrs = (C.c_void_p * max(n, 1))()
    lens = np.empty(max(n, 1), dtype=np.uint64)
    keep = []
    for i, b in enumerate(blobs):
        if isinstance(b, np.ndarray):
            a = np.ascontiguousarray(b, dtype=np.uint8)
        else:
            a = np.frombuffer(bytes(b), dtype=np.uint8)
        keep.append(a)
        ptrs[i] = a.ctypes.data if a.size else None
        lens[i] = a.size
    return keep, ptrs, lens


def pack_arena(type_, blobs, arena=None, threads=4, slot_align=64):
    """Host-only packing (vit_arena_pack_aligned).  Returns (arena, offsets, nblocks, used)."""
    lib = _abi.load()
    keep, ptrs, lens = _as_ptr_arrays(blobs)
    n = len(blobs)
    need = sum((int(lib.vit_object_blocks(type_.encode(), int(l))) * 64 + slot_align - 1) // slot_align * slot_align
               for l in lens[:n])
    if arena is None:                       # 64-byte aligned, as the pinned staging arenas are
        raw = np.zeros(max(need, 64) + 64, dtype=np.uint8)
        skip = (-raw.ctypes.data) % 64
        arena = raw[skip:skip + max(need, 64)]
    offsets = np.zeros(max(n, 1), dtype=np.uint64)
    nblocks = np.zeros(max(n, 1), dtype=np.uint32)
    used = C.c_uint64(0)
    rc = lib.vit_arena_pack_aligned(type_.encode(), ptrs, lens.ctypes.data, n, arena.ctypes.data, arena.size,
                                    offsets.ctypes.data, nblocks.ctypes.data, C.byref(used), threads, slot_align)
    if rc != 0:
        raise VitError(rc, "vit_arena_pack_aligned")
    return arena, offsets[:n], nblocks[:n], used.value


def bin_order(nblocks):
    nblocks = np.ascontiguousarray(nblocks, dtype=np.uint32)
    order = np.empty(max(len(nblocks), 1), dtype=np.uint32)
    rc = _abi.load().vit_bin_order(nblocks.ctypes.data, len(nblocks), order.ctypes.data)
    if rc < 0:
        raise VitError(rc, "vit_bin_order")
    return order[:len(nblocks)], bool(rc)


class DeviceBuffer:
    """Device memory owned through vit_device_alloc/free."""

    def __init__(self, hasher, dev, nbytes):
        self.h, self.dev, self.nbytes = hasher, dev, int(nbytes)
        p = C.c_void_p()
        hasher._check(hasher.lib.vit_device_alloc(hasher.handle, dev, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.h._check(self.h.lib.vit_device_upload(self.h.handle, self.dev, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, dtype=np.uint8, nbytes=None, offset=0):
        nbytes = self.nbytes - offset if nbytes is None else int(nbytes)
        out = np.empty(nbytes, dtype=np.uint8)
        self.h._check(self.h.lib.vit_device_download(self.h.handle, self.dev, out.ctypes.data,
                                                     self.ptr + offset, nbytes))
        return out.view(dtype)

    def free(self):
        if self.ptr:
            self.h.lib.vit_device_free(self.h.handle, self.dev, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Hasher:
    """vit_hasher handle.  devices=None: every visible GPU."""

    def __init__(self, devices=(0,)):
        self.lib = _abi.load()
        ids = list(devices) if devices is not None else []
        arr = (C.c_int * max(len(ids), 1))(*ids)
        h = C.c_void_p()
        rc = self.lib.vit_hasher_create(arr if ids else None, len(ids), C.byref(h))
        if rc != 0:
            raise VitError(rc, self.lib.vit_last_error(None).decode())
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.lib.vit_hasher_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise VitError(rc, self.lib.vit_last_error(self.handle).decode())

    @property
    def device_count(self):
        return self.lib.vit_hasher_device_count(self.handle)

    def set_option(self, name, value):
        self._check(self.lib.vit_hasher_set_option(self.handle, name.encode(), int(value)))

    def stat(self, name):
        v = C.c_uint64(0)
        self._check(self.lib.vit_hasher_get_stat(self.handle, name.encode(), C.byref(v)))
        return v.value

    # -- host-memory batches ------------------------------------------------
    def hash_objects(self, type_, blobs):
        keep, ptrs, lens = _as_ptr_arrays(blobs)
        n = len(blobs)
        out = np.zeros((max(n, 1), 20), dtype=np.uint8)
        self._check(self.lib.vit_hash_objects(self.handle, type_.encode(), ptrs, lens.ctypes.data, n,
                                              out.ctypes.data))
        return out[:n]

    def hash_blobs(self, blobs):
        keep, ptrs, lens = _as_ptr_arrays(blobs)
        n = len(blobs)
        out = np.zeros((max(n, 1), 20), dtype=np.uint8)
        self._check(self.lib.vit_hash_blobs(self.handle, ptrs, lens.ctypes.data, n, out.ctypes.data))
        return out[:n]

    def sha1_buffers(self, buffers):
        """Plain SHA-1 of each buffer (no Git framing)."""
        keep, ptrs, lens = _as_ptr_arrays(buffers)
        n = len(buffers)
        out = np.zeros((max(n, 1), 20), dtype=np.uint8)
        self._check(self.lib.vit_sha1_buffers(self.handle, ptrs, lens.ctypes.data, n, out.ctypes.data))
        return out[:n]

    def hash_arena(self, arena, offsets, nblocks, dev=0, arena_bytes=None):
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nblocks = np.ascontiguousarray(nblocks, dtype=np.uint32)
        n = len(offsets)
        out = np.zeros((max(n, 1), 20), dtype=np.uint8)
        self._check(self.lib.vit_hash_arena(self.handle, dev, arena.ctypes.data,
                                            arena.size if arena_bytes is None else arena_bytes,
                                            offsets.ctypes.data, nblocks.ctypes.data, n, out.ctypes.data))
        return out[:n]

    # -- device-resident batches ---------------------------------------------
    def alloc(self, nbytes, dev=0):
        return DeviceBuffer(self, dev, nbytes)

    def sync(self, dev=0):
        self._check(self.lib.vit_device_sync(self.handle, dev))

    def device_status(self, dev=0):
        s = C.c_uint32(0)
        self._check(self.lib.vit_device_status(self.handle, dev, C.byref(s)))
        return s.value

    def synth_arena(self, type_, seed, first_index, n, len_each, dev=0, lens=None, slot_align=64):
        """Generate a framed synthetic arena on the device.  Returns a dict of
        DeviceBuffers (arena, offsets, nblocks, digests) plus sizes.  slot_align
        (a multiple of 64) rounds every slot's start up to that boundary."""
        assert slot_align % 64 == 0 and slot_align > 0
        if slot_align != 64 and lens is None:
            lens = np.full(n, len_each, dtype=np.uint64)
        if lens is None:
            nb = object_blocks(type_, len_each)
            arena_bytes = nb * 64 * n
            d_lens = None
            offsets_host = None
        else:
            lens = np.ascontiguousarray(lens, dtype=np.uint64)
            assert len(lens) == n
            nbs = np.array([object_blocks(type_, int(l)) for l in lens], dtype=np.uint64) \
                if n < 200000 else _blocks_vec(type_, lens)
            q = np.uint64(slot_align)
            offsets_host = np.concatenate([[0], np.cumsum((nbs * np.uint64(64) + q - np.uint64(1)) // q * q)]).astype(np.uint64)
            arena_bytes = int(offsets_host[-1])
            offsets_host = offsets_host[:-1].copy()
            d_lens = self.alloc(8 * n, dev).upload(lens)
        arena = self.alloc(max(arena_bytes, 64), dev)
        offsets = self.alloc(8 * max(n, 1), dev)
        nblocks = self.alloc(4 * max(n, 1), dev)
        digests = self.alloc(20 * max(n, 1), dev)
        if offsets_host is not None:
            offsets.upload(offsets_host)
        self._check(self.lib.vit_synth_arena_device(
            self.handle, dev, type_.encode(), seed, first_index, n, len_each,
            d_lens.ptr if d_lens else None, offsets.ptr, nblocks.ptr, arena.ptr, arena_bytes, None))
        self.sync(dev)
        return dict(arena=arena, offsets=offsets, nblocks=nblocks, digests=digests, n=n,
                    arena_bytes=arena_bytes, lens=d_lens, dev=dev)

    def bin_order_device(self, batch):
        """Length-binning order of a device-resident batch, computed on the device; returns a DeviceBuffer."""
        order = self.alloc(4 * max(batch["n"], 1), batch["dev"])
        self._check(self.lib.vit_bin_order_device(self.handle, batch["dev"], batch["nblocks"].ptr, batch["n"],
                                                  order.ptr, None))
        return order

    def hash_device(self, batch, order=None):
        """Launch on a batch from synth_arena(); returns digests as (n, 20)."""
        self._check(self.lib.vit_hash_arena_device(
            self.handle, batch["dev"], batch["arena"].ptr, batch["arena_bytes"], batch["offsets"].ptr,
            batch["nblocks"].ptr, order.ptr if order is not None else None, batch["n"],
            batch["digests"].ptr, None))
        self.sync(batch["dev"])
        return batch["digests"].download(np.uint8, 20 * batch["n"]).reshape(-1, 20)

    def hash_device_static(self, batch):
        """The launch that allocates nothing (device-side binning into the library's static scratch)."""
        self._check(self.lib.vit_hash_arena_device_static(
            self.handle, batch["dev"], batch["arena"].ptr, batch["arena_bytes"], batch["offsets"].ptr,
            batch["nblocks"].ptr, batch["n"], batch["digests"].ptr, None))
        self.sync(batch["dev"])
        return batch["digests"].download(np.uint8, 20 * batch["n"]).reshape(-1, 20)

    def _deflate_device_prepare(self, batch, framed_lens):
        """Output buffer, offsets and lengths in device memory for a vit_deflate_arena_device call."""
        n = batch["n"]
        lib = self.lib
        cap = np.array([max(lib.vit_deflate_bound(int(L)), 16) for L in framed_lens], dtype=np.uint64)
        cap = (cap + np.uint64(15)) // np.uint64(16) * np.uint64(16)
        off = np.concatenate([[0], np.cumsum(cap)]).astype(np.uint64)
        d_out = self.alloc(int(off[-1]) + 16, batch["dev"])
        d_off = self.alloc(8 * n, batch["dev"]).upload(off[:n].copy())
        d_len = self.alloc(4 * n, batch["dev"])
        return {"n": n, "off": off, "d_out": d_out, "d_off": d_off, "d_len": d_len}

    def deflate_device_launch(self, batch, framed_lens, order=None, stream=None, prepared=None):
        """Queue vit_deflate_arena_device for a batch from synth_arena() on `stream` (a hipStream_t as an integer;
        None = the queue's own) and return what deflate_device_collect needs; nothing is waited for."""
        p = prepared if prepared is not None else self._deflate_device_prepare(batch, framed_lens)
        self._check(self.lib.vit_deflate_arena_device(self.handle, batch["dev"], batch["arena"].ptr, batch["arena_bytes"],
                                                      batch["offsets"].ptr, batch["nblocks"].ptr,
                                                      order.ptr if order is not None else None, p["n"], p["d_out"].ptr,
                                                      p["d_off"].ptr, p["d_len"].ptr, C.c_void_p(stream) if stream else None))
        p["order"] = order
        return p

    @staticmethod
    def deflate_device_collect(launched):
        """The streams of a finished deflate_device_launch (the caller has synchronised): list of bytes."""
        n, off = launched["n"], launched["off"]
        lens = launched["d_len"].download(np.uint32, 4 * n)
        raw = launched["d_out"].download(np.uint8, int(off[-1]))
        return [raw[int(off[i]):int(off[i]) + int(lens[i])].tobytes() for i in range(n)]

    def deflate_device(self, batch, framed_lens, order=None):
        """zlib streams of a batch from synth_arena() (vit_deflate_arena_device).  framed_lens: header + content
        length per object (sizes the output).  Returns (list of bytes, seconds on the device)."""
        import time
        prepared = self._deflate_device_prepare(batch, framed_lens)
        self.sync(batch["dev"])
        t0 = time.perf_counter()
        launched = self.deflate_device_launch(batch, framed_lens, order, prepared=prepared)
        self.sync(batch["dev"])
        dt = time.perf_counter() - t0
        return self.deflate_device_collect(launched), dt

    def deflate_objects(self, type_, blobs):
        """zlib streams of objects in host memory (vit_deflate_objects).  Returns (list of bytes, b'' where the
        object was left to the caller; seconds inside the call)."""
        import time
        keep, ptrs, lens = _as_ptr_arrays(blobs)
        n = len(blobs)
        need = self.lib.vit_deflate_out_bytes(self.handle, type_.encode(), lens.ctypes.data, n)
        out = np.zeros(max(need, 16), dtype=np.uint8)
        off = np.zeros(max(n, 1), dtype=np.uint64)
        zl = np.zeros(max(n, 1), dtype=np.uint32)
        t0 = time.perf_counter()
        self._check(self.lib.vit_deflate_objects(self.handle, type_.encode(), ptrs, lens.ctypes.data, n, out.ctypes.data,
                                                 out.size, off.ctypes.data, zl.ctypes.data))
        dt = time.perf_counter() - t0
        return [out[int(off[i]):int(off[i]) + int(zl[i])].tobytes() for i in range(n)], dt

    def time_hash_device(self, batch, order=None, warmup=1, iters=5):
        ms = (C.c_float * max(iters, 1))()
        self._check(self.lib.vit_time_hash_device(
            self.handle, batch["dev"], batch["arena"].ptr, batch["arena_bytes"], batch["offsets"].ptr,
            batch["nblocks"].ptr, order.ptr if order is not None else None, batch["n"],
            batch["digests"].ptr, warmup, iters, ms))
        return [float(ms[i]) for i in range(iters)]

    def valu_peak(self, n_lanes, blocks_per_lane, iters=3, dev=0):
        ms = C.c_float(0)
        self._check(self.lib.vit_valu_peak(self.handle, dev, n_lanes, blocks_per_lane, iters, C.byref(ms)))
        return float(ms.value)


def _blocks_vec(type_, lens):
    """Vectorised vit_object_blocks for large length arrays."""
    lens = lens.astype(np.uint64)
    digits = np.ones(len(lens), dtype=np.uint64)
    for k in range(1, 20):
        digits += (lens >= np.uint64(10 ** k)).astype(np.uint64)
    hl = np.uint64(len(type_) + 2) + digits
    return (hl + lens + np.uint64(9 + 63)) // np.uint64(64)
