"""ctypes doorway to liboracle.so (the C restatement) and, when it was built,
oracle/_ref/libvitref.so (the reference's own commit.cpp).

TEST INFRASTRUCTURE ONLY.  The functions mirror oracle/vit_oracle.h; see that
header for the reference lines each one restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

c_u8p = C.POINTER(C.c_uint8)
c_u64p = C.POINTER(C.c_uint64)


def build(ref=True):
    """Compile liboracle.so, and _ref/libvitref.so when /root/reference is present."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    if ref and os.path.isdir(os.environ.get("VIT_REFERENCE", "/root/reference")):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref",
                               "REF=" + os.environ.get("VIT_REFERENCE", "/root/reference")])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        L.vit_oracle_sha1.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.vit_oracle_object_id.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.vit_oracle_object_id_fill.argtypes = [C.c_char_p, C.c_uint8, C.c_uint64, C.c_void_p]
        L.vit_oracle_header.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p]
        L.vit_oracle_header.restype = C.c_size_t
        L.vit_oracle_hex.argtypes = [C.c_void_p, C.c_char_p]
        L.vit_oracle_unhex.argtypes = [C.c_char_p, C.c_void_p]
        L.vit_oracle_object_path.argtypes = [C.c_char_p, C.c_char_p]
        L.vit_oracle_write_tree.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, c_u64p, c_u64p]
        L.vit_oracle_store_object.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.vit_oracle_commit_text.argtypes = [C.c_char_p] * 5 + [C.c_longlong, C.c_longlong,
                                                                C.POINTER(C.c_size_t)]
        L.vit_oracle_commit_text.restype = C.c_void_p
        L.vit_oracle_synth_fill.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]
        L.vit_oracle_synth_fill_many.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                                 C.c_void_p, C.c_int]
        L.vit_oracle_synth_fill_var.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.vit_oracle_have_libcrypto.restype = C.c_int
        L.vit_oracle_libcrypto_sha1.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.vit_oracle_bench_hash_loop.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_uint64, C.c_int, C.c_int, C.c_void_p,
                                                 C.POINTER(C.c_int)]
        L.vit_oracle_bench_hash_loop.restype = C.c_double
        L.vit_oracle_bench_hash_loop_mt.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                    C.POINTER(C.c_int)]
        L.vit_oracle_bench_hash_loop_mt.restype = C.c_double
        _LIB = L
    return _LIB


def ref_lib():
    """The reference's own code (oracle/_ref/libvitref.so) or None if not built."""
    global _REF
    if _REF is None:
        path = os.path.join(HERE, "_ref", "libvitref.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.vitref_write_blob.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p]
        R.vitref_write_object.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_char_p]
        R.vitref_write_tree.argtypes = [C.c_char_p, C.c_char_p]
        R.vitref_hash_to_hex.argtypes = [C.c_void_p, C.c_char_p]
        R.vitref_hex_to_binary.argtypes = [C.c_char_p, C.c_void_p]
        R.vitref_object_path.argtypes = [C.c_char_p, C.c_char_p]
        R.vitref_read_object_content.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64]
        R.vitref_read_object_content.restype = C.c_uint64
        R.vitref_read_object.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64]
        R.vitref_read_object.restype = C.c_uint64
        if hasattr(R, "vitref_parse_tree"):
            R.vitref_parse_tree.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64]
            R.vitref_parse_tree.restype = C.c_uint64
            R.vitref_files_in_directory.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64]
            R.vitref_files_in_directory.restype = C.c_uint64
        _REF = R
    return _REF


def _buf(data):
    data = bytes(data)
    return (C.c_char * max(len(data), 1)).from_buffer_copy(data or b"\0"), len(data)


def sha1(data):
    b, n = _buf(data)
    out = C.create_string_buffer(20)
    lib().vit_oracle_sha1(b, n, out)
    return out.raw


def object_id(type_, data):
    """Raw 20-byte id of a Git-framed object (src/commit.cpp:90-94)."""
    b, n = _buf(data)
    out = C.create_string_buffer(20)
    lib().vit_oracle_object_id(type_.encode(), b, n, out)
    return out.raw


def object_id_fill(type_, fill, length):
    out = C.create_string_buffer(20)
    lib().vit_oracle_object_id_fill(type_.encode(), fill, length, out)
    return out.raw


def blob_id(data):
    return object_id("blob", data)


def header(type_, size):
    h = C.create_string_buffer(32)
    n = lib().vit_oracle_header(type_.encode(), size, h)
    return h.raw[:n]


def hex_(raw20):
    out = C.create_string_buffer(41)
    lib().vit_oracle_hex(bytes(raw20), out)
    return out.value.decode()


def unhex(hexstr):
    out = C.create_string_buffer(20)
    if lib().vit_oracle_unhex(hexstr.encode(), out) != 0:
        raise ValueError("bad hex")
    return out.raw


def object_path(hexstr):
    out = C.create_string_buffer(64)
    lib().vit_oracle_object_path(hexstr.encode(), out)
    return out.value.decode()


def tree_id(entries):
    """entries: iterable of (mode:str, name:bytes|str, raw20).  Returns (raw20, content)
    following src/commit.cpp:164-172 (sort by name as unsigned bytes)."""
    ents = []
    for mode, name, raw in entries:
        if isinstance(name, str):
            name = os.fsencode(name)
        ents.append((mode.encode(), name, bytes(raw)))
    ents.sort(key=lambda e: e[1])
    content = b"".join(m + b" " + n + b"\0" + r for m, n, r in ents)
    return object_id("tree", content), content


def write_tree(path, objects_dir=None):
    """Reference-order recursive tree id of a directory (src/commit.cpp:139-173).
    Returns (raw20, n_blobs, n_trees)."""
    out = C.create_string_buffer(20)
    nb, nt = C.c_uint64(0), C.c_uint64(0)
    rc = lib().vit_oracle_write_tree(os.fsencode(path),
                                     os.fsencode(objects_dir) if objects_dir else None,
                                     out, C.byref(nb), C.byref(nt))
    if rc != 0:
        raise OSError("vit_oracle_write_tree failed for %r" % (path,))
    return out.raw, nb.value, nt.value


def store_object(objects_dir, type_, data):
    b, n = _buf(data)
    out = C.create_string_buffer(20)
    if lib().vit_oracle_store_object(os.fsencode(objects_dir), type_.encode(), b, n, out) != 0:
        raise OSError("vit_oracle_store_object failed")
    return out.raw


def commit_text(tree_hex, parent_hex, message, author, email, t_author, t_committer):
    n = C.c_size_t(0)
    p = lib().vit_oracle_commit_text(tree_hex.encode(), (parent_hex or "").encode(),
                                     message.encode(), author.encode(), email.encode(),
                                     t_author, t_committer, C.byref(n))
    s = C.string_at(p, n.value)
    C.CDLL(None).free(C.c_void_p(p))
    return s


# ---------------------------------------------------------------- synthetic --
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GAMMA = 0x9E3779B97F4A7C15
_KEYMUL = 0xD1342543DE82EF95
_MASK = (1 << 64) - 1


def synth_bytes(seed, index, length):
    """numpy restatement of vit_oracle_synth_fill (counter-based splitmix64)."""
    if length == 0:
        return b""
    key = (seed ^ ((index * _KEYMUL) & _MASK)) & _MASK
    nw = (length + 7) // 8
    with np.errstate(over="ignore"):
        k = np.arange(1, nw + 1, dtype=np.uint64)
        z = np.uint64(key) + k * np.uint64(_GAMMA)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z.astype("<u8").tobytes()[:length]


def synth_bytes_c(seed, index, length):
    out = C.create_string_buffer(max(length, 1))
    lib().vit_oracle_synth_fill(seed, index, out, length)
    return out.raw[:length]


def synth_fill_many(seed, first, n, len_each, threads=8):
    """n blobs of len_each bytes, contiguous, as a numpy uint8 array."""
    arr = np.empty(n * len_each, dtype=np.uint8)
    lib().vit_oracle_synth_fill_many(seed, first, n, len_each, arr.ctypes.data, threads)
    return arr


def synth_fill_var(seed, first, lens, offsets, out, threads=8):
    """Objects of different lengths into `out` (numpy uint8): object i = stream (seed, first + i) at offsets[i]."""
    lens = np.ascontiguousarray(lens, dtype=np.uint64)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    lib().vit_oracle_synth_fill_var(seed, first, len(lens), lens.ctypes.data, offsets.ctypes.data, out.ctypes.data, threads)
    return out


def bench_hash_loop(blobs, offsets, lens, threads=1, want_libcrypto=True, type_="blob"):
    """Time the reference-equivalent hashing loop (src/commit.cpp:90-94).
    Returns (seconds, digests[n,20], used_libcrypto)."""
    blobs = np.ascontiguousarray(blobs, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    lens = np.ascontiguousarray(lens, dtype=np.uint64)
    n = len(lens)
    dig = np.zeros((n, 20), dtype=np.uint8)
    used = C.c_int(0)
    sec = lib().vit_oracle_bench_hash_loop(type_.encode(), blobs.ctypes.data, offsets.ctypes.data,
                                           lens.ctypes.data, n, threads, int(want_libcrypto),
                                           dig.ctypes.data, C.byref(used))
    return sec, dig, bool(used.value)


def bench_hash_loop_mt(blobs, offsets, lens, threads, reps=1, per_thread_ctx=False, want_libcrypto=True, type_="blob"):
    """The same loop on `threads` threads that exist before the clock starts (released together, stopped together),
    `reps` passes over the sample.  per_thread_ctx: one libcrypto digest context per thread instead of the one-shot
    SHA1() per object.  Returns (seconds for all passes, digests[n,20], 0 oracle SHA-1 / 1 SHA1() / 2 per-thread EVP)."""
    blobs = np.ascontiguousarray(blobs, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    lens = np.ascontiguousarray(lens, dtype=np.uint64)
    n = len(lens)
    dig = np.zeros((n, 20), dtype=np.uint8)
    used = C.c_int(0)
    sec = lib().vit_oracle_bench_hash_loop_mt(type_.encode(), blobs.ctypes.data, offsets.ctypes.data, lens.ctypes.data, n,
                                              threads, int(want_libcrypto), int(per_thread_ctx), reps, dig.ctypes.data,
                                              C.byref(used))
    return sec, dig, used.value


# ------------------------------------------------------- reference doorway --
def ref_write_blob(data):
    """The reference's writeBlob (needs CWD with .git/objects; returns hex)."""
    R = ref_lib()
    b, n = _buf(data)
    out = C.create_string_buffer(41)
    if R.vitref_write_blob(b, n, out) != 0:
        return ""
    return out.value.decode()


def ref_write_object(type_, data):
    R = ref_lib()
    b, n = _buf(data)
    out = C.create_string_buffer(41)
    if R.vitref_write_object(type_.encode(), b, n, out) != 0:
        return ""
    return out.value.decode()


def ref_write_tree(path):
    R = ref_lib()
    out = C.create_string_buffer(41)
    if R.vitref_write_tree(os.fsencode(path), out) != 0:
        return ""
    return out.value.decode()


def ref_parse_tree(hex_id):
    """The reference's parseTree (needs CWD with .git/objects): list of (mode, name, hex id, is_directory), bytes."""
    R = ref_lib()
    buf = C.create_string_buffer(1 << 20)
    n = R.vitref_parse_tree(hex_id.encode(), buf, len(buf))
    fields = buf.raw[:n].split(b"\0")[:-1]
    return [(fields[i], fields[i + 1], fields[i + 2].decode(), fields[i + 3] == b"d") for i in range(0, len(fields), 4)]


def ref_files_in_directory(directory="."):
    """The reference's FileUtils::getFilesInDirectory: list of paths (bytes), in its order."""
    R = ref_lib()
    buf = C.create_string_buffer(1 << 22)
    n = R.vitref_files_in_directory(os.fsencode(directory), buf, len(buf))
    return buf.raw[:n].split(b"\0")[:-1]
