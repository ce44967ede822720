#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REFERENCE'S OWN CODE.

Run in the build container only (needs /root/reference and oracle/_ref, built by
`make -C oracle ref`): every blob and tree id below is produced by the
reference's writeBlob / writeTree / writeObject (src/commit.cpp:88-173) through
oracle/_ref/libvitref.so, and cross-checked against real `git` where the two are
defined to agree.  The fixtures hold generator parameters and ids only — no
file contents and no reference source.

    python tests/golden/gen_golden.py
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402

SEED_BLOBS = 0x5EED00B0
SEED_C1 = 0x5EED0001


def boundary_lengths():
    """Lengths whose framed size lands on every SHA-1 padding edge, plus the
    lengths where the decimal header grows by a digit."""
    L = set(range(0, 201))
    for base in (4096, 65536, 1 << 20):
        for d in range(-70, 71):
            n = base + d
            hdr = 6 + len(str(n))
            if (hdr + n) % 64 in (54, 55, 56, 57, 63, 0, 1):
                L.add(n)
        L.add(base)
    for p in range(1, 8):
        L.update((10 ** p - 1, 10 ** p))
    return sorted(L)


def in_scratch_repo(fn):
    cwd = os.getcwd()
    td = tempfile.mkdtemp(prefix="vitgold_")
    try:
        os.makedirs(os.path.join(td, ".git", "objects"))
        os.chdir(td)
        return fn(td)
    finally:
        os.chdir(cwd)
        shutil.rmtree(td, ignore_errors=True)


def git_hash_object(data):
    return subprocess.run(["git", "hash-object", "--stdin"], input=data, stdout=subprocess.PIPE,
                          check=True).stdout.decode().strip()


def gen_blobs():
    lens = boundary_lengths()

    def work(_):
        out = []
        for L in lens:
            data = O.synth_bytes(SEED_BLOBS, L, L)
            ref = O.ref_write_blob(data)
            assert ref == hashlib.sha1(b"blob %d\0" % L + data).hexdigest(), L
            if L <= 200 or L in (4096, 65536):
                assert ref == git_hash_object(data), L
            out.append([L, ref])
        return out
    ids = in_scratch_repo(work)
    return {"generator": "content = synth_bytes(seed, index=L, length=L)", "seed": SEED_BLOBS,
            "source": "reference writeBlob (src/commit.cpp:116) via oracle/_ref; == hashlib; == git hash-object on a subset",
            "ids": ids}


def gen_typed_objects():
    """writeObject with the three type strings the reference uses (:116,:172,:275)."""
    def work(_):
        out = []
        for t in ("blob", "tree", "commit"):
            for L in (0, 1, 55, 56, 64, 200, 4096):
                data = O.synth_bytes(SEED_BLOBS ^ 0xFF, L, L)
                out.append([t, L, O.ref_write_object(t, data)])
        return out
    return {"seed": SEED_BLOBS ^ 0xFF, "generator": "content = synth_bytes(seed, index=L, length=L)",
            "source": "reference writeObject (src/commit.cpp:88-114) via oracle/_ref", "ids": in_scratch_repo(work)}


def materialize(spec, root):
    """spec: list of (relative path, seed, index, length) or (relative dir, None)."""
    for item in spec:
        path = os.path.join(root, item[0])
        if item[1] is None:
            os.makedirs(path, exist_ok=True)
            continue
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "wb") as f:
            f.write(O.synth_bytes(item[1], item[2], item[3]))
        if len(item) > 4 and item[4] == "x":
            os.chmod(path, 0o755)


def c1_spec():
    """Config 1: 1,000 files x 4,096 B in d00..d09 (100 each)."""
    return [("d%02d/f%04d.bin" % (i // 100, i), SEED_C1, i, 4096) for i in range(1000)]


def skew_spec(seed, n):
    """n files in t%02d/s%02d/ (3 levels, 20 x 10 directories), sizes skewed like a source tree: 5 % empty,
    75 % under 4 KiB, 17 % under 32 KiB, 2.9 % under 256 KiB, 0.1 % under 2 MiB.  Sizes come from the same
    counter-based stream as the contents (8 bytes of stream seed^0x51E5, index i), so the fixture depends on
    nothing but that generator.  tests/conftest.py holds the same function."""
    spec = []
    for i in range(n):
        r = int.from_bytes(O.synth_bytes(seed ^ 0x51E5, i, 8), "little")
        cls, v = r % 1000, r >> 16
        size = 0 if cls < 50 else v % 4096 if cls < 800 else v % 32768 if cls < 970 else v % 262144 if cls < 999 \
            else v % (2 << 20)
        spec.append(("t%02d/s%02d/f%05d.dat" % (i % 20, (i // 20) % 10, i), seed, i, size))
    return spec


def quirk_specs():
    s = 0x5EED0A11
    return {
        # directory "foo" next to file "foo.c": the reference sorts plain names (:164-165), git sorts "foo/"
        "dir_vs_file_prefix": [("foo/a.txt", s, 1, 10), ("foo.c", s, 2, 20), ("foo-bar", s, 3, 5)],
        # empty directory: the reference records an entry for the empty tree (:156-158,172), git drops it
        "empty_dir": [("keep/a", s, 4, 7), ("empty", None)],
        # executable bit is ignored: always 100644 (:154)
        "exec_bit": [("run.sh", s, 5, 33, "x"), ("data", s, 6, 0)],
        # nested three levels + names with bytes >= 0x80 (unsigned order) and spaces
        "nested_names": [("a/b/c/deep.bin", s, 7, 4096), ("a/b/x y.txt", s, 8, 100), ("a/été.txt", s, 9, 9),
                         ("a/z", s, 10, 1), ("Z", s, 11, 2), ("a.b", s, 12, 3), ("a0", s, 13, 4)],
        "single_empty_file": [("empty.txt", s, 14, 0)],
    }


def tree_record(spec, with_blob_ids=False):
    def work(td):
        wt = os.path.join(td, "wt")
        os.makedirs(os.path.join(wt, ".git", "objects"))
        materialize(spec, wt)
        os.chdir(wt)
        root = O.ref_write_tree(".")
        assert root, "reference writeTree failed"
        # every object the reference stored: (id, type) by inflating its own files
        import zlib
        objs = {}
        for d in os.listdir(".git/objects"):
            for f in os.listdir(os.path.join(".git/objects", d)):
                raw = zlib.decompress(open(os.path.join(".git/objects", d, f), "rb").read())
                objs[d + f] = raw.split(b" ", 1)[0].decode()
        trees = sorted(k for k, v in objs.items() if v == "tree")
        rec = {"root": root, "n_blobs": sum(1 for v in objs.values() if v == "blob"), "tree_ids": trees}
        if with_blob_ids:
            rec["blob_ids"] = [hashlib.sha1(b"blob %d\0" % it[3] + O.synth_bytes(it[1], it[2], it[3])).hexdigest()
                               for it in spec if it[1] is not None]
            assert all(b in objs for b in rec["blob_ids"])
        # what git itself says for the same worktree (may differ by design, see quirk list)
        shutil.rmtree(".git")
        subprocess.check_call(["git", "init", "-q", "."])
        subprocess.check_call(["git", "add", "-A"])
        rec["git_write_tree"] = subprocess.check_output(["git", "write-tree"]).decode().strip()
        return rec
    return in_scratch_repo(work)


def main():
    if O.ref_lib() is None:
        sys.exit("oracle/_ref/libvitref.so missing: run `make -C oracle ref` in the container that has /root/reference")
    out = {
        "kats": {
            "sha1": [["", "da39a3ee5e6b4b0d3255bfef95601890afd80709"],
                     ["abc", "a9993e364706816aba3e25717850c26c9cd0d89d"],
                     ["abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq",
                      "84983e441c3bd26ebaae4aa1f95129e5e54670f1"]],
            "sha1_million_a": "34aa973cd4c4daa4f61eeb2bdbad27316534016f",
            "git_empty_blob": "e69de29bb2d1d6434b8b29ae775ad8c2e48c5391",
            "git_empty_tree": "4b825dc642cb6eb9a060e54bf8d69288fbee4904",
            "git_hello_blob": ["hello\n", "ce013625030ba8dba906f756967f9e9ca394464a"],
        },
        "blobs": gen_blobs(),
        "typed": gen_typed_objects(),
    }
    # 64-bit length framing: "blob 4294967301\0" + that many 0x5A bytes, streamed through hashlib (OpenSSL)
    big = (1 << 32) + 5
    hh = hashlib.sha1(b"blob %d\0" % big)
    chunk = b"\x5a" * (1 << 20)
    left = big
    while left:
        take = min(left, len(chunk))
        hh.update(chunk[:take])
        left -= take
    out["framing_over_4g"] = {"fill": 0x5A, "length": big, "id": hh.hexdigest(),
                              "source": "hashlib (OpenSSL) streamed; the reference would need the whole string in RAM"}
    json.dump(out, open(os.path.join(HERE, "objects.json"), "w"), indent=0)

    trees = {"c1_1000x4096": dict(spec="c1", seed=SEED_C1, **tree_record(c1_spec(), with_blob_ids=True))}
    for name, spec in quirk_specs().items():
        trees[name] = dict(spec=[list(x) for x in spec], **tree_record(spec, with_blob_ids=True))
    trees["skew_10000"] = dict(spec="skew", seed=0x5EED0003, n=10000,
                               **tree_record(skew_spec(0x5EED0003, 10000), with_blob_ids=True))
    json.dump(trees, open(os.path.join(HERE, "trees.json"), "w"), indent=0)
    print("wrote objects.json (%d blob ids) and trees.json (%d worktrees)" % (len(out["blobs"]["ids"]), len(trees)))
    for k, v in trees.items():
        print("  %-22s reference %s  git %s%s" % (k, v["root"], v["git_write_tree"],
                                                   "" if v["root"] == v["git_write_tree"] else "   (differs by design)"))


if __name__ == "__main__":
    main()
