"""The source-level drop-in rehearsed: a program that calls the reference's free functions
(src/commit.hpp:8-25) links against this repository's object_store.cpp instead of the reference's
commit.cpp.  The helpers need no device; the writers are run on both sides of the store's small-work bound:
on the store's own threads (no device, CPU test) and through the GPU hasher."""
import hashlib
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "vit_amd", "bin", "option_a_driver")


def run(args, cwd, env=None):
    p = subprocess.run([DRIVER] + args, cwd=cwd, capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, VIT_DEVICES="0", **(env or {})))
    return p


def test_helpers_of_the_reference_interface_need_no_device(oracle, tmp_path):
    """hashToHexString, hexStringToBinary, binaryToHexString, getObjectPath (src/commit.cpp:52-84) and the
    read side called through the product's own free functions; a hasher is never created for them."""
    assert os.path.exists(DRIVER), "build with `python -m vit_amd.build`"
    p = run(["helpers"], str(tmp_path))
    assert p.returncode == 0, p.stderr
    out = dict(l.split(" ", 1) for l in p.stdout.splitlines() if " " in l)
    raw = bytes((i * 13 + 7) & 0xFF for i in range(20))
    assert out["hex"] == raw.hex() == oracle.hex_(raw)
    assert out["roundtrip"] == "ok"
    assert oracle.unhex(out["hex"]) == raw
    assert out["upper"] == "00ffaabb0102030405060708090a0b0c0d0e0f10"      # std::stoi accepts either case
    assert out["binhex"] == "00ff10"
    assert out["path"] == ".git/objects/ce/013625030ba8dba906f756967f9e9ca394464a" \
        == oracle.object_path("ce013625030ba8dba906f756967f9e9ca394464a")
    assert "badhex throws" in p.stdout
    assert "missing []" in p.stdout and "Object not found" in p.stderr
    assert "GPU hasher" not in p.stderr                                    # reading made no attempt to reach a device
    if oracle.ref_lib() is not None:                                       # the reference's own helpers, where built
        import ctypes as C
        R = oracle.ref_lib()
        buf = C.create_string_buffer(41)
        R.vitref_hash_to_hex(raw, buf)
        assert buf.value.decode() == out["hex"]
        path = C.create_string_buffer(128)
        R.vitref_object_path(b"ce013625030ba8dba906f756967f9e9ca394464a", path)
        assert path.value.decode() == out["path"]


def check_writers(oracle, tmp_path, env):
    wt = tmp_path / "wt"
    (wt / "sub").mkdir(parents=True)
    files = {"a.txt": b"hello\n", "sub/b.bin": oracle.synth_bytes(5, 1, 70000), "empty": b""}
    for name, data in files.items():
        (wt / name).write_bytes(data)
    (wt / ".git" / "objects").mkdir(parents=True)
    p = run(["store"] + list(files), str(wt), env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = [l.split() for l in p.stdout.splitlines()]
    blobs = [l for l in lines if l[0] == "blob"]
    for (name, data), l in zip(files.items(), blobs):
        want = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
        assert l[1] == want and l[2] == ".git/objects/%s/%s" % (want[:2], want[2:])
        assert os.path.exists(os.path.join(str(wt), l[2]))
    tree = [l for l in lines if l[0] == "tree"][0][1]
    want_tree, nb, nt = oracle.write_tree(str(wt), None)
    assert tree == want_tree.hex() and nb == 3 and nt == 2
    assert [l for l in lines if l[0] == "emptytree"][0][1] == "4b825dc642cb6eb9a060e54bf8d69288fbee4904"
    commit = [l for l in lines if l[0] == "commit"][0]
    assert len(commit[1]) == 40 and commit[-1] == "ok"
    return p


@pytest.mark.gpu
def test_writers_of_the_reference_interface_on_the_gpu(oracle, tmp_path):
    """writeBlob / writeObject / writeTree / writeCommit / readObject through the free functions, every id from the
    device (VIT_HOST_CROSSOVER_BYTES=0: no call is too small for it)."""
    check_writers(oracle, tmp_path, {"VIT_HOST_CROSSOVER_BYTES": "0"})


def test_writers_of_the_reference_interface_below_the_small_work_bound(oracle, tmp_path):
    """The same calls with the store's default bound: a few objects are hashed by the store's own threads, the GPU
    library is not even loaded (the program does not link it), and everything printed and stored is the same."""
    p = check_writers(oracle, tmp_path, {"VIT_HASH_LIBRARY": "/nonexistent/libvit_hash.so"})
    assert "GPU" not in p.stderr


def test_worker_threads_do_not_pin_the_host_programs_descriptors(oracle, tmp_path):
    """The store's workers give themselves private descriptor tables, which start as copies of the process's: they must
    drop those copies, or a pipe the host program closes after a writeTree never reports end-of-file to its reader."""
    wt = tmp_path / "wt"
    wt.mkdir()
    for i in range(40):
        (wt / ("f%02d" % i)).write_bytes(oracle.synth_bytes(77, i, 3000 + i))
    (wt / ".git" / "objects").mkdir(parents=True)
    p = run(["pipe"], str(wt))
    assert p.returncode == 0 and "pipe eof" in p.stdout, p.stdout + p.stderr
    assert p.stdout.split()[1] == oracle.write_tree(str(wt), None)[0].hex()
