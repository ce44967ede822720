"""The object store below its small-work bound: while it has no GPU hasher, a call whose objects one host core
hashes sooner than the HIP runtime starts is hashed by the store's own threads (the `vit` program does not even
link the GPU library; it loads it when a batch is large enough).  Same ids, same stored files, same stdout as on
the device; and above the bound a machine without a gfx950 still fails, loudly.  These tests need no GPU: the
bound keeps the device out of it (reference callers: src/main.cpp:90-120 hash-object, :155-163 write-tree)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VIT = os.path.join(ROOT, "vit_amd", "bin", "vit")


def vit(cwd, *args, env=None, check=True):
    e = dict(os.environ)
    if env:
        e.update(env)
    p = subprocess.run([VIT] + list(args), cwd=cwd, capture_output=True, env=e, timeout=240)
    if check:
        assert p.returncode == 0, p.stderr.decode()
    return p


def fresh_copy(src, dst):
    shutil.copytree(src, dst, symlinks=True)
    vit(dst, "init")
    return str(dst)


def same_files(a, b):
    n = 0
    for d in os.listdir(b):
        for f in os.listdir(os.path.join(b, d)):
            assert open(os.path.join(a, d, f), "rb").read() == open(os.path.join(b, d, f), "rb").read(), d + f
            n += 1
    return n


def test_the_program_does_not_link_the_gpu_library():
    out = subprocess.run(["ldd", VIT], capture_output=True, text=True).stdout
    assert "libz" in out and "amdhip" not in out and "vit_hash" not in out, out


def test_small_worktrees_get_reference_ids_and_files_from_host_threads(oracle, golden_trees, worktrees, tmp_path):
    """Every golden worktree under the bound (the largest holds some 60 MB, which a machine without SHA extensions
    measures itself out of: the bound is set to 1 GB for it; the others are below the 8 MiB under which no bound is
    even worked out): `vit write-tree` with the GPU library made unreachable prints the reference's root, stores the
    files the oracle stores, and says who hashed."""
    for name, rec in golden_trees.items():
        hidden = {"VIT_HASH_LIBRARY": "/nonexistent/libvit_hash.so"}
        if rec.get("spec") == "skew":
            hidden["VIT_HOST_CROSSOVER_BYTES"] = "1000000000"
        wt = fresh_copy(worktrees[name], tmp_path / name)
        p = vit(wt, "write-tree", "--stats", env=hidden)
        assert p.stdout.decode() == rec["root"] + "\n", name
        err = p.stderr.decode()
        assert "ids-on host-threads" in err and " 0 ids from the device" in err, err
        ref_dir = tmp_path / ("oracle_" + name)
        oracle.write_tree(worktrees[name], str(ref_dir))
        assert same_files(os.path.join(wt, ".git", "objects"), str(ref_dir)) >= len(set(rec["blob_ids"])) + len(set(rec["tree_ids"]))
        assert vit(wt, "write-tree", "--git-order", env=hidden).stdout.decode() == rec["git_write_tree"] + "\n", name


def test_hash_object_and_commit_tree_need_no_device(oracle, golden_objects, tmp_path):
    wt = tmp_path / "ho"
    wt.mkdir()
    vit(str(wt), "init")
    g = golden_objects["blobs"]
    for L, want in g["ids"][::17] + [g["ids"][0]]:
        (wt / "f.bin").write_bytes(oracle.synth_bytes(g["seed"], L, L))
        assert vit(str(wt), "hash-object", "-w", "f.bin").stdout.decode() == want + "\n", L
        assert vit(str(wt), "cat-file", "-p", want, check=(L > 0)).stdout == oracle.synth_bytes(g["seed"], L, L)
    tree = "4b825dc642cb6eb9a060e54bf8d69288fbee4904"
    got = vit(str(wt), "commit-tree", tree, "-p", "", "-m", "m", env={"VIT_COMMIT_TIME": "7", "VIT_AUTHOR": "a", "VIT_EMAIL": "e"})
    text = oracle.commit_text(tree, "", "m", "a", "e", 7, 7)
    assert got.stdout.decode().strip() == oracle.object_id("commit", text).hex()


def test_above_the_bound_a_machine_without_a_gpu_fails_loudly(golden_trees, worktrees, tmp_path):
    """The bound is a scheduling policy, not a CPU path: a worktree above it needs the device, and where there is none
    (or no GPU library) write-tree fails with the reason, prints no id and stores no tree."""
    import torch
    wt = fresh_copy(worktrees["c1_1000x4096"], tmp_path / "big")
    p = vit(wt, "write-tree", env={"VIT_HOST_CROSSOVER_BYTES": "1000000", "VIT_HASH_LIBRARY": "/nonexistent/libvit_hash.so"}, check=False)
    assert p.returncode == 1 and p.stdout == b""
    assert b"cannot load the GPU hasher" in p.stderr and b"no CPU path" in p.stderr and b"Failed to write tree" in p.stderr
    if not torch.cuda.is_available():
        p = vit(wt, "write-tree", env={"VIT_HOST_CROSSOVER_BYTES": "1000000"}, check=False)
        assert p.returncode == 1 and p.stdout == b""
        assert b"cannot create the GPU hasher" in p.stderr and b"no CPU path" in p.stderr
        p = vit(wt, "hash-object", "-w", "d00/f0000.bin", env={"VIT_HOST_CROSSOVER_BYTES": "0"}, check=False)
        assert p.returncode == 1 and p.stdout == b""
    # and just under the bound it is the host threads' again
    p = vit(wt, "write-tree", "--stats", env={"VIT_HOST_CROSSOVER_BYTES": "5000000", "VIT_HASH_LIBRARY": "/nonexistent/libvit_hash.so"})
    assert p.stdout.decode().strip() == golden_trees["c1_1000x4096"]["root"] and b"(crossover 5000000 bytes" in p.stderr


def test_unreadable_file_is_hashed_as_the_empty_blob_by_product_and_oracle(tmp_path):
    """reference src/commit.cpp:151-153: a regular file that cannot be opened leaves the stringstream empty, and
    writeTree hashes and stores the empty string for it.  Product and oracle both follow that.  Root opens any file,
    so under root both run as another user (copies of the program and of the oracle's library in a directory that
    user can reach)."""
    import stat
    import tempfile
    top = tempfile.mkdtemp(prefix="vit_unreadable_", dir="/tmp")
    try:
        os.chmod(top, 0o777)
        wt = os.path.join(top, "wt")
        os.makedirs(os.path.join(wt, "d"))
        open(os.path.join(wt, "d", "secret"), "wb").write(b"cannot be read")
        open(os.path.join(wt, "plain"), "wb").write(b"plain\n")
        os.makedirs(os.path.join(wt, ".git", "objects"))
        exe = shutil.copy(VIT, os.path.join(top, "vit"))
        lib = shutil.copy(os.path.join(ROOT, "oracle", "liboracle.so"), os.path.join(top, "liboracle.so"))
        for base, dirs, files in os.walk(top):
            for x in dirs + files:
                os.chmod(os.path.join(base, x), 0o777)
        os.chmod(os.path.join(wt, "d", "secret"), 0)
        drop = (lambda: os.setuid(65534)) if os.geteuid() == 0 else None
        env = dict(os.environ, VIT_HASH_LIBRARY="/nonexistent/libvit_hash.so")
        p = subprocess.run([exe, "write-tree"], cwd=wt, capture_output=True, env=env, preexec_fn=drop, timeout=60)
        assert p.returncode == 0, p.stderr.decode()
        code = ("import ctypes as C; L = C.CDLL(%r); out = C.create_string_buffer(20); "
                "rc = L.vit_oracle_write_tree(b'.', None, out, None, None); print(rc, out.raw.hex())" % lib)
        q = subprocess.run(["/usr/bin/python3", "-c", code], cwd=wt, capture_output=True, preexec_fn=drop, timeout=60)
        rc, want = q.stdout.decode().split()
        assert rc == "0", q.stderr.decode()
        assert p.stdout.decode().strip() == want
        # ... and that tree names the empty blob for the file
        sub = subprocess.run([exe, "ls-tree", want, "-l"], cwd=wt, capture_output=True, env=env).stdout.decode()
        d_id = [l.split()[2] for l in sub.splitlines() if l.endswith("\td")][0]
        inner = subprocess.run([exe, "ls-tree", d_id, "-l"], cwd=wt, capture_output=True, env=env).stdout.decode()
        assert "e69de29bb2d1d6434b8b29ae775ad8c2e48c5391\tsecret" in inner
        os.chmod(os.path.join(wt, "d", "secret"), stat.S_IRUSR)
    finally:
        shutil.rmtree(top, ignore_errors=True)
