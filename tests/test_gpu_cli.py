"""The C++ object store and the `vit` front end (vit_amd/bin/vit) on a GPU:
same stdout, ids, stored objects and failure behaviour as the reference's
hash-object / write-tree / commit-tree (src/main.cpp:90-186, src/commit.cpp:88-173)."""
import hashlib
import os
import shutil
import subprocess
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VIT = os.path.join(ROOT, "vit_amd", "bin", "vit")


def vit(cwd, *args, env=None, check=True, nofile=None):
    e = dict(os.environ)
    e.setdefault("VIT_DEVICES", "0")
    # these tests are about the device path: no call is too small for it (the store's small-work bound, which keeps
    # worktrees of this size on host threads by default, has tests/test_store_small_work.py and one test below)
    e.setdefault("VIT_HOST_CROSSOVER_BYTES", "0")
    if env:
        e.update(env)
    limit = None
    if nofile is not None:
        import resource
        limit = lambda: resource.setrlimit(resource.RLIMIT_NOFILE, (nofile, nofile))
    p = subprocess.run([VIT] + list(args), cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=240,
                       preexec_fn=limit)
    if check:
        assert p.returncode == 0, p.stderr.decode()
    return p


def stored_objects(wt):
    out = {}
    base = os.path.join(wt, ".git", "objects")
    for d in os.listdir(base):
        for f in os.listdir(os.path.join(base, d)):
            raw = zlib.decompress(open(os.path.join(base, d, f), "rb").read())
            assert hashlib.sha1(raw).hexdigest() == d + f, "object file name is not the SHA-1 of its content"
            out[d + f] = raw
    return out


def fresh_copy(src, dst):
    shutil.copytree(src, dst, symlinks=True)
    vit(dst, "init")
    return str(dst)


def test_binary_exists():
    assert os.path.exists(VIT), "build vit_amd/bin/vit with `python -m vit_amd.build`"


def test_both_sides_of_the_small_work_bound(golden_trees, worktrees, tmp_path):
    """Default bound: the 4 MB worktree of BASELINE configs[0] never starts the HIP runtime; a bound below its size
    sends the same worktree to the device.  Same root, and --stats says which side made the ids."""
    rec = golden_trees["c1_1000x4096"]
    wt = fresh_copy(worktrees["c1_1000x4096"], tmp_path / "host")
    p = vit(wt, "write-tree", "--stats", env={"VIT_HOST_CROSSOVER_BYTES": ""})
    assert p.stdout.decode().strip() == rec["root"]
    assert "ids-on host-threads" in p.stderr.decode() and " 0 ids from the device" in p.stderr.decode()
    wt = fresh_copy(worktrees["c1_1000x4096"], tmp_path / "gpu")
    p = vit(wt, "write-tree", "--stats", env={"VIT_HOST_CROSSOVER_BYTES": "1000000"})
    assert p.stdout.decode().strip() == rec["root"]
    assert "ids-on gpu" in p.stderr.decode() and " 0 from host threads" in p.stderr.decode()
    assert "1011 ids from the device" in p.stderr.decode(), p.stderr.decode()


def test_write_tree_matches_reference_on_every_golden_worktree(golden_trees, worktrees, tmp_path):
    for name, rec in golden_trees.items():
        wt = fresh_copy(worktrees[name], tmp_path / name)
        p = vit(wt, "write-tree")
        assert p.stdout.decode() == rec["root"] + "\n", name
        objs = stored_objects(wt)
        assert set(rec["tree_ids"]) <= set(objs), name
        assert set(rec["blob_ids"]) <= set(objs), name
        assert sum(1 for v in objs.values() if v.startswith(b"blob ")) == rec["n_blobs"], name
        # a second run rewrites the same objects and prints the same id (the reference always rewrites, :110)
        assert vit(wt, "write-tree").stdout.decode() == rec["root"] + "\n"


def test_stored_objects_are_byte_identical_to_the_oracles(oracle, golden_trees, worktrees, tmp_path):
    """Same zlib, same level, same framing => the loose-object files themselves match
    what the reference's writeObject stores (src/commit.cpp:98-111)."""
    wt = fresh_copy(worktrees["nested_names"], tmp_path / "wt")
    vit(wt, "write-tree")
    ref_dir = tmp_path / "oracle_objects"
    oracle.write_tree(worktrees["nested_names"], str(ref_dir))
    n = 0
    for d in os.listdir(ref_dir):
        for f in os.listdir(ref_dir / d):
            mine = os.path.join(wt, ".git", "objects", d, f)
            assert open(mine, "rb").read() == open(ref_dir / d / f, "rb").read(), d + f
            n += 1
    assert n >= len(golden_trees["nested_names"]["tree_ids"]) + len(golden_trees["nested_names"]["blob_ids"])


def test_device_deflate_stores_the_files_the_library_would(oracle, golden_trees, worktrees, tmp_path):
    """The streams the GPU lanes write (forced onto small batches here; by default only batches of 4096 small objects
    and more take that way) are the zlib library's: every loose-object file of a worktree equals the oracle's, for the
    10,000-file skewed worktree (objects above the device's size limit go to host threads within the same batch), the
    1,000-file one and the one with odd names; and with the device path off the files are the same again."""
    force = {"VIT_DEVICE_DEFLATE": "1", "VIT_DEVICE_DEFLATE_MIN": "1"}
    for name in ("nested_names", "c1_1000x4096", [k for k in golden_trees if golden_trees[k].get("spec") == "skew"][0]):
        wt = fresh_copy(worktrees[name], tmp_path / ("dd_" + name))
        p = vit(wt, "write-tree", "--stats", env=force)
        assert p.stdout.decode().strip() == golden_trees[name]["root"]
        stats = p.stderr.decode()
        done = int(stats.split(" s (")[1].split(" objects)")[0])
        assert done > 0, stats                                         # the device really made streams
        ref_dir = tmp_path / ("oracle_" + name)
        oracle.write_tree(worktrees[name], str(ref_dir))
        n = 0
        for d in os.listdir(ref_dir):
            for f in os.listdir(ref_dir / d):
                assert open(os.path.join(wt, ".git", "objects", d, f), "rb").read() == open(ref_dir / d / f, "rb").read(), d + f
                n += 1
        assert n >= len(set(golden_trees[name]["blob_ids"]))
        host = fresh_copy(worktrees[name], tmp_path / ("hd_" + name))
        q = vit(host, "write-tree", "--stats", "--host-deflate")
        assert " s (0 objects)" in q.stderr.decode()
        assert stored_objects(host).keys() == stored_objects(wt).keys()
        # five sixteenths of every batch on the device, the rest on host threads: the same files again
        part = fresh_copy(worktrees[name], tmp_path / ("pd_" + name))
        r = vit(part, "write-tree", "--stats", env=dict(force, VIT_DEVICE_SHARE="5"))
        some = int(r.stderr.decode().split(" s (")[1].split(" objects)")[0])
        assert 0 < some < done, (some, done)
        for d in os.listdir(ref_dir):
            for f in os.listdir(ref_dir / d):
                assert open(os.path.join(part, ".git", "objects", d, f), "rb").read() == open(ref_dir / d / f, "rb").read(), d + f


def test_device_deflate_failing_leaves_the_work_to_the_host(golden_trees, worktrees, tmp_path):
    """vit_deflate_objects failing (injected) costs nothing but a message, once: the batch and those after it are deflated
    by host threads, ids and stored objects as ever."""
    name = [k for k in golden_trees if golden_trees[k].get("spec") == "skew"][0]
    wt = fresh_copy(worktrees[name], tmp_path / "ddfail")
    p = vit(wt, "write-tree", "--stats", env={"VIT_DEVICE_DEFLATE": "1", "VIT_DEVICE_DEFLATE_MIN": "1", "VIT_BATCH_MB": "8",
                                              "VIT_DEBUG_FAIL_DEFLATE": "1"})
    assert p.stdout.decode().strip() == golden_trees[name]["root"]
    err = p.stderr.decode()
    assert err.count("deflate on the device failed") == 1 and " s (0 objects)" in err, err
    objs = stored_objects(wt)
    assert set(golden_trees[name]["blob_ids"]) <= set(objs) and set(golden_trees[name]["tree_ids"]) <= set(objs)


def test_device_deflate_that_differs_from_this_machines_zlib_is_dropped(oracle, golden_trees, worktrees, tmp_path):
    """The kernel restates zlib 1.2.11; the host threads of the same write-tree call whatever libz is linked.  The first
    batch the device makes is compared with the linked library on a sample that includes its first stream; here the
    library flips one byte in the middle of that stream (debug_corrupt_deflate: the Adler-32 trailer stays intact, as
    after a fault in the deflate itself).  One message, the device's streams dropped, every stored file the oracle's."""
    name = [k for k in golden_trees if golden_trees[k].get("spec") == "skew"][0]
    wt = fresh_copy(worktrees[name], tmp_path / "ddcorrupt")
    p = vit(wt, "write-tree", "--stats", env={"VIT_DEVICE_DEFLATE": "1", "VIT_DEVICE_DEFLATE_MIN": "1", "VIT_BATCH_MB": "8",
                                              "VIT_DEBUG_CORRUPT_DEFLATE": "1"})
    assert p.stdout.decode().strip() == golden_trees[name]["root"]
    err = p.stderr.decode()
    assert err.count("does not give the bytes of this machine's zlib") == 1 and " s (0 objects)" in err, err
    assert "zlib %s" % zlib.ZLIB_RUNTIME_VERSION in err
    ref_dir = tmp_path / "oracle_objects"
    oracle.write_tree(worktrees[name], str(ref_dir))
    n = 0
    for d in os.listdir(ref_dir):
        for f in os.listdir(ref_dir / d):
            assert open(os.path.join(wt, ".git", "objects", d, f), "rb").read() == open(ref_dir / d / f, "rb").read(), d + f
            n += 1
    assert n >= len(set(golden_trees[name]["blob_ids"]))


def test_stream_with_a_bad_trailer_is_deflated_again_by_the_host(oracle, golden_trees, worktrees, tmp_path):
    """Beyond the first batch's comparison, every device-made stream is looked at before it becomes a file that is
    never written again: length within zlib's bound, zlib's header bytes, a trailer equal to the Adler-32 of header +
    content.  VIT_DEBUG_BAD_TRAILER makes the store see a wrong trailer on every 100th stream: those objects are
    deflated by the host thread instead, counted, and the files are the oracle's."""
    name = "c1_1000x4096"
    wt = fresh_copy(worktrees[name], tmp_path / "badtrailer")
    p = vit(wt, "write-tree", "--stats", env={"VIT_DEVICE_DEFLATE": "1", "VIT_DEVICE_DEFLATE_MIN": "1", "VIT_DEBUG_BAD_TRAILER": "100"})
    assert p.stdout.decode().strip() == golden_trees[name]["root"]
    rejected = int(p.stderr.decode().split("device-streams-rejected ")[1].split()[0])
    assert 5 <= rejected <= 20, p.stderr.decode()
    ref_dir = tmp_path / "oracle_objects"
    oracle.write_tree(worktrees[name], str(ref_dir))
    for d in os.listdir(ref_dir):
        for f in os.listdir(ref_dir / d):
            assert open(os.path.join(wt, ".git", "objects", d, f), "rb").read() == open(ref_dir / d / f, "rb").read(), d + f


def test_write_tree_with_fewer_descriptors_than_fanout_directories(golden_trees, worktrees, tmp_path):
    """The storing threads keep a handle per objects/xx directory; where the descriptor limit is below the 256 of
    them (plus what the runtime holds), they give handles back and carry on: same id, every object stored."""
    name = [k for k in golden_trees if golden_trees[k].get("spec") == "skew"][0]
    wt = fresh_copy(worktrees[name], tmp_path / "fewfds")
    p = vit(wt, "write-tree", nofile=192)
    assert p.stdout.decode().strip() == golden_trees[name]["root"], p.stderr.decode()
    objs = stored_objects(wt)
    assert set(golden_trees[name]["blob_ids"]) <= set(objs) and set(golden_trees[name]["tree_ids"]) <= set(objs)


def test_git_reads_what_write_tree_stored(golden_trees, worktrees, tmp_path):
    if not shutil.which("git"):
        pytest.skip("no git here")
    wt = fresh_copy(worktrees["c1_1000x4096"], tmp_path / "c1")
    root = vit(wt, "write-tree").stdout.decode().strip()
    assert root == golden_trees["c1_1000x4096"]["git_write_tree"]       # on this worktree reference == git
    out = subprocess.check_output(["git", "-C", wt, "ls-tree", root]).decode().splitlines()
    assert len(out) == 10 and all(l.startswith("040000 tree ") for l in out)
    some_blob = golden_trees["c1_1000x4096"]["blob_ids"][123]
    assert subprocess.check_output(["git", "-C", wt, "cat-file", "-t", some_blob]).decode().strip() == "blob"


def test_hash_object_prints_the_reference_ids(oracle, golden_objects, tmp_path):
    wt = tmp_path / "ho"
    wt.mkdir()
    vit(str(wt), "init")
    g = golden_objects["blobs"]
    for L, want in g["ids"][::9] + [g["ids"][0]]:
        (wt / "f.bin").write_bytes(oracle.synth_bytes(g["seed"], L, L))
        assert vit(str(wt), "hash-object", "-w", "f.bin").stdout.decode() == want + "\n", L
        assert vit(str(wt), "cat-file", "-p", want, check=(L > 0)).stdout == oracle.synth_bytes(g["seed"], L, L)
    p = vit(str(wt), "hash-object", "-w", "missing", check=False)
    assert p.returncode == 1 and b"Failed to open file" in p.stderr
    assert vit(str(wt), "hash-object", "f.bin", check=False).returncode == 1


def test_commit_tree_with_pinned_clock(oracle, tmp_path):
    wt = tmp_path / "ct"
    wt.mkdir()
    vit(str(wt), "init")
    (wt / ".vitconfig").write_text("1\nalice\nalice@example.org\n")
    tree = "4b825dc642cb6eb9a060e54bf8d69288fbee4904"
    parent = "ce013625030ba8dba906f756967f9e9ca394464a"
    got = vit(str(wt), "commit-tree", tree, "-p", parent, "-m", "first", env={"VIT_COMMIT_TIME": "1700000000"}).stdout.decode().strip()
    text = oracle.commit_text(tree, parent, "first", "alice", "alice@example.org", 1700000000, 1700000000)
    assert got == oracle.object_id("commit", text).hex()
    assert vit(str(wt), "cat-file", "-p", got).stdout == text


def test_ls_tree_lists_sorted_entries(golden_trees, worktrees, tmp_path):
    wt = fresh_copy(worktrees["dir_vs_file_prefix"], tmp_path / "ls")
    root = vit(wt, "write-tree").stdout.decode().strip()
    names = vit(wt, "ls-tree", root, "--name-only").stdout.decode().split()
    assert names == ["foo", "foo-bar", "foo.c"]          # the reference's plain-name order (src/commit.cpp:164-165)
    long = vit(wt, "ls-tree", root, "-l").stdout.decode().splitlines()
    assert long[0].startswith("40000 tree ") and long[1].startswith("100644 blob ")


def test_write_tree_fails_like_the_reference_on_special_files(tmp_path):
    wt = tmp_path / "bad"
    (wt / "sub").mkdir(parents=True)
    (wt / "sub" / "ok.txt").write_text("x")
    os.mkfifo(wt / "sub" / "pipe")                       # neither regular file nor directory (src/commit.cpp:160)
    vit(str(wt), "init")
    p = vit(str(wt), "write-tree", check=False)
    assert p.returncode == 1 and p.stdout == b"" and b"Failed to write tree" in p.stderr


def test_symlinks_are_followed_and_dot_git_skipped_at_every_level(oracle, tmp_path):
    wt = tmp_path / "sl"
    (wt / "real").mkdir(parents=True)
    (wt / "real" / "a").write_bytes(b"A" * 100)
    (wt / "real" / ".git").mkdir()                       # a nested directory literally named .git is skipped (:145)
    (wt / "real" / ".git" / "junk").write_text("junk")
    os.symlink("real", wt / "link")                      # followed: hashed as a second copy of the directory
    os.symlink("real/a", wt / "alink")
    vit(str(wt), "init")
    got = vit(str(wt), "write-tree").stdout.decode().strip()
    assert got == oracle.write_tree(str(wt))[0].hex()


def test_diff_tree_finds_changes_by_id(oracle, golden_trees, worktrees, tmp_path):
    """Hash-based change detection: same verdicts as comparing contents
    (reference src/utils/change_analyzer.cpp:56-98), plus the added files."""
    wt = fresh_copy(worktrees["nested_names"], tmp_path / "dt")
    # The reference's parseTree splits "mode name" at the LAST space (src/commit.cpp:187), so a name with a
    # space is read back wrongly; that quirk is kept, and this test stays clear of it.
    os.remove(os.path.join(wt, "a", "b", "x y.txt"))
    root = vit(wt, "write-tree").stdout.decode().strip()
    p = vit(wt, "diff-tree", root)
    assert p.stdout == b"" and b"6 files analyzed, 0 changed" in p.stderr
    with open(os.path.join(wt, "a", "z"), "ab") as f:           # modified
        f.write(b"!")
    os.remove(os.path.join(wt, "a", "b", "c", "deep.bin"))      # deleted
    os.makedirs(os.path.join(wt, "new", "dir"))
    open(os.path.join(wt, "new", "dir", "n.txt"), "wb").write(b"new")        # added
    os.makedirs(os.path.join(wt, "build"))
    open(os.path.join(wt, "build", "ignored.o"), "wb").write(b"x")           # skipped like the reference does
    same = open(os.path.join(wt, "Z"), "rb").read()
    open(os.path.join(wt, "Z"), "wb").write(same)               # rewritten with identical bytes: not a change
    p = vit(wt, "diff-tree", root)
    assert p.stdout.decode().splitlines() == ["D\ta/b/c/deep.bin", "M\ta/z", "A\tnew/dir/n.txt"]
    # through a commit object as well
    commit = vit(wt, "commit-tree", root, "-p", "", "-m", "m", env={"VIT_COMMIT_TIME": "1"}).stdout.decode().strip()
    assert vit(wt, "diff-tree", commit).stdout == p.stdout
    assert vit(wt, "diff-tree", "0" * 40, check=False).returncode == 1


def test_diff_tree_after_500_random_changes_matches_the_oracle_with_ids_from_the_device(golden_trees, worktrees, tmp_path):
    """Row f4 against an oracle: oracle/change_oracle.py restates the reference's ChangeAnalyzer, which inflates every
    committed blob and compares CONTENTS (src/utils/change_analyzer.cpp:85-88); `vit diff-tree` compares the ids of one
    GPU batch over the worktree with the ids in the tree.  Same modified and deleted records after >= 500 random edits,
    deletions and additions to the 10,000-file skewed worktree, an identical rewrite, an emptied file and a name with
    a space (tests/_change_cases.py); added files = the records the reference builds and drops (:62-72)."""
    from oracle import change_oracle
    from _change_cases import mutate, run_and_compare
    name = [k for k in golden_trees if golden_trees[k].get("spec") == "skew"][0]
    wt = str(tmp_path / "wt")
    shutil.copytree(worktrees[name], wt)
    open(os.path.join(wt, "t00", "a name with spaces.txt"), "wb").write(b"spaced\n")
    vit(wt, "init")
    p = vit(wt, "write-tree", "--stats")
    assert "ids-on gpu" in p.stderr.decode()
    root = p.stdout.decode().strip()
    files = sorted(os.path.relpath(os.path.join(d, f), wt) for d, _, fs in os.walk(wt) for f in fs
                   if ".git" not in d and " " not in f)
    expect = mutate(wt, files, seed=12)
    got = run_and_compare(vit, change_oracle, wt, root, expect)
    assert len(got["M"]) + len(got["D"]) + len(got["A"]) >= 500
    assert "t00/spaces.txt" in got["D"] and "t00/a name with spaces.txt" in got["A"]


def test_medium_worktree_skewed_sizes(oracle, tmp_path):
    """~6,000 files with a heavy small-file skew and a few MiB-sized ones, 3 levels deep."""
    rng = np.random.default_rng(3)
    wt = tmp_path / "skew"
    sizes = np.minimum((np.exp(rng.normal(np.log(4500), 1.55, 6000))).astype(np.int64), 4 << 20)
    sizes[:3] = [0, 3 << 20, 1]
    for i, L in enumerate(sizes):
        d = wt / ("m%02d" % (i % 37)) / ("s%d" % (i % 5))
        d.mkdir(parents=True, exist_ok=True)
        (d / ("f%05d.dat" % i)).write_bytes(oracle.synth_bytes(0x5EED0003, i, int(L)))
    vit(str(wt), "init")
    p = vit(str(wt), "write-tree", "--stats")
    want, n_blobs, n_trees = oracle.write_tree(str(wt))
    assert p.stdout.decode().strip() == want.hex()
    assert ("blobs %d trees %d" % (n_blobs, n_trees)).encode() in p.stderr


def test_git_order_gives_gits_own_tree_ids(golden_trees, worktrees, tmp_path):
    """`write-tree --git-order` applies Git's rules where the reference's differ (directories sort as
    "name/", empty directories dropped, exec bit kept): on every golden worktree the id is the one
    `git add -A && git write-tree` printed when the fixtures were generated, and without the flag the
    reference's id stays."""
    for name, rec in golden_trees.items():
        wt = fresh_copy(worktrees[name], tmp_path / name)
        assert vit(wt, "write-tree", "--git-order").stdout.decode() == rec["git_write_tree"] + "\n", name
        assert vit(wt, "write-tree").stdout.decode() == rec["root"] + "\n", name


def test_git_order_with_symlinks_and_special_files_matches_live_git(oracle, tmp_path):
    """Symlinks become 120000 entries holding the target text, a fifo is passed over, a directory
    holding only empty directories vanishes: compared with git itself on the same worktree."""
    if not shutil.which("git"):
        pytest.skip("no git here")
    wt = tmp_path / "wt"
    (wt / "src" / "deep" / "hollow" / "hollower").mkdir(parents=True)
    (wt / "src" / "main.c").write_bytes(oracle.synth_bytes(61, 1, 5000))
    (wt / "src.old").write_bytes(oracle.synth_bytes(61, 2, 10))
    (wt / "src-x").write_bytes(b"")
    (wt / "tool.sh").write_bytes(b"#!/bin/sh\n")
    os.chmod(wt / "tool.sh", 0o755)
    os.symlink("src/main.c", wt / "link-to-file")
    os.symlink("src", wt / "link-to-dir")
    os.symlink("nowhere", wt / "src" / "dangling")
    os.mkfifo(wt / "pipe")
    vit(str(wt), "init")
    mine = vit(str(wt), "write-tree", "--git-order").stdout.decode().strip()
    objs = stored_objects(str(wt))
    shutil.rmtree(wt / ".git")
    env = dict(os.environ, GIT_CONFIG_NOSYSTEM="1", HOME=str(tmp_path))
    subprocess.check_call(["git", "init", "-q", "."], cwd=wt, env=env)
    subprocess.check_call(["git", "add", "-A"], cwd=wt, env=env)
    theirs = subprocess.check_output(["git", "write-tree"], cwd=wt, env=env).decode().strip()
    assert mine == theirs
    listing = subprocess.check_output(["git", "ls-tree", "-r", "-t", theirs], cwd=wt, env=env).decode()
    for line in listing.splitlines():                     # every object git names is one we stored
        assert line.split()[2] in objs, line
    assert "120000" in listing and "100755" in listing and "hollow" not in listing


def test_write_tree_flags_no_store_and_rewrite(golden_trees, worktrees, tmp_path):
    """--no-store prints the id and writes nothing; a plain run then stores; a second plain run skips what
    is there (already-stored = all of it); --rewrite writes every object again, as the reference always does."""
    rec = golden_trees["nested_names"]
    wt = fresh_copy(worktrees["nested_names"], tmp_path / "wt")
    assert vit(wt, "write-tree", "--no-store").stdout.decode() == rec["root"] + "\n"
    assert stored_objects(wt) == {}
    assert vit(wt, "write-tree").stdout.decode() == rec["root"] + "\n"
    objs = stored_objects(wt)
    total = len(objs)
    assert total == len(set(rec["tree_ids"]) | set(rec["blob_ids"]))
    some = os.path.join(wt, ".git", "objects", rec["root"][:2], rec["root"][2:])
    first = os.stat(some).st_ino
    p = vit(wt, "write-tree", "--stats")
    assert "already-stored %d" % (rec["n_blobs"] + len(rec["tree_ids"])) in p.stderr.decode() or \
        "already-stored %d" % total in p.stderr.decode()
    assert os.stat(some).st_ino == first                  # untouched
    p = vit(wt, "write-tree", "--rewrite", "--stats")
    assert p.stdout.decode() == rec["root"] + "\n" and "already-stored 0" in p.stderr.decode()
    assert os.stat(some).st_ino != first                  # a new file was renamed into place
    assert stored_objects(wt) == objs


def test_other_deflate_levels_store_the_same_objects(golden_trees, worktrees, tmp_path):
    """--deflate-level: files differ from the reference's in their bytes, not in what they hold: same ids, every
    stored object inflates to "<type> <len>\\0" + content (stored_objects checks name == SHA-1 of that), git reads them."""
    name = "c1_1000x4096"
    rec = golden_trees[name]
    ref = fresh_copy(worktrees[name], tmp_path / "ref")
    assert vit(ref, "write-tree").stdout.decode().strip() == rec["root"]
    want = stored_objects(ref)
    sizes = {}
    for level in (0, 1, 9):
        wt = fresh_copy(worktrees[name], tmp_path / ("l%d" % level))
        assert vit(wt, "write-tree", "--deflate-level=%d" % level).stdout.decode().strip() == rec["root"]
        got = stored_objects(wt)
        assert got == want
        base = os.path.join(wt, ".git", "objects")
        sizes[level] = sum(os.path.getsize(os.path.join(base, d, f)) for d in os.listdir(base) for f in os.listdir(os.path.join(base, d)))
        if shutil.which("git"):
            blob = rec["blob_ids"][0]
            out = subprocess.run(["git", "--git-dir", os.path.join(wt, ".git"), "cat-file", "-t", blob], capture_output=True, text=True)
            assert out.stdout.strip() == "blob", out.stderr
    assert sizes[0] >= sizes[1] >= sizes[9] * 0.99
