"""Row f4, hash-based change detection (`vit diff-tree`), against an ORACLE: oracle/change_oracle.py restates the
reference's content-comparing ChangeAnalyzer (src/utils/change_analyzer.cpp:40-103) and the two routines it is fed
by, which are pinned here by the reference's own code through tests/golden/change_detect.json (parseTree,
src/commit.cpp:175-200; FileUtils::getFilesInDirectory, src/utils/file_utils.cpp:55-80).  The product's ids come
from the store's host threads here (worktrees below its small-work bound, no GPU needed); tests/test_gpu_cli.py
runs the same comparison with every id from the device."""
import json
import os
import shutil
import stat
import subprocess

import pytest

from _change_cases import mutate, run_and_compare

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VIT = os.path.join(ROOT, "vit_amd", "bin", "vit")
HOST_SIDE = {"VIT_HASH_LIBRARY": "/nonexistent/libvit_hash.so", "VIT_HOST_CROSSOVER_BYTES": "4000000000"}


def vit(cwd, *args, check=True):
    p = subprocess.run([VIT] + list(args), cwd=cwd, capture_output=True, env=dict(os.environ, **HOST_SIDE), timeout=240)
    if check:
        assert p.returncode == 0, p.stderr.decode()
    return p


@pytest.fixture(scope="module")
def changes():
    from oracle import change_oracle
    return change_oracle


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "change_detect.json")))


def materialize_spec(spec, root, skip=()):
    for path, content in spec:
        if path in skip:
            continue
        full = os.path.join(root, path)
        if content is None:
            os.makedirs(full, exist_ok=True)
            continue
        os.makedirs(os.path.dirname(full) or root, exist_ok=True)
        if content.startswith("->"):
            os.symlink(content[2:], full)
        else:
            with open(full, "w") as f:
                f.write(content)


def test_tree_parsing_and_listing_are_the_references(oracle, changes, golden, tmp_path):
    """The fixture holds what the reference's writeTree, parseTree and getFilesInDirectory gave for this worktree."""
    wt = tmp_path / "wt"
    wt.mkdir()
    materialize_spec(golden["spec"], str(wt), skip=("dangling",))
    objects = tmp_path / "objects"
    root, _, _ = oracle.write_tree(str(wt), str(objects))
    assert root.hex() == golden["root"]
    for tree, entries in golden["trees"].items():
        got = changes.parse_tree(str(objects), tree)
        assert [[m.decode("latin-1"), n.decode("latin-1"), h, d] for m, n, h, d in got] == entries, tree
    os.symlink("nowhere", wt / "dangling")
    here = os.getcwd()
    os.chdir(wt)
    try:
        listed = sorted(p.decode("latin-1") for p in changes.files_in_directory("."))
    finally:
        os.chdir(here)
    assert listed == golden["files_in_directory"]
    # the last-space split, spelled out: the entry of "with space.txt" comes back as "space.txt" with mode "100644 with"
    top = {n: (m, d) for m, n, h, d in golden["trees"][golden["root"]]}
    assert top["space.txt"] == ("100644 with", False) and top["here"] == ("100644 two  spaces", False)
    if oracle.ref_lib() is not None and hasattr(oracle.ref_lib(), "vitref_parse_tree"):      # live, where the reference is built
        os.makedirs(wt / ".git", exist_ok=True)
        shutil.copytree(objects, wt / ".git" / "objects")
        os.chdir(wt)
        try:
            for tree in golden["trees"]:
                assert oracle.ref_parse_tree(tree) == changes.parse_tree(str(objects), tree)
            assert sorted(oracle.ref_files_in_directory(".")) == sorted(changes.files_in_directory("."))
        finally:
            os.chdir(here)


def test_diff_tree_on_the_corner_case_worktree(oracle, changes, golden, tmp_path):
    """Every quirk at once, product against oracle: names with spaces (committed path "deleted", real file "added"),
    build/ and node_modules/ never listed but present in the tree (so "deleted"), a directory symlink hashed by
    write-tree but not descended into by the listing, a backslash normalized to a slash, an unreadable file."""
    wt = tmp_path / "wt"
    wt.mkdir()
    materialize_spec(golden["spec"], str(wt), skip=("dangling",))
    vit(str(wt), "init")
    root = vit(str(wt), "write-tree").stdout.decode().strip()
    assert root == golden["root"]
    got = run_and_compare(vit, changes, str(wt), root)
    assert "space.txt" in got["D"] and "with space.txt" in got["A"] and "build/out.o" in got["D"]
    assert "link-to-dir/deep file.md" not in got["A"] and "link-to-dir/c/leaf" in got["D"] and "back\\slash" in got["D"]
    assert not got["M"]
    with open(wt / "top.txt", "a") as f:
        f.write("changed\n")
    os.remove(wt / "a" / "one.c")                          # link-to-file now dangles: no longer a regular file
    os.chmod(wt / "sub" / "kept", 0)                       # (root opens it all the same: then it is simply unchanged)
    got = run_and_compare(vit, changes, str(wt), root)
    assert "top.txt" in got["M"] and {"a/one.c", "link-to-file"} <= got["D"]
    assert ("sub/kept" not in got["M"]) and ("sub/kept" not in got["D"])
    os.chmod(wt / "sub" / "kept", stat.S_IRUSR | stat.S_IWUSR)


def test_diff_tree_after_500_random_changes_matches_the_oracle(oracle, changes, golden_trees, worktrees, tmp_path):
    """The 10,000-file skewed golden worktree, stored, then >= 500 random edits, deletions and additions, one identical
    rewrite, one emptied file, one name with a space: the product's id comparison gives the reference's records."""
    name = [k for k in golden_trees if golden_trees[k].get("spec") == "skew"][0]
    wt = str(tmp_path / "wt")
    shutil.copytree(worktrees[name], wt)
    open(os.path.join(wt, "t00", "a name with spaces.txt"), "wb").write(b"spaced\n")
    vit(wt, "init")
    root = vit(wt, "write-tree").stdout.decode().strip()
    files = sorted(os.path.relpath(os.path.join(d, f), wt) for d, _, fs in os.walk(wt) for f in fs
                   if ".git" not in d and " " not in f)
    assert len(files) == golden_trees[name]["n"]
    expect = mutate(wt, files)
    got = run_and_compare(vit, changes, wt, root, expect)
    assert len(got["M"]) + len(got["D"]) + len(got["A"]) >= 500
    assert "t00/spaces.txt" in got["D"] and "t00/a name with spaces.txt" in got["A"]
    # through a commit object as well
    commit = subprocess.run([VIT, "commit-tree", root, "-p", "", "-m", "m"], cwd=wt, capture_output=True,
                            env=dict(os.environ, VIT_COMMIT_TIME="1", **HOST_SIDE)).stdout.decode().strip()
    assert vit(wt, "diff-tree", commit).stdout == vit(wt, "diff-tree", root).stdout
