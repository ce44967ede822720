"""Shared by the change-detection tests (CPU: ids from the store's host threads; GPU: ids from the device):
mutate a copy of a stored worktree and compare `vit diff-tree` with the oracle's restatement of the reference's
content-comparing ChangeAnalyzer (oracle/change_oracle.py; reference src/utils/change_analyzer.cpp:40-103)."""
import os

import numpy as np


def mutate(wt, files, seed=11, edits=260, deletes=160, adds=120):
    """>= 500 random changes to the files listed (relative paths), plus the corner cases the reference's code has:
    a file rewritten with identical bytes (no change), a name with a space (its tree entry is parsed back wrongly,
    src/commit.cpp:187: the committed path shows as deleted and the real file as added), a file replaced by an empty
    one, a new file under build/ (never listed), a file that becomes unreadable in place."""
    rng = np.random.default_rng(seed)
    pick = rng.permutation(len(files))
    todo = [files[i] for i in pick[:edits + deletes + 2]]
    expect = {"M": set(), "D": set(), "A": set()}
    for k, rel in enumerate(todo[:edits]):
        p = os.path.join(wt, rel)
        data = open(p, "rb").read()
        how = k % 4
        if how == 0:
            data = data + b"+"                              # grows by a byte
        elif how == 1 and len(data) > 0:
            data = data[:-1]                                # shrinks by a byte
        elif how == 2 and len(data) > 0:
            i = int(rng.integers(0, len(data)))
            data = data[:i] + bytes([data[i] ^ 0x40]) + data[i + 1:]     # one bit, same length
        else:
            data = b"replaced %d\n" % k
        open(p, "wb").write(data)
        expect["M"].add(rel)
    for rel in todo[edits:edits + deletes]:
        os.remove(os.path.join(wt, rel))
        expect["D"].add(rel)
    same = todo[edits + deletes]
    data = open(os.path.join(wt, same), "rb").read()
    open(os.path.join(wt, same), "wb").write(data)          # identical rewrite: new inode and mtime, same content
    emptied = todo[edits + deletes + 1]
    was_empty = os.path.getsize(os.path.join(wt, emptied)) == 0
    open(os.path.join(wt, emptied), "wb").close()
    if not was_empty:
        expect["M"].add(emptied)
    for k in range(adds):
        rel = "added/n%02d/new%03d.txt" % (k % 7, k)
        os.makedirs(os.path.dirname(os.path.join(wt, rel)), exist_ok=True)
        open(os.path.join(wt, rel), "wb").write(b"new file %d\n" % k)
        expect["A"].add(rel)
    os.makedirs(os.path.join(wt, "build", "x"), exist_ok=True)
    open(os.path.join(wt, "build", "x", "skipped.o"), "wb").write(b"never listed")
    return expect


def run_and_compare(vit, oracle_changes, wt, root, expect=None):
    """`vit diff-tree root` in wt against the oracle: modified and deleted sets equal the reference's records; added =
    the files the reference builds a record for and drops (the product's documented extension); same total."""
    p = vit(wt, "diff-tree", root)
    got = {"M": set(), "D": set(), "A": set()}
    for line in p.stdout.decode("latin-1").splitlines():
        kind, path = line.split("\t", 1)
        got[kind].add(path)
    want = oracle_changes.analyze_changes(os.path.join(wt, ".git", "objects"), root, wt)
    dec = lambda s: {x.decode("latin-1") for x in s}
    assert got["M"] == dec(want["modified"]), (sorted(got["M"] ^ dec(want["modified"]))[:10])
    assert got["D"] == dec(want["deleted"]), (sorted(got["D"] ^ dec(want["deleted"]))[:10])
    assert got["A"] == dec(want["added_then_dropped"]), (sorted(got["A"] ^ dec(want["added_then_dropped"]))[:10])
    changed = len(got["M"]) + len(got["D"]) + len(got["A"])
    assert ("%d files analyzed, %d changed" % (want["total"], changed)).encode() in p.stderr, p.stderr[-300:]
    if expect is not None:
        for k in "MDA":
            assert expect[k] <= got[k], (k, sorted(expect[k] - got[k])[:10])
    return got
