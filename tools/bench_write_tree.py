#!/usr/bin/env python3
"""write-tree on synthetic worktrees: the GPU-backed `vit write-tree` next to the
CPU path (the oracle's restatement of the reference's writeTree, and the
reference's own code when oracle/_ref was shipped with the tree).

    python tools/bench_write_tree.py c1            # 1,000 x 4 KiB in 10 dirs   (BASELINE configs[0])
    python tools/bench_write_tree.py c3            # ~80k files, ~1.2 GB, lognormal sizes, 3 levels (configs[2])
    python tools/bench_write_tree.py c3 --files 20000

Measurement tool (uses the oracle as the checker); prints one JSON line.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402

VIT = os.path.join(ROOT, "vit_amd", "bin", "vit")


def make_c1(root):
    n = 1000
    blob = O.synth_fill_many(0x5EED0001, 0, n, 4096, threads=8)
    for i in range(n):
        d = os.path.join(root, "d%02d" % (i // 100))
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "f%04d.bin" % i), "wb") as f:
            f.write(blob[i * 4096:(i + 1) * 4096].tobytes())
    return n, n * 4096


def _c2_part(args):
    root, lo, hi = args
    step = 4096
    for base in range(lo, hi, step):
        top = min(base + step, hi)
        blob = O.synth_fill_many(0x5EED0002, base, top - base, 4096, threads=1)
        for i in range(base, top):
            d = os.path.join(root, "d%03d" % (i // 1000))
            if i % 1000 == 0 or i == lo:
                os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "f%07d.bin" % i), "wb") as f:
                f.write(blob[(i - base) * 4096:(i - base + 1) * 4096].tobytes())
    return hi - lo


def make_c2(root, n_files):
    """The headline worktree: n files x 4,096 B (blob i = synthetic stream i of seed 0x5EED0002, the same
    blobs bench.py hashes), 1,000 per directory; on this layout the reference's tree id is also git's."""
    import multiprocessing as mp
    procs = 8
    cuts = [n_files * k // procs // 1000 * 1000 for k in range(procs)] + [n_files]
    with mp.Pool(procs) as pool:
        done = 0
        for part in pool.imap_unordered(_c2_part, [(root, cuts[k], cuts[k + 1]) for k in range(procs)]):
            done += part
            print("generated %d files" % done, file=sys.stderr, flush=True)
    return n_files, n_files * 4096


_TEXT = None


def _text_pool():
    """80 MB of text-like bytes (words drawn with a Zipf law), the same in every process."""
    global _TEXT
    if _TEXT is None:
        rng = np.random.default_rng(12)
        vocab = [bytes(rng.integers(97, 123, int(k)).astype(np.uint8)) for k in rng.integers(2, 10, 3000)]
        vocab += [b"    ", b"\n", b"{", b"}", b"(", b")", b";", b"return ", b"const ", b"int ", b"if (", b"for ("]
        pick = np.minimum(rng.zipf(1.3, 12_000_000), len(vocab)) - 1
        _TEXT = b" ".join(vocab[int(k)] for k in pick)
    return _TEXT


def _c2text_part(args):
    root, lo, hi = args
    text = _text_pool()
    rng = np.random.default_rng(lo + 1)
    starts = rng.integers(0, len(text) - 4096, hi - lo)
    for i in range(lo, hi):
        d = os.path.join(root, "d%03d" % (i // 1000))
        if i % 1000 == 0 or i == lo:
            os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "f%07d.c" % i), "wb") as f:
            s = int(starts[i - lo])
            f.write(b"/* %07d */" % i + text[s:s + 4096 - 13])
    return hi - lo


def make_c2_text(root, n_files):
    """The headline layout with text-like content: n files x 4,096 B that zlib brings down to about a third, where the
    random blobs of make_c2 do not compress at all; 1,000 per directory."""
    import multiprocessing as mp
    procs = 8
    cuts = [n_files * k // procs // 1000 * 1000 for k in range(procs)] + [n_files]
    with mp.Pool(procs) as pool:
        for part in pool.imap_unordered(_c2text_part, [(root, cuts[k], cuts[k + 1]) for k in range(procs)]):
            pass
    return n_files, n_files * 4096


def make_c3(root, n_files):
    """Linux-kernel-like: lognormal sizes (median 4.5 KB, sigma 1.55) clipped to 32 MiB and rescaled to
    ~15 KB mean; ~n/16 directories, three levels; names avoid the cases where the reference and git differ."""
    rng = np.random.default_rng(0x5EED0003)
    sizes = np.exp(rng.normal(np.log(4500.0), 1.55, n_files))
    sizes = np.minimum(sizes * (15000.0 / sizes.mean()), 32 << 20).astype(np.int64)
    total = 0
    big = O.synth_fill_many(0x5EED0003, 0, 1, int(sizes.max()) + 8, threads=1)
    for i, L in enumerate(sizes):
        top, mid = i % 64, (i // 64) % 80
        d = os.path.join(root, "sub%02d" % top, "mod%02d" % mid)
        if not os.path.isdir(d):
            os.makedirs(d, exist_ok=True)
        # content: a window of one long synthetic stream, offset by the file index (cheap and distinct)
        off = (i * 7919) % 4096
        L = int(L)
        with open(os.path.join(d, "file%06d.c" % i), "wb") as f:
            f.write(big[off:off + L].tobytes() if off + L <= len(big) else big[:L].tobytes())
            f.write(b"")
        total += L
    return n_files, total


def run_vit(wt, args=()):
    subprocess.check_call([VIT, "init"], cwd=wt, stdout=subprocess.DEVNULL)
    env = dict(os.environ)
    env.setdefault("VIT_DEVICES", "0")
    t0 = time.perf_counter()
    p = subprocess.run([VIT, "write-tree", "--stats"] + list(args), cwd=wt, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, check=True)
    dt = time.perf_counter() - t0
    return p.stdout.decode().strip(), dt, p.stderr.decode().strip().splitlines()[-1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=["c1", "c2", "c2text", "c3"])
    ap.add_argument("--files", type=int, default=None, help="c2: default 1,000,000; c3: default 80,000")
    ap.add_argument("--git", action="store_true", help="also run real `git add -A && git write-tree` on the worktree")
    ap.add_argument("--dir", default=None, help="where to build the worktree (default: a temp dir, /dev/shm if present)")
    ap.add_argument("--skip-reference", action="store_true")
    ap.add_argument("--host-deflate", action="store_true", help="also run `vit write-tree --host-deflate` (zlib on host threads only)")
    ap.add_argument("--levels", type=int, nargs="*", default=[],
                    help="also run `vit write-tree --deflate-level=N` for these zlib levels (same ids, other stored bytes)")
    ap.add_argument("--make-only", action="store_true", help="build the worktree under --dir/wt and stop (bench.py's generator)")
    a = ap.parse_args()
    if a.make_only:
        wt = os.path.join(a.dir, "wt")
        os.makedirs(wt, exist_ok=True)
        n, total = (make_c1(wt) if a.config == "c1" else make_c2(wt, a.files or 1_000_000) if a.config == "c2"
                    else make_c2_text(wt, a.files or 200_000) if a.config == "c2text" else make_c3(wt, a.files or 80000))
        print(json.dumps({"files": n, "bytes": total}))
        return
    base = a.dir or tempfile.mkdtemp(prefix="vitwt_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    wt = os.path.join(base, "wt")
    os.makedirs(wt, exist_ok=True)
    try:
        t0 = time.perf_counter()
        if a.config == "c1":
            n, total = make_c1(wt)
        elif a.config == "c2":
            n, total = make_c2(wt, a.files or 1_000_000)
        elif a.config == "c2text":
            n, total = make_c2_text(wt, a.files or 200_000)
        else:
            n, total = make_c3(wt, a.files or 80000)
        gen_s = time.perf_counter() - t0
        out = {"config": a.config, "files": n, "bytes": total, "generate_s": round(gen_s, 2)}
        say = lambda m: print(m, file=sys.stderr, flush=True)
        say("worktree ready (%d files, %.1f s); CPU restatement with store ..." % (n, gen_s))

        # CPU path, single thread, as the reference does it: hash + deflate + store per object
        objdir = os.path.join(base, "oracle_objects")
        t0 = time.perf_counter()
        want, nb, nt = O.write_tree(wt, objdir)
        cpu_s = time.perf_counter() - t0
        shutil.rmtree(objdir, ignore_errors=True)
        say("... %.1f s; CPU restatement, ids only ..." % cpu_s)
        t0 = time.perf_counter()
        want2, _, _ = O.write_tree(wt, None)
        cpu_hash_s = time.perf_counter() - t0
        assert want == want2
        out["cpu_restatement"] = {"root": want.hex(), "blobs": nb, "trees": nt, "hash_deflate_store_s": round(cpu_s, 3),
                                  "hash_only_s": round(cpu_hash_s, 3), "cores": 1,
                                  "objects_per_s": round((nb + nt) / cpu_s, 1)}
        if O.ref_lib() is not None and not a.skip_reference:
            say("reference's own writeTree ...")
            cwd = os.getcwd()
            os.makedirs(os.path.join(wt, ".git", "objects"), exist_ok=True)
            os.chdir(wt)
            try:
                t0 = time.perf_counter()
                ref = O.ref_write_tree(".")
                ref_s = time.perf_counter() - t0
            finally:
                os.chdir(cwd)
            shutil.rmtree(os.path.join(wt, ".git"))
            out["reference_code"] = {"root": ref, "write_tree_s": round(ref_s, 3), "cores": 1,
                                     "objects_per_s": round((nb + nt) / ref_s, 1)}
            assert ref == want.hex()

        say("vit write-tree (GPU) ...")
        root, dt, stats = run_vit(wt)
        shutil.rmtree(os.path.join(wt, ".git"))
        root2, dt2, stats2 = run_vit(wt, ["--no-store"])
        out["gpu_write_tree"] = {"root": root, "wall_s": round(dt, 3), "stats": stats,
                                 "objects_per_s": round((nb + nt) / dt, 1),
                                 "ids_only_wall_s": round(dt2, 3), "ids_only_stats": stats2}
        out["match"] = bool(root == want.hex() == root2)
        if a.host_deflate:
            shutil.rmtree(os.path.join(wt, ".git"), ignore_errors=True)
            r4, dt4, st4 = run_vit(wt, ["--host-deflate"])
            out["gpu_write_tree_host_deflate"] = {"root": r4, "wall_s": round(dt4, 3), "stats": st4}
            out["match"] = out["match"] and r4 == root
        for level in a.levels:
            shutil.rmtree(os.path.join(wt, ".git"), ignore_errors=True)
            r3, dt3, st3 = run_vit(wt, ["--deflate-level=%d" % level])
            out["gpu_write_tree_level_%d" % level] = {"root": r3, "wall_s": round(dt3, 3), "stats": st3,
                                                      "objects_per_s": round((nb + nt) / dt3, 1)}
            out["match"] = out["match"] and r3 == root
        if a.git and shutil.which("git"):
            say("git add -A && git write-tree ...")
            shutil.rmtree(os.path.join(wt, ".git"), ignore_errors=True)
            t0 = time.perf_counter()
            subprocess.check_call(["git", "init", "-q", "."], cwd=wt)
            subprocess.check_call(["git", "-c", "core.fsync=none", "-c", "gc.auto=0", "add", "-A"], cwd=wt)
            g = subprocess.check_output(["git", "write-tree"], cwd=wt).decode().strip()
            out["git"] = {"write_tree": g, "seconds": round(time.perf_counter() - t0, 2), "equals_vit": g == root}
        print(json.dumps(out), flush=True)
        if not out["match"]:
            sys.exit(1)
    finally:
        if a.dir is None:
            shutil.rmtree(base, ignore_errors=True)


if __name__ == "__main__":
    main()
