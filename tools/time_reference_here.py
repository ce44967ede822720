#!/usr/bin/env python3
"""Time the reference's OWN code (oracle/_ref/libvitref.so = its commit.cpp compiled as it lies) in the build
container: writeBlob on 4 KiB blobs and writeTree on BASELINE configs[0].  The reference build does not travel
to the GPU box, so these numbers are container numbers (CPU model and core count recorded beside them);
the on-box CPU baseline in bench.py is the restatement (kind "port").

    python tools/time_reference_here.py > profiles/r02_reference_container.json
"""
import json
import os
import platform
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402


def main():
    if O.ref_lib() is None:
        print(json.dumps({"error": "oracle/_ref/libvitref.so is not built here (make -C oracle ref)"}))
        return 1
    cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
    out = {"where": "build container", "cpu": cpu[0] if cpu else platform.processor(), "logical_cpus": os.cpu_count(),
           "threads_used": 1, "kind": "reference"}
    cwd = os.getcwd()
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=base) as td:
        os.makedirs(os.path.join(td, ".git", "objects"))
        os.chdir(td)
        try:
            n = 4000
            blobs = O.synth_fill_many(0x5EED0002, 0, n, 4096, threads=4)
            t0 = time.perf_counter()
            for i in range(n):
                O.ref_write_blob(blobs[i * 4096:(i + 1) * 4096].tobytes())
            dt = time.perf_counter() - t0
            out["write_blob_4k"] = {"objects": n, "seconds": round(dt, 3), "objects_per_s": round(n / dt, 1),
                                    "GBps": round(n * 4096 / dt / 1e9, 5),
                                    "what": "writeBlob = SHA1 + zlib compress + loose-object file (src/commit.cpp:88-116)"}
        finally:
            os.chdir(cwd)
    with tempfile.TemporaryDirectory(dir=base) as td:
        import importlib.util
        spec = importlib.util.spec_from_file_location("bwt", os.path.join(ROOT, "tools", "bench_write_tree.py"))
        bwt = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bwt)
        wt = os.path.join(td, "wt")
        os.makedirs(os.path.join(wt, ".git", "objects"))
        bwt.make_c1(wt)
        os.chdir(wt)
        try:
            t0 = time.perf_counter()
            root = O.ref_write_tree(".")
            dt = time.perf_counter() - t0
        finally:
            os.chdir(cwd)
        want, nb, nt = O.write_tree(wt, None)
        out["write_tree_c1"] = {"files": 1000, "seconds": round(dt, 3), "objects_per_s": round((nb + nt) / dt, 1),
                                "root": root, "restatement_agrees": root == want.hex(),
                                "what": "writeTree on 1,000 x 4 KiB in 10 directories (src/commit.cpp:139-173)"}
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
