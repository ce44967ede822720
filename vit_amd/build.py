"""Build the native parts in-tree with hipcc (gfx950 only).

    python -m vit_amd.build            # libvit_hash.so + the vit CLI
    python -m vit_amd.build --force

Outputs (git-ignored, shipped to the GPU box with the tree):
    vit_amd/libvit_hash.so    C ABI of include/vit_hash.h (kernels + host pipeline)
    vit_amd/bin/vit           hash-object / write-tree front end (C++ object store); links the host-side
                              sources only and opens libvit_hash.so when a batch is large enough for the GPU
    vit_amd/bin/option_a_driver   test program: the reference's free functions over this object store
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvit_hash.so")
CLI = os.path.join(HERE, "bin", "vit")
DRIVER = os.path.join(HERE, "bin", "option_a_driver")
DRIVER_SOURCE = os.path.join(ROOT, "tests", "option_a_driver.cpp")

# host-only sources (no HIP): also linked straight into the CLI, which loads the GPU library only when a batch needs it
HOST_SOURCES = ["packer.cpp", "sha1_host.cpp", "numa.cpp"]
LIB_SOURCES = ["kernels.hip", "deflate_kernels.hip", "hasher.cpp", "pipeline.cpp", "balance.cpp", "deflate_host.cpp",
               "bench_support.cpp"] + HOST_SOURCES
CLI_SOURCES = ["object_store.cpp", "change_detect.cpp", "vit_cli.cpp"]
COMMON = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "-I" + os.path.join(ROOT, "include")]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the HIP library cannot be built")
    return exe


def _cxx():
    """Host-only programs are compiled and linked by the C++ compiler itself: hipcc links the HIP runtime into
    whatever it links, and a program that merely loads that runtime pays some 15 ms at every start."""
    return shutil.which("g++") or shutil.which("c++") or os.path.join(os.path.dirname(_hipcc()), "amdclang++")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    deps = list(sources)
    for d in (CSRC, os.path.join(ROOT, "include")):
        deps += [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".h", ".hpp"))]
    return any(os.path.getmtime(s) > t for s in deps)


def build_lib(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in LIB_SOURCES]
    if not force and not _stale(LIB, srcs):
        return LIB
    cmd = [_hipcc()] + COMMON + ["--offload-arch=gfx950", "-shared", "-o", LIB] + srcs + ["-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_cli(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in CLI_SOURCES + HOST_SOURCES]
    if not all(os.path.exists(s) for s in srcs):
        return None
    if not force and not _stale(CLI, srcs) and os.path.getmtime(CLI) >= os.path.getmtime(LIB):
        return CLI
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    cmd = [_cxx()] + COMMON + ["-o", CLI] + srcs + ["-Wl,-rpath,$ORIGIN/..", "-lz", "-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return CLI


def build_driver(force=False, verbose=False):
    """tests/option_a_driver.cpp: a caller of the reference's free functions linked against object_store.cpp
    (the drop-in in which this file takes the place of the reference's commit.cpp), for tests/test_option_a.py."""
    srcs = [DRIVER_SOURCE, os.path.join(CSRC, "object_store.cpp")] + [os.path.join(CSRC, s) for s in HOST_SOURCES]
    if not os.path.exists(DRIVER_SOURCE):
        return None
    if not force and not _stale(DRIVER, srcs) and os.path.getmtime(DRIVER) >= os.path.getmtime(LIB):
        return DRIVER
    os.makedirs(os.path.dirname(DRIVER), exist_ok=True)
    cmd = [_cxx()] + COMMON + ["-I" + CSRC, "-o", DRIVER] + srcs + ["-Wl,-rpath,$ORIGIN/..", "-lz", "-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return DRIVER


def build_all(force=False, verbose=False):
    build_lib(force, verbose)
    build_cli(force, verbose)
    build_driver(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
