import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_objects():
    return json.load(open(os.path.join(GOLDEN, "objects.json")))


@pytest.fixture(scope="session")
def golden_trees():
    return json.load(open(os.path.join(GOLDEN, "trees.json")))


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def _session_hasher():
    """One hasher on GPU 0 for the whole GPU session (fails loudly without a GPU)."""
    import vit_amd
    h = vit_amd.Hasher((0,))
    yield h
    h.close()


@pytest.fixture
def hasher(_session_hasher):
    """The session's hasher with every option back at its default."""
    h = _session_hasher
    for name, value in (("kernel", 0), ("chunk_bytes", 64 << 20), ("chunk_object_multiple", 8192), ("trace", 0),
                        ("host_threshold_bytes", 0), ("host_auto", 1), ("sched_waves", 1), ("sched_min_groups", 4096), ("sched_tail_percent", 75),
                        ("debug_fail_at_chunk", 0), ("lane_rate", 0), ("core_rate", 0), ("core_rate_multi", 0), ("host_bias_percent", 100), ("deflate_max_bytes", 16382), ("host_min_bytes", 16 << 10), ("copy_rate", 0), ("pack_rate", 0)):
        h.set_option(name, value)
    return h


def materialize(spec, root, oracle):
    """Write a worktree from a golden spec: (path, seed, index, length[, 'x']) or (dir, None)."""
    for item in spec:
        path = os.path.join(root, item[0])
        if item[1] is None:
            os.makedirs(path, exist_ok=True)
            continue
        os.makedirs(os.path.dirname(path) or root, exist_ok=True)
        with open(path, "wb") as f:
            f.write(oracle.synth_bytes(item[1], item[2], item[3]))
        if len(item) > 4 and item[4] == "x":
            os.chmod(path, 0o755)


def c1_spec(seed):
    return [("d%02d/f%04d.bin" % (i // 100, i), seed, i, 4096) for i in range(1000)]


def skew_spec(seed, n, oracle):
    """The skewed worktree of tests/golden/gen_golden.py (same function): n files in t%02d/s%02d/, sizes from the
    shared counter-based stream: 5 % empty, 75 % under 4 KiB, 17 % under 32 KiB, 2.9 % under 256 KiB, 0.1 % under 2 MiB."""
    spec = []
    for i in range(n):
        r = int.from_bytes(oracle.synth_bytes(seed ^ 0x51E5, i, 8), "little")
        cls, v = r % 1000, r >> 16
        size = 0 if cls < 50 else v % 4096 if cls < 800 else v % 32768 if cls < 970 else v % 262144 if cls < 999 \
            else v % (2 << 20)
        spec.append(("t%02d/s%02d/f%05d.dat" % (i % 20, (i // 20) % 10, i), seed, i, size))
    return spec


@pytest.fixture(scope="session")
def worktrees(tmp_path_factory, golden_trees, oracle):
    """Every golden worktree materialised once: name -> directory."""
    out = {}
    for name, rec in golden_trees.items():
        d = tmp_path_factory.mktemp("wt_" + name)
        spec = c1_spec(rec["seed"]) if rec["spec"] == "c1" else \
            skew_spec(rec["seed"], rec["n"], oracle) if rec["spec"] == "skew" else rec["spec"]
        materialize(spec, str(d), oracle)
        out[name] = str(d)
    return out
