"""The N > 1 path on CPU: two gloo ranks agree on a byte-balanced, disjoint and
complete partition of one object list, hash their shards independently (the
oracle stands in for the GPU here — this test is about the sharding and the
reductions, not the kernel), and combine timing and checksum summaries with the
same helpers bench.py uses on RCCL."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    sys.path.insert(0, %r)
    from vit_amd import dist as vdist
    from oracle import pyoracle as O
    import vit_amd

    rank, local_rank, world = vdist.env()
    dist = vdist.init(backend="gloo")
    assert dist is not None and dist.get_world_size() == world == 2

    rng = np.random.default_rng(7)                       # same list on every rank
    lens = np.concatenate([rng.integers(0, 600, 900), [3 << 20, 2 << 20, 1 << 20]]).astype(np.int64)
    rng.shuffle(lens)
    nblocks = np.array([vit_amd.object_blocks("blob", int(L)) for L in lens], dtype=np.uint32)

    mine = vdist.strong_shard(rank, world, nblocks)
    digests = np.array([np.frombuffer(O.blob_id(O.synth_bytes(99, int(i), int(lens[i]))), dtype=np.uint8)
                        for i in mine]).reshape(-1, 20)
    total = vdist.sum_over_ranks(dist, len(mine))
    load = int(nblocks[mine].astype(np.int64).sum())
    slowest = vdist.max_over_ranks(dist, 1.0 + rank)     # rank 1 pretends to be slower
    summary = vdist.combine_summaries(dist, vdist.xor_summary(digests))
    dist.barrier()
    # rank 0 works alone for a while; the other rank waits on the store, not in a collective
    import time
    t0 = time.time()
    if rank == 0:
        time.sleep(1.0)
        vdist.release_ranks(dist)
    else:
        vdist.wait_for_rank0(dist, timeout_s=60)
        assert time.time() - t0 > 0.5
    print(json.dumps({"rank": rank, "mine": mine.tolist(), "total": total, "load": load, "slowest": slowest,
                      "summary": summary.tolist(), "first": vdist.weak_first_index(rank, 1000)}), flush=True)
    dist.destroy_process_group()
""") % ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_sharding_and_reductions(tmp_path, oracle):
    import json
    import vit_amd
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(free_port())
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=240)
        assert p.returncode == 0, e.decode()[-2000:]
        outs.append(json.loads(o.decode().strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])

    rng = np.random.default_rng(7)
    lens = np.concatenate([rng.integers(0, 600, 900), [3 << 20, 2 << 20, 1 << 20]]).astype(np.int64)
    rng.shuffle(lens)
    n = len(lens)
    a, b = set(outs[0]["mine"]), set(outs[1]["mine"])
    assert not (a & b) and a | b == set(range(n))                      # disjoint and complete
    assert outs[0]["total"] == outs[1]["total"] == n
    loads = [outs[0]["load"], outs[1]["load"]]
    assert max(loads) <= 0.55 * sum(loads)                             # balanced by bytes despite three huge objects
    assert outs[0]["slowest"] == outs[1]["slowest"] == 2.0             # max over ranks, same everywhere
    want = np.zeros(20, dtype=np.uint8)
    for i in range(n):
        want ^= np.frombuffer(oracle.blob_id(oracle.synth_bytes(99, i, int(lens[i]))), dtype=np.uint8)
    assert outs[0]["summary"] == outs[1]["summary"] == want.tolist()   # checksum of checksums over the whole job
    assert [o["first"] for o in outs] == [0, 1000]


@pytest.mark.gpu
def test_bench_runs_as_two_ranks_on_one_gpu():
    """bench.py under torch.distributed.run with two ranks, as the scaling run launches it; both ranks
    use GPU 0 here, so the timing barrier goes over gloo (RCCL refuses two ranks on one device).  The
    line must count both ranks' blobs, and each rank hashes a different range of the synthetic stream."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--blobs", "200000",
           "--dist-backend", "gloo", "--no-cpu-baseline", "--e2e-objects", "30000"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                   # rank 0 alone prints
    assert out.stdout.strip().splitlines() == lines, out.stdout   # and nothing else reaches stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["scaling"] == "weak" and d["device_status"] == 0
    assert abs(d["objects_per_s"] - 2 * 200000 * 5 / (d["ms_per_step"] * 5e-3)) / d["objects_per_s"] < 0.01
    assert abs(d["value"] - d["objects_per_s"] * 4096 / 1e9) / d["value"] < 0.01
    # the host-memory legs ran after the timed region, on one hasher with a queue per rank's GPU (GPU 0 twice here)
    e = d["end_to_end_host_memory"]
    assert e["uniform_4k"]["matches_resident_digests"]
    m = e["mixed_slice"]
    assert m["queues"] == 2 and m["device_ids"] == [0, 0] and m["matches"] and m["scaling"] == "strong"
    assert m["objects_gpu"] + m["objects_host"] == 30000
    w = e["mixed_slice_weak"]                             # N times the objects through the same N queues
    assert w["scaling"] == "weak" and w["queues"] == 2 and w["matches"] and w["objects_gpu"] + w["objects_host"] == 60000
