"""Development probe: quick parity + timing of both kernels on one GPU.
Not part of the product or the test-suite; run with gpurun."""
import hashlib
import sys
import time

import numpy as np

import vit_amd


def git_id(type_, b):
    return hashlib.sha1(("%s %d" % (type_, len(b))).encode() + b"\0" + b).digest()


def main():
    n_big = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    h = vit_amd.Hasher((0,))
    print("host sha accelerated:", h.stat("host_sha_accelerated"), flush=True)
    lens = list(range(0, 300)) + [4095, 4096, 4097, 65536, 100000, 1 << 20]
    blobs = [vit_amd.synth_bytes(11, i, L) for i, L in enumerate(lens)]
    want = np.array([np.frombuffer(git_id("blob", b), dtype=np.uint8) for b in blobs])
    for k in (vit_amd.KERNEL_LDS, vit_amd.KERNEL_LANE):
        h.set_option("kernel", k)
        got = h.hash_blobs(blobs)
        bad = np.nonzero((got != want).any(axis=1))[0]
        print("kernel", k, "host-batch parity:", "OK" if len(bad) == 0 else "BAD at %s" % [lens[i] for i in bad[:10]],
              flush=True)

    # device generator vs host packer
    b = h.synth_arena("blob", 0x5EED0002, 0, 4096, 4096)
    arena = b["arena"].download(np.uint8, b["arena_bytes"])
    blobs = [vit_amd.synth_bytes(0x5EED0002, i, 4096) for i in range(4096)]
    ref_arena, off, nb, used = vit_amd.pack_arena("blob", blobs)
    print("generator == packer:", bool((arena == ref_arena[:used]).all()), flush=True)
    want = np.array([np.frombuffer(git_id("blob", x), dtype=np.uint8) for x in blobs])
    for k in (vit_amd.KERNEL_LDS, vit_amd.KERNEL_LANE):
        h.set_option("kernel", k)
        got = h.hash_device(b)
        print("kernel", k, "device parity:", bool((got == want).all()), "status", h.device_status(), flush=True)

    # throughput
    t0 = time.time()
    big = h.synth_arena("blob", 0x5EED0002, 0, n_big, 4096)
    print("synth %d x 4096: %.3fs" % (n_big, time.time() - t0), flush=True)
    payload = n_big * 4096
    for k in (vit_amd.KERNEL_LDS, vit_amd.KERNEL_LANE):
        h.set_option("kernel", k)
        ms = h.time_hash_device(big, warmup=1, iters=5)
        best = min(ms)
        print("kernel", k, "ms:", ["%.3f" % m for m in ms], "payload GB/s %.1f" % (payload / best / 1e6),
              "Mobj/s %.1f" % (n_big / best / 1e3), flush=True)
    # sample check of the big batch
    dig = big["digests"].download(np.uint8, 20 * n_big).reshape(-1, 20)
    for i in (0, 1, 63, 64, n_big // 2, n_big - 1):
        assert dig[i].tobytes() == git_id("blob", vit_amd.synth_bytes(0x5EED0002, i, 4096)), i
    print("big sample parity OK", flush=True)

    lanes = 256 * 2048
    for bpl in (64, 256):
        ms = h.valu_peak(lanes, bpl, iters=3)
        blocks = lanes * bpl
        print("valu_peak lanes=%d blocks/lane=%d: %.3f ms -> %.2f Gblocks/s = %.2f TB/s-equivalent, %.1f Tops/s at 597 ops/block"
              % (lanes, bpl, ms, blocks / ms / 1e6, blocks * 64 / ms / 1e9, blocks * 597 / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
