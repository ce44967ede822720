/*
 * vit_hash_bench.h -- measurement, test and diagnostic entry points of libvit_hash.so.
 *
 * Nothing here is needed to put the library behind the reference's object-store
 * functions (reference src/commit.hpp:8-18): that boundary is vit_hash.h.  These
 * are the generators, timers and plans the benchmark and the tests use, and
 * device-memory helpers for callers that have no HIP binding of their own (the
 * Python test harness).  None has a counterpart in the reference, which ships no
 * benchmark, no tests and no device code.
 *
 * Options of vit_hasher_set_option meant for tests and diagnostics only:
 *  "debug_fail_at_chunk"  the N-th staged chunk of every later call fails as if a HIP
 *                         call had (0 = off)
 *  "debug_fail_deflate"   1: vit_deflate_objects fails at once, as if the device had
 *  "debug_corrupt_deflate" 1: vit_deflate_objects flips one byte in the middle of the first
 *                         stream it returns (what a silent device fault would look like to
 *                         the caller, whose checks it is there to exercise)
 *  "trace"                1: vit_hash_objects prints one line per staged chunk to stderr
 *                         (host and device timestamps of pack, copy, kernel and read-back)
 *  "wave_trace_ptr"       a device address with room for 32 bytes per wave (4 waves per
 *                         256 objects); while it is set, every wave of a launch records
 *                         there when it started and ended (100 MHz clock), its hardware id
 *                         (SIMD, CU, SE, XCD) and its longest object in blocks; 0 = off.
 *                         The deflate kernels that run one workgroup per object take the
 *                         same address for something else: fourteen 64-bit counters (128 bytes), to the first eight of which
 *                         the wave of every workgroup that runs the serial parts adds its
 *                         shader-clock cycles by phase (load and checksum, chains, waiting for
 *                         the searches, decisions, symbols to scratch, the group's blocks by
 *                         lanes, stored bytes; last: objects done), and one searching wave of every workgroup its own to
 *                         the other six: tools/probe_deflate_phases.py
 */
#ifndef VIT_HASH_BENCH_H
#define VIT_HASH_BENCH_H

#include "vit_hash.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- plans, host-only ----------------------------------------------------- */

/* The balance "host_auto" strikes, as a plan: on_host[i] = 1 if object i (content
 * length len[i]) would be hashed on host threads beside the GPU batch, given
 * bytes-per-second rates of one GPU lane, one host core hashing, one pinned
 * host-to-device copy and one host core packing, the number of device queues and
 * the host thread budget.  Objects under 16 KiB never move; from the longest
 * down, objects move while the host side (hashing what moved plus packing what
 * stayed, over the threads) would still finish before the GPU side (its bytes
 * through 0.8 x the copy rate of the queues -- n_queues x copy_rate, or
 * copy_rate_total when that is given (non-zero) and less: what the queues get
 * when they all copy at once --, never less than its longest object on one
 * lane); a run of equal lengths may be split.  Host-only; needs no device.  The
 * reference hashes every object on its one thread (src/commit.cpp:93-94). */
int vit_host_split_plan(const uint64_t *len, uint64_t n, uint64_t lane_rate, uint64_t core_rate,
                        uint64_t copy_rate, uint64_t copy_rate_total, uint64_t pack_rate,
                        int n_queues, int threads, uint8_t *on_host);

/* Launch shape of the SHA-1 kernel for n objects (n <= 2^32 - 1): workgroups
 * and threads per workgroup.  Host-only; lets the launch arithmetic be checked
 * at sizes no test can allocate. */
int vit_launch_shape(int kernel, uint64_t n, uint32_t *blocks, uint32_t *threads);

/* Where the pool's workers would be pinned, as a plan (the "numa_pin" option): given the
 * PCI bus ids ("0000:c1:00.0") of the GPUs behind n_queues device queues and a number of pool
 * workers, queue_node[k] = NUMA node of queue k's GPU (-1 unknown) and worker_node[w] = node
 * worker w is pinned to (-1: not pinned).  Nodes get workers in proportion to their queues;
 * nothing is pinned when a node is unknown or has fewer CPUs than workers it would get.
 * sysfs_root: NULL = "/sys" (tests pass a directory that holds bus/pci/devices/<id>/numa_node
 * and devices/system/node/node<N>/cpulist files).  Host-only; needs no device.  Here every
 * CPU of a node counts; the hasher itself also leaves out CPUs the process may not run on. */
int vit_numa_plan(const char *sysfs_root, const char *const *pci_bus_ids, int n_queues, int workers,
                  int *queue_node, int *worker_node);
/* CPUs of a NUMA node: up to `cap` of them into cpus[]; returns how many the node has. */
int vit_numa_node_cpus(const char *sysfs_root, int node, int *cpus, int cap);

/* ---- device memory, for callers without a HIP binding of their own -------- */
int vit_device_alloc(vit_hasher *h, int dev, uint64_t bytes, void **out);
int vit_device_free(vit_hasher *h, int dev, void *p);
int vit_device_upload(vit_hasher *h, int dev, void *d_dst, const void *src, uint64_t bytes);
int vit_device_download(vit_hasher *h, int dev, void *dst, const void *d_src, uint64_t bytes);
int vit_device_sync(vit_hasher *h, int dev);

/* ---- synthetic inputs and measurement ------------------------------------ */

/* Fill a device arena with n framed, padded objects whose content is the
 * counter-based splitmix64 stream (seed, first_index + i).  d_lens == NULL:
 * every object has len_each bytes, and d_offsets/d_nblocks are outputs;
 * otherwise d_lens and d_offsets are inputs and d_nblocks is an output.
 * The reference ships no benchmark inputs; this generator is this library's. */
int vit_synth_arena_device(vit_hasher *h, int dev, const char *type, uint64_t seed,
                           uint64_t first_index, uint64_t n, uint64_t len_each,
                           const uint64_t *d_lens, uint64_t *d_offsets, uint32_t *d_nblocks,
                           void *d_arena, uint64_t arena_bytes, void *stream);
/* Host restatement of the same byte stream (for checking the generator). */
void vit_synth_bytes(uint64_t seed, uint64_t index, uint8_t *dst, uint64_t len);
/* ... for n objects in one call, on `threads` threads: object i = stream (seed, first_index + i),
 * lens[i] bytes at dst + offsets[i].  For workloads of millions of objects in host memory. */
int vit_synth_fill_host(uint64_t seed, uint64_t first_index, uint64_t n, const uint64_t *lens,
                        const uint64_t *offsets, uint8_t *dst, int threads);

/* Launch the device-resident hash `iters` times after `warmup` untimed
 * launches and report each launch's duration in milliseconds, measured with
 * HIP events on the launch stream. */
int vit_time_hash_device(vit_hasher *h, int dev, const void *d_arena, uint64_t arena_bytes,
                         const uint64_t *d_offsets, const uint32_t *d_nblocks,
                         const uint32_t *d_order, uint64_t n, uint8_t *d_digests,
                         int warmup, int iters, float *ms_each);

/* Compute-only ceiling of the SHA-1 instruction stream on this device: n_lanes
 * lanes each run blocks_per_lane compressions on register data.  *ms = one
 * launch (best of iters). */
int vit_valu_peak(vit_hasher *h, int dev, uint32_t n_lanes, uint32_t blocks_per_lane,
                  int iters, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* VIT_HASH_BENCH_H */
