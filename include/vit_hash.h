/*
 * vit_hash.h — C ABI of the MI355X batched Git-object hasher (libvit_hash.so).
 *
 * What it replaces.  The reference VCS computes every object id inline, one
 * object at a time, with OpenSSL's one-shot SHA1() inside writeObject()
 * (reference src/commit.cpp:88-95, declared src/commit.hpp:13; called per file
 * from writeTree(), src/commit.cpp:139-173, and from writeBlob(), :116).  The
 * reference has no plugin or FFI boundary at that point; this header IS the
 * boundary a maintainer would cut there: the host keeps framing, tree assembly,
 * deflate and storage, and hands batches of objects to the functions below,
 * which return the same 20-byte ids SHA1() would have produced.
 *
 * Conventions.  Plain C types only.  Every function returns VIT_OK (0) or a
 * negative VIT_ERR_* code and never throws or aborts; the text of the last
 * error is kept per hasher (vit_last_error).  Digests are 20 raw bytes per
 * object, in input order.  A vit_hasher may be used by one thread at a time.
 * There is no CPU fallback: without a usable gfx950 device vit_hasher_create()
 * fails with VIT_ERR_NODEVICE.
 */
#ifndef VIT_HASH_H
#define VIT_HASH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIT_OK             0
#define VIT_ERR_ARG       -1   /* null pointer, bad type string, size mismatch          */
#define VIT_ERR_HIP       -2   /* a HIP runtime call failed (text in vit_last_error)    */
#define VIT_ERR_NOMEM     -3   /* host or device allocation failed                      */
#define VIT_ERR_RANGE     -4   /* arena too small / offset out of bounds / misaligned   */
#define VIT_ERR_NODEVICE  -5   /* no gfx950 device visible                              */
#define VIT_ERR_DEVICE_STATUS -6 /* a kernel flagged bad input (see vit_device_status)  */

#define VIT_DIGEST_BYTES 20

#define VIT_KERNEL_LDS   0     /* wave-cooperative loads staged through LDS (default)   */
#define VIT_KERNEL_LANE  1     /* every lane loads its own blocks                       */

typedef struct vit_hasher vit_hasher;

/* ---- lifetime ------------------------------------------------------------ */

/* Create a hasher over the given HIP device ordinals (n_devices == 0: every
 * visible device; an ordinal may repeat, which gives two independent queues on
 * one GPU).  Objects of a batch are sharded across the devices by bytes; no
 * inter-GPU exchange takes place.  No counterpart in the reference (it has no
 * device state); stands where the reference would link libcrypto
 * (reference CMakeLists.txt:17-22). */
int vit_hasher_create(const int *device_ids, int n_devices, vit_hasher **out);
void vit_hasher_destroy(vit_hasher *h);

/* Text of the last error on this hasher (h == NULL: last vit_hasher_create
 * failure on the calling thread).  Reference convention: message on std::cerr
 * and an empty result (src/commit.cpp:101-104). */
const char *vit_last_error(const vit_hasher *h);

int vit_hasher_device_count(const vit_hasher *h);

/* Options:
 *  "kernel"               VIT_KERNEL_*
 *  "host_threshold_bytes" objects larger than this are hashed on host threads
 *                         while the GPU batch runs (one SHA-1 stream is serial;
 *                         a lane sustains ~50 MB/s, a host core ~2 GB/s);
 *                         0 = never, every object takes a GPU lane
 *  "host_auto"            1 (default): derive that threshold per call by
 *                         balancing the two sides: from the longest object
 *                         down, objects of 16 KiB or more move to the host
 *                         threads while the host side (its bytes over the
 *                         threads; its longest object on one core) would
 *                         still finish before the GPU side (its bytes over
 *                         pack + PCIe; its longest object on one lane)
 *  "host_min_bytes"       under "host_auto", objects shorter than this never move to the
 *                         host threads (default 16 KiB: batches of small files, the bulk of a
 *                         worktree, stay the GPU's, and the host threads stay free for
 *                         reads, deflate and stores; 1 lets the balance share any batch,
 *                         which a pure hashing workload from host memory gains from)
 *  "chunk_bytes"          smallest pinned staging buffer per queue and stream
 *                         (default 64 MiB); chunks grow with the longest
 *                         object they may hold, up to 1 GiB
 *  "chunk_object_multiple" that growth: chunk = longest object x this
 *                         (default 8192; 0 = fixed chunk_bytes)
 *  "pack_threads"         the host thread budget of a call: ONE pool of threads per
 *                         hasher packs the chunks of every device queue (first) and
 *                         hashes the host-lane objects (when no packing waits); the
 *                         calling thread and the per-queue threads count against it,
 *                         so the number of busy threads does not grow with the
 *                         number of GPUs (default: the cores, at most 16)
 *  "lane_rate" "core_rate" "core_rate_multi" "copy_rate" "pack_rate"
 *                         bytes per second of one GPU lane, one host core hashing one
 *                         object, one host core hashing four objects side by side (their
 *                         SHA-1 streams interleaved: where the CPU pipelines its SHA
 *                         instructions that is about twice the bytes, and host-lane
 *                         objects that outnumber the threads then go in such groups),
 *                         one pinned host-to-device copy and one host core packing:
 *                         what "host_auto" balances with.  They are measured on this
 *                         machine the first time a batch needs them (a few ms);
 *                         setting one fixes it, 0 has it measured again
 *  "host_bias_percent"    the factor (x100) on the host side's estimated time in that balance.
 *                         The hasher adjusts it after every large call by how much later or
 *                         earlier than the device queues the host threads finished (the rates
 *                         are measured one at a time; a batch runs them against one memory
 *                         system at once); setting it starts the adjustment from there, 0 = 100
 *  "sched_waves"          resident waves per SIMD of the software-scheduled launch that batches in
 *                         binned order (an order array is given, or computed by the library)
 *                         of at least "sched_min_groups" 64-object groups take: a fixed grid
 *                         of waves pulls groups longest-first, one leader wave per SIMD, so a
 *                         long object has its SIMD to itself (default 1; 2 adds a spare wave
 *                         per SIMD; 0 = always the hardware dispatcher)
 *  "sched_min_groups"     default 4096
 *  "sched_tail_percent"   ... and only when the batch's longest object is at least this many
 *                         percent of one SIMD's mean share of the work (arena bytes / 64 / 64
 *                         lanes / SIMDs, in blocks): the scheduled launch trades about a
 *                         quarter of the chip's rate for never letting a long object share its
 *                         SIMD, which pays from about 0.78 of that mean on and costs batches of
 *                         short objects a factor of three (default 75; 0 = whatever the lengths).
 *                         Where the lengths exist in device memory only (vit_hash_arena_device
 *                         with an order, vit_hash_arena_device_static) both launches are queued
 *                         and their waves decide: those of the one not chosen leave at once
 *  "sched_share_below"    a spare wave works only while its SIMD's leader is on a group of at
 *                         most this many blocks (default 128)
 *  "copy_rate_total"      what ALL device queues of the hasher get out of pinned host-to-device
 *                         copies when they copy at once, bytes per second: with one queue the
 *                         same as "copy_rate"; with several it is measured by one copy on every
 *                         queue at the same time, and is what the balance divides the GPU side's
 *                         bytes by when it is less than the queues' own rates added up (queues
 *                         that share a PCIe link, a socket's root complex or the host's DRAM do
 *                         not each get what one of them gets alone); setting fixes, 0 measures
 *  "numa_pin"             1 (default): the pool's workers are pinned, in proportion to the queues
 *                         there, to the CPUs of the NUMA nodes the GPUs hang off, and a queue's
 *                         packing goes to the workers of its node first (its pinned staging arena
 *                         lies in that node's memory); nothing is pinned where a GPU's node is
 *                         unknown or a node has fewer usable CPUs than workers it would get.
 *                         0: workers run where the scheduler puts them
 *  "deflate_max_bytes" "deflate_lanes" "deflate_chunk_bytes"   see vit_deflate_objects,
 *                         vit_deflate_arena_device
 * Options for tests and diagnostics ("debug_*", "trace", "wave_trace_ptr") are
 * listed in vit_hash_bench.h. */
int vit_hasher_set_option(vit_hasher *h, const char *name, uint64_t value);

/* Counters since creation: "objects_gpu", "objects_host", "bytes_host",
 * "blocks_gpu", "bytes_h2d", "launches"; of the last call: "last_kernel_ns",
 * "last_pack_ns", "last_total_ns", "last_plan_ns" (its serial preamble: the host/GPU split, block counts, sharding),
 * "last_gpu_done_ns", "last_host_done_ns" (when the device queues and the
 * last host-lane task of the last call had finished, from its start), "last_host_threshold" (shortest length moved to
 * the host threads, less one), "last_host_objects"; "host_bias_percent"; the four rates above and
 * "calibrate_ns" (what measuring them cost); "pool_threads", "pool_peak_busy"
 * (most pool threads inside a task at once); "host_sha_accelerated" (1 when
 * the host lane uses the CPU's SHA extensions); "copy_rate_total";
 * "workers_pinned" (pool workers pinned to a NUMA node); and per device queue k,
 * "queue_numa_node:k" (node of the GPU's PCIe function) and "arena_numa_node:k"
 * (node the first page of the queue's pinned staging arena lies on), each
 * 2^64 - 1 when unknown. */
int vit_hasher_get_stat(const vit_hasher *h, const char *name, uint64_t *value);

/* ---- framing (reference src/commit.cpp:90) ------------------------------- */

/* Length of "<type> <len>\0": strlen(type) + 1 + decimal digits + 1. */
uint32_t vit_header_len(const char *type, uint64_t len);
/* Write that header into out (>= 32 bytes); returns its length. */
uint32_t vit_write_header(const char *type, uint64_t len, uint8_t *out);
/* 64-byte SHA-1 blocks of the framed, padded object: ceil((hdr + len + 9) / 64). */
uint64_t vit_object_blocks(const char *type, uint64_t len);

/* ---- batch hashing from host memory -------------------------------------- */

/* ids of n objects of one type: digest i = SHA1("<type> <len[i]>\0" + data[i]).
 * Replaces n calls of the SHA1() at reference src/commit.cpp:93-94.  The
 * library packs the objects into pinned 64-byte-block arenas, copies them to
 * the device(s) on HIP streams (packing of chunk k+1 overlaps copy and hashing
 * of chunk k) and returns when all digests are written. */
int vit_hash_objects(vit_hasher *h, const char *type, const uint8_t *const *data,
                     const uint64_t *len, uint64_t n, uint8_t *digests);

/* Same with type "blob": what writeBlob() hashes (reference src/commit.cpp:116). */
int vit_hash_blobs(vit_hasher *h, const uint8_t *const *data, const uint64_t *len,
                   uint64_t n, uint8_t *digests);

/* Plain SHA-1 of n buffers, no Git framing: digest i = SHA1(data[i], len[i]).
 * This is the reference's call itself, n at a time: src/commit.cpp:93-94 passes
 * SHA1() the already framed "<type> <len>\0" + content string.  Same pipeline
 * and scheduling as vit_hash_objects. */
int vit_sha1_buffers(vit_hasher *h, const uint8_t *const *data, const uint64_t *len,
                     uint64_t n, uint8_t *digests);

/* id of ONE object computed on the calling host thread (SHA-NI when the CPU has
 * it).  This is the routine the scheduler applies to objects above
 * "host_threshold_bytes"; exported so that path can be checked on its own.
 * Same result as the SHA1() at reference src/commit.cpp:93-94. */
int vit_host_object_id(const char *type, const uint8_t *data, uint64_t len, uint8_t *digest);

/* ids of n objects on the calling host thread, up to four of them at a time with
 * their SHA-1 block functions interleaved (one stream's rounds are a dependent
 * chain; a core with pipelined SHA instructions gets through more than twice the
 * bytes that way).  The routine the scheduler applies to host-lane objects that
 * outnumber the threads; exported so that path can be checked on its own.
 * digests: n x 20 bytes, in input order.  Same results as n calls of
 * vit_host_object_id, i.e. as the SHA1() at reference src/commit.cpp:93-94. */
int vit_host_object_ids(const char *type, const uint8_t *const *data, const uint64_t *len, uint64_t n,
                        uint8_t *digests);

/* ---- pre-packed arenas ---------------------------------------------------- */

/* Pinned host memory for arenas (hipHostMalloc); free with vit_arena_free. */
int vit_arena_alloc(vit_hasher *h, uint64_t bytes, void **out);
int vit_arena_free(vit_hasher *h, void *arena);

/* Pack n objects back to back into `arena`: object i starts at offsets[i] (a
 * multiple of 64) and occupies nblocks[i]*64 bytes holding header, content,
 * 0x80, zero fill and the 64-bit big-endian bit length, i.e. exactly the words
 * SHA-1 compresses, each 4-byte big-endian message word stored as a native
 * little-endian dword (the 16 byte swaps per block are paid here, during the
 * copy, and not by the kernel).  *used_bytes = total.  VIT_ERR_RANGE if arena_bytes is too
 * small (nothing written).  Host-only; needs no device.  Restates the framing
 * of reference src/commit.cpp:90-91 plus FIPS 180-4 5.1.1 padding. */
int vit_arena_pack(const char *type, const uint8_t *const *data, const uint64_t *len,
                   uint64_t n, void *arena, uint64_t arena_bytes, uint64_t *offsets,
                   uint32_t *nblocks, uint64_t *used_bytes, int threads);

/* Same, with every slot's start rounded up to a multiple of slot_align (64, or
 * a larger power of two).  For an arena that stays resident in device memory
 * use 128: the device L2 works on 128-byte lines and the kernel reads 128
 * bytes per object and step, so a slot that starts mid-line splits every one
 * of its reads in two and, once the line has left the cache between two steps,
 * fetches it again (measured on 1 M slots of 65 blocks: 15 % more bytes fetched
 * from HBM at 64 than at 128).  The library's own staging arenas, which cross
 * PCIe, stay at 64: there the padding would cost more than the re-reads. */
int vit_arena_pack_aligned(const char *type, const uint8_t *const *data, const uint64_t *len,
                           uint64_t n, void *arena, uint64_t arena_bytes, uint64_t *offsets,
                           uint32_t *nblocks, uint64_t *used_bytes, int threads,
                           uint32_t slot_align);

/* Length binning: order[] = object indices sorted by block count, longest
 * first, ties in index order, so the 64 lanes of a wave finish together.
 * Returns 1 if the order differs from identity, 0 if all counts are equal
 * (order is then the identity and may be omitted), negative on error. */
int vit_bin_order(const uint32_t *nblocks, uint64_t n, uint32_t *order);

/* The same permutation computed on the device, for arenas whose block counts
 * never were in host memory: d_nblocks and d_order are device pointers on
 * queue `dev`; asynchronous on `stream` (NULL = the queue's own).  The library
 * keeps the sort's scratch memory (about 8 bytes per object) with the queue
 * (it allocates on first use and when n grows: see
 * vit_hash_arena_device_static for a launch that bins without allocating). */
int vit_bin_order_device(vit_hasher *h, int dev, const uint32_t *d_nblocks, uint64_t n,
                         uint32_t *d_order, void *stream);

/* ---- loose-object contents on the device ------------------------------------- */

/* The zlib stream of every object of a device-resident arena: the bytes the
 * reference stores, compress(header + content) at the default level
 * (src/commit.cpp:98-105), with the zlib library's own choices, so that the files
 * are identical to the reference's.  Object i's stream is written to
 * d_out + d_out_offsets[i], where the caller has left room for
 * vit_deflate_bound(framed length) bytes, and its length to d_out_lens[i].
 * A length of 0 means the object was not done here and is the host's to deflate:
 * its framed length (header + content) exceeds option "deflate_max_bytes", or its
 * slot was bad (status word, as for the hashing launches).  The arena is the one
 * the hashing entry points take (message words as native dwords, padded blocks);
 * all pointers are device pointers on queue `dev`; asynchronous on `stream`
 * (NULL = the queue's own).  d_order (optional, as for vit_hash_arena_device:
 * object indices by length, vit_bin_order) evens out the launches' tails.
 * Two kernels share the work.  An object of up to 16382 framed bytes (one deflate
 * block; the default "deflate_max_bytes") is deflated by one WORKGROUP with the
 * object, its hash chains and the block's symbols in LDS: one wave builds the
 * chains and runs zlib's decisions, the others walk the chains of the positions
 * ahead of it; for 32 objects at a time the Huffman trees are then built one
 * object per lane and the blocks' bits written by all threads.  Device memory
 * sees the object once, its symbols and tree arrays in a scratch of the
 * workgroups in flight (0.6 GB per queue, allocated on first use and kept), and
 * its stream once.  With
 * "deflate_max_bytes" raised (at most 65536, the range in which zlib's window
 * never slides) the longer objects are deflated one per LANE with 181 KiB of
 * tables each in device memory, which the library then allocates for
 * "deflate_lanes" objects in flight (default 4096: 0.74 GB) and keeps with the
 * queue.  Launches on one queue share the scratch and the tables and are made to
 * wait for one another, whatever streams they are on. */
int vit_deflate_arena_device(vit_hasher *h, int dev, const void *d_arena, uint64_t arena_bytes,
                             const uint64_t *d_offsets, const uint32_t *d_nblocks,
                             const uint32_t *d_order, uint64_t n,
                             uint8_t *d_out, const uint64_t *d_out_offsets, uint32_t *d_out_lens,
                             void *stream);

/* The same for objects in host memory, on the hasher's first device queue: chunks
 * of "deflate_chunk_bytes" packed bytes (default 64 MiB) alternate between two
 * sets of buffers and two streams, so that the host packs one chunk and brings
 * home the streams of another while the device deflates a third.  Objects whose
 * framed length exceeds option "deflate_max_bytes" (default 16382, at most 65536)
 * are not done here.
 * out: host memory of at least vit_deflate_out_bytes() bytes.  Pinned memory
 * (vit_arena_alloc) receives the streams straight from the device; into any
 * other memory they go through a pinned buffer of the library's, from which the
 * hasher's threads copy each stream's own bytes (pages of `out` behind a
 * stream's end are never touched).  On return stream i is out[out_off[i] ..
 * out_off[i] + out_len[i]); out_len[i] = 0 means the object is the caller's to
 * deflate. */
int vit_deflate_objects(vit_hasher *h, const char *type, const uint8_t *const *data,
                        const uint64_t *len, uint64_t n, uint8_t *out, uint64_t out_cap,
                        uint64_t *out_off, uint32_t *out_len);
uint64_t vit_deflate_out_bytes(const vit_hasher *h, const char *type, const uint64_t *len, uint64_t n);

/* Room a stream of that framed length may need (what compressBound() gives);
 * 0 if the length is beyond what the device does. */
uint32_t vit_deflate_bound(uint64_t framed_len);

/* The multi-GPU partition vit_hash_objects applies: queue_of[i] = device queue
 * (0 .. n_queues-1) that hashes object i, balanced by block count (bytes), not
 * by object count; objects that are large against a fair share are placed
 * first, longest to the least loaded queue.  Host-only; needs no device.  The
 * reference is single-threaded and has no counterpart (src/commit.cpp:143). */
int vit_shard_plan(const uint32_t *nblocks, uint64_t n, int n_queues, uint32_t *queue_of);

/* Hash a packed arena that lives in host memory on device queue `dev`
 * (0 <= dev < vit_hasher_device_count).  Offsets and block counts are checked
 * against arena_bytes on the host before anything is launched.  An arena whose
 * slots ascend (what vit_arena_pack writes) goes over in pieces of "chunk_bytes",
 * the copy of one piece beside the kernel of the piece before. */
int vit_hash_arena(vit_hasher *h, int dev, const void *arena, uint64_t arena_bytes,
                   const uint64_t *offsets, const uint32_t *nblocks, uint64_t n,
                   uint8_t *digests);

/* Hash a packed arena that is already resident in device memory; every pointer
 * is a device pointer, d_order may be NULL, digests are written to device
 * memory.  Asynchronous on `stream` (a hipStream_t; NULL = the queue's own
 * stream).  Input that would read outside the arena is not hashed; the kernel
 * raises a status bit instead (vit_device_status).  This is the launch the
 * throughput benchmark times.  With d_order and at least "sched_min_groups"
 * groups it is the software-scheduled launch; of those, a queue expects at
 * most 13 in flight at once (each holds one of the queue's scheduler states). */
int vit_hash_arena_device(vit_hasher *h, int dev, const void *d_arena, uint64_t arena_bytes,
                          const uint64_t *d_offsets, const uint32_t *d_nblocks,
                          const uint32_t *d_order, uint64_t n, uint8_t *d_digests,
                          void *stream);

/* The same launch for callers that cannot allocate (tight serving loops, harnesses
 * that forbid hipMalloc, code that must not perturb the device allocator): it length-bins
 * the batch on the device and hashes it in that order, like vit_bin_order_device
 * followed by vit_hash_arena_device, but calls no hipMalloc/hipFree or any
 * allocating routine.  The order array, the binning counters and the status
 * word live in a file-scope device array inside the library: about 8 MiB that
 * every GPU the library runs on holds until the process ends.  Limit: n <=
 * VIT_STATIC_MAX_OBJECTS per call (VIT_ERR_RANGE beyond, nothing launched).
 * All calls on one GPU share that GPU's array, so the library runs them one after
 * the other: each call's launches wait, on the device, for the previous call's
 * (through one event per GPU, made when the GPU's first queue is; a lock keeps two
 * host threads' calls from interleaving), whatever hasher, queue or stream they
 * come through.  Calls on one GPU therefore never overlap each other; they
 * overlap everything else.  A stream under graph capture is not supported: the
 * event that orders the calls would be captured with it.  Digests are those of
 * vit_hash_arena_device.  Asynchronous on `stream`; status bits are read and
 * cleared by vit_device_status.
 * No counterpart in the reference (its SHA1() call allocates nothing and has no
 * device state, src/commit.cpp:93-94). */
#define VIT_STATIC_MAX_OBJECTS (1u << 21)
int vit_hash_arena_device_static(vit_hasher *h, int dev, const void *d_arena, uint64_t arena_bytes,
                                 const uint64_t *d_offsets, const uint32_t *d_nblocks, uint64_t n,
                                 uint8_t *d_digests, void *stream);

/* Wait for queue `dev`'s own streams (a launch put on a caller's stream is the
 * caller's to wait for first), then read and clear its status word (0 = clean;
 * 1 out of bounds, 2 misaligned offset, 4 bad order entry).  No other stream of
 * the device is stalled. */
int vit_device_status(vit_hasher *h, int dev, uint32_t *status);

#ifdef __cplusplus
}
#endif
#endif /* VIT_HASH_H */
