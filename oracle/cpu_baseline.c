/*
 * cpu_baseline.c — the reference's per-object hashing loop, timed on host cores.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg).
 *
 * Restates src/commit.cpp:90-94 of the reference: for every object build
 * header = type + ' ' + to_string(size) + '\0', concatenate header + content
 * into one fresh buffer (the reference's `full` std::string, :91), and call the
 * one-shot SHA1() on it.  SHA1() is resolved at run time from the system's
 * libcrypto (the same third-party function the reference links, :16,94) with
 * dlopen, so no OpenSSL headers are needed; when libcrypto is absent the
 * oracle's own FIPS 180-4 restatement is timed instead and the caller is told.
 * The reference is single-threaded; `threads` > 1 shards the objects statically
 * to show what all host cores could do.
 */
#define _POSIX_C_SOURCE 200809L
#define _DEFAULT_SOURCE
#include "vit_oracle.h"

#include <dlfcn.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef unsigned char *(*sha1_fn)(const unsigned char *, size_t, unsigned char *);

static sha1_fn resolve_libcrypto_sha1(void)
{
    static const char *names[] = { "libcrypto.so.3", "libcrypto.so.1.1", "libcrypto.so", NULL };
    for (int i = 0; names[i]; ++i) {
        void *h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!h) continue;
        sha1_fn f = (sha1_fn)dlsym(h, "SHA1");
        if (f) return f;
    }
    return NULL;
}

/* 1 if libcrypto's SHA1 can be resolved on this machine. */
int vit_oracle_have_libcrypto(void) { return resolve_libcrypto_sha1() != NULL; }

/* One-shot libcrypto SHA1 (for cross-checking the restatement); -1 if absent. */
int vit_oracle_libcrypto_sha1(const void *data, size_t len, uint8_t out[20])
{
    sha1_fn f = resolve_libcrypto_sha1();
    if (!f) return -1;
    f((const unsigned char *)data, len, out);
    return 0;
}

typedef struct {
    const uint8_t *blobs; const uint64_t *offsets; const uint64_t *lens;
    uint64_t begin, end; uint8_t *digests; sha1_fn fn; const char *type;
} shard_t;

static void *shard_main(void *arg)
{
    shard_t *s = (shard_t *)arg;
    for (uint64_t i = s->begin; i < s->end; ++i) {
        char hdr[32];
        size_t hl = vit_oracle_header(s->type, s->lens[i], hdr);          /* :90 */
        size_t full_len = hl + (size_t)s->lens[i];
        unsigned char *full = (unsigned char *)malloc(full_len);           /* :91 */
        memcpy(full, hdr, hl);
        memcpy(full + hl, s->blobs + s->offsets[i], (size_t)s->lens[i]);
        if (s->fn) s->fn(full, full_len, s->digests + 20 * i);             /* :94 */
        else vit_oracle_sha1(full, full_len, s->digests + 20 * i);
        free(full);
    }
    return NULL;
}

/*
 * Hash n objects (content i = blobs[offsets[i] .. +lens[i])) the way the
 * reference does, on `threads` host threads.  Returns elapsed wall seconds;
 * *used_libcrypto tells which SHA-1 ran.  digests: n*20 bytes.
 */
double vit_oracle_bench_hash_loop(const char *type, const uint8_t *blobs,
                                  const uint64_t *offsets, const uint64_t *lens,
                                  uint64_t n, int threads, int want_libcrypto,
                                  uint8_t *digests, int *used_libcrypto)
{
    sha1_fn fn = want_libcrypto ? resolve_libcrypto_sha1() : NULL;
    if (used_libcrypto) *used_libcrypto = fn != NULL;
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > n && n) threads = (int)n;
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    shard_t *sh = (shard_t *)malloc(sizeof(shard_t) * (size_t)threads);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; ++t) {
        sh[t] = (shard_t){ blobs, offsets, lens, n * (uint64_t)t / (uint64_t)threads,
                           n * (uint64_t)(t + 1) / (uint64_t)threads, digests, fn, type };
        if (threads == 1) shard_main(&sh[t]);
        else pthread_create(&tid[t], NULL, shard_main, &sh[t]);
    }
    if (threads > 1) for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(tid); free(sh);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* Fill n blobs of len_each synthetic bytes (blob i = stream index first+i). */
typedef struct { uint64_t seed, first, begin, end, len; uint8_t *dst; } fill_t;
static void *fill_main(void *arg)
{
    fill_t *f = (fill_t *)arg;
    for (uint64_t i = f->begin; i < f->end; ++i)
        vit_oracle_synth_fill(f->seed, f->first + i, f->dst + i * f->len, f->len);
    return NULL;
}
void vit_oracle_synth_fill_many(uint64_t seed, uint64_t first, uint64_t n, uint64_t len_each,
                                uint8_t *dst, int threads)
{
    if (threads < 1) threads = 1;
    pthread_t tid[256]; fill_t f[256];
    if (threads > 256) threads = 256;
    for (int t = 0; t < threads; ++t) {
        f[t] = (fill_t){ seed, first, n * (uint64_t)t / (uint64_t)threads,
                         n * (uint64_t)(t + 1) / (uint64_t)threads, len_each, dst };
        pthread_create(&tid[t], NULL, fill_main, &f[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
}
