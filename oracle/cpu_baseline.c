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
#define _XOPEN_SOURCE 700
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

/* libcrypto's EVP interface, for the variant that keeps one digest context per thread (below). */
typedef struct {
    void *(*md_fetch)(void *, const char *, const char *);
    void *(*ctx_new)(void);
    void (*ctx_free)(void *);
    int (*init)(void *, const void *, void *);
    int (*update)(void *, const void *, size_t);
    int (*final)(void *, unsigned char *, unsigned int *);
    void *sha1_md;
} evp_t;

static const evp_t *resolve_evp(void)
{
    static evp_t e;
    static int tried = 0, ok = 0;
    if (tried) return ok ? &e : NULL;
    tried = 1;
    static const char *names[] = { "libcrypto.so.3", "libcrypto.so.1.1", "libcrypto.so", NULL };
    for (int i = 0; names[i] && !ok; ++i) {
        void *h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!h) continue;
        *(void **)&e.md_fetch = dlsym(h, "EVP_MD_fetch");
        *(void **)&e.ctx_new = dlsym(h, "EVP_MD_CTX_new");
        *(void **)&e.ctx_free = dlsym(h, "EVP_MD_CTX_free");
        *(void **)&e.init = dlsym(h, "EVP_DigestInit_ex");
        *(void **)&e.update = dlsym(h, "EVP_DigestUpdate");
        *(void **)&e.final = dlsym(h, "EVP_DigestFinal_ex");
        if (e.md_fetch && e.ctx_new && e.ctx_free && e.init && e.update && e.final) {
            e.sha1_md = e.md_fetch(NULL, "SHA1", NULL);
            ok = e.sha1_md != NULL;
        }
    }
    return ok ? &e : NULL;
}

typedef struct {
    const uint8_t *blobs; const uint64_t *offsets; const uint64_t *lens;
    uint64_t begin, end; uint8_t *digests; sha1_fn fn; const evp_t *evp; const char *type;
    int reps; pthread_barrier_t *gate;
} shard_t;

static void *shard_main(void *arg)
{
    shard_t *s = (shard_t *)arg;
    void *ctx = s->evp ? s->evp->ctx_new() : NULL;
    if (s->gate) pthread_barrier_wait(s->gate);                            /* every thread starts together */
    for (int r = 0; r < s->reps; ++r)
    for (uint64_t i = s->begin; i < s->end; ++i) {
        char hdr[32];
        size_t hl = vit_oracle_header(s->type, s->lens[i], hdr);          /* :90 */
        size_t full_len = hl + (size_t)s->lens[i];
        unsigned char *full = (unsigned char *)malloc(full_len);           /* :91 */
        memcpy(full, hdr, hl);
        memcpy(full + hl, s->blobs + s->offsets[i], (size_t)s->lens[i]);
        if (ctx) {
            unsigned int dl = 0;
            s->evp->init(ctx, s->evp->sha1_md, NULL);
            s->evp->update(ctx, full, full_len);
            s->evp->final(ctx, s->digests + 20 * i, &dl);
        }
        else if (s->fn) s->fn(full, full_len, s->digests + 20 * i);        /* :94 */
        else vit_oracle_sha1(full, full_len, s->digests + 20 * i);
        free(full);
    }
    if (s->gate) pthread_barrier_wait(s->gate);                            /* ... and the clock stops when the last has ended */
    if (ctx) s->evp->ctx_free(ctx);
    return NULL;
}

/*
 * Hash n objects (content i = blobs[offsets[i] .. +lens[i])) the way the
 * reference does, `reps` times over, on `threads` host threads.  Returns elapsed wall
 * seconds for all the passes; *used_libcrypto tells which SHA-1 ran.  digests: n*20 bytes.
 * The threads exist before the clock starts: they are released together by a
 * barrier and the clock stops at a second barrier, so neither their creation
 * nor their joining is in the time.
 * how: 0 = the reference's call, libcrypto's one-shot SHA1() per object (the oracle's own
 *          SHA-1 when there is no libcrypto, or want_libcrypto is 0);
 *      1 = the same loop with one digest context per thread and the algorithm looked up once
 *          (EVP_MD_fetch, EVP_DigestInit_ex/Update/Final): OpenSSL 3's one-shot SHA1() looks
 *          the algorithm up on every call, under a lock that threads contend for, so the
 *          reference's loop does not scale over cores as its arithmetic would.  Falls back
 *          to 0 where the EVP entry points are missing.
 */
double vit_oracle_bench_hash_loop_mt(const char *type, const uint8_t *blobs,
                                     const uint64_t *offsets, const uint64_t *lens,
                                     uint64_t n, int threads, int want_libcrypto, int how, int reps,
                                     uint8_t *digests, int *used_libcrypto)
{
    sha1_fn fn = want_libcrypto ? resolve_libcrypto_sha1() : NULL;
    const evp_t *evp = (how == 1 && want_libcrypto) ? resolve_evp() : NULL;
    if (used_libcrypto) *used_libcrypto = evp ? 2 : fn != NULL;
    if (threads < 1) threads = 1;
    if (reps < 1) reps = 1;
    if ((uint64_t)threads > n && n) threads = (int)n;
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    shard_t *sh = (shard_t *)malloc(sizeof(shard_t) * (size_t)threads);
    pthread_barrier_t gate;
    pthread_barrier_init(&gate, NULL, (unsigned)threads + 1);
    for (int t = 0; t < threads; ++t) {
        sh[t] = (shard_t){ blobs, offsets, lens, n * (uint64_t)t / (uint64_t)threads,
                           n * (uint64_t)(t + 1) / (uint64_t)threads, digests, fn, evp, type, reps, &gate };
        pthread_create(&tid[t], NULL, shard_main, &sh[t]);
    }
    struct timespec t0, t1;
    pthread_barrier_wait(&gate);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    pthread_barrier_wait(&gate);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
    pthread_barrier_destroy(&gate);
    free(tid); free(sh);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* One pass on the calling thread (the reference's own, single-threaded form), or on `threads`. */
double vit_oracle_bench_hash_loop(const char *type, const uint8_t *blobs,
                                  const uint64_t *offsets, const uint64_t *lens,
                                  uint64_t n, int threads, int want_libcrypto,
                                  uint8_t *digests, int *used_libcrypto)
{
    if (threads > 1)
        return vit_oracle_bench_hash_loop_mt(type, blobs, offsets, lens, n, threads, want_libcrypto, 0, 1, digests, used_libcrypto);
    sha1_fn fn = want_libcrypto ? resolve_libcrypto_sha1() : NULL;
    if (used_libcrypto) *used_libcrypto = fn != NULL;
    shard_t one = { blobs, offsets, lens, 0, n, digests, fn, NULL, type, 1, NULL };
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    shard_main(&one);
    clock_gettime(CLOCK_MONOTONIC, &t1);
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

/* The same for objects of different lengths: object i (stream index first+i) at dst + offsets[i], lens[i] bytes. */
typedef struct { uint64_t seed, first, begin, end; const uint64_t *lens, *offsets; uint8_t *dst; } fillv_t;
static void *fillv_main(void *arg)
{
    fillv_t *f = (fillv_t *)arg;
    for (uint64_t i = f->begin; i < f->end; ++i)
        vit_oracle_synth_fill(f->seed, f->first + i, f->dst + f->offsets[i], f->lens[i]);
    return NULL;
}
void vit_oracle_synth_fill_var(uint64_t seed, uint64_t first, uint64_t n, const uint64_t *lens,
                               const uint64_t *offsets, uint8_t *dst, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t tid[256]; fillv_t f[256];
    /* ranges of about equal bytes: offsets[] ascend */
    uint64_t total = n ? offsets[n - 1] + lens[n - 1] : 0, at = 0;
    for (int t = 0; t < threads; ++t) {
        uint64_t end = at;
        uint64_t goal = total / (uint64_t)threads * (uint64_t)(t + 1);
        while (end < n && (t == threads - 1 || offsets[end] < goal)) ++end;
        f[t] = (fillv_t){ seed, first, at, end, lens, offsets, dst };
        pthread_create(&tid[t], NULL, fillv_main, &f[t]);
        at = end;
    }
    for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
}
