/*
 * vit_oracle.c — CPU restatement of the reference's content-addressing path.
 * TEST INFRASTRUCTURE ONLY (see vit_oracle.h).  Plain C99 + zlib + POSIX.
 */
#define _POSIX_C_SOURCE 200809L
#define _DEFAULT_SOURCE
#include "vit_oracle.h"

#include <dirent.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <zlib.h>

/* ------------------------------------------------------------------ SHA-1 --
 * FIPS 180-4 section 6.1.2, written in the textbook form (80-entry schedule,
 * four explicit f/K phases) so that it shares no structure with the GPU
 * kernel's rolling 16-word schedule.  Replaces OpenSSL SHA1() as called at
 * src/commit.cpp:93-94.
 */
static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

static void sha1_block(uint32_t h[5], const uint8_t p[64])
{
    uint32_t w[80];
    for (int t = 0; t < 16; ++t)
        w[t] = ((uint32_t)p[4 * t] << 24) | ((uint32_t)p[4 * t + 1] << 16) |
               ((uint32_t)p[4 * t + 2] << 8) | (uint32_t)p[4 * t + 3];
    for (int t = 16; t < 80; ++t)
        w[t] = rotl(w[t - 3] ^ w[t - 8] ^ w[t - 14] ^ w[t - 16], 1);

    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
    for (int t = 0; t < 80; ++t) {
        uint32_t f, k;
        if (t < 20)      { f = (b & c) | (~b & d);            k = 0x5A827999u; }
        else if (t < 40) { f = b ^ c ^ d;                     k = 0x6ED9EBA1u; }
        else if (t < 60) { f = (b & c) | (b & d) | (c & d);   k = 0x8F1BBCDCu; }
        else             { f = b ^ c ^ d;                     k = 0xCA62C1D6u; }
        uint32_t tmp = rotl(a, 5) + f + e + k + w[t];
        e = d; d = c; c = rotl(b, 30); b = a; a = tmp;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e;
}

void vit_oracle_sha1_init(vit_oracle_sha1_ctx *c)
{
    c->h[0] = 0x67452301u; c->h[1] = 0xEFCDAB89u; c->h[2] = 0x98BADCFEu;
    c->h[3] = 0x10325476u; c->h[4] = 0xC3D2E1F0u;
    c->nbytes = 0;
}

void vit_oracle_sha1_update(vit_oracle_sha1_ctx *c, const void *data, size_t len)
{
    const uint8_t *p = (const uint8_t *)data;
    size_t fill = (size_t)(c->nbytes & 63);
    c->nbytes += len;
    if (fill) {
        size_t take = 64 - fill;
        if (take > len) take = len;
        memcpy(c->buf + fill, p, take);
        p += take; len -= take; fill += take;
        if (fill < 64) return;
        sha1_block(c->h, c->buf);
    }
    while (len >= 64) { sha1_block(c->h, p); p += 64; len -= 64; }
    if (len) memcpy(c->buf, p, len);
}

void vit_oracle_sha1_final(vit_oracle_sha1_ctx *c, uint8_t out[20])
{
    /* FIPS 180-4 5.1.1: append 0x80, zeros to 56 mod 64, 64-bit big-endian bit length. */
    uint64_t bits = c->nbytes * 8u;
    size_t fill = (size_t)(c->nbytes & 63);
    c->buf[fill++] = 0x80;
    if (fill > 56) {
        memset(c->buf + fill, 0, 64 - fill);
        sha1_block(c->h, c->buf);
        fill = 0;
    }
    memset(c->buf + fill, 0, 56 - fill);
    for (int i = 0; i < 8; ++i) c->buf[56 + i] = (uint8_t)(bits >> (56 - 8 * i));
    sha1_block(c->h, c->buf);
    for (int i = 0; i < 5; ++i) {
        out[4 * i]     = (uint8_t)(c->h[i] >> 24);
        out[4 * i + 1] = (uint8_t)(c->h[i] >> 16);
        out[4 * i + 2] = (uint8_t)(c->h[i] >> 8);
        out[4 * i + 3] = (uint8_t)(c->h[i]);
    }
}

void vit_oracle_sha1(const void *data, size_t len, uint8_t out[20])
{
    vit_oracle_sha1_ctx c;
    vit_oracle_sha1_init(&c);
    vit_oracle_sha1_update(&c, data, len);
    vit_oracle_sha1_final(&c, out);
}

/* --------------------------------------------------------- object framing -- */

/* src/commit.cpp:90 — type + ' ' + std::to_string(content.size()) + '\0' */
size_t vit_oracle_header(const char *type, uint64_t size, char hdr[32])
{
    int n = snprintf(hdr, 32, "%s %llu", type, (unsigned long long)size);
    return (size_t)n + 1;   /* the terminating NUL is part of the header */
}

/* src/commit.cpp:90-94 */
void vit_oracle_object_id(const char *type, const void *content, uint64_t len,
                          uint8_t out[20])
{
    char hdr[32];
    size_t hl = vit_oracle_header(type, len, hdr);
    vit_oracle_sha1_ctx c;
    vit_oracle_sha1_init(&c);
    vit_oracle_sha1_update(&c, hdr, hl);
    vit_oracle_sha1_update(&c, content, (size_t)len);
    vit_oracle_sha1_final(&c, out);
}

void vit_oracle_object_id_fill(const char *type, uint8_t fill, uint64_t len,
                               uint8_t out[20])
{
    char hdr[32];
    size_t hl = vit_oracle_header(type, len, hdr);
    uint8_t local[1 << 12];
    memset(local, fill, sizeof local);
    vit_oracle_sha1_ctx c;
    vit_oracle_sha1_init(&c);
    vit_oracle_sha1_update(&c, hdr, hl);
    while (len) {
        size_t take = len > sizeof local ? sizeof local : (size_t)len;
        vit_oracle_sha1_update(&c, local, take);
        len -= take;
    }
    vit_oracle_sha1_final(&c, out);
}

/* src/commit.cpp:52-59 */
void vit_oracle_hex(const uint8_t in[20], char out[41])
{
    static const char d[] = "0123456789abcdef";
    for (int i = 0; i < 20; ++i) { out[2 * i] = d[in[i] >> 4]; out[2 * i + 1] = d[in[i] & 15]; }
    out[40] = 0;
}

static int hexval(char ch)
{
    if (ch >= '0' && ch <= '9') return ch - '0';
    if (ch >= 'a' && ch <= 'f') return ch - 'a' + 10;
    if (ch >= 'A' && ch <= 'F') return ch - 'A' + 10;
    return -1;
}

/* src/commit.cpp:61-70 */
int vit_oracle_unhex(const char *hex, uint8_t out[20])
{
    for (int i = 0; i < 20; ++i) {
        int hi = hexval(hex[2 * i]), lo = hi < 0 ? -1 : hexval(hex[2 * i + 1]);
        if (hi < 0 || lo < 0) return -1;
        out[i] = (uint8_t)(hi * 16 + lo);
    }
    return 0;
}

/* src/commit.cpp:81-84 */
void vit_oracle_object_path(const char *hex, char out[64])
{
    snprintf(out, 64, ".git/objects/%.2s/%s", hex, hex + 2);
}

/* ------------------------------------------------------------------ trees -- */

/* src/commit.cpp:164-165: std::string operator< == lexicographic unsigned bytes,
 * shorter string first on a common prefix. */
static int entry_cmp(const void *pa, const void *pb)
{
    const vit_oracle_tree_entry *a = (const vit_oracle_tree_entry *)pa;
    const vit_oracle_tree_entry *b = (const vit_oracle_tree_entry *)pb;
    size_t la = strlen(a->name), lb = strlen(b->name);
    int r = memcmp(a->name, b->name, la < lb ? la : lb);
    if (r) return r;
    return la < lb ? -1 : la > lb;
}

/* src/commit.cpp:164-172 */
void vit_oracle_tree_id(vit_oracle_tree_entry *entries, size_t n, uint8_t out[20],
                        uint8_t **content_out, size_t *content_len)
{
    qsort(entries, n, sizeof *entries, entry_cmp);
    size_t total = 0;
    for (size_t i = 0; i < n; ++i)
        total += strlen(entries[i].mode) + 1 + strlen(entries[i].name) + 1 + 20;
    uint8_t *buf = (uint8_t *)malloc(total ? total : 1), *p = buf;
    for (size_t i = 0; i < n; ++i) {
        size_t lm = strlen(entries[i].mode), ln = strlen(entries[i].name);
        memcpy(p, entries[i].mode, lm); p += lm;
        *p++ = ' ';
        memcpy(p, entries[i].name, ln); p += ln;
        *p++ = 0;
        memcpy(p, entries[i].id, 20); p += 20;
    }
    vit_oracle_object_id("tree", buf, total, out);
    if (content_out) { *content_out = buf; if (content_len) *content_len = total; }
    else free(buf);
}

static int slurp(const char *path, uint8_t **data, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    size_t cap = 1 << 16, n = 0;
    uint8_t *buf = (uint8_t *)malloc(cap);
    for (;;) {
        if (n == cap) { cap *= 2; buf = (uint8_t *)realloc(buf, cap); }
        size_t r = fread(buf + n, 1, cap - n, f);
        n += r;
        if (r == 0) break;
    }
    fclose(f);
    *data = buf; *len = n;
    return 0;
}

static int mkdir_p(const char *path)
{
    char tmp[4096];
    size_t l = strlen(path);
    if (l >= sizeof tmp) return -1;
    memcpy(tmp, path, l + 1);
    for (char *p = tmp + 1; *p; ++p)
        if (*p == '/') { *p = 0; if (mkdir(tmp, 0777) && errno != EEXIST) return -1; *p = '/'; }
    if (mkdir(tmp, 0777) && errno != EEXIST) return -1;
    return 0;
}

/* src/commit.cpp:98-111 */
int vit_oracle_store_object(const char *objects_dir, const char *type,
                            const void *content, uint64_t len, uint8_t id_out[20])
{
    char hdr[32];
    size_t hl = vit_oracle_header(type, len, hdr);
    size_t full_len = hl + (size_t)len;
    uint8_t *full = (uint8_t *)malloc(full_len);
    memcpy(full, hdr, hl);
    if (len) memcpy(full + hl, content, (size_t)len);
    vit_oracle_sha1(full, full_len, id_out);

    uLongf dst = compressBound((uLong)full_len);
    uint8_t *z = (uint8_t *)malloc(dst);
    int rc = compress(z, &dst, full, (uLong)full_len);   /* :100, default level */
    free(full);
    if (rc != Z_OK) { free(z); return -1; }

    char hex[41], dir[4096], path[4200];
    vit_oracle_hex(id_out, hex);
    snprintf(dir, sizeof dir, "%s/%.2s", objects_dir, hex);
    if (mkdir_p(dir)) { free(z); return -1; }
    snprintf(path, sizeof path, "%s/%s", dir, hex + 2);
    FILE *f = fopen(path, "wb");
    if (!f) { free(z); return -1; }
    fwrite(z, 1, dst, f);
    fclose(f);
    free(z);
    return 0;
}

/* src/commit.cpp:139-173 */
int vit_oracle_write_tree(const char *dir, const char *objects_dir, uint8_t out[20],
                          uint64_t *n_blobs, uint64_t *n_trees)
{
    DIR *d = opendir(dir);
    if (!d) return -1;
    size_t cap = 16, n = 0;
    vit_oracle_tree_entry *ents = (vit_oracle_tree_entry *)malloc(cap * sizeof *ents);
    int rc = 0;
    struct dirent *de;
    while ((de = readdir(d))) {
        const char *name = de->d_name;
        if (!strcmp(name, ".") || !strcmp(name, "..")) continue;   /* directory_iterator skips these */
        if (!strcmp(name, ".git")) continue;                       /* :145 */
        char path[4096];
        snprintf(path, sizeof path, "%s/%s", dir, name);
        struct stat st;
        if (stat(path, &st)) { rc = -1; break; }                   /* follows symlinks, as :150,156 */
        if (n == cap) { cap *= 2; ents = (vit_oracle_tree_entry *)realloc(ents, cap * sizeof *ents); }
        vit_oracle_tree_entry *e = &ents[n];
        if (S_ISREG(st.st_mode)) {                                  /* :150-154 */
            uint8_t *data; size_t len;
            /* :151-152: a file that cannot be opened leaves the reference's stringstream empty, and the empty
             * string is hashed and stored like any other content: the entry points at the empty blob. */
            if (slurp(path, &data, &len)) { data = (uint8_t *)malloc(1); len = 0; }
            if (objects_dir) {
                if (vit_oracle_store_object(objects_dir, "blob", data, len, e->id)) { free(data); rc = -1; break; }
            } else vit_oracle_object_id("blob", data, len, e->id);
            free(data);
            e->mode = "100644";
            if (n_blobs) ++*n_blobs;
        } else if (S_ISDIR(st.st_mode)) {                           /* :156-158 */
            if (vit_oracle_write_tree(path, objects_dir, e->id, n_blobs, n_trees)) { rc = -1; break; }
            e->mode = "40000";
        } else { rc = -1; break; }                                  /* :160 empty hash -> {} */
        e->name = strdup(name);
        ++n;
    }
    closedir(d);
    if (rc == 0) {
        uint8_t *content = NULL; size_t clen = 0;
        vit_oracle_tree_id(ents, n, out, &content, &clen);          /* :164-172 */
        if (objects_dir) {
            uint8_t id2[20];
            if (vit_oracle_store_object(objects_dir, "tree", content, clen, id2)) rc = -1;
        }
        free(content);
        if (n_trees) ++*n_trees;
    }
    for (size_t i = 0; i < n; ++i) free((void *)ents[i].name);
    free(ents);
    return rc;
}

/* src/commit.cpp:263-276 */
char *vit_oracle_commit_text(const char *tree_hex, const char *parent_hex,
                             const char *message, const char *author,
                             const char *email, long long t_author,
                             long long t_committer, size_t *len_out)
{
    size_t cap = strlen(tree_hex) + (parent_hex ? strlen(parent_hex) : 0) +
                 strlen(message) + 2 * (strlen(author) + strlen(email)) + 160;
    char *s = (char *)malloc(cap);
    int n = snprintf(s, cap, "tree %s\n", tree_hex);
    if (parent_hex && *parent_hex) n += snprintf(s + n, cap - n, "parent %s\n", parent_hex);
    n += snprintf(s + n, cap - n, "author %s <%s> %lld\n", author, email, t_author);
    n += snprintf(s + n, cap - n, "committer %s <%s> %lld\n\n", author, email, t_committer);
    n += snprintf(s + n, cap - n, "%s\n", message);
    if (len_out) *len_out = (size_t)n;
    return s;
}

/* -------------------------------------------------------- synthetic bytes --
 * Counter-based splitmix64: word k of blob `index` is the k-th output of a
 * splitmix64 stream whose initial state is seed ^ (index * 0xD1342543DE82EF95);
 * bytes are the little-endian bytes of those words.
 */
static uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void vit_oracle_synth_fill(uint64_t seed, uint64_t index, uint8_t *dst, uint64_t len)
{
    uint64_t key = seed ^ (index * 0xD1342543DE82EF95ull);
    uint64_t k = 0, i = 0;
    while (i < len) {
        uint64_t w = mix64(key + (k + 1) * 0x9E3779B97F4A7C15ull);
        for (int b = 0; b < 8 && i < len; ++b, ++i) dst[i] = (uint8_t)(w >> (8 * b));
        ++k;
    }
}
