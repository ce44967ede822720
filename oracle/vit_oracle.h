/*
 * vit_oracle.h — CPU restatement of the reference's content-addressing path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing that ships (the HIP library, the C++ object
 * store, the CLI) links or calls this file; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg do, and there only as the checker.
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference).  The SHA-1 compression function itself is NOT in the
 * reference: the reference calls OpenSSL libcrypto SHA1() (src/commit.cpp:16,94;
 * dependency "openssl" in vcpkg.json:5, version unpinned).  It is restated here
 * from the published algorithm, FIPS 180-4 section 6.1, and pinned by the FIPS
 * known-answer vectors, Git's well-known ids, real `git hash-object`, and the
 * outputs of the reference's own commit.cpp compiled into oracle/_ref
 * (tests/golden/).
 */
#ifndef VIT_ORACLE_H
#define VIT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint32_t h[5];
    uint64_t nbytes;      /* total message bytes absorbed so far */
    uint8_t  buf[64];     /* partial block */
} vit_oracle_sha1_ctx;

/* FIPS 180-4 6.1 (what OpenSSL SHA1() computes at src/commit.cpp:94). */
void vit_oracle_sha1_init(vit_oracle_sha1_ctx *c);
void vit_oracle_sha1_update(vit_oracle_sha1_ctx *c, const void *data, size_t len);
void vit_oracle_sha1_final(vit_oracle_sha1_ctx *c, uint8_t out[20]);
void vit_oracle_sha1(const void *data, size_t len, uint8_t out[20]);

/* src/commit.cpp:90: header = type + ' ' + to_string(size) + '\0'.
 * Writes the header (NUL included) into hdr (>= 32 bytes), returns its length. */
size_t vit_oracle_header(const char *type, uint64_t size, char hdr[32]);

/* src/commit.cpp:90-94: SHA1(header + content) -> 20 raw bytes. */
void vit_oracle_object_id(const char *type, const void *content, uint64_t len,
                          uint8_t out[20]);

/* Same digest for an object whose content is `len` copies of the byte `fill`,
 * streamed so a >4 GiB length needs no buffer (checks 64-bit length framing;
 * the reference would build the whole std::string, src/commit.cpp:90-94). */
void vit_oracle_object_id_fill(const char *type, uint8_t fill, uint64_t len,
                               uint8_t out[20]);

/* src/commit.cpp:52-59 hashToHexString: 20 bytes -> 40 lowercase hex + NUL. */
void vit_oracle_hex(const uint8_t in[20], char out[41]);
/* src/commit.cpp:61-70 hexStringToBinary: 40 hex -> 20 bytes; 0 ok, -1 bad hex. */
int  vit_oracle_unhex(const char *hex, uint8_t out[20]);
/* src/commit.cpp:81-84 getObjectPath: ".git/objects/aa/bbbb..." into out (>= 64). */
void vit_oracle_object_path(const char *hex, char out[64]);

/* One tree entry as src/commit.hpp:31-35 carries it (hash kept raw here). */
typedef struct {
    const char *mode;      /* "100644" or "40000"  (src/commit.cpp:154,158) */
    const char *name;
    uint8_t     id[20];
} vit_oracle_tree_entry;

/* src/commit.cpp:164-172: sort by name (unsigned byte order), serialise
 * mode ' ' name '\0' raw20, hash as "tree".  Sorts `entries` in place.
 * If content_out != NULL, *content_out receives a malloc'd copy of the tree
 * content (caller frees) and *content_len its length. */
void vit_oracle_tree_id(vit_oracle_tree_entry *entries, size_t n, uint8_t out[20],
                        uint8_t **content_out, size_t *content_len);

/* src/commit.cpp:139-173 writeTree: recursive walk of `dir`, skipping entries
 * named ".git" (:145), regular files -> blob/100644 (:150-154), directories ->
 * recurse/40000 (:156-158); stat() follows symlinks as std::filesystem's
 * is_regular_file()/is_directory() do.  Entries that are neither are an error
 * (:160, empty hash -> return {}).  If objects_dir != NULL every object is also
 * deflated and stored under it (src/commit.cpp:98-111).  Returns 0 and the root
 * id, or -1 on failure.  n_blobs/n_trees (optional) count objects hashed. */
int vit_oracle_write_tree(const char *dir, const char *objects_dir, uint8_t out[20],
                          uint64_t *n_blobs, uint64_t *n_trees);

/* src/commit.cpp:98-111: zlib compress() of header+content (default level),
 * create "<objects_dir>/aa", write "<objects_dir>/aa/bbbb...".  0 ok, -1 error. */
int vit_oracle_store_object(const char *objects_dir, const char *type,
                            const void *content, uint64_t len, uint8_t id_out[20]);

/* src/commit.cpp:263-276 writeCommit text with the clock injected (the
 * reference calls time(nullptr) twice, :272-273).  Returns malloc'd text. */
char *vit_oracle_commit_text(const char *tree_hex, const char *parent_hex,
                             const char *message, const char *author,
                             const char *email, long long t_author,
                             long long t_committer, size_t *len_out);

/* Synthetic-input generator shared by tests, bench and the library's own
 * generator (counter-based splitmix64; not from the reference, which has no
 * benchmark inputs).  Byte j of blob `index` under `seed`. */
void vit_oracle_synth_fill(uint64_t seed, uint64_t index, uint8_t *dst, uint64_t len);

#ifdef __cplusplus
}
#endif
#endif
