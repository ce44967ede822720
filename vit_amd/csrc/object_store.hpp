// object_store.hpp — the reference's object-store interface over the batched
// GPU hasher.
//
// The free functions at the bottom carry the reference's names, signatures and
// conventions (reference src/commit.hpp:8-18): ids are 40 lowercase hex
// characters, failure is the empty string with a message on std::cerr, objects
// go to the CWD-relative ".git/objects/aa/bbbb…" as zlib streams of
// "<type> <len>\0" + content.  What differs is where the SHA-1 runs (one GPU
// batch through include/vit_hash.h instead of one OpenSSL call per object) and
// that writeTree walks first, hashes every blob of the worktree in one batch,
// then assembles the trees bottom-up, one batch per depth level, while a host
// thread pool deflates and stores the objects.
#pragma once
#include <atomic>
#include <cstdint>
#include <ctime>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "work_pool.h"

struct vit_hasher;
namespace vit { struct GpuApi; }

/* ---------- data structures (reference src/commit.hpp:31-35) ---------- */
struct TreeEntry {
    std::string mode;
    std::string hash;
    std::string filename;
};

namespace vit {

struct StoreStats {
    uint64_t device_deflated = 0, blobs = 0, trees = 0, bytes = 0;
    uint64_t skipped = 0;                      // objects already in the store (not deflated or written again)
    uint64_t device_rejected = 0;              // streams the device made that failed the store's checks (deflated again by host threads)
    uint64_t gpu_hashed = 0, host_hashed = 0;  // ids computed by the device / by the store's own threads (see setHostCrossoverBytes)
    bool ids_on_host = false;                  // writeTree: which of the two the worktree went to
    uint64_t crossover_bytes = 0;              // ... and the bound it was compared with (0: so small that none was worked out)
    // writeTree overlaps its stages: read_s = what the hashing thread waited for file reads, hash_s = GPU batches
    // (pack + copy + kernel + read-back), deflate_store_s = from the first batch's hand-over to the last object on disk
    double init_s = 0, walk_s = 0, read_s = 0, hash_s = 0, device_deflate_s = 0, deflate_store_s = 0, trees_s = 0, tree_hash_s = 0, total_s = 0;
};

class ObjectStore {
public:
    // objects_dir: where loose objects go (reference: ".git/objects", src/commit.cpp:83,108).
    // devices: HIP ordinals for the hasher (empty = every visible GPU).
    explicit ObjectStore(std::string objects_dir = ".git/objects", std::vector<int> devices = {});
    ~ObjectStore();
    ObjectStore(const ObjectStore&) = delete;
    ObjectStore& operator=(const ObjectStore&) = delete;

    // false (with the reason on std::cerr) when no GPU hasher could be created.  The hasher is created by
    // the first call that needs the device; readObject / readObjectContent never do, and neither does a
    // call small enough for the store's own threads (setHostCrossoverBytes).
    bool ok() { return ensureHasher(); }

    // reference src/commit.cpp:88-114 / :116
    std::string writeObject(const std::string& type, const std::string& content);
    std::string writeBlob(const std::string& content) { return writeObject("blob", content); }
    // Many objects of one type in one GPU batch; ids in input order, {} on failure.
    std::vector<std::string> writeObjects(const std::string& type, const std::vector<std::string>& contents);

    // reference src/commit.cpp:139-173, batched (same ids, same stored objects).
    std::string writeTree(const std::string& directoryPath);

    // reference src/commit.cpp:263-276; `now` replaces its two time(nullptr) calls.
    std::string writeCommit(const std::string& treeHash, const std::string& parentHash, const std::string& message,
                            const std::string& author, const std::string& email, std::time_t now = std::time(nullptr));

    // reference src/commit.cpp:120-136
    std::string readObject(const std::string& hash) const;
    std::string readObjectContent(const std::string& hash) const;

    // Hash without storing (ids only); raw 20-byte digests appended to out.
    bool hashObjects(const std::string& type, const std::vector<const std::string*>& contents, std::vector<uint8_t>& out);

    void setStoreObjects(bool on) { store_ = on; }          // false: compute ids only (benchmarks)
    // true (default): an object whose file already exists is not deflated and written again — its name
    // is the hash of its content, so the bytes would be the same.  false: always rewrite, as the
    // reference does (src/commit.cpp:108-111).
    void setSkipExisting(bool on) { skip_existing_ = on; }
    // false (default): writeTree builds the tree the REFERENCE builds (src/commit.cpp:139-173), which differs
    // from Git's where a directory name is a prefix of a sibling's, for empty directories, executables and
    // symlinks.  true: Git's own rules instead: directories sort as "name/", empty directories are left out,
    // executables get mode 100755, symlinks are stored as 120000 blobs holding the link target, other special
    // files are ignored; the id is then what `git add -A && git write-tree` gives on a worktree without ignore rules.
    void setGitTreeRules(bool on) { git_rules_ = on; }
    // zlib level of the stored objects.  -1 (default) = what the reference's compress() uses (src/commit.cpp:100): the
    // stored files are then byte for byte the reference's.  0..9: any other level gives files that differ from the
    // reference's in their bytes but hold the same object (they inflate to the same "<type> <len>\0" + content, so the
    // reference's readObject and git read them); level 1 deflates about three times faster, level 0 stores.
    void setDeflateLevel(int level) { deflate_level_ = level < -1 ? -1 : (level > 9 ? 9 : level); }
    // Where the small objects of a write-tree are deflated.  true (default): batches that hold at least 16384 objects
    // of up to 16 KiB (one deflate block) are deflated on the GPU, by the kernel that writes the zlib library's own bytes
    // (vit_deflate_objects; their longer objects, and smaller batches, still go to host threads), so that the host
    // threads are left with creating files.  Only at the default level: any other level is the host's.
    void setDeviceDeflate(bool on) { device_deflate_ = on; }
    // While the store has no GPU hasher yet, a call (a writeTree: the whole worktree) of fewer bytes than this has
    // its ids computed by the store's own threads and starts no HIP runtime.  Default: what one host core hashes in the
    // 40 ms the runtime takes to start, measured on first need (about 100 MB with SHA extensions).  0: every call goes
    // to the device.  $VIT_HOST_CROSSOVER_BYTES sets it for the default store.
    void setHostCrossoverBytes(uint64_t bytes) { crossover_bytes_ = bytes; crossover_set_ = true; }
    void setIoThreads(int n) { io_threads_ = n < 1 ? 1 : n; }
    void setBatchBytes(uint64_t b) { batch_bytes_ = b < 1 ? 1 : b; }
    const StoreStats& lastStats() const { return stats_; }
    vit_hasher* hasher() { return ensureHasher() ? hasher_ : nullptr; }
    std::string objectPath(const std::string& hash) const;

private:
    struct StoreOutcome { std::atomic<int> failed{0}; std::atomic<uint64_t> skipped{0}, rejected{0}; };   // failed: 1 compression, 2 write
    enum class Side { GPU, HOST };
    Side sideFor(uint64_t bytes);                        // who computes the ids of a call of that many bytes
    uint64_t hostCrossoverBytes();
    void hashOnHost(const std::string& type, const std::vector<const uint8_t*>& ptr, const std::vector<uint64_t>& len, uint8_t* out);
    bool ensureHasher();
    bool ensureDeflateHasher();
    void tuneHasher();                                   // per call: which objects may leave the GPU for host threads
    WorkPool& pool();
    // Streams made on the device for a batch: stream i = buf[off[i] .. off[i] + len[i]); len[i] == 0: none.
    struct DeviceStreams { const uint8_t* buf; const uint64_t* off; const uint32_t* len; };
    WorkPool::Ticket storeAsync(const std::string& type, const std::vector<std::string>& contents,
                                const std::vector<uint8_t>& raw, StoreOutcome& outcome, DeviceStreams z);
    bool storeFinished(StoreOutcome& outcome);

    std::string objects_dir_;
    std::vector<int> devices_;
    std::unique_ptr<WorkPool> pool_;
    bool hasher_failed_ = false;
    const GpuApi* gpu_ = nullptr;                         // libvit_hash.so, loaded with the first hasher
    vit_hasher* hasher_ = nullptr;
    Side tree_side_ = Side::GPU;                          // during a writeTree: where its ids are computed
    uint64_t crossover_bytes_ = 0;
    bool crossover_set_ = false;
    vit_hasher* deflate_hasher_ = nullptr;                // second set of streams and staging, for the device deflate
    bool deflate_hasher_failed_ = false;
    bool store_ = true;
    bool skip_existing_ = true;
    bool git_rules_ = false;
    int deflate_level_ = -1;
    bool device_deflate_ = true;
    int io_threads_ = 8;
    // Blobs per hashing batch, by bytes.  128 MiB: a dozen or more batches on a worktree of gigabytes, so that
    // reading, hashing and storing overlap, and three batches of contents in memory at most.
    uint64_t batch_bytes_ = 128ull << 20;
    StoreStats stats_;
    std::atomic<uint64_t> store_calls_{0};
    bool in_write_tree_ = false;
};

// Version of the zlib this program deflates with on the host (zlibVersion()).
const char* zlibVersionString();

// The store behind the free functions below: ".git/objects" relative to the
// current directory, devices from $VIT_DEVICES ("0,1,…"; unset = all).
ObjectStore& defaultStore();

}  // namespace vit

/* ---------- Low-level object helpers (reference src/commit.hpp:8-18) ---------- */
std::string hashToHexString(const unsigned char* hash);
std::string hexStringToBinary(const std::string& hexString);
std::string binaryToHexString(const std::string& binary);
std::string getObjectPath(const std::string& hash);

std::string writeObject(const std::string& type, const std::string& content);
std::string writeBlob(const std::string& content);
std::string writeTree(const std::string& directoryPath);

std::string readObject(const std::string& hash);
std::string readObjectContent(const std::string& hash);

std::string writeCommit(const std::string& treeHash, const std::string& parentHash, const std::string& message,
                        const std::string& author, const std::string& email);
