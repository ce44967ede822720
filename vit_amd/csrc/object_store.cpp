// object_store.cpp — see object_store.hpp.
#include "object_store.hpp"

#include "../../include/vit_hash.h"
#include "gpu_api.h"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <future>
#include <deque>
#include <condition_variable>
#include <iostream>
#include <mutex>
#include <queue>
#include <sstream>
#include <thread>
#include <cerrno>
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace fs = std::filesystem;

namespace {

using clk = std::chrono::steady_clock;
double since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

// The store's worker threads (file reads, deflate, object writes) are a
// vit::WorkPool: groups of indexed tasks at two priorities, the waiting thread
// helps.  Every worker gives itself a file-descriptor table of its own first:
// open() and close() take a lock on the table, and with one table for the whole
// process sixteen threads reading small files get through fewer files per second
// than four do (measured on tmpfs: 1.0 us per file on 16 threads with the shared
// table, 0.41 us on 8 with private ones).  A worker only ever uses descriptors it
// opened itself.  The new table starts as a COPY of the process's, and a copy of a
// descriptor keeps its file open: a pipe the host program closes later would not
// deliver end-of-file, a lock would stay held, a deleted file would keep its blocks,
// for as long as the pool lives.  So the worker closes every inherited descriptor
// above stderr in its own table at once (that closes only its own references).
thread_local bool tl_private_fd_table = false;
void privateFdTable()
{
#ifdef __linux__
    if (::unshare(CLONE_FILES) != 0) return;
    tl_private_fd_table = true;
#ifdef SYS_close_range
    if (::syscall(SYS_close_range, 3u, ~0u, 0u) == 0) return;
#endif
    // an older kernel: close what /proc lists for this thread (the listing's own descriptor goes last)
    std::vector<int> open_now;
    std::error_code ec;
    for (fs::directory_iterator it("/proc/thread-self/fd", ec), end; !ec && it != end; it.increment(ec)) {
        const int fd = std::atoi(it->path().filename().c_str());
        if (fd > 2) open_now.push_back(fd);
    }
    for (int fd : open_now) ::close(fd);
#endif
}

// Run fn(i) for i in [0, n) on the pool, in slices, and wait.  fn must not throw.
void forEach(vit::WorkPool& pool, size_t n, const std::function<void(size_t)>& fn,
             vit::WorkPool::Priority prio = vit::WorkPool::HIGH)
{
    if (n == 0) return;
    const size_t slices = std::min(n, size_t(pool.threads() + 1) * 4);
    pool.run(prio, slices, [&](size_t s) { for (size_t i = n * s / slices; i < n * (s + 1) / slices; ++i) fn(i); });
}

const char kHex[] = "0123456789abcdef";

std::string toHex(const uint8_t* raw, size_t n)
{
    std::string s(2 * n, '0');
    for (size_t i = 0; i < n; ++i) { s[2 * i] = kHex[raw[i] >> 4]; s[2 * i + 1] = kHex[raw[i] & 15]; }
    return s;
}

int hexNibble(char c)
{
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}

// The zlib stream compress() produces at the default level (reference
// src/commit.cpp:98-105), without what compress() costs per call: it sets up and
// tears down a deflate state (about 270 KB of allocations) for every object and
// wants header and content copied into one buffer first.  One state per thread
// is reset between objects, and header and content are fed as two pieces;
// deflate's output does not depend on how its input is cut, so the bytes are
// the ones compress() gives (tests compare them).
class Deflater {
public:
    Deflater() { std::memset(&zs_, 0, sizeof zs_); }
    ~Deflater() { if (ok_) deflateEnd(&zs_); }
    Deflater(const Deflater&) = delete;
    Deflater& operator=(const Deflater&) = delete;

    // level: zlib's 0..9, or Z_DEFAULT_COMPRESSION (-1), which is what compress() uses
    bool run(int level, const std::string& type, const std::string& content, std::string& out)
    {
        if (!ok_ || level != level_) {
            if (ok_) deflateEnd(&zs_);
            std::memset(&zs_, 0, sizeof zs_);
            ok_ = deflateInit(&zs_, level) == Z_OK;
            level_ = level;
        }
        if (!ok_ || deflateReset(&zs_) != Z_OK) return false;
        std::string header = type;
        header += ' ';
        header += std::to_string(content.size());
        header += '\0';
        out.resize(deflateBound(&zs_, uLong(header.size() + content.size())));
        zs_.next_out = reinterpret_cast<Bytef*>(&out[0]);
        zs_.avail_out = uInt(out.size());
        zs_.next_in = reinterpret_cast<Bytef*>(&header[0]);
        zs_.avail_in = uInt(header.size());
        if (deflate(&zs_, Z_NO_FLUSH) != Z_OK) return false;
        // avail_in is 32 bits wide: a content of 4 GiB or more goes in slices
        const Bytef* p = reinterpret_cast<const Bytef*>(content.data());
        size_t left = content.size();
        for (;;) {
            const size_t take = std::min<size_t>(left, 1u << 30);
            zs_.next_in = const_cast<Bytef*>(p);
            zs_.avail_in = uInt(take);
            p += take;
            left -= take;
            const int rc = deflate(&zs_, left ? Z_NO_FLUSH : Z_FINISH);
            if (!left) { if (rc != Z_STREAM_END) return false; break; }
            if (rc != Z_OK) return false;
        }
        out.resize(out.size() - zs_.avail_out);
        return true;
    }

private:
    z_stream zs_;
    bool ok_ = false;
    int level_ = 0;
};

bool deflateFramed(int level, const std::string& type, const std::string& content, std::string& out)
{
    static thread_local Deflater deflater;
    return deflater.run(level, type, content, out);
}

// Handles of the objects/xx directories, one set per thread (see storeAsync).
struct FanoutDirs {
    const void* owner = nullptr;                          // the store the handles belong to
    uint64_t generation = 0;
    int fd[256];
    FanoutDirs() { for (int& f : fd) f = -1; }
    ~FanoutDirs() { drop(); }
    void drop() { for (int& f : fd) if (f >= 0) { ::close(f); f = -1; } }
    int get(const void* store, uint64_t gen, const std::string& objects_dir, uint8_t byte)
    {
        if (!tl_private_fd_table) return -1;              // handles are kept only in a table of the thread's own: in a shared one
                                                          // every thread's 256 would count against the one limit
        if (owner != store || generation != gen) { drop(); owner = store; generation = gen; }
        if (fd[byte] < 0) {
            const std::string dir = objects_dir + '/' + toHex(&byte, 1);
            fd[byte] = ::open(dir.c_str(), O_RDONLY | O_DIRECTORY | O_CLOEXEC);
            if (fd[byte] < 0 && (errno == EMFILE || errno == ENFILE)) {      // a descriptor limit below the 256 directories: give the
                drop();                                                        // handles back and start over
                fd[byte] = ::open(dir.c_str(), O_RDONLY | O_DIRECTORY | O_CLOEXEC);
            }
        }
        return fd[byte];
    }
};

// The calling thread's handles.
FanoutDirs& threadDirs()
{
    static thread_local FanoutDirs dirs;
    return dirs;
}

// Whole file into a string.  size_hint is the size seen during the walk: the
// buffer is one byte larger, so a file that still has that size is read, and
// its end detected, by a single read() (open + read + close per file; a million
// small files are bound by system calls, not by bytes).
std::string slurp(const std::string& path, uint64_t size_hint)
{
    // reference src/commit.cpp:151-152: an unreadable file yields empty content.
    int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0 && (errno == EMFILE || errno == ENFILE)) {  // out of descriptors is not "unreadable": this thread gives its
        threadDirs().drop();                               // directory handles back and tries again
        fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
    }
    if (fd < 0) return {};
    std::string s(size_t(size_hint) + 1, '\0');
    size_t got = 0;
    for (;;) {
        if (got == s.size()) s.resize(s.size() * 2);                       // the file grew since the walk
        const ssize_t r = ::read(fd, &s[got], s.size() - got);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) break;
        got += size_t(r);
        if (got < s.size() && got >= size_hint) break;                      // short read at or past the expected size: EOF
    }
    ::close(fd);
    s.resize(got);
    return s;
}

std::vector<int> devicesFromEnv()
{
    std::vector<int> ids;
    if (const char* e = std::getenv("VIT_DEVICES")) {
        std::stringstream ss(e);
        std::string tok;
        while (std::getline(ss, tok, ','))
            if (!tok.empty()) ids.push_back(std::atoi(tok.c_str()));
    }
    return ids;
}

enum Kind : uint8_t { REGULAR, EXECUTABLE, SYMLINK, DIRECTORY };   // the last three only under Git's rules
struct Ent { std::string name; Kind kind; size_t ref; bool isDir() const { return kind == DIRECTORY; } };
struct DirNode { std::string path; int depth; std::vector<Ent> ents; uint8_t id[20]; bool empty = false; };
struct FileRec { std::string path; uint64_t size; Kind kind; uint8_t id[20]; };

// What one directory listing found, before indices are assigned.
struct Listed { std::string name; std::string path; Kind kind; uint64_t size; };

const char* modeOf(Kind k)
{
    switch (k) {
    case EXECUTABLE: return "100755";
    case SYMLINK:    return "120000";
    case DIRECTORY:  return "40000";
    default:         return "100644";
    }
}

// Git's view of one directory: nothing is followed; a symlink is an entry of
// its own, the executable bit is kept, sockets, fifos and devices are passed over.
bool listDirGit(const std::string& dir, std::vector<Listed>& out)
{
    std::error_code ec;
    fs::directory_iterator it(dir, ec), end;
    if (ec) { std::cerr << "cannot read directory " << dir << ": " << ec.message() << '\n'; return false; }
    for (; it != end; it.increment(ec)) {
        if (ec) return false;
        std::string name = it->path().filename().string();
        if (name == ".git") continue;
        std::error_code sec;
        const fs::file_status st = it->symlink_status(sec);
        if (sec) return false;
        if (fs::is_symlink(st)) {
            out.push_back(Listed{std::move(name), it->path().string(), SYMLINK, 0});
        } else if (fs::is_regular_file(st)) {
            uint64_t sz = it->file_size(sec);
            if (sec) sz = 0;
            const bool x = (st.permissions() & fs::perms::owner_exec) != fs::perms::none;
            out.push_back(Listed{std::move(name), it->path().string(), x ? EXECUTABLE : REGULAR, sz});
        } else if (fs::is_directory(st)) {
            out.push_back(Listed{std::move(name), it->path().string(), DIRECTORY, 0});
        }
    }
    return true;
}

// List one directory with the reference's rules (src/commit.cpp:143-161): skip
// entries named ".git"; status checks follow symlinks; anything that is neither
// a regular file nor a directory makes the whole tree fail (:160).
bool listDir(const std::string& dir, std::vector<Listed>& out)
{
    std::error_code ec;
    fs::directory_iterator it(dir, ec), end;
    if (ec) { std::cerr << "cannot read directory " << dir << ": " << ec.message() << '\n'; return false; }
    for (; it != end; it.increment(ec)) {
        if (ec) return false;
        std::string name = it->path().filename().string();
        if (name == ".git") continue;
        std::error_code sec;
        if (it->is_regular_file(sec)) {
            uint64_t sz = it->file_size(sec);
            if (sec) sz = 0;
            out.push_back(Listed{std::move(name), it->path().string(), REGULAR, sz});
        } else if (it->is_directory(sec)) {
            out.push_back(Listed{std::move(name), it->path().string(), DIRECTORY, 0});
        } else {
            return false;
        }
    }
    return true;
}

// Breadth-first walk: all directories of one depth are listed in parallel on
// the pool, then their entries get file / directory indices serially.  dirs[0]
// is the root.  (The reference recurses depth-first, one entry at a time; the
// order of discovery does not enter any id, entries are sorted by name later.)
bool walk(const std::string& root, vit::WorkPool& pool, bool git_rules, std::vector<DirNode>& dirs, std::vector<FileRec>& files)
{
    dirs.push_back(DirNode{root, 0, {}, {}});
    std::vector<size_t> level(1, 0);
    for (int depth = 0; !level.empty(); ++depth) {
        std::vector<std::vector<Listed>> found(level.size());
        std::atomic<bool> ok{true};
        forEach(pool, level.size(), [&](size_t k) {
            try { if (!(git_rules ? listDirGit : listDir)(dirs[level[k]].path, found[k])) ok = false; }
            catch (const std::exception&) { ok = false; }
        });
        if (!ok) return false;
        std::vector<size_t> next;
        for (size_t k = 0; k < level.size(); ++k) {
            std::vector<Ent> ents;
            ents.reserve(found[k].size());
            for (Listed& e : found[k]) {
                if (e.kind == DIRECTORY) {
                    ents.push_back(Ent{std::move(e.name), DIRECTORY, dirs.size()});
                    next.push_back(dirs.size());
                    dirs.push_back(DirNode{std::move(e.path), depth + 1, {}, {}});
                } else {
                    ents.push_back(Ent{std::move(e.name), e.kind, files.size()});
                    files.push_back(FileRec{std::move(e.path), e.size, e.kind, {}});
                }
            }
            dirs[level[k]].ents = std::move(ents);
        }
        level.swap(next);
    }
    return true;
}

// One thread that runs what it is given, in order.  (The device deflate of successive batches: its
// hasher serves one caller at a time, and a batch's stores are queued when its streams are there.)
class SerialWorker {
public:
    SerialWorker() : t_([this] { loop(); }) {}
    ~SerialWorker()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        t_.join();
    }
    std::shared_future<void> post(std::function<void()> f)
    {
        std::packaged_task<void()> task(std::move(f));
        std::shared_future<void> done = task.get_future().share();
        { std::lock_guard<std::mutex> lk(m_); q_.push_back(std::move(task)); }
        cv_.notify_one();
        return done;
    }
private:
    void loop()
    {
        for (;;) {
            std::packaged_task<void()> task;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;                    // stop asked for and nothing left
                task = std::move(q_.front());
                q_.pop_front();
            }
            task();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::packaged_task<void()>> q_;
    bool stop_ = false;
    std::thread t_;
};

}  // namespace

namespace vit {

const GpuApi* GpuApi::load(std::string* why)
{
    static std::mutex m;
    static GpuApi api;
    static bool tried = false, ok = false;
    static std::string reason;
    std::lock_guard<std::mutex> lk(m);
    if (!tried) {
        tried = true;
        std::vector<std::string> where;
        if (const char* e = std::getenv("VIT_HASH_LIBRARY")) where.push_back(e);
        else {
            where.push_back("libvit_hash.so");
            std::error_code ec;
            const fs::path exe = fs::read_symlink("/proc/self/exe", ec);
            if (!ec) {
                where.push_back((exe.parent_path() / "libvit_hash.so").string());
                where.push_back((exe.parent_path().parent_path() / "libvit_hash.so").string());
            }
        }
        void* lib = nullptr;
        for (const std::string& w : where) {
            lib = ::dlopen(w.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
            const char* e = ::dlerror();
            reason += (reason.empty() ? "" : "; ") + std::string(e ? e : "dlopen failed");
        }
        if (lib) {
            ok = true;
            auto sym = [&](const char* name) -> void* {
                void* p = ::dlsym(lib, name);
                if (!p) { ok = false; reason = std::string("libvit_hash.so has no ") + name; }
                return p;
            };
            api.hasher_create     = reinterpret_cast<decltype(api.hasher_create)>(sym("vit_hasher_create"));
            api.hasher_destroy    = reinterpret_cast<decltype(api.hasher_destroy)>(sym("vit_hasher_destroy"));
            api.last_error        = reinterpret_cast<decltype(api.last_error)>(sym("vit_last_error"));
            api.set_option        = reinterpret_cast<decltype(api.set_option)>(sym("vit_hasher_set_option"));
            api.get_stat          = reinterpret_cast<decltype(api.get_stat)>(sym("vit_hasher_get_stat"));
            api.hash_objects      = reinterpret_cast<decltype(api.hash_objects)>(sym("vit_hash_objects"));
            api.arena_alloc       = reinterpret_cast<decltype(api.arena_alloc)>(sym("vit_arena_alloc"));
            api.arena_free        = reinterpret_cast<decltype(api.arena_free)>(sym("vit_arena_free"));
            api.deflate_objects   = reinterpret_cast<decltype(api.deflate_objects)>(sym("vit_deflate_objects"));
            api.deflate_out_bytes = reinterpret_cast<decltype(api.deflate_out_bytes)>(sym("vit_deflate_out_bytes"));
        }
    }
    if (!ok && why) *why = reason;
    return ok ? &api : nullptr;
}

ObjectStore::ObjectStore(std::string objects_dir, std::vector<int> devices)
    : objects_dir_(std::move(objects_dir)), devices_(std::move(devices))
{
    unsigned hw = std::thread::hardware_concurrency();
    io_threads_ = int(hw ? std::min(hw, 16u) : 4);
    if (const char* e = std::getenv("VIT_DEFLATE_LEVEL")) setDeflateLevel(std::atoi(e));
    if (const char* e = std::getenv("VIT_DEVICE_DEFLATE")) setDeviceDeflate(std::atoi(e) != 0);
    if (const char* e = std::getenv("VIT_HOST_CROSSOVER_BYTES")) if (*e) setHostCrossoverBytes(std::strtoull(e, nullptr, 10));
}

ObjectStore::~ObjectStore()
{
    pool_.reset();
    if (deflate_hasher_) gpu_->hasher_destroy(deflate_hasher_);
    if (hasher_) gpu_->hasher_destroy(hasher_);
}

// The GPU hasher is made by the first call that needs the device: reading objects
// back (cat-file, ls-tree) never does, and a call whose objects one host core
// hashes sooner than the HIP runtime starts does not either (sideFor).
bool ObjectStore::ensureHasher()
{
    if (hasher_) return true;
    if (hasher_failed_) return false;
    std::string why;
    gpu_ = GpuApi::load(&why);
    if (!gpu_) {
        std::cerr << "vit: cannot load the GPU hasher (" << why << "); there is no CPU path for a batch this large\n";
        hasher_failed_ = true;
        return false;
    }
    if (gpu_->hasher_create(devices_.empty() ? nullptr : devices_.data(), int(devices_.size()), &hasher_) != VIT_OK) {
        std::cerr << "vit: cannot create the GPU hasher: " << gpu_->last_error(nullptr) << '\n';
        hasher_ = nullptr;
        hasher_failed_ = true;
        return false;
    }
    // Tuning knobs for experiments; the defaults are the library's.
    if (const char* e = std::getenv("VIT_HOST_THRESHOLD")) gpu_->set_option(hasher_, "host_threshold_bytes", std::strtoull(e, nullptr, 10));
    if (const char* e = std::getenv("VIT_CHUNK_MB")) gpu_->set_option(hasher_, "chunk_bytes", std::strtoull(e, nullptr, 10) << 20);
    if (const char* e = std::getenv("VIT_KERNEL")) gpu_->set_option(hasher_, "kernel", std::strtoull(e, nullptr, 10));
    // The store's own pool (reads, deflate, writes) keeps the cores busy while a
    // batch is hashed: the hasher's packing threads get a modest share.
    gpu_->set_option(hasher_, "pack_threads", uint64_t(std::max(2, io_threads_ / 2)));
    return true;
}

// Small work never pays for the GPU.  Starting the HIP runtime under a process's first hasher
// costs 40 ms on a warm machine and up to 180 ms on a cold one (loading the library another 15),
// more than the reference needs for a whole small worktree.  While this store has no hasher yet, a
// call whose objects ONE host core would hash within that start-up time is hashed by the store's
// own threads instead (vit_host_object_ids, four SHA-1 streams interleaved per core).  The bound is
// measured, not assumed: start-up (the 40 ms of a warm machine) x this machine's one-core rate,
// about 100 MB with SHA extensions.  A larger call creates the hasher, and from then on every
// call of the store goes to the device, whose batches cost microseconds once it is up.  This is a
// scheduling policy of the store, above the C ABI: the library itself still has no CPU path, and a
// call above the bound on a machine without a gfx950 fails as before.
uint64_t ObjectStore::hostCrossoverBytes()
{
    if (crossover_set_) return crossover_bytes_;
    static const uint64_t measured = [] {
        constexpr double kHipStartSeconds = 0.040;
        std::vector<uint8_t> buf(1u << 20, 0x5A);
        uint8_t id[20];
        double best = 1e30;
        for (int r = 0; r < 2; ++r) {
            const auto t = clk::now();
            vit_host_object_id("blob", buf.data(), buf.size(), id);
            best = std::min(best, since(t));
        }
        return uint64_t(kHipStartSeconds * double(buf.size()) / best);
    }();
    return measured;
}

ObjectStore::Side ObjectStore::sideFor(uint64_t bytes)
{
    if (hasher_) return Side::GPU;
    if (crossover_set_ && bytes >= crossover_bytes_) return Side::GPU;          // 0 = always the device
    if (!crossover_set_ && bytes >= (8u << 20) && bytes >= hostCrossoverBytes()) return Side::GPU;   // under 8 MiB: not worth measuring
    return Side::HOST;
}

// ids on the store's threads: slices of consecutive objects, each through vit_host_object_ids.
void ObjectStore::hashOnHost(const std::string& type, const std::vector<const uint8_t*>& ptr, const std::vector<uint64_t>& len,
                             uint8_t* out)
{
    const size_t n = ptr.size();
    uint64_t bytes = 0;
    for (uint64_t l : len) bytes += l;
    const size_t slices = std::max<size_t>(1, std::min<size_t>({n / 4, size_t(io_threads_) * 4, size_t(bytes >> 16) + 1}));
    const char* t = type.c_str();
    auto part = [&](size_t s) {
        const size_t b = n * s / slices, e = n * (s + 1) / slices;
        if (b < e) vit_host_object_ids(t, ptr.data() + b, len.data() + b, e - b, out + b * 20);
    };
    if (slices == 1) part(0);
    else pool().run(WorkPool::HIGH, slices, part);
    stats_.host_hashed += n;
}

// A second hasher on the same device for the deflate kernel: its own streams and staging memory, so that a
// thread of its own can make the streams of one batch while this one hashes the next.
bool ObjectStore::ensureDeflateHasher()
{
    if (deflate_hasher_) return true;
    if (deflate_hasher_failed_) return false;
    const int dev = devices_.empty() ? 0 : devices_[0];
    if (!gpu_ || gpu_->hasher_create(&dev, 1, &deflate_hasher_) != VIT_OK) {
        std::cerr << "vit: no second hasher for the device deflate (" << (gpu_ ? gpu_->last_error(nullptr) : "no GPU library") << "); deflating on host threads\n";
        deflate_hasher_ = nullptr;
        deflate_hasher_failed_ = true;
        return false;
    }
    gpu_->set_option(deflate_hasher_, "pack_threads", uint64_t(std::max(2, io_threads_ / 4)));
    if (const char* e = std::getenv("VIT_DEVICE_DEFLATE_LANES")) gpu_->set_option(deflate_hasher_, "deflate_lanes", std::strtoull(e, nullptr, 10));
    if (const char* e = std::getenv("VIT_DEVICE_DEFLATE_MAX")) gpu_->set_option(deflate_hasher_, "deflate_max_bytes", std::strtoull(e, nullptr, 10));
    if (std::getenv("VIT_DEBUG_FAIL_DEFLATE")) gpu_->set_option(deflate_hasher_, "debug_fail_deflate", 1);       // tests
    if (std::getenv("VIT_DEBUG_CORRUPT_DEFLATE")) gpu_->set_option(deflate_hasher_, "debug_corrupt_deflate", 1); // tests
    return true;
}

// Who hashes the mid-sized objects of a batch.  While objects are being stored
// the host threads are the scarce side (deflate and file operations are most of
// a write-tree) and the GPU the idle one: only objects long enough to hold a GPU
// lane for milliseconds (256 KiB: 5 ms) go to host threads then.  When ids are
// all that is wanted the threads are free and the library's own minimum stands
// (a 128 MiB batch of mixed sizes: 2.6 ms against 8).
void ObjectStore::tuneHasher()
{
    uint64_t min_bytes = store_ ? (256u << 10) : (16u << 10);
    if (const char* e = std::getenv("VIT_HOST_MIN_BYTES")) min_bytes = std::strtoull(e, nullptr, 10);
    gpu_->set_option(hasher_, "host_min_bytes", min_bytes);
}

vit::WorkPool& ObjectStore::pool()
{
    if (!pool_ || pool_->threads() != io_threads_ - 1) pool_.reset(new WorkPool(std::max(0, io_threads_ - 1), [](int, int) { privateFdTable(); }));
    return *pool_;
}

std::string ObjectStore::objectPath(const std::string& hash) const
{
    return objects_dir_ + '/' + hash.substr(0, 2) + '/' + hash.substr(2);
}

bool ObjectStore::hashObjects(const std::string& type, const std::vector<const std::string*>& contents,
                              std::vector<uint8_t>& out)
{
    const size_t n = contents.size();
    std::vector<const uint8_t*> ptr(n);
    std::vector<uint64_t> len(n);
    uint64_t bytes = 0;
    for (size_t i = 0; i < n; ++i) {
        ptr[i] = reinterpret_cast<const uint8_t*>(contents[i]->data());
        len[i] = contents[i]->size();
        bytes += len[i];
    }
    out.resize(n * 20);
    // inside a writeTree the side was chosen once, for the whole worktree
    const Side side = in_write_tree_ ? tree_side_ : sideFor(bytes);
    if (side == Side::HOST) {
        if (vit_header_len(type.c_str(), 0) == 0) { std::cerr << "vit: hashing failed: bad object type\n"; return false; }
        hashOnHost(type, ptr, len, out.data());
        return true;
    }
    if (!ensureHasher()) return false;
    tuneHasher();
    if (gpu_->hash_objects(hasher_, type.c_str(), ptr.data(), len.data(), n, out.data()) != VIT_OK) {
        std::cerr << "vit: hashing failed: " << gpu_->last_error(hasher_) << '\n';
        return false;
    }
    stats_.gpu_hashed += n;
    return true;
}

// A loose object that is already in place is a regular file with something in it (an empty or
// truncated-to-nothing file, as a crashed non-atomic writer such as the reference's leaves behind, is written
// again); objects are written atomically (temporary name, then rename: a reader or a crash never sees half an
// object).  Both go through a directory handle: a million objects go into 256 directories, and naming each of them by
// its full path has the kernel walk .../.git/objects/xx afresh for every stat, create and rename.  Every worker
// thread keeps its own handles (the pool's threads have private descriptor tables).
namespace {

// One number per top-level storing call, over every store of the process: a thread's cached handles are never taken
// for those of another call, even of a store that has since been destroyed and another created at its address.
uint64_t nextStoreCall()
{
    static std::atomic<uint64_t> calls{0};
    return ++calls;
}

bool storedAt(int dirfd, const char* name)
{
    struct stat st;
    return ::fstatat(dirfd, name, &st, 0) == 0 && S_ISREG(st.st_mode) && st.st_size > 0;
}

// 0, or the errno of the step that failed.
int writeFileAtomicAt(int dirfd, const char* name, const char* data, size_t size)
{
    static std::atomic<uint64_t> serial{0};
    const std::string tmp_name = std::string(name) + ".tmp" + std::to_string(getpid()) + '_' + std::to_string(serial++);
    const char* const tmp = tmp_name.c_str();
    const int fd = ::openat(dirfd, tmp, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (fd < 0) return errno ? errno : EIO;
    size_t put = 0;
    while (put < size) {
        const ssize_t w = ::write(fd, data + put, size - put);
        if (w < 0 && errno == EINTR) continue;
        if (w <= 0) { const int e = w < 0 && errno ? errno : EIO; ::close(fd); ::unlinkat(dirfd, tmp, 0); return e; }
        put += size_t(w);
    }
    if (::close(fd) != 0 || ::renameat(dirfd, tmp, dirfd, name) != 0) { const int e = errno ? errno : EIO; ::unlinkat(dirfd, tmp, 0); return e; }
    return 0;
}
}  // namespace

// Queue deflate + write of objects whose ids are known; `contents` and `raw`
// must stay alive until the ticket has been waited for.  Low priority: file
// reads of the next batch, which feed the GPU, go first.
WorkPool::Ticket ObjectStore::storeAsync(const std::string& type, const std::vector<std::string>& contents,
                                         const std::vector<uint8_t>& raw, StoreOutcome& outcome, DeviceStreams dz)
{
    const size_t n = contents.size();
    bool need[256] = {false};                             // the "aa" fan-out directories this batch touches
    for (size_t i = 0; i < n; ++i) need[raw[i * 20]] = true;
    for (int d = 0; d < 256; ++d)
        if (need[d]) {
            const uint8_t byte = uint8_t(d);
            std::error_code ec;
            fs::create_directories(objects_dir_ + '/' + toHex(&byte, 1), ec);
        }
    const size_t slices = std::min(n, size_t(io_threads_) * 8);
    const std::string* c = contents.data();
    const uint8_t* r = raw.data();
    const int level = deflate_level_;
    const uint64_t dirs_generation = store_calls_;        // handles are good for one top-level call: between calls the directories may have been replaced
    const char* bt = std::getenv("VIT_DEBUG_BAD_TRAILER");
    const size_t bad_trailer_every = bt ? size_t(std::strtoull(bt, nullptr, 10)) : 0;
    return pool().submit(WorkPool::LOW, slices, [this, type, c, r, n, slices, level, dz, dirs_generation, bad_trailer_every, &outcome](size_t s) {
        for (size_t i = n * s / slices; i < n * (s + 1) / slices; ++i) {
            try {
                FanoutDirs& dirs = threadDirs();
                int dirfd = dirs.get(this, dirs_generation, objects_dir_, r[i * 20]);
                char leaf[40];
                static const char hex[] = "0123456789abcdef";
                for (int b = 1; b < 20; ++b) { leaf[2 * b - 2] = hex[r[i * 20 + b] >> 4]; leaf[2 * b - 1] = hex[r[i * 20 + b] & 15]; }
                leaf[38] = 0;
                const char* name = leaf;
                std::string full;
                if (dirfd < 0) {                           // no handle to be had: name the object by its whole path instead
                    full = objects_dir_ + '/' + toHex(&r[i * 20], 1) + '/' + leaf;
                    name = full.c_str();
                    dirfd = AT_FDCWD;
                }
                if (skip_existing_ && storedAt(dirfd, name)) { ++outcome.skipped; continue; }
                auto put = [&](const char* data, size_t size) {
                    int err = writeFileAtomicAt(dirfd, name, data, size);
                    if (err == EMFILE || err == ENFILE) {  // the handles themselves have used the descriptors up: give them back
                        dirs.drop();
                        if (full.empty()) full = objects_dir_ + '/' + toHex(&r[i * 20], 1) + '/' + leaf;
                        err = writeFileAtomicAt(AT_FDCWD, full.c_str(), data, size);
                    }
                    if (err) outcome.failed = 2;
                };
                if (dz.buf && dz.len[i]) {                 // the device has made this object's stream already
                    // With skip_existing a stored file is never written again, so a stream is looked at before it
                    // becomes one: no longer than zlib's own bound, zlib's two header bytes for the default level,
                    // and a trailer that is the Adler-32 of header + content as computed here (a microsecond for
                    // 4 KiB, beside 20 of file operations).  What fails is deflated by this thread instead.
                    const uint8_t* z = dz.buf + dz.off[i];
                    const uint32_t zn = dz.len[i];
                    char hdr[40];
                    const int hl = std::snprintf(hdr, sizeof hdr, "%s %llu", type.c_str(), (unsigned long long)c[i].size()) + 1;
                    uLong ad = adler32(adler32(0L, Z_NULL, 0), reinterpret_cast<const Bytef*>(hdr), uInt(hl));
                    for (size_t done = 0; done < c[i].size();) {
                        const size_t take = std::min<size_t>(c[i].size() - done, 1u << 30);
                        ad = adler32(ad, reinterpret_cast<const Bytef*>(c[i].data() + done), uInt(take));
                        done += take;
                    }
                    if (bad_trailer_every && i % bad_trailer_every == 0) ad ^= 1;        // tests: as if the stream's trailer were wrong
                    const bool sane = zn >= 8 && zn <= compressBound(uLong(size_t(hl) + c[i].size())) && z[0] == 0x78 && z[1] == 0x9c &&
                                      ((uint32_t(z[zn - 4]) << 24) | (uint32_t(z[zn - 3]) << 16) | (uint32_t(z[zn - 2]) << 8) | z[zn - 1]) == uint32_t(ad);
                    if (sane) { put(reinterpret_cast<const char*>(z), zn); continue; }
                    ++outcome.rejected;
                }
                std::string z;
                if (!deflateFramed(level, type, c[i], z)) { outcome.failed = 1; continue; }
                put(z.data(), z.size());
            } catch (const std::exception&) { outcome.failed = 2; }      // out of memory in deflate or the path strings
        }
    });
}

bool ObjectStore::storeFinished(StoreOutcome& outcome)
{
    stats_.skipped += outcome.skipped.exchange(0);
    stats_.device_rejected += outcome.rejected.exchange(0);
    if (outcome.failed == 1) { std::cerr << "Compression failed\n"; return false; }
    if (outcome.failed == 2) { std::cerr << "vit: writing objects under " << objects_dir_ << " failed\n"; return false; }
    return true;
}

std::vector<std::string> ObjectStore::writeObjects(const std::string& type, const std::vector<std::string>& contents)
{
    const size_t n = contents.size();
    std::vector<std::string> ids;
    if (n == 0) return ids;
    if (!in_write_tree_) store_calls_ = nextStoreCall();
    std::vector<const std::string*> view(n);
    for (size_t i = 0; i < n; ++i) view[i] = &contents[i];

    // Ids first: the GPU batch takes milliseconds per GB, and knowing the id lets
    // the (far more expensive) deflate be skipped for objects the store already
    // holds.  The reference always rewrites (src/commit.cpp:110); the stored
    // bytes are the same either way.
    const auto t0 = clk::now();
    std::vector<uint8_t> raw;
    if (!hashObjects(type, view, raw)) return ids;
    stats_.hash_s += since(t0);

    ids.resize(n);
    for (size_t i = 0; i < n; ++i) ids[i] = toHex(&raw[i * 20], 20);
    if (store_) {
        const auto t1 = clk::now();
        StoreOutcome outcome;
        pool().wait(storeAsync(type, contents, raw, outcome, DeviceStreams{nullptr, nullptr, nullptr}));
        stats_.deflate_store_s += since(t1);
        if (!storeFinished(outcome)) return {};
    }
    return ids;
}

std::string ObjectStore::writeObject(const std::string& type, const std::string& content)
{
    std::vector<std::string> one(1, content);
    std::vector<std::string> ids = writeObjects(type, one);
    return ids.empty() ? std::string() : ids[0];
}

std::string ObjectStore::writeTree(const std::string& directoryPath)
{
    stats_ = StoreStats();
    store_calls_ = nextStoreCall();
    struct InTree { bool& f; InTree(bool& x) : f(x) { f = true; } ~InTree() { f = false; } } in_tree(in_write_tree_);
    const auto t_all = clk::now();

    std::vector<DirNode> dirs;
    std::vector<FileRec> files;
    WorkPool& workers = pool();
    const size_t root = 0;
    auto t0 = clk::now();
    if (!walk(directoryPath, workers, git_rules_, dirs, files)) return {};
    stats_.walk_s = since(t0);

    // Who computes the ids, decided once for the whole worktree (sideFor): the device, unless this store has no
    // hasher yet and the files add up to less than one host core hashes while the HIP runtime would start.
    uint64_t worktree_bytes = 0;
    for (const FileRec& f : files) worktree_bytes += f.size;
    tree_side_ = sideFor(worktree_bytes);
    stats_.ids_on_host = tree_side_ == Side::HOST;
    stats_.crossover_bytes = crossover_set_ ? crossover_bytes_ : (worktree_bytes >= (8u << 20) ? hostCrossoverBytes() : 0);
    if (tree_side_ == Side::GPU) {
        t0 = clk::now();
        if (!ensureHasher()) return {};
        stats_.init_s = since(t0);                       // the HIP runtime starting up under the first hasher, mostly
    }

    // ---- blobs, in batches of bounded size, three stages in flight at once:
    //   file reads of batch k+1 (pool, high priority: they feed the GPU)
    //   ids of batch k on the GPU (this thread; packing on the hasher's threads)
    //   deflate + store of batches k-1, k (pool, low priority: by far the most work)
    // A batch's contents live in one of three slots until its objects are stored.
    // With the device deflating there is a fourth stage (streams of batch k on the GPU, through a hasher and a
    // thread of its own, while this thread reads and hashes batch k+1), hence a fourth slot.  Whether a batch goes
    // that way is decided batch by batch: the kernels take one object per workgroup, a thousand of which are on the
    // chip at a time, so a few thousand small objects keep it busy; fewer are done sooner by the host threads.
    const bool may_device = tree_side_ == Side::GPU && device_deflate_ && store_ && (deflate_level_ == -1 || deflate_level_ == 6);
    uint64_t device_min_objects = 4096, device_max_bytes = 16382;      // up to there a workgroup deflates an object in LDS, no device tables
    if (const char* e = std::getenv("VIT_DEVICE_DEFLATE_MIN")) device_min_objects = std::strtoull(e, nullptr, 10);
    if (const char* e = std::getenv("VIT_DEVICE_DEFLATE_MAX")) device_max_bytes = std::strtoull(e, nullptr, 10);
    const size_t ns = may_device ? 4 : 3;
    uint64_t batch_bytes = batch_bytes_;
    if (const char* e = std::getenv("VIT_BATCH_MB")) batch_bytes = std::max<uint64_t>(1, std::strtoull(e, nullptr, 10)) << 20;
    struct Slot {
        std::vector<std::string> contents;
        std::vector<uint8_t> raw;
        WorkPool::Ticket read, store;
        std::shared_future<void> deflate;                 // device deflate of the batch; queues the batch's stores when done
        StoreOutcome outcome;
        void* zbuf = nullptr; uint64_t zcap = 0;          // pinned: the streams the device made for this batch
        std::vector<uint64_t> zoff; std::vector<uint32_t> zlen;
    } slot[4];
    struct FreePinned {                                   // on every way out
        const GpuApi* gpu; vit_hasher* h; Slot* s;
        ~FreePinned() { for (int k = 0; k < 4; ++k) if (s[k].zbuf && gpu) gpu->arena_free(h, s[k].zbuf); }
    } free_pinned{gpu_, hasher_, slot};
    std::vector<std::pair<size_t, size_t>> batches;
    for (size_t pos = 0; pos < files.size();) {
        size_t end = pos;
        uint64_t bytes = 0;
        while (end < files.size() && (end == pos || bytes + files[end].size <= batch_bytes)) bytes += files[end++].size;
        batches.emplace_back(pos, end);
        pos = end;
    }
    bool ok = true;
    const bool timeline = std::getenv("VIT_TIMELINE") != nullptr;   // diagnostics: when each stage of each batch ended
    auto mark = [&, t_all](const char* what, size_t k) {
        if (timeline) std::fprintf(stderr, "[vit timeline] %8.1f ms  batch %zu %s\n", since(t_all) * 1e3, k, what);
    };
    // Of a batch's small objects, the share (in sixteenths) that the device deflates; the host threads do the rest next
    // to their file operations.  How fast a lane deflates depends on the content (incompressible bytes: a position has
    // no hash chain to walk; text: several candidates per position, three to four times the work), how fast the host
    // threads are on the machine and its file system, so the share follows what the pipeline waits for: when a slot is
    // needed again and its batch is still with the device, the device gets less of the next batches; when the device is
    // long done and the host threads are still storing, it gets more.
    std::atomic<bool> device_failed{false};
    std::atomic<bool> device_checked{false};              // the first batch's streams have been compared with this machine's zlib
    std::atomic<int> device_share{16};
    if (const char* e = std::getenv("VIT_DEVICE_SHARE")) device_share = std::max(1, std::min(16, std::atoi(e)));
    const bool adapt_share = std::getenv("VIT_DEVICE_SHARE") == nullptr;
    // The pool's tasks and the deflate thread hold pointers into this frame (slots, files, the flags above).  Whatever
    // way this function is left, an exception on this thread included (out of memory while a batch's views are built),
    // they have finished first: declared after everything they reference, so destroyed before any of it.
    std::unique_ptr<SerialWorker> deflater;
    struct Drain {
        std::unique_ptr<SerialWorker>& deflater; WorkPool& workers; Slot* s;
        ~Drain()
        {
            deflater.reset();                             // runs what was posted (each posts its batch's stores), then joins
            for (int k = 0; k < 4; ++k) {
                if (s[k].read) workers.wait(s[k].read);
                if (s[k].store) workers.wait(s[k].store);
            }
        }
    } drain{deflater, workers, slot};
    auto finish = [&](Slot& s) {                          // wait for a slot's stores; false if any failed
        double waited_device = 0;
        if (s.deflate.valid()) {
            const auto t = clk::now();
            s.deflate.wait();
            s.deflate = std::shared_future<void>();
            waited_device = since(t);
        }
        if (!s.store) return true;
        const auto t = clk::now();
        workers.wait(s.store);
        s.store.reset();
        if (adapt_share) {
            if (waited_device > 0.005) device_share = std::max(4, device_share.load() - 2);
            else if (since(t) > 0.005 && device_share < 16) device_share = device_share.load() + 1;
        }
        mark("a slot's objects are stored", size_t(&s - slot));
        return storeFinished(s.outcome);
    };
    auto start_read = [&](size_t k) {
        Slot& s = slot[k % ns];
        const size_t pos = batches[k].first, cnt = batches[k].second - pos;
        s.contents.assign(cnt, std::string());
        std::string* out = s.contents.data();
        const FileRec* in = files.data() + pos;
        const size_t slices = std::min(cnt, size_t(io_threads_) * 8);
        s.read = workers.submit(WorkPool::HIGH, slices, [out, in, cnt, slices](size_t sl) {
            for (size_t i = cnt * sl / slices; i < cnt * (sl + 1) / slices; ++i) {
                try {
                    if (in[i].kind == SYMLINK) {           // Git stores the link's target text as the blob
                        std::error_code ec;
                        out[i] = fs::read_symlink(in[i].path, ec).string();
                    } else {
                        out[i] = slurp(in[i].path, in[i].size);
                    }
                } catch (const std::exception&) { out[i].clear(); }    // as an unreadable file: empty content (:151-152)
            }
        });
    };
    const auto t_store = clk::now();
    if (!batches.empty()) start_read(0);
    for (size_t k = 0; k < batches.size() && ok; ++k) {
        Slot& s = slot[k % ns];
        if (k + 1 < batches.size()) {
            ok = finish(slot[(k + 1) % ns]);
            mark("slot free for the read of the next", k);               // batch k-2 used that slot: its objects must be on disk
            if (!ok) break;
            start_read(k + 1);
        }
        t0 = clk::now();
        workers.wait(s.read);
        stats_.read_s += since(t0);
        mark("read", k);
        t0 = clk::now();
        std::vector<const std::string*> view(s.contents.size());
        for (size_t i = 0; i < view.size(); ++i) view[i] = &s.contents[i];
        if (!hashObjects("blob", view, s.raw)) { ok = false; break; }
        stats_.hash_s += since(t0);
        mark("hashed", k);
        const size_t pos = batches[k].first;
        for (size_t i = 0; i < s.contents.size(); ++i) {
            std::memcpy(files[pos + i].id, &s.raw[i * 20], 20);
            stats_.bytes += s.contents[i].size();
        }
        stats_.blobs += s.contents.size();
        bool on_device = false;
        if (may_device) {
            uint64_t small = 0;
            for (const std::string& c : s.contents) small += c.size() + 32 <= device_max_bytes;
            on_device = small >= device_min_objects && !device_failed.load() && ensureDeflateHasher();
        }
        if (on_device) {
            if (!deflater) deflater.reset(new SerialWorker());
            Slot* sp = &s;
            s.deflate = deflater->post([this, sp, k, &mark, &device_share, &device_failed, &device_checked] {
                Slot& s = *sp;
                mark("deflate begins", k);
                DeviceStreams dz{nullptr, nullptr, nullptr};
                try {
                    const auto t = clk::now();
                    const size_t all = s.contents.size();
                    const size_t share = size_t(device_share.load());
                    std::vector<uint32_t> pick;                        // the device's objects: `share` of every sixteen
                    pick.reserve(all);
                    for (size_t i = 0; i < all; ++i) if (i % 16 < share) pick.push_back(uint32_t(i));
                    const size_t cnt = pick.size();
                    std::vector<const uint8_t*> ptr(cnt);
                    std::vector<uint64_t> len(cnt);
                    for (size_t j = 0; j < cnt; ++j) { const std::string& c = s.contents[pick[j]]; ptr[j] = reinterpret_cast<const uint8_t*>(c.data()); len[j] = c.size(); }
                    const uint64_t need = gpu_->deflate_out_bytes(deflate_hasher_, "blob", len.data(), cnt);
                    if (need > s.zcap) {
                        if (s.zbuf) { gpu_->arena_free(hasher_, s.zbuf); s.zbuf = nullptr; s.zcap = 0; }
                        if (gpu_->arena_alloc(hasher_, need + need / 4, &s.zbuf) == VIT_OK) s.zcap = need + need / 4;
                    }
                    std::vector<uint64_t> off(cnt, 0);
                    std::vector<uint32_t> zl(cnt, 0);
                    s.zoff.assign(all, 0); s.zlen.assign(all, 0);
                    bool made = need && s.zbuf &&
                        gpu_->deflate_objects(deflate_hasher_, "blob", ptr.data(), len.data(), cnt, static_cast<uint8_t*>(s.zbuf), s.zcap,
                                              off.data(), zl.data()) == VIT_OK;
                    if (need && !made) {
                        // once is enough: the batches that follow go to the host threads without trying (no memory for
                        // the tables on a GPU that others use, a device error)
                        if (!device_failed.exchange(true))
                            std::cerr << "vit: deflate on the device failed (" << gpu_->last_error(deflate_hasher_) << "); zlib on host threads from here on\n";
                    }
                    // The kernel restates zlib 1.2.11's default-level deflate; the host threads of this same write-tree
                    // call whatever libz the program is linked with, and the reference's zlib is not pinned at all.  On
                    // the first batch the device has made, a sample of its streams (the first, the last and six between)
                    // is compared with what this machine's library gives for the same objects.  Any difference (another
                    // deflate behind the zlib interface, or a fault on the device) and the device's streams are dropped,
                    // once and with a message: one store never holds files from two deflates that disagree.
                    if (made && !device_checked.exchange(true)) {
                        std::vector<size_t> done;
                        for (size_t j = 0; j < cnt; ++j) if (zl[j]) done.push_back(j);
                        for (size_t t = 0; t < 8 && !done.empty() && made; ++t) {
                            const size_t j = done[(done.size() - 1) * t / 7];
                            std::string z;
                            const uint8_t* mine = static_cast<const uint8_t*>(s.zbuf) + off[j];
                            if (!deflateFramed(deflate_level_, "blob", s.contents[pick[j]], z) || z.size() != zl[j] ||
                                std::memcmp(z.data(), mine, z.size()) != 0)
                                made = false;
                        }
                        if (!made && !device_failed.exchange(true))
                            std::cerr << "vit: the device's deflate does not give the bytes of this machine's zlib (" << zlibVersion()
                                      << "); zlib on host threads from here on\n";
                    }
                    if (made && !device_failed.load()) {
                        for (size_t j = 0; j < cnt; ++j) { s.zoff[pick[j]] = off[j]; s.zlen[pick[j]] = zl[j]; stats_.device_deflated += zl[j] != 0; }
                        dz = DeviceStreams{static_cast<const uint8_t*>(s.zbuf), s.zoff.data(), s.zlen.data()};
                    }
                    stats_.device_deflate_s += since(t);
                } catch (const std::exception&) { dz = DeviceStreams{nullptr, nullptr, nullptr}; }      // out of memory: host threads do the batch
                mark("deflated", k);
                s.store = storeAsync("blob", s.contents, s.raw, s.outcome, dz);
            });
        } else if (store_) {
            s.store = storeAsync("blob", s.contents, s.raw, s.outcome, DeviceStreams{nullptr, nullptr, nullptr});
        }
    }
    for (Slot& s : slot) {                                // also on failure: no job may outlive the slots
        if (s.read) workers.wait(s.read);
        if (!finish(s)) ok = false;
    }
    stats_.deflate_store_s += since(t_store);
    if (!ok) return {};

    // ---- trees: deepest level first, every directory of a level in one batch
    const auto t_trees = clk::now();
    const double blob_hash_s = stats_.hash_s, blob_store_s = stats_.deflate_store_s;
    int max_depth = 0;
    for (const DirNode& d : dirs) max_depth = std::max(max_depth, d.depth);
    for (int depth = max_depth; depth >= 0; --depth) {
        std::vector<size_t> level;
        for (size_t i = 0; i < dirs.size(); ++i) if (dirs[i].depth == depth) level.push_back(i);
        std::vector<std::string> contents(level.size());
        for (size_t k = 0; k < level.size(); ++k) {
            DirNode& d = dirs[level[k]];
            if (git_rules_) {
                // Git compares a directory as "name/" and has no entry for a directory without files below it
                d.ents.erase(std::remove_if(d.ents.begin(), d.ents.end(),
                                            [&](const Ent& e) { return e.isDir() && dirs[e.ref].empty; }), d.ents.end());
                std::sort(d.ents.begin(), d.ents.end(), [](const Ent& a, const Ent& b) {
                    return (a.isDir() ? a.name + '/' : a.name) < (b.isDir() ? b.name + '/' : b.name);
                });
                d.empty = d.ents.empty();
            } else {
                // reference src/commit.cpp:164-165: plain std::string order of the names
                std::sort(d.ents.begin(), d.ents.end(), [](const Ent& a, const Ent& b) { return a.name < b.name; });
            }
            std::string& c = contents[k];
            for (const Ent& e : d.ents) {                  // :167-171
                c += git_rules_ ? modeOf(e.kind) : (e.isDir() ? "40000" : "100644");
                c += ' ';
                c += e.name;
                c += '\0';
                c.append(reinterpret_cast<const char*>(e.isDir() ? dirs[e.ref].id : files[e.ref].id), 20);
            }
        }
        // Under Git's rules an empty directory has no tree object at all (the root excepted: an empty
        // worktree is the empty tree); hashing it anyway is harmless, storing it is not wanted.
        std::vector<std::string> ids;
        if (git_rules_ && depth > 0) {
            std::vector<std::string> kept;
            std::vector<size_t> at;
            for (size_t k = 0; k < level.size(); ++k)
                if (!dirs[level[k]].empty) { at.push_back(k); kept.push_back(std::move(contents[k])); }
            std::vector<std::string> some = writeObjects("tree", kept);
            if (some.size() != kept.size()) return {};
            ids.assign(level.size(), std::string(40, '0'));
            for (size_t j = 0; j < at.size(); ++j) ids[at[j]] = some[j];
        } else {
            ids = writeObjects("tree", contents);
            if (ids.size() != contents.size()) return {};
        }
        for (size_t k = 0; k < level.size(); ++k)
            for (int b = 0; b < 20; ++b)
                dirs[level[k]].id[b] = uint8_t(hexNibble(ids[k][2 * b]) * 16 + hexNibble(ids[k][2 * b + 1]));
        stats_.trees += level.size();
    }
    stats_.trees_s = since(t_trees);
    stats_.tree_hash_s = stats_.hash_s - blob_hash_s;               // what writeObjects added for the trees
    stats_.hash_s = blob_hash_s;
    stats_.deflate_store_s = blob_store_s;
    stats_.total_s = since(t_all);
    return toHex(dirs[root].id, 20);
}

std::string ObjectStore::writeCommit(const std::string& tree, const std::string& parent, const std::string& message,
                                     const std::string& author, const std::string& email, std::time_t now)
{
    const std::string stamp = std::to_string(static_cast<long long>(now));
    std::string c = "tree " + tree + '\n';
    if (!parent.empty()) c += "parent " + parent + '\n';
    c += "author " + author + " <" + email + "> " + stamp + '\n';
    c += "committer " + author + " <" + email + "> " + stamp + "\n\n";
    c += message + '\n';
    return writeObject("commit", c);
}

std::string ObjectStore::readObject(const std::string& hash) const
{
    if (hash.size() < 3) return {};
    std::ifstream in(objectPath(hash), std::ios::binary);
    if (!in) { std::cerr << "Object not found: " << hash << '\n'; return {}; }
    std::ostringstream ss;
    ss << in.rdbuf();
    const std::string z = ss.str();
    z_stream zs{};
    if (inflateInit(&zs) != Z_OK) { std::cerr << "inflateInit failed\n"; return {}; }
    zs.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(z.data()));
    zs.avail_in = uInt(z.size());
    std::string out;
    std::vector<char> buf(1 << 16);
    int rc;
    do {
        zs.next_out = reinterpret_cast<Bytef*>(buf.data());
        zs.avail_out = uInt(buf.size());
        rc = inflate(&zs, Z_NO_FLUSH);
        if (rc != Z_OK && rc != Z_STREAM_END) { std::cerr << "inflate error: " << rc << '\n'; inflateEnd(&zs); return {}; }
        out.append(buf.data(), buf.size() - zs.avail_out);
    } while (rc != Z_STREAM_END);
    inflateEnd(&zs);
    return out;
}

std::string ObjectStore::readObjectContent(const std::string& hash) const
{
    const std::string full = readObject(hash);
    const size_t nul = full.find('\0');
    return nul == std::string::npos ? full : full.substr(nul + 1);
}

const char* zlibVersionString() { return zlibVersion(); }

ObjectStore& defaultStore()
{
    static ObjectStore store(".git/objects", devicesFromEnv());
    return store;
}

}  // namespace vit

/* ---------- the reference's free functions ---------- */

std::string hashToHexString(const unsigned char* hash) { return toHex(hash, 20); }

std::string hexStringToBinary(const std::string& hex)
{
    std::string bin(20, '\0');
    for (int i = 0; i < 20; ++i) {
        const int hi = hexNibble(hex.at(size_t(2 * i))), lo = hexNibble(hex.at(size_t(2 * i + 1)));
        if (hi < 0 || lo < 0) throw std::invalid_argument("hexStringToBinary: not a hex digit");   // std::stoi throws too
        bin[size_t(i)] = char(hi * 16 + lo);
    }
    return bin;
}

std::string binaryToHexString(const std::string& bin)
{
    return toHex(reinterpret_cast<const uint8_t*>(bin.data()), bin.size());
}

std::string getObjectPath(const std::string& hash) { return ".git/objects/" + hash.substr(0, 2) + '/' + hash.substr(2); }

std::string writeObject(const std::string& type, const std::string& content) { return vit::defaultStore().writeObject(type, content); }
std::string writeBlob(const std::string& content) { return vit::defaultStore().writeBlob(content); }
std::string writeTree(const std::string& directoryPath) { return vit::defaultStore().writeTree(directoryPath); }
std::string readObject(const std::string& hash) { return vit::defaultStore().readObject(hash); }
std::string readObjectContent(const std::string& hash) { return vit::defaultStore().readObjectContent(hash); }
std::string writeCommit(const std::string& treeHash, const std::string& parentHash, const std::string& message,
                        const std::string& author, const std::string& email)
{
    return vit::defaultStore().writeCommit(treeHash, parentHash, message, author, email);
}
