// object_store.cpp — see object_store.hpp.
#include "object_store.hpp"

#include "../../include/vit_hash.h"

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
#include <iostream>
#include <mutex>
#include <queue>
#include <sstream>
#include <thread>
#include <unistd.h>

namespace fs = std::filesystem;

namespace {

using clk = std::chrono::steady_clock;
double since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

// Fixed-size worker pool for file reads, deflate and object writes.
class Pool {
public:
    explicit Pool(int n)
    {
        for (int i = 0; i < n; ++i)
            workers_.emplace_back([this] {
                for (;;) {
                    std::function<void()> job;
                    {
                        std::unique_lock<std::mutex> lk(m_);
                        cv_.wait(lk, [this] { return stop_ || !jobs_.empty(); });
                        if (jobs_.empty()) return;
                        job = std::move(jobs_.front());
                        jobs_.pop();
                    }
                    job();
                    {
                        std::lock_guard<std::mutex> lk(m_);
                        if (--pending_ == 0) idle_.notify_all();
                    }
                }
            });
    }
    ~Pool()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    void submit(std::function<void()> f)
    {
        { std::lock_guard<std::mutex> lk(m_); jobs_.push(std::move(f)); ++pending_; }
        cv_.notify_one();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(m_);
        idle_.wait(lk, [this] { return pending_ == 0; });
    }
    // Run fn(i) for i in [0, n) on the pool, in slices, and wait.
    void forEach(size_t n, const std::function<void(size_t)>& fn)
    {
        if (n == 0) return;
        const size_t slices = std::min(n, workers_.size() * 4);
        for (size_t s = 0; s < slices; ++s)
            submit([=, &fn] { for (size_t i = n * s / slices; i < n * (s + 1) / slices; ++i) fn(i); });
        wait();
    }

private:
    std::vector<std::thread> workers_;
    std::queue<std::function<void()>> jobs_;
    std::mutex m_;
    std::condition_variable cv_, idle_;
    size_t pending_ = 0;
    bool stop_ = false;
};

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

std::string frame(const std::string& type, const std::string& content)
{
    std::string full;
    full.reserve(type.size() + 24 + content.size());
    full += type;
    full += ' ';
    full += std::to_string(content.size());
    full += '\0';
    full += content;
    return full;
}

// zlib compress() at the default level, as reference src/commit.cpp:98-105.
bool deflateFramed(const std::string& type, const std::string& content, std::string& out)
{
    const std::string full = frame(type, content);
    uLongf dst = compressBound(uLong(full.size()));
    out.resize(dst);
    if (compress(reinterpret_cast<Bytef*>(&out[0]), &dst, reinterpret_cast<const Bytef*>(full.data()),
                 uLong(full.size())) != Z_OK)
        return false;
    out.resize(dst);
    return true;
}

// Write atomically: a reader (or a crash) never sees a half-written object.
bool writeFileAtomic(const std::string& path, const std::string& bytes)
{
    static std::atomic<uint64_t> serial{0};
    const std::string tmp = path + ".tmp" + std::to_string(uint64_t(getpid())) + "_" + std::to_string(serial++);
    {
        std::ofstream f(tmp, std::ios::binary);
        if (!f) return false;
        f.write(bytes.data(), std::streamsize(bytes.size()));
        if (!f) { std::remove(tmp.c_str()); return false; }
    }
    std::error_code ec;
    fs::rename(tmp, path, ec);
    if (ec) { std::remove(tmp.c_str()); return false; }
    return true;
}

std::string slurp(const std::string& path)
{
    // reference src/commit.cpp:151-152: an unreadable file yields empty content.
    std::ifstream f(path, std::ios::binary);
    if (!f) return {};
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    if (n <= 0) { std::ostringstream ss; f.clear(); f.seekg(0); ss << f.rdbuf(); return ss.str(); }
    std::string s(size_t(n), '\0');
    f.seekg(0);
    f.read(&s[0], n);
    s.resize(size_t(f.gcount()));
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

struct Ent { std::string name; bool is_dir; size_t ref; };
struct DirNode { std::string path; int depth; std::vector<Ent> ents; uint8_t id[20]; };
struct FileRec { std::string path; uint64_t size; uint8_t id[20]; };

// Depth-first walk with the reference's rules (src/commit.cpp:143-161): skip
// entries named ".git"; status checks follow symlinks; anything that is neither
// a regular file nor a directory makes the whole tree fail.
bool walk(const std::string& dir, int depth, std::vector<DirNode>& dirs, std::vector<FileRec>& files, size_t& self)
{
    self = dirs.size();
    dirs.push_back(DirNode{dir, depth, {}, {}});
    std::error_code ec;
    fs::directory_iterator it(dir, ec), end;
    if (ec) { std::cerr << "cannot read directory " << dir << ": " << ec.message() << '\n'; return false; }
    std::vector<Ent> ents;
    for (; it != end; it.increment(ec)) {
        if (ec) return false;
        const std::string name = it->path().filename().string();
        if (name == ".git") continue;
        std::error_code sec;
        if (it->is_regular_file(sec)) {
            uint64_t sz = it->file_size(sec);
            if (sec) sz = 0;
            ents.push_back(Ent{name, false, files.size()});
            files.push_back(FileRec{it->path().string(), sz, {}});
        } else if (it->is_directory(sec)) {
            size_t child;
            if (!walk(it->path().string(), depth + 1, dirs, files, child)) return false;
            ents.push_back(Ent{name, true, child});
        } else {
            return false;                                 // reference: empty hash -> return {}
        }
    }
    dirs[self].ents = std::move(ents);
    return true;
}

}  // namespace

namespace vit {

ObjectStore::ObjectStore(std::string objects_dir, std::vector<int> devices) : objects_dir_(std::move(objects_dir))
{
    unsigned hw = std::thread::hardware_concurrency();
    io_threads_ = int(hw ? std::min(hw, 16u) : 4);
    if (vit_hasher_create(devices.empty() ? nullptr : devices.data(), int(devices.size()), &hasher_) != VIT_OK) {
        std::cerr << "vit: cannot create the GPU hasher: " << vit_last_error(nullptr) << '\n';
        hasher_ = nullptr;
        return;
    }
    // Tuning knobs for experiments; the defaults are the library's.
    if (const char* e = std::getenv("VIT_HOST_THRESHOLD")) vit_hasher_set_option(hasher_, "host_threshold_bytes", std::strtoull(e, nullptr, 10));
    if (const char* e = std::getenv("VIT_CHUNK_MB")) vit_hasher_set_option(hasher_, "chunk_bytes", std::strtoull(e, nullptr, 10) << 20);
    if (const char* e = std::getenv("VIT_KERNEL")) vit_hasher_set_option(hasher_, "kernel", std::strtoull(e, nullptr, 10));
}

ObjectStore::~ObjectStore()
{
    if (hasher_) vit_hasher_destroy(hasher_);
}

std::string ObjectStore::objectPath(const std::string& hash) const
{
    return objects_dir_ + '/' + hash.substr(0, 2) + '/' + hash.substr(2);
}

bool ObjectStore::hashObjects(const std::string& type, const std::vector<const std::string*>& contents,
                              std::vector<uint8_t>& out)
{
    if (!hasher_) return false;
    const size_t n = contents.size();
    std::vector<const uint8_t*> ptr(n);
    std::vector<uint64_t> len(n);
    for (size_t i = 0; i < n; ++i) {
        ptr[i] = reinterpret_cast<const uint8_t*>(contents[i]->data());
        len[i] = contents[i]->size();
    }
    out.resize(n * 20);
    if (vit_hash_objects(hasher_, type.c_str(), ptr.data(), len.data(), n, out.data()) != VIT_OK) {
        std::cerr << "vit: hashing failed: " << vit_last_error(hasher_) << '\n';
        return false;
    }
    return true;
}

bool ObjectStore::storeFramed(const std::string& hex, const std::string& type, const std::string& content) const
{
    std::string z;
    if (!deflateFramed(type, content, z)) { std::cerr << "Compression failed\n"; return false; }
    std::error_code ec;
    fs::create_directories(objects_dir_ + '/' + hex.substr(0, 2), ec);
    return writeFileAtomic(objectPath(hex), z);
}

std::vector<std::string> ObjectStore::writeObjects(const std::string& type, const std::vector<std::string>& contents)
{
    const size_t n = contents.size();
    std::vector<std::string> ids;
    if (n == 0 || !hasher_) return ids;
    std::vector<const std::string*> view(n);
    for (size_t i = 0; i < n; ++i) view[i] = &contents[i];

    // Deflate needs no id, so it runs on the pool while the GPU hashes; only the
    // file name waits for the digest.
    std::vector<std::string> z(store_ ? n : 0);
    std::atomic<bool> z_ok{true};
    Pool pool(store_ && n > 1 ? io_threads_ : 1);
    const auto t0 = clk::now();
    if (store_)
        for (size_t s = 0, slices = std::min<size_t>(n, size_t(io_threads_) * 4); s < slices; ++s)
            pool.submit([&, s, slices] {
                for (size_t i = n * s / slices; i < n * (s + 1) / slices; ++i)
                    if (!deflateFramed(type, contents[i], z[i])) z_ok = false;
            });
    std::vector<uint8_t> raw;
    const bool hashed = hashObjects(type, view, raw);
    stats_.hash_s += since(t0);
    pool.wait();
    if (!hashed) return ids;
    if (!z_ok) { std::cerr << "Compression failed\n"; return ids; }

    ids.resize(n);
    for (size_t i = 0; i < n; ++i) ids[i] = toHex(&raw[i * 20], 20);
    if (store_) {
        std::atomic<bool> w_ok{true};
        pool.forEach(n, [&](size_t i) {
            std::error_code ec;
            fs::create_directories(objects_dir_ + '/' + ids[i].substr(0, 2), ec);
            if (!writeFileAtomic(objectPath(ids[i]), z[i])) w_ok = false;
        });
        if (!w_ok) { std::cerr << "vit: writing objects under " << objects_dir_ << " failed\n"; return {}; }
    }
    stats_.deflate_store_s += since(t0);
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
    const auto t_all = clk::now();
    if (!hasher_) return {};

    std::vector<DirNode> dirs;
    std::vector<FileRec> files;
    size_t root;
    auto t0 = clk::now();
    if (!walk(directoryPath, 0, dirs, files, root)) return {};
    stats_.walk_s = since(t0);

    // ---- blobs: bounded batches: read (pool) -> hash on GPU || deflate (pool) -> store (pool)
    Pool reader(io_threads_);
    for (size_t pos = 0; pos < files.size();) {
        size_t end = pos;
        uint64_t bytes = 0;
        while (end < files.size() && (end == pos || bytes + files[end].size <= batch_bytes_)) bytes += files[end++].size;
        std::vector<std::string> contents(end - pos);
        t0 = clk::now();
        reader.forEach(end - pos, [&](size_t i) { contents[i] = slurp(files[pos + i].path); });
        stats_.read_s += since(t0);
        std::vector<std::string> ids = writeObjects("blob", contents);
        if (ids.size() != contents.size()) return {};
        for (size_t i = 0; i < ids.size(); ++i) {
            for (int k = 0; k < 20; ++k)
                files[pos + i].id[k] = uint8_t(hexNibble(ids[i][2 * k]) * 16 + hexNibble(ids[i][2 * k + 1]));
            stats_.bytes += contents[i].size();
        }
        stats_.blobs += ids.size();
        pos = end;
    }

    // ---- trees: deepest level first, every directory of a level in one batch
    int max_depth = 0;
    for (const DirNode& d : dirs) max_depth = std::max(max_depth, d.depth);
    for (int depth = max_depth; depth >= 0; --depth) {
        std::vector<size_t> level;
        for (size_t i = 0; i < dirs.size(); ++i) if (dirs[i].depth == depth) level.push_back(i);
        std::vector<std::string> contents(level.size());
        for (size_t k = 0; k < level.size(); ++k) {
            DirNode& d = dirs[level[k]];
            // reference src/commit.cpp:164-165: plain std::string order of the names
            std::sort(d.ents.begin(), d.ents.end(), [](const Ent& a, const Ent& b) { return a.name < b.name; });
            std::string& c = contents[k];
            for (const Ent& e : d.ents) {                  // :167-171
                c += e.is_dir ? "40000" : "100644";
                c += ' ';
                c += e.name;
                c += '\0';
                c.append(reinterpret_cast<const char*>(e.is_dir ? dirs[e.ref].id : files[e.ref].id), 20);
            }
        }
        std::vector<std::string> ids = writeObjects("tree", contents);
        if (ids.size() != contents.size()) return {};
        for (size_t k = 0; k < level.size(); ++k)
            for (int b = 0; b < 20; ++b)
                dirs[level[k]].id[b] = uint8_t(hexNibble(ids[k][2 * b]) * 16 + hexNibble(ids[k][2 * b + 1]));
        stats_.trees += level.size();
    }
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
