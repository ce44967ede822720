// sha1_host.h — streaming SHA-1 on a host core.
//
// Used only for objects the scheduler keeps off the GPU because a single SHA-1
// stream is serial: one GPU lane hashes tens of MB/s, a host core with SHA
// extensions more than a GB/s, so an object above the hasher's
// host_threshold_bytes is hashed here, concurrently with the GPU batch.
// Same function the reference gets from OpenSSL SHA1() (src/commit.cpp:93-94).
#pragma once
#include <cstddef>
#include <cstdint>

namespace vit {

class HostSha1 {
public:
    HostSha1();
    void update(const void* data, size_t len);
    void finish(uint8_t out[20]);
    // True when the SHA-NI block function is in use on this CPU.
    static bool accelerated();

    // For host_object_ids: whole 64-byte blocks that were fed to the block
    // function directly; only between blocks (no partial block buffered).
    uint32_t* state() { return h_; }
    size_t to_block_boundary() const { return size_t((64 - (nbytes_ & 63)) & 63); }   // bytes that complete the buffered block
    void account_blocks(uint64_t nblocks) { nbytes_ += nblocks * 64; }

private:
    uint32_t h_[5];
    uint64_t nbytes_;
    uint8_t  buf_[64];
};

// Git object id of (type, content) on the host.
void host_object_id(const char* type, const uint8_t* content, uint64_t len, uint8_t out[20]);

// The same for `count` objects (any count) on ONE core, up to kHostStreams of
// them at a time with their block functions interleaved: sha1rnds4 has a
// latency of several cycles and one stream's rounds form a dependent chain, so
// a core running four independent streams side by side gets through about 2.3
// times the bytes of one (measured on an EPYC: 2.6, 4.6, 5.2, 5.9 GB/s for one
// to four streams; an Intel core whose sha1rnds4 is not pipelined gains
// nothing and loses nothing).  out: count x 20 bytes.  Objects of similar
// length interleave best: the streams run in lock step until the shortest ends.
constexpr int kHostStreams = 4;
void host_object_ids(const char* type, const uint8_t* const* content, const uint64_t* len, size_t count, uint8_t* out);

}  // namespace vit
