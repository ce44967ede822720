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

private:
    uint32_t h_[5];
    uint64_t nbytes_;
    uint8_t  buf_[64];
};

// Git object id of (type, content) on the host.
void host_object_id(const char* type, const uint8_t* content, uint64_t len, uint8_t out[20]);

}  // namespace vit
