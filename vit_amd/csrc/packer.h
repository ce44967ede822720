// packer.h — host-side arena packing primitives (see packer.cpp).
#pragma once
#include <cstdint>

namespace vit {

// strlen(type) if type is a usable object type name (1..8 bytes, no space), else 0.
uint32_t type_len_checked(const char* type);
// "<type> <len>\0" into out; returns the header length.  type_len 0 = a raw
// buffer without Git framing: nothing is written, 0 is returned.
uint32_t write_header(const char* type, uint32_t type_len, uint64_t len, uint8_t* out);
// Header + content + SHA-1 padding into a slot of nblocks*64 bytes.
void pack_one(const char* type, uint32_t type_len, const uint8_t* data, uint64_t len, uint8_t* slot,
              uint64_t nblocks);
// pack_one writes with non-temporal stores: each packing thread calls this once
// after its last slot, before another agent (a copy engine) reads the arena.
void pack_fence();
// order[] = indices by block count, longest first, stable; false if all equal.
bool bin_order(const uint32_t* nblocks, uint64_t n, uint32_t* order);

}  // namespace vit
