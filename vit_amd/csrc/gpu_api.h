// gpu_api.h — libvit_hash.so as the object store sees it: loaded when a batch needs it.
//
// A program that links the GPU library pays for the HIP runtime's shared objects
// at every start (about 15 ms of loading and static initialisers, before main),
// and for the runtime's own start-up (40 to 180 ms) under the first hasher.  One
// object, or a worktree a single core hashes in less time than that, is not worth
// either; `vit hash-object` and `vit cat-file` would pay for nothing.  The object
// store therefore links only the host-side sources (framing, host SHA-1) and
// opens the GPU library, through the table below, the first time a batch is large
// enough to go to the device.  The table holds exactly the entry points of
// include/vit_hash.h the store calls.
#pragma once
#include "../../include/vit_hash.h"

#include <string>

namespace vit {

struct GpuApi {
    decltype(&vit_hasher_create)      hasher_create = nullptr;
    decltype(&vit_hasher_destroy)     hasher_destroy = nullptr;
    decltype(&vit_last_error)         last_error = nullptr;
    decltype(&vit_hasher_set_option)  set_option = nullptr;
    decltype(&vit_hasher_get_stat)    get_stat = nullptr;
    decltype(&vit_hash_objects)       hash_objects = nullptr;
    decltype(&vit_arena_alloc)        arena_alloc = nullptr;
    decltype(&vit_arena_free)         arena_free = nullptr;
    decltype(&vit_deflate_objects)    deflate_objects = nullptr;
    decltype(&vit_deflate_out_bytes)  deflate_out_bytes = nullptr;

    // The library, opened once per process: $VIT_HASH_LIBRARY if set, else "libvit_hash.so" by
    // the loader's rules (the program's run path, LD_LIBRARY_PATH, the system's directories),
    // else next to the program or one directory above it.  nullptr, with the loader's reasons in
    // *why, when it cannot be opened or lacks a symbol: there is no other way to the device.
    static const GpuApi* load(std::string* why);
};

}  // namespace vit
