// ref_shim.cpp — C-callable doorway into the reference's own object-store code.
//
// TEST INFRASTRUCTURE ONLY.  This file is compiled TOGETHER WITH the reference's
// src/commit.cpp, src/branch.cpp and src/utils/file_utils.cpp, taken where they lie under /root/reference
// (never copied into this repository), into oracle/_ref/libvitref.so by
// oracle/Makefile.  It contains no algorithm: each function forwards to the
// reference function named in its comment.  Used to pin the CPU restatement
// (vit_oracle.c), to generate tests/golden/, and as the "reference" CPU baseline.
//
// The reference's writeObject() stores into the CWD-relative ".git/objects/"
// (src/commit.cpp:108-111), so callers chdir() into a scratch repository first.
#include "commit.hpp"
#include "utils/file_utils.hpp"

#include <cstdint>
#include <cstring>
#include <string>

extern "C" {

// writeBlob, src/commit.cpp:116.  hex_out >= 41 bytes; returns 0, or -1 if the
// reference returned the empty string (its failure convention, :103,160).
int vitref_write_blob(const void* data, uint64_t len, char* hex_out)
{
    std::string h = writeBlob(std::string(static_cast<const char*>(data), len));
    if (h.empty()) return -1;
    std::memcpy(hex_out, h.c_str(), h.size() + 1);
    return 0;
}

// writeObject, src/commit.cpp:88-114.
int vitref_write_object(const char* type, const void* data, uint64_t len, char* hex_out)
{
    std::string h = writeObject(type, std::string(static_cast<const char*>(data), len));
    if (h.empty()) return -1;
    std::memcpy(hex_out, h.c_str(), h.size() + 1);
    return 0;
}

// writeTree, src/commit.cpp:139-173.
int vitref_write_tree(const char* dir, char* hex_out)
{
    std::string h = writeTree(dir);
    if (h.empty()) return -1;
    std::memcpy(hex_out, h.c_str(), h.size() + 1);
    return 0;
}

// hashToHexString, src/commit.cpp:52-59.
void vitref_hash_to_hex(const unsigned char* raw20, char* hex_out)
{
    std::string h = hashToHexString(raw20);
    std::memcpy(hex_out, h.c_str(), h.size() + 1);
}

// hexStringToBinary, src/commit.cpp:61-70.
void vitref_hex_to_binary(const char* hex, unsigned char* raw20)
{
    std::string b = hexStringToBinary(hex);
    std::memcpy(raw20, b.data(), 20);
}

// getObjectPath, src/commit.cpp:81-84.  out >= 64 bytes.
void vitref_object_path(const char* hex, char* out)
{
    std::string p = getObjectPath(hex);
    std::memcpy(out, p.c_str(), p.size() + 1);
}

// readObjectContent, src/commit.cpp:131-136 (inflate + strip header); returns
// the content length, copies at most cap bytes.
uint64_t vitref_read_object_content(const char* hex, void* out, uint64_t cap)
{
    std::string c = readObjectContent(hex);
    std::memcpy(out, c.data(), c.size() < cap ? c.size() : cap);
    return c.size();
}

// readObject, src/commit.cpp:120-129 (inflate, header kept).
uint64_t vitref_read_object(const char* hex, void* out, uint64_t cap)
{
    std::string c = readObject(hex);
    std::memcpy(out, c.data(), c.size() < cap ? c.size() : cap);
    return c.size();
}

// parseTree, src/commit.cpp:175-200.  Every entry as four NUL-terminated fields (mode, name, hex id,
// "d" or "f"); returns the bytes that takes, copies at most cap of them.
uint64_t vitref_parse_tree(const char* hex, void* out, uint64_t cap)
{
    std::string flat;
    for (const FileInfo& f : parseTree(hex)) {
        flat += f.mode; flat += '\0';
        flat += f.name; flat += '\0';
        flat += f.hash; flat += '\0';
        flat += f.isDirectory ? "d" : "f"; flat += '\0';
    }
    std::memcpy(out, flat.data(), flat.size() < cap ? flat.size() : cap);
    return flat.size();
}

// FileUtils::getFilesInDirectory, src/utils/file_utils.cpp:55-80.  The paths, each NUL-terminated;
// returns the bytes that takes, copies at most cap of them.
uint64_t vitref_files_in_directory(const char* dir, void* out, uint64_t cap)
{
    std::string flat;
    for (const std::string& p : vit::utils::FileUtils::getFilesInDirectory(dir)) { flat += p; flat += '\0'; }
    std::memcpy(out, flat.data(), flat.size() < cap ? flat.size() : cap);
    return flat.size();
}

}  // extern "C"
