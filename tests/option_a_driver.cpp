// A caller written against the REFERENCE's interface (the free functions of its src/commit.hpp:8-25),
// linked against this repository's object_store.cpp instead of the reference's commit.cpp: the rehearsal
// of the source-level drop-in (the reference's callers unchanged, its commit.cpp replaced).  Built by vit_amd/build.py to
// vit_amd/bin/option_a_driver; run by tests/test_option_a.py.
//
//   option_a_driver helpers                      the helpers that need no device (hex, paths)
//   option_a_driver store <file>...              writeBlob each file, then writeTree("."), readObject, writeCommit
//   option_a_driver pipe                         writeTree("."), then close a pipe's write end: its reader must see end-of-file
//                                                (the store's worker threads must not hold copies of the host program's descriptors)
#include "object_store.hpp"

#include <poll.h>
#include <unistd.h>

#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    if (!std::strcmp(argv[1], "helpers")) {
        unsigned char raw[20];
        for (int i = 0; i < 20; ++i) raw[i] = static_cast<unsigned char>(i * 13 + 7);
        const std::string hex = hashToHexString(raw);                              // src/commit.cpp:52-59
        std::cout << "hex " << hex << '\n';
        const std::string bin = hexStringToBinary(hex);                            // :61-70
        std::cout << "roundtrip " << (bin.size() == 20 && std::memcmp(bin.data(), raw, 20) == 0 ? "ok" : "BAD") << '\n';
        std::cout << "upper " << binaryToHexString(hexStringToBinary("00FFAaBb0102030405060708090a0b0c0d0e0f10")) << '\n';
        std::cout << "binhex " << binaryToHexString(std::string("\x00\xff\x10", 3)) << '\n';   // :72-79, any length
        std::cout << "path " << getObjectPath("ce013625030ba8dba906f756967f9e9ca394464a") << '\n';   // :81-84
        try { hexStringToBinary("zz"); std::cout << "badhex accepted\n"; }
        catch (const std::exception&) { std::cout << "badhex throws\n"; }          // std::stoi throws in the reference too
        // reading needs no device either: a missing object is an empty string and a message, as :120-123
        std::cout << "missing [" << readObjectContent("0000000000000000000000000000000000000000") << "]\n";
        return 0;
    }
    if (!std::strcmp(argv[1], "store")) {
        for (int i = 2; i < argc; ++i) {
            std::ifstream in(argv[i], std::ios::binary);
            std::stringstream ss;
            ss << in.rdbuf();
            const std::string id = writeBlob(ss.str());                            // :116
            if (id.empty()) return 1;
            std::cout << "blob " << id << ' ' << getObjectPath(id) << '\n';
            if (readObjectContent(id) != ss.str()) { std::cout << "readback differs\n"; return 1; }   // :131-136
            const std::string full = readObject(id);                               // :120-129
            if (full.compare(0, 5, "blob ") != 0) { std::cout << "bad framing\n"; return 1; }
        }
        const std::string tree = writeTree(".");                                   // :139-173
        if (tree.empty()) return 1;
        std::cout << "tree " << tree << '\n';
        const std::string typed = writeObject("tree", "");                         // :88-114
        std::cout << "emptytree " << typed << '\n';
        const std::string commit = writeCommit(tree, "", "first", "A U Thor", "a@example.org");   // :263-276
        if (commit.empty()) return 1;
        const std::string text = readObjectContent(commit);
        std::cout << "commit " << commit << " tree-line " << (text.compare(0, 5 + 40, "tree " + tree) == 0 ? "ok" : "BAD") << '\n';
        return 0;
    }
    if (!std::strcmp(argv[1], "pipe")) {
        int fd[2];
        if (::pipe(fd) != 0) return 1;
        const std::string tree = writeTree(".");            // starts the store's pool, which lives on in the default store
        if (tree.empty()) return 1;
        std::cout << "tree " << tree << '\n';
        ::close(fd[1]);
        struct pollfd p = {fd[0], POLLIN, 0};
        char byte;
        const bool eof = ::poll(&p, 1, 2000) == 1 && ::read(fd[0], &byte, 1) == 0;
        std::cout << "pipe " << (eof ? "eof" : "STILL-OPEN") << '\n';
        return eof ? 0 : 1;
    }
    return 2;
}
