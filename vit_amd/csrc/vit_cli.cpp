// vit_cli.cpp — the plumbing commands of the reference that sit on the
// content-addressing path, on top of the GPU-backed object store:
//
//   vit init
//   vit hash-object -w <file>            reference src/main.cpp:90-120
//   vit write-tree [--stats] [--no-store] [--rewrite] [--git-order] [--deflate-level=N]    reference src/main.cpp:155-163
//                                        (--git-order: Git's tree rules where the reference's differ;
//                                         --deflate-level: zlib level of the stored objects, default the reference's)
//   vit commit-tree <tree> -p <parent> -m <message>   reference src/main.cpp:165-186
//   vit cat-file -p <hash>               reference src/main.cpp:68-88
//   vit ls-tree <hash> --name-only|-l    reference src/main.cpp:122-153
//   vit diff-tree <tree-or-commit>       which worktree files differ from a stored tree, decided by id
//                                        (hash-based form of reference src/utils/change_analyzer.cpp:56-98)
//
// stdout carries exactly what the reference prints (the id and a newline);
// the exit status is 0 on success, 1 on failure.  Author name and e-mail come
// from the reference's ".vitconfig" (three whitespace separated fields,
// src/main.cpp:35-41) or $VIT_AUTHOR / $VIT_EMAIL; $VIT_COMMIT_TIME pins the
// clock for reproducible commits; $VIT_DEVICES picks the GPUs.
#include "change_detect.hpp"
#include "object_store.hpp"

#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <sstream>

namespace {

struct Config { bool localAI = true; std::string userName, userEmail; };

Config loadConfig()
{
    Config c;
    std::ifstream f(".vitconfig");
    if (f) { f >> c.localAI; f >> c.userName; f >> c.userEmail; }
    if (const char* e = std::getenv("VIT_AUTHOR")) c.userName = e;
    if (const char* e = std::getenv("VIT_EMAIL")) c.userEmail = e;
    return c;
}

bool cmdInit()
{
    std::error_code ec;
    for (const char* d : {".git", ".git/objects", ".git/refs", ".git/refs/heads"}) {
        std::filesystem::create_directory(d, ec);
        if (ec) { std::cerr << ec.message() << '\n'; return false; }
    }
    std::ofstream head(".git/HEAD");
    if (!head) { std::cerr << "Failed to create .git/HEAD file.\n"; return false; }
    head << "ref: refs/heads/main\n";
    std::cout << "Initialized vit directory\n";
    return true;
}

bool cmdHashObject(int argc, char** argv)
{
    if (argc < 4) { std::cerr << "Usage: hash-object -w <file>\n"; return false; }
    if (std::string(argv[2]) != "-w") { std::cerr << "Unknown flag: " << argv[2] << '\n'; return false; }
    std::ifstream in(argv[3], std::ios::binary);
    if (!in.is_open()) { std::cerr << "Failed to open file: " << argv[3] << '\n'; return false; }
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string id = writeBlob(ss.str());
    if (id.empty()) return false;
    std::cout << id << '\n';
    return true;
}

bool cmdWriteTree(int argc, char** argv)
{
    bool stats = false;
    for (int i = 2; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--stats")) stats = true;
        else if (!std::strcmp(argv[i], "--no-store")) vit::defaultStore().setStoreObjects(false);
        else if (!std::strcmp(argv[i], "--rewrite")) vit::defaultStore().setSkipExisting(false);
        else if (!std::strcmp(argv[i], "--git-order")) vit::defaultStore().setGitTreeRules(true);
        else if (!std::strcmp(argv[i], "--device-deflate")) vit::defaultStore().setDeviceDeflate(true);
        else if (!std::strcmp(argv[i], "--host-deflate")) vit::defaultStore().setDeviceDeflate(false);
        else if (!std::strncmp(argv[i], "--deflate-level=", 16)) vit::defaultStore().setDeflateLevel(std::atoi(argv[i] + 16));
        else { std::cerr << "Unknown flag: " << argv[i] << '\n'; return false; }
    }
    const std::string id = writeTree(".");
    if (id.empty()) { std::cerr << "Failed to write tree\n"; return false; }
    std::cout << id << '\n';
    if (stats) {
        const vit::StoreStats& s = vit::defaultStore().lastStats();
        std::cerr << "blobs " << s.blobs << " trees " << s.trees << " bytes " << s.bytes << " already-stored " << s.skipped
                  << " | ids-on " << (s.ids_on_host ? "host-threads" : "gpu") << " (crossover " << s.crossover_bytes << " bytes; " << s.gpu_hashed
                  << " ids from the device, " << s.host_hashed << " from host threads) zlib " << vit::zlibVersionString()
                  << " device-streams-rejected " << s.device_rejected
                  << " | init " << s.init_s << " s walk " << s.walk_s
                  << " s read-wait " << s.read_s << " s hash " << s.hash_s << " s device-deflate " << s.device_deflate_s << " s (" << s.device_deflated
                  << " objects) blobs-pipeline(read|hash|deflate+store) " << s.deflate_store_s
                  << " s trees(ids+deflate+store, level by level) " << s.trees_s << " s of which ids " << s.tree_hash_s
                  << " s total " << s.total_s << " s\n";
    }
    return true;
}

bool cmdCommitTree(int argc, char** argv)
{
    if (argc < 7) { std::cerr << "Usage: commit-tree <tree_sha> -p <commit_sha> -m <message>\n"; return false; }
    const Config cfg = loadConfig();
    std::time_t now = std::time(nullptr);
    if (const char* e = std::getenv("VIT_COMMIT_TIME")) now = std::time_t(std::atoll(e));
    const std::string id = vit::defaultStore().writeCommit(argv[2], argv[4], argv[6], cfg.userName, cfg.userEmail, now);
    if (id.empty()) { std::cerr << "Failed to create commit\n"; return false; }
    std::cout << id << '\n';
    return true;
}

bool cmdCatFile(int argc, char** argv)
{
    if (argc < 4) { std::cerr << "Usage: cat-file -p <hash>\n"; return false; }
    if (std::string(argv[2]) != "-p") { std::cerr << "Unknown flag: " << argv[2] << '\n'; return false; }
    const std::string content = readObjectContent(argv[3]);
    if (content.empty()) return false;
    std::cout << content;
    return true;
}

bool cmdLsTree(int argc, char** argv)
{
    if (argc < 4) { std::cerr << "Usage: ls-tree <hash> [--name-only|-l]\n"; return false; }
    const std::string flag = argv[3];
    if (flag != "--name-only" && flag != "-l") { std::cerr << "Unknown flag: " << flag << '\n'; return false; }
    const std::string content = readObjectContent(argv[2]);
    if (content.empty()) { std::cerr << "Tree not found or empty: " << argv[2] << '\n'; return false; }
    for (size_t pos = 0; pos < content.size();) {
        const size_t nul = content.find('\0', pos);
        if (nul == std::string::npos || nul + 21 > content.size()) break;
        const std::string head = content.substr(pos, nul - pos);
        const size_t sp = head.find_last_of(' ');
        if (sp != std::string::npos) {
            const std::string mode = head.substr(0, sp), name = head.substr(sp + 1);
            if (flag == "--name-only") std::cout << name << '\n';
            else std::cout << mode << ' ' << (mode.rfind("400", 0) == 0 ? "tree" : "blob") << ' '
                           << binaryToHexString(content.substr(nul + 1, 20)) << '\t' << name << '\n';
        }
        pos = nul + 21;
    }
    return true;
}

bool cmdDiffTree(int argc, char** argv)
{
    if (argc < 3) { std::cerr << "Usage: diff-tree <tree-or-commit hash>\n"; return false; }
    std::string tree = argv[2];
    const std::string obj = readObject(tree);
    if (obj.empty()) return false;
    if (obj.rfind("commit ", 0) == 0) {                     // take the commit's tree line
        const size_t at = obj.find("tree ");
        if (at == std::string::npos || at + 45 > obj.size()) { std::cerr << "Invalid commit: " << tree << '\n'; return false; }
        tree = obj.substr(at + 5, 40);
    }
    const vit::ChangeSet cs = vit::detectChanges(vit::defaultStore(), tree, ".");
    if (!cs.ok) return false;
    for (const vit::PathChange& c : cs.changes)
        std::cout << (c.changeType == vit::ChangeType::ADDED ? 'A' : c.changeType == vit::ChangeType::DELETED ? 'D' : 'M')
                  << '\t' << c.filePath << '\n';
    std::cerr << cs.totalFilesAnalyzed << " files analyzed, " << cs.changes.size() << " changed\n";
    return true;
}

}  // namespace

int main(int argc, char** argv)
{
    std::cout << std::unitbuf;
    std::cerr << std::unitbuf;
    if (argc < 2) { std::cerr << "No command provided.\n"; return EXIT_FAILURE; }
    const std::string cmd = argv[1];
    bool ok;
    if (cmd == "init") ok = cmdInit();
    else if (cmd == "hash-object") ok = cmdHashObject(argc, argv);
    else if (cmd == "write-tree") ok = cmdWriteTree(argc, argv);
    else if (cmd == "commit-tree") ok = cmdCommitTree(argc, argv);
    else if (cmd == "cat-file") ok = cmdCatFile(argc, argv);
    else if (cmd == "ls-tree") ok = cmdLsTree(argc, argv);
    else if (cmd == "diff-tree") ok = cmdDiffTree(argc, argv);
    else { std::cerr << "Unknown command " << cmd << '\n'; return EXIT_FAILURE; }
    return ok ? EXIT_SUCCESS : EXIT_FAILURE;
}
