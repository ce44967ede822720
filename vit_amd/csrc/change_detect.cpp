// change_detect.cpp — see change_detect.hpp.
#include "change_detect.hpp"

#include "object_store.hpp"

#include <algorithm>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <set>
#include <sstream>

namespace fs = std::filesystem;

namespace vit {

std::vector<TreeItem> parseTree(const ObjectStore& store, const std::string& treeHash)
{
    std::vector<TreeItem> out;
    const std::string content = store.readObjectContent(treeHash);
    for (size_t pos = 0; pos < content.size();) {
        const size_t nul = content.find('\0', pos);
        if (nul == std::string::npos || nul + 21 > content.size()) break;
        const std::string head = content.substr(pos, nul - pos);
        const size_t sp = head.find_last_of(' ');
        if (sp != std::string::npos) {
            TreeItem it;
            it.mode = head.substr(0, sp);
            it.name = head.substr(sp + 1);
            it.hash = binaryToHexString(content.substr(nul + 1, 20));
            it.isDirectory = it.mode.compare(0, 3, "400") == 0;
            out.push_back(std::move(it));
        }
        pos = nul + 21;
    }
    return out;
}

static void collect(const ObjectStore& store, const std::string& treeHash, const std::string& base,
                    std::unordered_map<std::string, std::string>& map)
{
    for (const TreeItem& it : parseTree(store, treeHash)) {
        const std::string full = base.empty() ? it.name : base + '/' + it.name;
        if (it.isDirectory) collect(store, it.hash, full, map);
        else map[full] = it.hash;
    }
}

std::unordered_map<std::string, std::string> treeFileMap(const ObjectStore& store, const std::string& treeHash)
{
    std::unordered_map<std::string, std::string> map;
    collect(store, treeHash, "", map);
    return map;
}

std::vector<std::string> worktreeFiles(const std::string& root)
{
    std::vector<std::string> files;
    std::set<std::string> seen;
    std::error_code ec;
    // the reference's iterator has its default options (src/utils/file_utils.cpp:60): a symlink to a directory is not
    // descended into (and is no regular file, so it is passed over); a symlink to a file counts as the file
    fs::recursive_directory_iterator it(root, ec), end;
    if (ec) { std::cerr << "Error scanning directory " << root << ": " << ec.message() << '\n'; return files; }
    for (; it != end; it.increment(ec)) {
        if (ec) break;
        std::error_code sec;
        if (!it->is_regular_file(sec)) continue;
        std::string p = it->path().string();
        if (p.find("/.git/") != std::string::npos || p.find("/build/") != std::string::npos ||
            p.find("/node_modules/") != std::string::npos)
            continue;
        // relative to root, without a leading "./"
        if (p.compare(0, root.size(), root) == 0) p.erase(0, root.size());
        while (!p.empty() && p[0] == '/') p.erase(0, 1);
        std::replace(p.begin(), p.end(), '\\', '/');
        if (seen.insert(p).second) files.push_back(p);
    }
    return files;
}

ChangeSet detectChanges(ObjectStore& store, const std::string& treeHash, const std::string& root)
{
    ChangeSet out;
    const auto committed = treeFileMap(store, treeHash);
    const std::vector<std::string> working = worktreeFiles(root);

    std::set<std::string> all;
    for (const auto& kv : committed) all.insert(kv.first);
    for (const auto& p : working) all.insert(p);
    out.totalFilesAnalyzed = all.size();

    // One batch over every worktree file (ids only, nothing is stored).  A file that cannot be opened: the reference's
    // readFile throws (src/utils/file_utils.cpp:11-15) and its caller records nothing for the path (:94-96).
    std::vector<std::string> contents(working.size());
    std::vector<bool> unreadable(working.size(), false);
    for (size_t i = 0; i < working.size(); ++i) {
        std::ifstream f(root + '/' + working[i], std::ios::binary);
        if (!f.is_open()) { unreadable[i] = true; continue; }
        std::ostringstream ss;
        ss << f.rdbuf();
        contents[i] = ss.str();
    }
    std::vector<const std::string*> view(contents.size());
    for (size_t i = 0; i < contents.size(); ++i) view[i] = &contents[i];
    std::vector<uint8_t> raw;
    if (!view.empty() && !store.hashObjects("blob", view, raw)) { out.ok = false; return out; }

    std::set<std::string> skipped;
    for (size_t i = 0; i < working.size(); ++i)
        if (unreadable[i] && committed.count(working[i])) {
            std::cerr << "Warning: Could not analyze file " << working[i] << ": Failed to open file: " << working[i] << '\n';
            skipped.insert(working[i]);
        }
    std::unordered_map<std::string, std::string> now;
    for (size_t i = 0; i < working.size(); ++i) now[working[i]] = hashToHexString(&raw[i * 20]);

    for (const std::string& path : all) {
        const auto c = committed.find(path);
        const auto w = now.find(path);
        if (c == committed.end()) out.changes.push_back({path, ChangeType::ADDED, "", w->second});
        else if (w == now.end()) out.changes.push_back({path, ChangeType::DELETED, c->second, ""});
        else if (skipped.count(path)) continue;
        else if (c->second != w->second) out.changes.push_back({path, ChangeType::MODIFIED, c->second, w->second});
    }
    return out;
}

}  // namespace vit
