// change_detect.hpp — which worktree files differ from a stored tree, by id.
//
// The reference's ChangeAnalyzer (src/utils/change_analyzer.cpp:11-103) decides
// "modified" by inflating the committed blob and comparing the two contents as
// strings (:85-88), one file at a time.  With a batched hasher the same
// question is a comparison of 20-byte ids: the committed id is already in the
// tree (collectTreeFileMap, :136-149), the worktree id comes from one GPU
// batch over all candidate files, and only files that really changed need
// their old content read back.
#pragma once
#include <cstddef>
#include <string>
#include <unordered_map>
#include <vector>

namespace vit {

class ObjectStore;

enum class ChangeType { ADDED, MODIFIED, DELETED };       // reference change_analyzer.hpp:15-19

struct PathChange {
    std::string filePath;
    ChangeType  changeType;
    std::string oldHash;                                   // empty for ADDED
    std::string newHash;                                   // empty for DELETED
};

struct ChangeSet {
    std::vector<PathChange> changes;                       // sorted by path
    size_t totalFilesAnalyzed = 0;                         // union of committed and worktree paths (:54)
    bool ok = true;
    bool hasChanges() const { return !changes.empty(); }
};

struct TreeItem { std::string mode, name, hash; bool isDirectory; };

// Entries of one stored tree object (reference parseTree, src/commit.cpp:175-200).
std::vector<TreeItem> parseTree(const ObjectStore& store, const std::string& treeHash);

// path -> blob id of every file under a stored tree (reference
// getCommitFileMap / collectTreeFileMap, change_analyzer.cpp:130-149).
std::unordered_map<std::string, std::string> treeFileMap(const ObjectStore& store, const std::string& treeHash);

// Regular files under `root`, relative, '/' separated, skipping paths that
// contain "/.git/", "/build/" or "/node_modules/" (reference
// FileUtils::getFilesInDirectory, src/utils/file_utils.cpp:55-80, and
// normalizeFilePath, change_analyzer.cpp:160-172).
std::vector<std::string> worktreeFiles(const std::string& root);

// Compare the worktree under `root` with the stored tree `treeHash`.
// Classification follows change_analyzer.cpp:56-98; unlike the reference, which
// builds ADDED records and then drops them (:62-72), added files are reported.
ChangeSet detectChanges(ObjectStore& store, const std::string& treeHash, const std::string& root = ".");

}  // namespace vit
