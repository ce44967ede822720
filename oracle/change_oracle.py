"""CPU restatement of the reference's change detection: which worktree files differ from a
committed tree, decided the reference's way, by inflating the committed blob and comparing the
two CONTENTS (the product decides the same question by comparing 20-byte ids from one GPU batch).

TEST INFRASTRUCTURE ONLY: imported by tests/ and by tests/golden/gen_change_golden.py, never by
the product.  Pure Python on purpose: the work is directory listings, zlib and dictionary
look-ups on worktrees of at most some ten thousand files.

Each function names the reference lines it follows.  What pins it: parse_tree and
files_in_directory are checked against the reference's OWN parseTree (src/commit.cpp:175-200) and
FileUtils::getFilesInDirectory (src/utils/file_utils.cpp:55-80), compiled in place into
oracle/_ref/libvitref.so, through the fixture tests/golden/change_detect.json.  The
classification loop itself (src/utils/change_analyzer.cpp:40-103) cannot be compiled here: its
header pulls in the AI client (change_analyzer.hpp:8 -> ai/openai_client.hpp -> curl,
nlohmann-json, both absent), and the reference holds no test or fixture for it: for that loop
parity is UNPINNED, the restatement follows the source text only.
"""
import os
import zlib


def object_path(objects_dir, hex_id):
    """getObjectPath, src/commit.cpp:81-84 (relative to an objects directory instead of the CWD)."""
    return os.path.join(objects_dir, hex_id[:2], hex_id[2:])


def read_object_content(objects_dir, hex_id):
    """readObject + readObjectContent, src/commit.cpp:120-136: inflate, drop everything up to the first NUL.
    A missing object is the empty string (the reference prints "Object not found" and returns {})."""
    try:
        raw = open(object_path(objects_dir, hex_id), "rb").read()
    except OSError:
        return b""
    full = zlib.decompress(raw)
    nul = full.find(b"\0")
    return full if nul < 0 else full[nul + 1:]


def parse_tree(objects_dir, tree_hex):
    """parseTree, src/commit.cpp:175-200 -> list of (mode, name, hex id, is_directory).
    The header of an entry is split at its LAST space (:187), so a name that holds a space comes back as its last
    word, with the rest glued to the mode; an entry without any space is passed over; is_directory is "mode starts
    with 400" (:194)."""
    content = read_object_content(objects_dir, tree_hex)
    out = []
    pos = 0
    while pos < len(content):
        nul = content.find(b"\0", pos)
        if nul < 0:
            break
        header = content[pos:nul]
        sp = header.rfind(b" ")
        if sp >= 0:
            mode, name = header[:sp], header[sp + 1:]
            out.append((mode, name, content[nul + 1:nul + 21].hex(), mode[:3] == b"400"))
        pos = nul + 1 + 20
    return out


def commit_file_map(objects_dir, tree_hex, base=b""):
    """getCommitFileMap / collectTreeFileMap, src/utils/change_analyzer.cpp:130-149: path -> blob id."""
    files = {}
    for mode, name, hex_id, is_dir in parse_tree(objects_dir, tree_hex):
        full = name if not base else base + b"/" + name
        if is_dir:
            files.update(commit_file_map(objects_dir, hex_id, full))
        else:
            files[full] = hex_id
    return files


def find_file_in_tree(objects_dir, tree_hex, path):
    """findFileInTree, src/utils/change_analyzer.cpp:109-128: the committed CONTENT of a path; b"" if not found."""
    entries = parse_tree(objects_dir, tree_hex)
    for mode, name, hex_id, is_dir in entries:
        if name == path and not is_dir:
            return read_object_content(objects_dir, hex_id)
    for mode, name, hex_id, is_dir in entries:
        if is_dir and path.startswith(name + b"/"):
            return find_file_in_tree(objects_dir, hex_id, path[len(name) + 1:])
    return b""


def files_in_directory(directory):
    """FileUtils::getFilesInDirectory, src/utils/file_utils.cpp:55-80: every entry of a recursive walk that is a regular
    file (a symlink to one counts: is_regular_file follows it; directory symlinks are not descended into: the
    iterator's default options), as the iterator spells its path (directory + "/" + ...), except paths that contain
    "/.git/", "/build/" or "/node_modules/"; each path once."""
    directory = os.fsencode(directory)
    found, seen = [], set()
    stack = [directory]
    while stack:
        d = stack.pop()
        try:
            entries = list(os.scandir(d))
        except OSError:
            continue
        for e in entries:
            path = d + b"/" + e.name
            if e.is_dir(follow_symlinks=False):
                stack.append(path)
            try:
                regular = e.is_file(follow_symlinks=True)
            except OSError:
                regular = False
            if not regular:
                continue
            if b"/.git/" in path or b"/build/" in path or b"/node_modules/" in path:
                continue
            if path not in seen:
                seen.add(path)
                found.append(path)
    return found


def normalize(path):
    """normalizeFilePath, src/utils/change_analyzer.cpp:160-172: drop one leading "./", backslashes to slashes."""
    if path.startswith(b"./"):
        path = path[2:]
    return path.replace(b"\\", b"/")


def analyze_changes(objects_dir, tree_hex, worktree="."):
    """ChangeAnalyzer::analyzeChanges against a given tree, with sourceOnly off, src/utils/change_analyzer.cpp:45-103.
    Run with the worktree as the current directory, as the reference does (it lists "." and opens the normalized
    paths relative to it); `worktree` is made current for the call.
    Returns {"modified": set, "deleted": set, "added_then_dropped": set, "total": int}:
    the reference RECORDS modified and deleted files (:74-97); for a file that is only in the worktree it builds an
    ADDED record and lets it go out of scope (:62-72), so added files are not in its result."""
    here = os.getcwd()
    objects_dir = os.path.abspath(objects_dir)
    os.chdir(worktree)
    try:
        commit_files = commit_file_map(objects_dir, tree_hex)                        # :45
        working = [normalize(p) for p in files_in_directory(".")]                    # :46, :151-158
        all_paths = set(commit_files) | set(working)                                 # :49-51
        working_set = set(working)
        modified, deleted, added = set(), set(), set()
        for path in all_paths:                                                       # :56
            in_commit, in_working = path in commit_files, path in working_set        # :59-60
            if not in_commit and in_working:
                added.add(path)                                                      # :62-72: built, never pushed
            elif in_commit and not in_working:
                deleted.add(path)                                                    # :74-78
            elif in_commit and in_working:
                old = find_file_in_tree(objects_dir, tree_hex, path)                 # :85
                try:
                    new = open(path, "rb").read()                                    # :86 (readFile, file_utils.cpp:11-19)
                except OSError:
                    continue                                                         # readFile throws: caught at :94, nothing recorded
                if old != new:                                                       # :88
                    modified.add(path)
        return {"modified": modified, "deleted": deleted, "added_then_dropped": added, "total": len(all_paths)}   # :54
    finally:
        os.chdir(here)
