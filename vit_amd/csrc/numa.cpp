// numa.cpp — see numa.h.  Host-only.
#include "numa.h"

#include "../../include/vit_hash_bench.h"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace vit {
namespace numa {

std::vector<int> parse_cpulist(const std::string& text)
{
    std::vector<int> out;
    size_t i = 0;
    auto number = [&](int* v) {
        if (i >= text.size() || !std::isdigit(static_cast<unsigned char>(text[i]))) return false;
        long x = 0;
        while (i < text.size() && std::isdigit(static_cast<unsigned char>(text[i])) && x < (1 << 24)) x = x * 10 + (text[i++] - '0');
        *v = int(x);
        return true;
    };
    while (i < text.size()) {
        int lo, hi;
        if (!number(&lo)) break;
        hi = lo;
        if (i < text.size() && text[i] == '-') { ++i; if (!number(&hi)) break; }
        if (hi < lo || hi - lo > (1 << 16)) break;
        for (int c = lo; c <= hi; ++c) out.push_back(c);
        if (i < text.size() && text[i] == ',') ++i; else break;
    }
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());
    return out;
}

static bool first_line(const std::string& path, std::string* line)
{
    std::ifstream f(path);
    if (!f) return false;
    std::getline(f, *line);
    return true;
}

std::vector<int> node_cpus(const std::string& sysfs_root, int node)
{
    std::string line;
    if (node < 0 || !first_line(sysfs_root + "/devices/system/node/node" + std::to_string(node) + "/cpulist", &line)) return {};
    return parse_cpulist(line);
}

int pci_device_node(const std::string& sysfs_root, const std::string& pci_bus_id)
{
    std::string id = pci_bus_id, line;
    for (char& c : id) c = char(std::tolower(static_cast<unsigned char>(c)));   // sysfs spells the hex digits in lower case
    if (id.empty() || !first_line(sysfs_root + "/bus/pci/devices/" + id + "/numa_node", &line)) return -1;
    char* end = nullptr;
    const long v = std::strtol(line.c_str(), &end, 10);
    return end == line.c_str() || v < 0 || v > 4095 ? -1 : int(v);
}

std::vector<int> plan_worker_nodes(const std::vector<int>& queue_node, int workers,
                                   const std::vector<std::vector<int>>& allowed_cpus_of_node)
{
    std::vector<int> none(size_t(std::max(workers, 0)), -1);
    if (workers <= 0 || queue_node.empty()) return none;
    std::map<int, int> queues_on;                          // node -> queues there
    for (int n : queue_node) {
        if (n < 0) return none;
        ++queues_on[n];
    }
    // Largest-remainder shares of the workers, by queues per node; every node at least one while workers last.
    const int nq = int(queue_node.size());
    std::map<int, int> share;
    int given = 0;
    for (const auto& kv : queues_on) { share[kv.first] = workers * kv.second / nq; given += share[kv.first]; }
    std::vector<std::pair<int, int>> rest;                 // (-remainder, node)
    for (const auto& kv : queues_on) rest.emplace_back(-((workers * kv.second) % nq), kv.first);
    std::sort(rest.begin(), rest.end());
    for (size_t k = 0; given < workers && k < rest.size(); ++k, ++given) ++share[rest[k].second];
    for (auto& kv : share) {
        if (kv.second > 0) continue;
        auto richest = std::max_element(share.begin(), share.end(), [](const auto& a, const auto& b) { return a.second < b.second; });
        if (richest->second > 1) { --richest->second; kv.second = 1; }
    }
    for (const auto& kv : share) {
        const size_t node = size_t(kv.first);
        const size_t usable = node < allowed_cpus_of_node.size() ? allowed_cpus_of_node[node].size() : 0;
        if (kv.second > 0 && usable < size_t(kv.second)) return none;
    }
    // Hand the nodes out in turn (node a, node b, node a, ...), so a prefix of the workers is spread like the whole.
    std::vector<int> out;
    std::map<int, int> left = share;
    while (int(out.size()) < workers) {
        bool any = false;
        for (auto& kv : left)
            if (kv.second > 0 && int(out.size()) < workers) { out.push_back(kv.first); --kv.second; any = true; }
        if (!any) break;
    }
    out.resize(size_t(workers), -1);
    return out;
}

}  // namespace numa
}  // namespace vit

extern "C" {

int vit_numa_plan(const char* sysfs_root, const char* const* pci_bus_ids, int n_queues, int workers, int* queue_node,
                  int* worker_node)
{
    if (n_queues < 1 || workers < 0 || !pci_bus_ids || !queue_node || (workers && !worker_node)) return VIT_ERR_ARG;
    try {
        const std::string root = sysfs_root ? sysfs_root : "/sys";
        std::vector<int> qn(size_t(n_queues), -1);
        int top = -1;
        for (int k = 0; k < n_queues; ++k) {
            qn[size_t(k)] = pci_bus_ids[k] ? vit::numa::pci_device_node(root, pci_bus_ids[k]) : -1;
            queue_node[k] = qn[size_t(k)];
            top = std::max(top, qn[size_t(k)]);
        }
        std::vector<std::vector<int>> cpus(size_t(top + 1));
        for (int n = 0; n <= top; ++n) cpus[size_t(n)] = vit::numa::node_cpus(root, n);
        const std::vector<int> wn = vit::numa::plan_worker_nodes(qn, workers, cpus);
        for (int w = 0; w < workers; ++w) worker_node[w] = wn[size_t(w)];
        return VIT_OK;
    } catch (const std::exception&) { return VIT_ERR_NOMEM; }
}

int vit_numa_node_cpus(const char* sysfs_root, int node, int* cpus, int cap)
{
    if (cap < 0 || (cap && !cpus)) return VIT_ERR_ARG;
    try {
        const std::vector<int> v = vit::numa::node_cpus(sysfs_root ? sysfs_root : "/sys", node);
        for (size_t i = 0; i < v.size() && i < size_t(cap); ++i) cpus[i] = v[i];
        return int(v.size());
    } catch (const std::exception&) { return VIT_ERR_NOMEM; }
}

}  // extern "C"
