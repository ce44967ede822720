// numa.h — where the host threads that feed a GPU should run.
//
// A GPU hangs off one socket's PCIe root.  The pinned staging arena of its queue
// is placed on that socket's memory by the runtime; a packing thread on the
// other socket writes every byte of it across the inter-socket link, and with
// eight GPUs on two sockets half of all packing would.  The hasher therefore
// reads each GPU's NUMA node from sysfs, pins a share of its pool's workers to
// the CPUs of each node that has a queue, and hands a queue's packing pieces to
// the workers of its node first.  Everything here is plain file parsing and
// arithmetic, no device needed; the sysfs root can be replaced for tests.
#pragma once
#include <string>
#include <vector>

namespace vit {
namespace numa {

// "0-3,8,10-11" -> {0,1,2,3,8,10,11}; anything unparsable ends the list.
std::vector<int> parse_cpulist(const std::string& text);

// CPUs of a node: <sysfs>/devices/system/node/node<N>/cpulist; empty when there is no such node.
std::vector<int> node_cpus(const std::string& sysfs_root, int node);

// NUMA node of a PCI function: <sysfs>/bus/pci/devices/<dddd:bb:dd.f>/numa_node;
// -1 when the file is missing or says -1 (a machine with one node).
int pci_device_node(const std::string& sysfs_root, const std::string& pci_bus_id);

// How many workers each node gets, and which: worker w is pinned to worker_node[w] (-1 = not
// pinned).  Nodes get workers in proportion to the queues they hold (at least one each, while
// there are workers), handed out in turn so that any prefix of the workers is spread the same
// way.  Nothing is pinned when the queues name fewer than two distinct nodes, when a queue's
// node is unknown, or when a node has fewer usable CPUs (allowed[n], after the process's own
// affinity mask) than the workers it would get: the pool then runs as the scheduler places it.
std::vector<int> plan_worker_nodes(const std::vector<int>& queue_node, int workers,
                                   const std::vector<std::vector<int>>& allowed_cpus_of_node);

}  // namespace numa
}  // namespace vit
