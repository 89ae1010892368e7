// affinity.h — pin the host threads that feed a GPU to the CPUs of that GPU's NUMA node ("one worker thread per GPU pinned to the
// GPU's NUMA node"). The reference is a single-threaded Win32 program (main.cu:546-590) and has nothing of the kind; on an 8-GPU MI355X node the pinned
// staging buffers, the zlib work and the launch thread of a GPU should sit on the socket its PCIe root hangs off.
// The GPU's PCI address comes from the library (nct_device_pci_bus_id); Linux publishes the rest under /sys/bus/pci/devices/<addr>/{numa_node,local_cpulist}.
// NCT_SYSFS_ROOT replaces "/sys" (tests).
#pragma once
#include <sched.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace affinity {

// "0-3,8,10-11" -> {0,1,2,3,8,10,11}; tolerant of whitespace / a trailing newline; nonsense -> empty
inline std::vector<int> parse_cpulist(const std::string& s) {
    std::vector<int> out;
    size_t i = 0;
    auto num = [&](long& v) { if (i >= s.size() || s[i] < '0' || s[i] > '9') return false; v = 0; while (i < s.size() && s[i] >= '0' && s[i] <= '9') { v = v * 10 + (s[i] - '0'); if (v > 1 << 20) return false; ++i; } return true; };
    while (i < s.size()) {
        while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == ',')) ++i;
        if (i >= s.size()) break;
        long a, b;
        if (!num(a)) return {};
        b = a;
        if (i < s.size() && s[i] == '-') { ++i; if (!num(b) || b < a) return {}; }
        if (b - a > 4096) return {};
        for (long c = a; c <= b; ++c) out.push_back((int)c);
    }
    return out;
}

inline std::string sysfs_root() { const char* r = getenv("NCT_SYSFS_ROOT"); return r && *r ? r : "/sys"; }

inline bool read_small(const std::string& path, std::string& out) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[4096]; const size_t n = fread(buf, 1, sizeof buf - 1, f); fclose(f);
    buf[n] = 0; out = buf;
    return true;
}

struct GpuLocality { int numa_node = -1; std::vector<int> cpus; };

// numa_node may be -1 (single-node box / no ACPI proximity info): local_cpulist then names every CPU, which makes pinning a no-op — fine
inline GpuLocality gpu_locality(const std::string& pci_addr) {
    GpuLocality g; std::string s;
    const std::string dir = sysfs_root() + "/bus/pci/devices/" + pci_addr + "/";
    if (read_small(dir + "numa_node", s)) g.numa_node = atoi(s.c_str());
    if (read_small(dir + "local_cpulist", s)) g.cpus = parse_cpulist(s);
    if (g.cpus.empty() && g.numa_node >= 0 && read_small(sysfs_root() + "/devices/system/node/node" + std::to_string(g.numa_node) + "/cpulist", s)) g.cpus = parse_cpulist(s);
    return g;
}

// restrict the calling thread to `cpus` (intersected with what the process may use); false = left unpinned
inline bool pin_current_thread(const std::vector<int>& cpus) {
    if (cpus.empty()) return false;
    cpu_set_t allowed, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
    int n = 0;
    for (int c : cpus) if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) { CPU_SET(c, &want); ++n; }
    if (n == 0) return false;
    return sched_setaffinity(0, sizeof want, &want) == 0;
}

inline std::string cpus_to_string(const std::vector<int>& cpus) {       // compact "a-b,c" form for the log line
    std::string out;
    for (size_t i = 0; i < cpus.size();) {
        size_t j = i;
        while (j + 1 < cpus.size() && cpus[j + 1] == cpus[j] + 1) ++j;
        if (!out.empty()) out += ",";
        out += std::to_string(cpus[i]);
        if (j > i) out += "-" + std::to_string(cpus[j]);
        i = j + 1;
    }
    return out;
}

}  // namespace affinity
