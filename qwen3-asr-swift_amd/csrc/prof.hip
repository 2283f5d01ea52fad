#include "prof.h"
#include <map>
#include <vector>

namespace {
struct Entry {
    int64_t launches = 0, timed = 0;
    double work = 0, work_timed = 0;
    std::vector<int> slots;
};
struct State {
    bool on = false;
    int stride = 1;
    hipStream_t s = nullptr;
    std::map<std::string, Entry> entries;
    std::vector<hipEvent_t> ev;  // 2 per slot
    size_t used = 0;
} g;
}  // namespace

bool prof_active() { return g.on; }

void prof_start(hipStream_t s, int stride) {
    g.on = true;
    g.stride = stride < 1 ? 1 : stride;
    g.s = s;
    g.entries.clear();
    g.used = 0;
}

int prof_begin(const char* name, double work, hipStream_t s) {
    Entry& e = g.entries[name];
    const bool timed = (e.launches % g.stride) == 0;
    e.launches += 1;
    e.work += work;
    if (!timed) return -1;
    if (g.used + 2 > g.ev.size()) {
        const size_t old = g.ev.size();
        g.ev.resize(old + 1024);
        for (size_t i = old; i < g.ev.size(); ++i) HIP_CHECK(hipEventCreate(&g.ev[i]));
    }
    const int slot = (int)g.used;
    g.used += 2;
    e.timed += 1;
    e.work_timed += work;
    e.slots.push_back(slot);
    HIP_CHECK(hipEventRecord(g.ev[slot], s));
    return slot;
}

void prof_end(int slot, hipStream_t s) { HIP_CHECK(hipEventRecord(g.ev[slot + 1], s)); }

std::string prof_stop() {
    std::string out = "[";
    if (g.on) {
        HIP_CHECK(hipDeviceSynchronize());
        bool first = true;
        for (auto& kv : g.entries) {
            double ms = 0;
            for (int slot : kv.second.slots) {
                float t = 0;
                HIP_CHECK(hipEventElapsedTime(&t, g.ev[slot], g.ev[slot + 1]));
                ms += t;
            }
            char buf[512];
            snprintf(buf, sizeof buf, "%s{\"name\": \"%s\", \"launches\": %lld, \"timed\": %lld, \"ms_timed\": %.6f, \"work\": %.6e, \"work_timed\": %.6e}",
                     first ? "" : ", ", kv.first.c_str(), (long long)kv.second.launches, (long long)kv.second.timed, ms, kv.second.work,
                     kv.second.work_timed);
            out += buf;
            first = false;
        }
    }
    out += "]";
    g.on = false;
    g.entries.clear();
    g.used = 0;
    return out;
}
