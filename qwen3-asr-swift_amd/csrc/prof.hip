#include "prof.h"
#include <atomic>
#include <map>
#include <memory>
#include <vector>

namespace {
struct Entry {
    int64_t launches = 0, timed = 0;
    double work = 0, work_timed = 0;
    std::vector<int> slots;
};
struct State {
    int stride = 1;
    std::map<std::string, Entry> entries;
    std::vector<hipEvent_t> ev;  // 2 per slot
    size_t used = 0;
    ~State() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
};
// Never destructed: a profile left running at process exit must not have its events destroyed from a static
// destructor, after the HIP runtime's own exit handlers (a model's destruction stops its profile).
std::mutex& g_mu = *new std::mutex();
std::map<hipStream_t, std::shared_ptr<State>>& g_by_stream = *new std::map<hipStream_t, std::shared_ptr<State>>();
std::atomic<int> g_running{0};   // profiles in progress: the launchers' fast path when none is

std::shared_ptr<State> find(hipStream_t s) {
    const auto it = g_by_stream.find(s);
    return it == g_by_stream.end() ? nullptr : it->second;
}
}  // namespace

int device_cus() {
    static PerDeviceOnce once;
    static int cus[PerDeviceOnce::MAX_DEVICES];
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    once.run([&] { HIP_CHECK(hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev)); });
    return cus[dev];
}

bool prof_active(hipStream_t s) {
    if (g_running.load(std::memory_order_relaxed) == 0) return false;
    std::lock_guard<std::mutex> lk(g_mu);
    return find(s) != nullptr;
}

void prof_start(hipStream_t s0, int stride) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (const auto old = find(s0)) {   // restarted: drop every stream of the old profile
        for (auto it = g_by_stream.begin(); it != g_by_stream.end();)
            it = it->second == old ? g_by_stream.erase(it) : std::next(it);
    } else {
        g_running.fetch_add(1);
    }
    auto st = std::make_shared<State>();
    st->stride = stride < 1 ? 1 : stride;
    g_by_stream[s0] = st;
}

void prof_attach(hipStream_t s, hipStream_t extra) {
    std::lock_guard<std::mutex> lk(g_mu);
    const auto st = find(s);
    if (st && extra) g_by_stream[extra] = st;
}

int prof_begin(const char* name, double work, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    const auto st = find(s);
    if (!st) return -1;
    Entry& e = st->entries[name];
    const bool timed = (e.launches % st->stride) == 0;
    e.launches += 1;
    e.work += work;
    if (!timed) return -1;
    if (st->used + 2 > st->ev.size()) {
        const size_t old = st->ev.size();
        st->ev.resize(old + 1024);
        for (size_t i = old; i < st->ev.size(); ++i) HIP_CHECK(hipEventCreate(&st->ev[i]));
    }
    const int slot = (int)st->used;
    st->used += 2;
    e.timed += 1;
    e.work_timed += work;
    e.slots.push_back(slot);
    HIP_CHECK(hipEventRecord(st->ev[slot], s));
    return slot;
}

void prof_end(int slot, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    const auto st = find(s);
    if (st) HIP_CHECK(hipEventRecord(st->ev[slot + 1], s));
}

bool prof_next_timed(const char* name, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    const auto st = find(s);
    if (!st) return false;
    const auto it = st->entries.find(name);
    return it == st->entries.end() || (it->second.launches % st->stride) == 0;
}

void prof_add_work(const char* name, double total, double timed, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    const auto st = find(s);
    if (!st) return;
    Entry& e = st->entries[name];
    e.work += total;
    e.work_timed += timed;
}

std::string prof_stop(hipStream_t s) {
    std::shared_ptr<State> st;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        st = find(s);
        if (!st) return "[]";
        for (auto it = g_by_stream.begin(); it != g_by_stream.end();)
            it = it->second == st ? g_by_stream.erase(it) : std::next(it);
        g_running.fetch_sub(1);
    }
    std::string out = "[";
    HIP_CHECK(hipDeviceSynchronize());
    bool first = true;
    for (auto& kv : st->entries) {
        double ms = 0;
        for (int slot : kv.second.slots) {
            float t = 0;
            HIP_CHECK(hipEventElapsedTime(&t, st->ev[slot], st->ev[slot + 1]));
            ms += t;
        }
        char buf[512];
        snprintf(buf, sizeof buf, "%s{\"name\": \"%s\", \"launches\": %lld, \"timed\": %lld, \"ms_timed\": %.6f, \"work\": %.6e, \"work_timed\": %.6e}",
                 first ? "" : ", ", kv.first.c_str(), (long long)kv.second.launches, (long long)kv.second.timed, ms, kv.second.work,
                 kv.second.work_timed);
        out += buf;
        first = false;
    }
    out += "]";
    return out;
}
