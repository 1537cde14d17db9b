// HIP-event profiler behind mdpt_profile_enable / mdpt_profile_report (include/mdpt.h).
#include "mdpt_prof.h"

#include <stdio.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Rec { int slot; double flops; hipEvent_t a, b; };
struct Slot { std::string name; };
std::atomic<bool> g_on{false};
int g_gen = 0;  // bumped by every mdpt_profile_enable(): a scope opened before a reset must not close a record of the new table
std::vector<Rec> g_recs;
std::vector<Slot> g_slots;
std::map<std::string, int> g_slot_index;
std::vector<hipEvent_t> g_pool;
size_t g_pool_next = 0;
std::mutex g_mu;

hipEvent_t take_event() {
    if (g_pool_next == g_pool.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        g_pool.push_back(e);
    }
    return g_pool[g_pool_next++];
}
}  // namespace

bool mdpt_prof_on() { return g_on.load(std::memory_order_relaxed); }

int mdpt_prof_begin(const char* name, double flops, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_on.load(std::memory_order_relaxed)) return -1;
    auto it = g_slot_index.find(name);
    int slot;
    if (it == g_slot_index.end()) {
        slot = (int)g_slots.size();
        g_slots.push_back({name});
        g_slot_index[name] = slot;
    } else {
        slot = it->second;
    }
    Rec r;
    r.slot = slot; r.flops = flops; r.a = take_event(); r.b = take_event();
    hipEventRecord(r.a, stream);
    g_recs.push_back(r);
    return ((g_gen & 0x7F) << 24) | ((int)g_recs.size() - 1);  // (generation, index): see mdpt_prof_end
}

void mdpt_prof_end(int record, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (record < 0 || ((record >> 24) & 0x7F) != (g_gen & 0x7F)) return;  // profiling was reset while this scope was open
    const int idx = record & 0xFFFFFF;
    if (idx < (int)g_recs.size()) hipEventRecord(g_recs[idx].b, stream);
}

extern "C" int mdpt_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(g_mu);
    g_on.store(on != 0, std::memory_order_relaxed);
    ++g_gen;
    g_recs.clear();
    g_pool_next = 0;
    return 0;
}

// JSON: {"kernels":[{"name":..,"launches":n,"total_ms":t,"avg_us":a,"gflop":g,"tflops":x}, ...]} sorted by total time
extern "C" int mdpt_profile_report(char* buf, size_t cap) {
    if (!buf || cap < 64) return -1;
    std::lock_guard<std::mutex> lock(g_mu);
    struct Acc { int n = 0; double ms = 0, flops = 0; };
    std::vector<Acc> acc(g_slots.size());
    for (const Rec& r : g_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        acc[r.slot].n++;
        acc[r.slot].ms += ms;
        acc[r.slot].flops += r.flops;
    }
    std::vector<int> order;
    for (int i = 0; i < (int)acc.size(); ++i)
        if (acc[i].n) order.push_back(i);
    for (size_t i = 0; i < order.size(); ++i)
        for (size_t j = i + 1; j < order.size(); ++j)
            if (acc[order[j]].ms > acc[order[i]].ms) { int t = order[i]; order[i] = order[j]; order[j] = t; }
    std::string out = "{\"kernels\":[";
    char line[512];
    for (size_t i = 0; i < order.size(); ++i) {
        const Acc& a = acc[order[i]];
        snprintf(line, sizeof(line), "%s{\"name\":\"%s\",\"launches\":%d,\"total_ms\":%.4f,\"avg_us\":%.3f,\"gflop\":%.3f,\"tflops\":%.3f}",
                 i ? "," : "", g_slots[order[i]].name.c_str(), a.n, a.ms, a.ms * 1e3 / a.n, a.flops * 1e-9,
                 a.ms > 0 ? a.flops / (a.ms * 1e-3) * 1e-12 : 0.0);
        out += line;
    }
    out += "]}";
    if (out.size() + 1 > cap) return -2;
    memcpy(buf, out.c_str(), out.size() + 1);
    return 0;
}
