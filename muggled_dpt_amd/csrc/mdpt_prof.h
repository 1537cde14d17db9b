// Optional per-launch HIP-event profiler (off by default; bench.py turns it on for the timed region).
// Events are recorded on the SAME stream the kernel is launched on, so durations are device-side and need no host sync.
#pragma once
#include <hip/hip_runtime.h>

void mdpt_prof_begin(const char* name, double flops, hipStream_t stream);
void mdpt_prof_end(hipStream_t stream);
bool mdpt_prof_on();

struct MdptProfScope {
    hipStream_t s;
    bool on;
    MdptProfScope(const char* name, double flops, hipStream_t stream) : s(stream), on(mdpt_prof_on()) {
        if (on) mdpt_prof_begin(name, flops, s);
    }
    ~MdptProfScope() {
        if (on) mdpt_prof_end(s);
    }
};
