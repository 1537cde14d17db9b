// Optional per-launch HIP-event profiler (off by default; bench.py turns it on for the timed region).
// Events are recorded on the SAME stream the kernel is launched on, so durations are device-side and need no host sync.
#pragma once
#include <hip/hip_runtime.h>

int mdpt_prof_begin(const char* name, double flops, hipStream_t stream);  // returns this scope's record handle = (profiling generation, index); -1: profiling off
void mdpt_prof_end(int record, hipStream_t stream);
bool mdpt_prof_on();

// Each scope closes ITS OWN record (nested scopes and launches from several host threads - the two half-batch streams of one forward are
// driven by one thread, but two handles may run on two threads - cannot close each other's); the record table is guarded by a mutex.
struct MdptProfScope {
    hipStream_t s;
    int rec;
    MdptProfScope(const char* name, double flops, hipStream_t stream) : s(stream), rec(mdpt_prof_on() ? mdpt_prof_begin(name, flops, stream) : -1) {}
    ~MdptProfScope() {
        if (rec >= 0) mdpt_prof_end(rec, s);
    }
};
