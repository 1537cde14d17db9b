// "Can these two streams run kernels at the same time?" - asked of the GPU, once per (handle, caller stream), when mdpt_forward picks its side stream.
//
// Why it has to be measured: the HIP runtime multiplexes the streams of a process onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default;
// a new stream joins the queue with the fewest users once all exist). Two streams on one queue execute in submission order, so a side stream that
// lands on the caller's queue silently turns the split batch into two half batches run back to back (measured: SwinV2-L batch 16 14.2 ms instead
// of 10.2 ms, ViT-L batch 32 43.7 instead of 41.1, for every third or fourth handle of a process - tools/probes/gpu_side_stream_queue.py).
// Nothing in the API tells which queue a stream got, and the other ways out measured worse: a stream of another PRIORITY class has its own queue
// pool but the two priorities do not overlap (ViT-L 42.9 ms = the unsplit time; SwinV2-L up to 17.3 ms), hipExtStreamCreateWithCUMask streams
// own a queue but are always blocking with respect to the NULL stream PyTorch computes on.
//
// The test: a waiter on the caller's stream polls a flag for at most ~150 us; a setter on the candidate stream raises it. Concurrent streams:
// the waiter sees the flag within microseconds. Same queue: the setter cannot start before the waiter ends, the waiter times out.

#include "mdpt_kernels.h"
#include "mdpt_prof.h"

namespace {

__global__ void queue_probe_wait_kernel(unsigned* flag, unsigned* seen, unsigned long long timeout_ticks) {
    const unsigned long long t0 = wall_clock64();  // constant-rate counter (100 MHz)
    unsigned got = 0;
    for (int it = 0; it < (1 << 22) && !got; ++it) {
        got = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (wall_clock64() - t0 > timeout_ticks) break;
        __builtin_amdgcn_s_sleep(8);
    }
    *seen = got;
}

__global__ void queue_probe_set_kernel(unsigned* flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

}  // namespace

// flag / seen: two device words (zeroed here, on `waiter_stream`); `ready` orders the candidate behind the zeroing. Asynchronous: the caller
// reads `seen` after synchronising with `waiter_stream`.
int mdpt_launch_queue_probe(unsigned* flag, unsigned* seen, hipStream_t waiter_stream, hipStream_t candidate, hipEvent_t ready) {
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(unsigned), waiter_stream);
    if (e == hipSuccess) e = hipMemsetAsync(seen, 0, sizeof(unsigned), waiter_stream);
    if (e == hipSuccess) e = hipEventRecord(ready, waiter_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(candidate, ready, 0);
    if (e != hipSuccess) return (int)e;
    {
        MdptProfScope prof("queue_probe_wait_kernel", 0.0, waiter_stream);
        hipLaunchKernelGGL(queue_probe_wait_kernel, dim3(1), dim3(1), 0, waiter_stream, flag, seen, 15000ull);
    }
    hipLaunchKernelGGL(queue_probe_set_kernel, dim3(1), dim3(1), 0, candidate, flag);
    return (int)hipGetLastError();
}
