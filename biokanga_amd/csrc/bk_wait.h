// bk_wait.h - waits on the GPU that leave the CPU to others.
#pragma once
#include <hip/hip_runtime.h>

namespace bk {

// An event a host thread can sleep on: hipStreamSynchronize and events without this flag spin on the CPU until the GPU is done, and a
// pipeline has three threads that do little else - eight ranks of them are 24 spinning threads against a quota of 16 CPUs.
inline hipError_t make_wait_event(hipEvent_t *ev) { return hipEventCreateWithFlags(ev, hipEventDisableTiming | hipEventBlockingSync); }

// waits for everything enqueued on `s` so far, asleep
inline hipError_t wait_stream(hipStream_t s, hipEvent_t ev)
{
    hipError_t e = hipEventRecord(ev, s);
    if (e == hipSuccess) e = hipEventSynchronize(ev);
    return e;
}

}  // namespace bk
