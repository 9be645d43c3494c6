// bk_wait.h - waits on the GPU that leave the CPU to others.
#pragma once
#include <hip/hip_runtime.h>

namespace bk {

// An event a host thread can sleep on: hipStreamSynchronize and events without this flag spin on the CPU until the GPU is done, and a
// pipeline has three threads that do little else - eight ranks of them are 24 spinning threads against a quota of 16 CPUs.
// (Polling the event with naps of 50 us in between was measured against it in round 5 and lost: 22 ms per 50 M-read step in waits instead
// of 7 - a nap under the box's CPU quota lasts far longer than asked for.)
inline hipError_t make_wait_event(hipEvent_t *ev) { return hipEventCreateWithFlags(ev, hipEventDisableTiming | hipEventBlockingSync); }

inline hipError_t wait_event(hipEvent_t ev) { return hipEventSynchronize(ev); }

// waits for everything enqueued on `s` so far, asleep
inline hipError_t wait_stream(hipStream_t s, hipEvent_t ev)
{
    hipError_t e = hipEventRecord(ev, s);
    if (e == hipSuccess) e = wait_event(ev);
    return e;
}

}  // namespace bk
