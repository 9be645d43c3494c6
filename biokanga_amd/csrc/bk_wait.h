// bk_wait.h - how host threads wait for the GPU.
#pragma once
#include <atomic>
#include <stdlib.h>

#include <hip/hip_runtime.h>

#include "bk_cpus.h"
#include "bk_env.h"

namespace bk {

// hipStreamSynchronize spins on the CPU until the GPU is done: the shortest wait there is, and what a pipeline's three threads do when
// the process has CPUs to spare.  When it has not, the threads sleep on events made with hipEventBlockingSync instead, so that they do
// not burn the CPU time the threads with work need - under a cgroup quota every runnable thread beyond it gets the whole group
// throttled.  "Not to spare": about four threads per context of this process (a pipeline's three and the caller) times the ranks the
// launcher started on this node (LOCAL_WORLD_SIZE; they share the quota) exceed the CPUs the process may use (affinity mask, quota).
// Sleeping is not free: measured on the C2 host-in / host-out steps (round 5, profiles/NOTES.md) an event wait comes back about 10 ms
// later than the spin does - 470 against 579 M reads/s - whether the event has the blocking flag or not, and polling with 50 us naps is
// no better under the box's CPU quota.  Hence the choice by need.
inline std::atomic<int> &live_contexts() { static std::atomic<int> n{0}; return n; }

inline bool sleepy_waits()
{
    static const int cpus = effective_cpus();
    static const int ranks = env::local_world_size();
    const int ctxs = live_contexts().load(std::memory_order_relaxed);
    return 4 * (ctxs > 1 ? ctxs : 1) * ranks > cpus;
}

// (every wait event can be slept on; whether a wait does, is decided when it happens)
inline hipError_t make_wait_event(hipEvent_t *ev) { return hipEventCreateWithFlags(ev, hipEventDisableTiming | hipEventBlockingSync); }

inline hipError_t wait_event(hipEvent_t ev) { return hipEventSynchronize(ev); }

// waits for everything enqueued on `s` so far
inline hipError_t wait_stream(hipStream_t s, hipEvent_t ev)
{
    if (!sleepy_waits()) return hipStreamSynchronize(s);
    hipError_t e = hipEventRecord(ev, s);
    if (e == hipSuccess) e = wait_event(ev);
    return e;
}

}  // namespace bk
