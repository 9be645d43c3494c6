// bk_dev_prof.h - in-kernel section timers of the experiments (off unless the library is built with -DBK_PROF=1 / 2).
#pragma once
#include <hip/hip_runtime.h>

namespace bk {

// -DBK_PROF=1 (k_flat) / 2 (k_search_a_ilp): where a block's time goes - thread 0 adds the cycles between its section marks to
// g_prof (summed over the blocks, read with bk_debug_prof(); `BK_DIAG=1 python bench.py` prints them)
#ifdef BK_PROF
static __device__ unsigned long long g_prof[64 * 16];      // (one per kernel file: BK_PROF picks the kernel)
#endif
#ifdef BK_PROF
#define PROF_AT(k) do { if (threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const long long now_ = clock64(); pt[k] = now_ - last_; last_ = now_; } } while (0)
#define PROF_BEGIN long long pt[10] = {0,0,0,0,0,0,0,0,0,0}, last_ = clock64()
#define PROF_END do { if (threadIdx.x == 0) { for (int k_ = 0; k_ < 10; k_++) atomicAdd(&g_prof[(blockIdx.x & 63) * 16 + k_], (unsigned long long)pt[k_]); atomicAdd(&g_prof[(blockIdx.x & 63) * 16 + 10], 1ULL); } } while (0)
#else
#define PROF_AT(k) do { } while (0)
#define PROF_BEGIN do { } while (0)
#define PROF_END do { } while (0)
#endif
#if defined(BK_PROF) && BK_PROF == 1
#define PROF(k) PROF_AT(k)
#else
#define PROF(k) do { } while (0)
#endif
#if defined(BK_PROF) && BK_PROF == 2
#define PROFS(k) PROF_AT(k)
#else
#define PROFS(k) do { } while (0)
#endif
#ifdef BK_PROF
// sums the per-stripe section counters of THIS kernel file's g_prof into out16
static inline int prof_read(unsigned long long *out16)
{
    unsigned long long h[64 * 16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prof), sizeof(h)) != hipSuccess) return 1;
    for (int k = 0; k < 16; k++) { out16[k] = 0; for (int s0 = 0; s0 < 64; s0++) out16[k] += h[s0 * 16 + k]; }
    return 0;
}
#endif

}  // namespace bk
