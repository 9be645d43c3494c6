// bk_env.h - every environment variable the library and the command line read, in ONE place.  None of them changes a result: they switch
// diagnostics on, or let a test reach with a small input a code path that a production run only takes at scale.  (tools/README.md has the
// same table for users.)  Everything else is an option of the command line or a knob of bk_ctx_tune.
#pragma once
#include <cstdlib>

namespace bk {
namespace env {

inline bool set(const char *name) { return getenv(name) != nullptr; }
inline unsigned long long u64(const char *name, unsigned long long dflt)
{
    const char *e = getenv(name);
    if (!e) return dflt;
    const unsigned long long v = strtoull(e, nullptr, 10);
    return v ? v : dflt;
}

// ---- diagnostics
inline bool timing() { static const bool v = set("BK_TIMING"); return v; }                     // stage clocks on stderr; every hipMalloc / hipFree of more than 2 ms
inline bool debug_phases() { return set("BK_DEBUG"); }                                           // the phase loop reads its counts back and prints them per phase
inline int poison() { static const int v = set("BK_POISON") ? atoi(getenv("BK_POISON")) : -1; return v; }      // every device allocation filled with this byte first
inline bool inflate_debug() { static const bool v = set("BK_INFLATE_DEBUG"); return v; }       // pieces, sizes and stage times of a several-thread inflate
// ---- thresholds lowered by tests (0 / absent: the production value)
inline unsigned long long grow_after_reads() { return u64("BK_GROW_AFTER_READS", 0); }           // BK_CTX_GROW_IMAGE: the worker starts after this many reads and the next batch waits for it
inline unsigned long long table_slices() { return u64("BK_TABLE_SLICES", 0); }                   // the suffix array of a small .sfx uploaded in this many slices, tables made behind each
inline unsigned long long sa_wide_chunk() { return u64("BK_SA_WIDE_CHUNK", 0); }                 // chunk size of the 40-bit suffix sort
inline unsigned long long sam_device_min(unsigned long long dflt) { const char *e = getenv("BK_SAM_DEVICE_MIN"); return e ? strtoull(e, nullptr, 10) : dflt; }   // records from which plain SAM is formatted on the device
inline unsigned long long sam_early_min(unsigned long long dflt) { const char *e = getenv("BK_SAM_EARLY_MIN"); return e ? strtoull(e, nullptr, 10) : dflt; }     // input bytes from which the SAM file is started before the reads are parsed
inline bool sam_device_fail() { return set("BK_SAM_DEVICE_FAIL"); }                              // the device declines the SAM records after its head start
inline unsigned long long inflate_piece_min(unsigned long long dflt) { const char *e = getenv("BK_INFLATE_PIECE_MIN"); return e ? strtoull(e, nullptr, 10) : dflt; }   // compressed bytes per thread from which one gzip member is inflated by several threads
// ---- the launcher's, not ours: ranks started on this node (how host threads wait depends on how many share the process's CPUs, bk_wait.h)
inline int local_world_size() { static const int v = [] { const char *e = getenv("LOCAL_WORLD_SIZE"); const int r = e ? atoi(e) : 1; return r > 1 ? r : 1; }(); return v; }

}  // namespace env
}  // namespace bk
