// bk_dev_sets.h - the reference's set of seen target starts (SfxArrayV2.cpp:5932-5950) as the wave-per-read kernels keep it: epoch-tagged
// open addressing in HBM (k_heavy, k_wave's HASH form), a per-wave set in LDS in front of it (k_wave).
#pragma once
#include "bk_dev_util.h"

namespace bk {

__device__ __forceinline__ uint32_t hash_key(uint32_t key, uint32_t mask)
{
    return (key * 2654435761u) & mask;      // table size is a power of two
}

__device__ __forceinline__ bool htab_contains(unsigned long long *tab, uint32_t mask, uint32_t epoch, uint32_t key)
{
    unsigned long long mine = ((unsigned long long)epoch << 32) | key;
    uint32_t h = hash_key(key, mask);
    for (;;) {
        unsigned long long v = __hip_atomic_load(&tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(v >> 32) != epoch) return false;
        if (v == mine) return true;
        h = (h + 1) & mask;
    }
}

__device__ __forceinline__ void htab_insert(unsigned long long *tab, uint32_t mask, uint32_t epoch, uint32_t key)
{
    unsigned long long mine = ((unsigned long long)epoch << 32) | key;
    uint32_t h = hash_key(key, mask);
    for (;;) {
        unsigned long long v = __hip_atomic_load(&tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(v >> 32) != epoch) {
            unsigned long long old = atomicCAS(&tab[h], v, mine);
            if (old == v) return;
            continue;                       // somebody else took the slot: look at it again
        }
        if (v == mine) return;
        h = (h + 1) & mask;
    }
}

// The wave kernel's form for a strand pass that already lives in the HBM table: look-up and insert are one walk of the probe sequence
// (two dependent random lines per candidate round become one).  A key is inserted before the kernel knows whether its candidate is
// processed at all - the MaxIter cut and the copy check come later in the round - so a lane whose candidate is cut off takes its key
// back: the slot keeps this pass's epoch with kTombBit set, occupied for the probing of others, matching nothing.  (Epochs stay below
// kTombBit; entries of other passes - k_heavy's included, which shares the tables - read as stale to everybody, as before.)
constexpr uint32_t kTombBit = 0x80000000u;

__device__ __forceinline__ bool htab_find_or_insert(unsigned long long *tab, uint32_t mask, uint32_t epoch, uint32_t key, uint32_t &slot)
{
    const unsigned long long mine = ((unsigned long long)epoch << 32) | key;
    uint32_t h = hash_key(key, mask);
    for (;;) {
        const unsigned long long v = __hip_atomic_load(&tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t e = (uint32_t)(v >> 32);
        if (e == epoch) {
            if (v == mine) return true;             // seen
        } else if (e != (epoch | kTombBit)) {       // stale or empty: take it
            if (atomicCAS(&tab[h], v, mine) == v) { slot = h; return false; }
            continue;                               // somebody else took the slot: look at it again
        }
        h = (h + 1) & mask;
    }
}

__device__ __forceinline__ void htab_retract(unsigned long long *tab, uint32_t slot, uint32_t epoch)
{
    __hip_atomic_store(&tab[slot], (unsigned long long)(epoch | kTombBit) << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 5-byte indexes only: two candidates of one 64-candidate round whose target starts lie a multiple of 2^32 bases apart carry the
// same truncated key (SfxArrayV2.cpp:5932).  The reference, walking them one after the other, takes the later one for seen; the
// hash set is only consulted for what EARLIER rounds left in it, so the round is checked against itself here.
__device__ __forceinline__ bool same_key_earlier_in_round(bool cand, uint32_t key, int lane)
{
    // Every lane narrows the set of candidates that agree with its key, eight key bits at a time (a ballot per bit: the lanes whose bit
    // is set; a lane keeps the side it is on).  After 16 bits a round of 64 random keys has 0.03 pairs left on average, so almost every
    // round ends at the second look; a set that survives all 32 bits holds equal keys, and the later lanes of it are the duplicates.
    // (The first form of this - six bits, then a loop over every lane that still had company, 40 of 64 on average - was a quarter of the
    // wave kernel's time on 5-byte indexes.)
    uint64_t peers = __ballot(cand);
    if ((peers & (peers - 1)) == 0) return false;               // fewer than two candidates
    bool more = true;
#pragma unroll
    for (int chunk = 0; chunk < 4 && more; chunk++) {
#pragma unroll
        for (int bit = 8 * chunk; bit < 8 * chunk + 8; bit++) {
            const bool one = (key >> bit) & 1;
            const uint64_t bm = __ballot(cand && one);
            peers &= one ? bm : ~bm;
        }
        more = __ballot(cand && (peers & (peers - 1)) != 0) != 0;  // somebody still has company
    }
    if (!more) return false;
    return cand && (peers & ((1ULL << lane) - 1ULL)) != 0;
}

// the wave kernel's LDS set of seen keys (HASH form)
#ifndef BK_LDS_SET
#define BK_LDS_SET 2048
#endif
constexpr uint32_t kLdsSet = BK_LDS_SET, kLdsSetFill = kLdsSet * 3 / 4, kLdsEmpty = 0xFFFFFFFFu;

// Open addressing by buckets of four keys (one 16-byte LDS read per probe): nearly every look-up is for a key that is NOT there (2 % of a
// pass's candidates are repeats of earlier ones), and such a look-up walks to the first free slot - 8.5 slots on average at three
// quarters full with a slot per probe, the slowest of a round's 64 lanes some dozens, a quarter of the 5-byte forms' time
// (profiles/r06_zb_kwave_C5_round_sections.txt).  With four slots per probe it is one or two.  A key lives in the first bucket of its
// probe sequence that had a free slot when it came; keys never leave (the set is cleared per strand pass), so a look-up ends at the first
// bucket with a free slot.  The bucket comes from the product's HIGH bits.
#ifndef BK_LDS_SET_BUCKETS
#define BK_LDS_SET_BUCKETS 1
#endif
constexpr uint32_t kLdsBuckets = kLdsSet / 4;
typedef uint32_t lset_u32x4 __attribute__((ext_vector_type(4)));
static_assert((kLdsSet & (kLdsSet - 1)) == 0 && kLdsSet >= 64, "the set's size is a power of two");

__device__ __forceinline__ uint32_t lset_bucket(uint32_t key)
{
    return (uint32_t)(((uint64_t)(key * 2654435761u) * kLdsBuckets) >> 32);
}

__device__ __forceinline__ bool lset_contains(const uint32_t *set, uint32_t key)
{
#if BK_LDS_SET_BUCKETS
    uint32_t b = lset_bucket(key);
    for (;;) {
        const lset_u32x4 v = *reinterpret_cast<const volatile lset_u32x4 *>(set + 4 * b);
        if (v.x == key || v.y == key || v.z == key || v.w == key) return true;
        if (v.x == kLdsEmpty || v.y == kLdsEmpty || v.z == kLdsEmpty || v.w == kLdsEmpty) return false;
        b = (b + 1) & (kLdsBuckets - 1);
    }
#else
    uint32_t h = hash_key(key, kLdsSet - 1);
    for (;;) {
        const uint32_t v = __hip_atomic_load(&set[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (v == kLdsEmpty) return false;
        if (v == key) return true;
        h = (h + 1) & (kLdsSet - 1);
    }
#endif
}

__device__ __forceinline__ void lset_insert(uint32_t *set, uint32_t key)
{
#if BK_LDS_SET_BUCKETS
    uint32_t b = lset_bucket(key);
    for (;;) {
        const lset_u32x4 v = *reinterpret_cast<const volatile lset_u32x4 *>(set + 4 * b);
        if (v.x == key || v.y == key || v.z == key || v.w == key) return;
        // (the first free slot of the bucket; another lane of the round may take it first: the bucket is looked at again)
        const int i = v.x == kLdsEmpty ? 0 : (v.y == kLdsEmpty ? 1 : (v.z == kLdsEmpty ? 2 : (v.w == kLdsEmpty ? 3 : 4)));
        if (i == 4) { b = (b + 1) & (kLdsBuckets - 1); continue; }
        const uint32_t old = atomicCAS(&set[4 * b + i], kLdsEmpty, key);
        if (old == kLdsEmpty || old == key) return;
    }
#else
    uint32_t h = hash_key(key, kLdsSet - 1);
    for (;;) {
        const uint32_t old = atomicCAS(&set[h], kLdsEmpty, key);
        if (old == kLdsEmpty || old == key) return;
        h = (h + 1) & (kLdsSet - 1);
    }
#endif
}

}  // namespace bk
