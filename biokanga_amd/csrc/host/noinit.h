// noinit.h - std::vector whose resize() leaves trivially-constructible elements uninitialised.  The big host buffers (parsed
// bases, the read store, sort keys) are sized once and then filled by every host thread: a value-initialising resize() would
// first zero gigabytes - and take every page fault - on one thread.
#pragma once
#include <sys/mman.h>

#include <cstdlib>
#include <memory>
#include <new>
#include <utility>
#include <vector>

namespace bk {

// Buffers of 64 MB and more are aligned to 2 MB and marked for transparent huge pages (the GPU boxes run with THP "madvise"): a
// 5 GB read store then takes 2 500 page faults to touch instead of 1.3 M, and its pages go back to the kernel as quickly at exit -
// with 4 KB pages the process spent 60 ms per GB on that alone (profiles/NOTES.md, round 4).
constexpr size_t kHugeFrom = 64u << 20, kHugePage = 2u << 20;

template <typename T>
struct NoInitAlloc : std::allocator<T> {
    template <typename U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <typename U> NoInitAlloc(const NoInitAlloc<U> &) {}
    T *allocate(size_t n)
    {
        const size_t bytes = n * sizeof(T);
        if (bytes < kHugeFrom) return std::allocator<T>::allocate(n);
        void *p = nullptr;
        if (posix_memalign(&p, kHugePage, (bytes + kHugePage - 1) & ~(kHugePage - 1)) != 0 || !p) throw std::bad_alloc();
        (void)madvise(p, (bytes + kHugePage - 1) & ~(kHugePage - 1), MADV_HUGEPAGE);
        return static_cast<T *>(p);
    }
    void deallocate(T *p, size_t n)
    {
        if (n * sizeof(T) < kHugeFrom) std::allocator<T>::deallocate(p, n);
        else free(p);
    }
    template <typename U, typename... A>
    void construct(U *p, A &&...a)
    {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
        else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};

template <typename T> using RawVec = std::vector<T, NoInitAlloc<T>>;

}  // namespace bk
