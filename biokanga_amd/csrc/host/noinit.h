// noinit.h - std::vector whose resize() leaves trivially-constructible elements uninitialised.  The big host buffers (parsed
// bases, the read store, sort keys) are sized once and then filled by every host thread: a value-initialising resize() would
// first zero gigabytes - and take every page fault - on one thread.
#pragma once
#include <memory>
#include <utility>
#include <vector>

namespace bk {

template <typename T>
struct NoInitAlloc : std::allocator<T> {
    template <typename U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <typename U> NoInitAlloc(const NoInitAlloc<U> &) {}
    template <typename U, typename... A>
    void construct(U *p, A &&...a)
    {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
        else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};

template <typename T> using RawVec = std::vector<T, NoInitAlloc<T>>;

}  // namespace bk
