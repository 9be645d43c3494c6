// mtqsort.h - replica of the ORDER produced by the reference's CMTqsort (libbiokanga/MTqsort.cpp:313-479).
//
// Why: the final alignment order comes from CAligner::SortReadHits(eRSMHitMatch) whose comparator
// (SortHitMatch, biokanga/Aligner.cpp:10069-10114) has no ReadID tie-break, so the order of equal
// elements is whatever the sort algorithm leaves - and byte-identical SAM needs exactly that order.
// The reference's threads run the same routine on whole sub-partitions (> 50 000 elements), so the
// result is a pure function of the input order; a single-threaded run of the same scheme gives it:
//   n < 25 000            glibc qsort(), a stable merge sort for these sizes -> std::stable_sort
//   otherwise             median-of-3 quicksort with this exact partition scheme; partitions of
//                         <= 16 elements finished by a max-selection sort
#pragma once
#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace bk {

template <typename T, typename Cmp>   // Cmp(a,b) -> <0, 0, >0
void selection_finish(T *left, T *right, Cmp cmp)
{
    while (right > left) {
        T *mx = left;
        for (T *p = left + 1; p <= right; ++p)
            if (cmp(*p, *mx) > 0) mx = p;
        std::swap(*mx, *right);
        --right;
    }
}

// One span and everything below it; sub-spans of more than `share_above` elements are handed to
// `share` (another thread's work) instead of the local stack.  Partitions never overlap, so the
// result does not depend on who sorts which span or in what order.
template <typename T, typename Cmp, typename Share>
void ref_order_span(T *lo, T *hi, Cmp cmp, size_t share_above, Share share)
{
    struct Span { T *l, *r; };
    std::vector<Span> stack;
    stack.reserve(128);
    auto defer = [&](T *l, T *r) {
        if (share_above && (size_t)(r - l) + 1 > share_above) share(l, r);
        else stack.push_back({l, r});
    };
    for (;;) {
        size_t cnt = (size_t)(hi - lo) + 1;
        bool descend = false;
        if (cnt <= 16)            // cMergeSortThres
            selection_finish(lo, hi, cmp);
        else {
            T *mid = lo + cnt / 2;
            if (cmp(*lo, *mid) > 0) std::swap(*lo, *mid);
            if (cmp(*lo, *hi) > 0) std::swap(*lo, *hi);
            if (cmp(*mid, *hi) > 0) std::swap(*mid, *hi);
            T *pl = lo, *ph = hi;
            for (;;) {
                if (mid > pl)
                    do { ++pl; } while (pl < mid && cmp(*pl, *mid) <= 0);
                if (mid <= pl)
                    do { ++pl; } while (pl <= hi && cmp(*pl, *mid) <= 0);
                do { --ph; } while (ph > mid && cmp(*ph, *mid) > 0);
                if (ph < pl) break;
                std::swap(*pl, *ph);
                if (mid == ph) mid = pl;
            }
            ++ph;
            if (mid < ph)
                do { --ph; } while (ph > mid && cmp(*ph, *mid) == 0);
            if (mid >= ph)
                do { --ph; } while (ph > lo && cmp(*ph, *mid) == 0);
            // larger side is deferred, smaller side is continued with
            if (ph - lo >= hi - pl) {
                if (lo < ph) defer(lo, ph);
                if (pl < hi) { lo = pl; descend = true; }
            } else {
                if (pl < hi) defer(pl, hi);
                if (lo < ph) { hi = ph; descend = true; }
            }
        }
        if (descend) continue;
        if (stack.empty()) break;
        lo = stack.back().l;
        hi = stack.back().r;
        stack.pop_back();
    }
}

template <typename T, typename Cmp>
void ref_order_sort(T *a, int64_t n, Cmp cmp, int nthreads = 1)
{
    if (n < 2) return;
    if (n < 25000) {            // cMinUseLibQsort
        std::stable_sort(a, a + n, [&](const T &x, const T &y) { return cmp(x, y) < 0; });
        return;
    }
    if (nthreads <= 1 || n < 200000) {
        ref_order_span(a, a + (n - 1), cmp, 0, [](T *, T *) {});
        return;
    }
    // the reference's own threads also take whole sub-partitions (MTqsort.cpp:313-479)
    struct Span { T *l, *r; };
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Span> queue;
    int active = 0;
    queue.push_back({a, a + (n - 1)});
    auto share = [&](T *l, T *r) {
        { std::lock_guard<std::mutex> g(mu); queue.push_back({l, r}); }
        cv.notify_one();
    };
    auto worker = [&]() {
        for (;;) {
            Span sp;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return !queue.empty() || active == 0; });
                if (queue.empty()) { cv.notify_all(); return; }
                sp = queue.back();
                queue.pop_back();
                active++;
            }
            ref_order_span(sp.l, sp.r, cmp, (size_t)65536, share);
            {
                std::lock_guard<std::mutex> g(mu);
                active--;
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; t++) th.emplace_back(worker);
    worker();
    for (auto &t : th) t.join();
}

}  // namespace bk
