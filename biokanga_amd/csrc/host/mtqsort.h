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
#include <cstdlib>
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

// The partition loop of the scheme from a given state (pl, ph, mid: the pivot element, which moves when it is swapped), to its end:
// afterwards [lo, ph] and [pl, hi] are the two sides still to sort.  The serial sort enters it with pl = lo, ph = hi; the parallel
// partition below enters it after its threads have done most of the swaps.
template <typename T, typename Cmp>
inline void hoare_run(T *lo, T *hi, T *&pl, T *&ph, T *mid, Cmp cmp)
{
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
}

// One partition step of a LARGE span by all threads, with exactly the serial outcome.  In the serial loop pl stops at the elements
// greater than the pivot, in ascending position (L_1 < L_2 < ..), ph at those not greater, in descending position (R_1 > R_2 > ..;
// the pivot's own position among them), and the k-th stops swap while L_k < R_k: which elements pair up depends only on the array as
// it stands after the median-of-three step.  So: blocks count their L's and R's, prefix sums give every k its two blocks, all k whose
// blocks lie apart (L's left of R's: certainly L_k < R_k) are swapped by the threads, each over its own range of k - its L's and R's
// lie in two position ranges no other thread touches - and the serial loop finishes from there (pl = L_k0, ph = R_k0): a few blocks.
template <typename T, typename Cmp>
void par_partition(T *lo, T *hi, Cmp cmp, int nthreads, T *&out_pl, T *&out_ph)
{
    const size_t cnt = (size_t)(hi - lo) + 1;
    T *mid = lo + cnt / 2;
    if (cmp(*lo, *mid) > 0) std::swap(*lo, *mid);
    if (cmp(*lo, *hi) > 0) std::swap(*lo, *hi);
    if (cmp(*mid, *hi) > 0) std::swap(*mid, *hi);
    const T pivot = *mid;                              // (a copy: the element itself may move under the threads)
    const size_t nb = (size_t)nthreads * 8;
    const size_t bs = (cnt + nb - 1) / nb;
    // L candidates live in [lo + 1, hi], R candidates in [lo, hi - 1]
    std::vector<uint64_t> cl(nb + 1, 0), cr(nb + 1, 0);          // cl[b + 1]: L's in block b; then prefix sums
    auto run_threads = [&](auto fn) {
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; t++) th.emplace_back(fn, t);
        fn(0);
        for (auto &x : th) x.join();
    };
    // (every thread works on copies of what its loops read: the originals live in this function's frame, which the calling thread - a
    // worker too - writes to; a shared cache line per element would cost more than the threads bring)
    run_threads([&](int t) {
        const T pv = pivot;
        T *const lo_ = lo;
        const size_t cnt_ = cnt, bs_ = bs, nb_ = nb, nt_ = (size_t)nthreads;
        Cmp cmp_ = cmp;
        for (size_t b = (size_t)t; b < nb_; b += nt_) {
            const size_t i0 = b * bs_, i1 = std::min(cnt_, i0 + bs_);
            uint64_t nl = 0, nr = 0;
            for (size_t i = i0; i < i1; i++) {
                const bool gt = cmp_(lo_[i], pv) > 0;
                nl += (gt && i >= 1) ? 1 : 0;
                nr += (!gt && i + 1 < cnt_) ? 1 : 0;
            }
            cl[b + 1] = nl; cr[b + 1] = nr;
        }
    });
    // CL[b] = L's in blocks < b (ascending rank base); CRr[b] = R's in blocks > b (descending rank base)
    std::vector<uint64_t> CL(nb + 1, 0), CRr(nb + 1, 0);
    for (size_t b = 0; b < nb; b++) CL[b + 1] = CL[b] + cl[b + 1];
    for (size_t b = nb; b-- > 0;) CRr[b] = (b + 1 < nb ? CRr[b + 1] : 0) + (b + 1 < nb ? cr[b + 2] : 0);
    const uint64_t totL = CL[nb], totR = CRr[0] + cr[1];
    // block of the k-th L (1-based): the b with CL[b] < k <= CL[b + 1]; block of the k-th R: the b with CRr[b] < k <= CRr[b] + cr[b + 1]
    auto block_of_L = [&](uint64_t k) { size_t a = 0, z = nb; while (z - a > 1) { const size_t m = (a + z) / 2; if (CL[m] < k) a = m; else z = m; } return a; };
    auto block_of_R = [&](uint64_t k) { size_t a = 0, z = nb; while (z - a > 1) { const size_t m = (a + z) / 2; if (CRr[m] + cr[m + 1] >= k) a = m; else z = m; } return a; };
    uint64_t k0 = 0;
    {
        uint64_t a = 0, z = std::min(totL, totR) + 1;          // invariant: k = a qualifies (or is 0), k = z does not
        while (z - a > 1) {
            const uint64_t m = (a + z) / 2;
            if (block_of_L(m) < block_of_R(m)) a = m; else z = m;
        }
        k0 = a;
    }
    T *pl = lo, *ph = hi;
    if (k0) {
        // thread t swaps the pairs k in (k0 * t / nt, k0 * (t + 1) / nt].  First every thread finds where its range starts (nothing has
        // moved yet: the ranks inside a block are those of the counting pass), then all swap.
        std::vector<T *> moved((size_t)nthreads, nullptr), last_l((size_t)nthreads, nullptr), last_r((size_t)nthreads, nullptr);
        std::vector<T *> first_l((size_t)nthreads, nullptr), first_r((size_t)nthreads, nullptr);
        T *const pivot_at = mid;
        run_threads([&](int t) {
            const uint64_t ka = k0 * (uint64_t)t / (uint64_t)nthreads, kb = k0 * (uint64_t)(t + 1) / (uint64_t)nthreads;
            if (kb <= ka) return;
            // position of L_{ka + 1}: inside its block, the (ka + 1 - CL[b])-th L from the block's start
            size_t b = block_of_L(ka + 1);
            uint64_t skip = ka + 1 - CL[b];
            T *l = lo + b * bs;
            for (;; ++l) { if (l > lo && cmp(*l, pivot) > 0 && --skip == 0) break; }
            // position of R_{ka + 1}: inside its block, the (ka + 1 - CRr[b])-th R from the block's end
            b = block_of_R(ka + 1);
            skip = ka + 1 - CRr[b];
            T *r = lo + std::min(cnt, b * bs + bs) - 1;
            for (;; --r) { if (r < hi && cmp(*r, pivot) <= 0 && --skip == 0) break; }
            first_l[(size_t)t] = l; first_r[(size_t)t] = r;
        });
        run_threads([&](int t) {
            const uint64_t ka = k0 * (uint64_t)t / (uint64_t)nthreads, kb = k0 * (uint64_t)(t + 1) / (uint64_t)nthreads;
            if (kb <= ka) return;
            T *l = first_l[(size_t)t], *r = first_r[(size_t)t], *mv = nullptr;
            const T pv = pivot;
            T *const pivot_at_ = pivot_at;
            Cmp cmp_ = cmp;
            for (uint64_t k = ka + 1;; k++) {
                if (r == pivot_at_) mv = l;                    // the pivot element goes to l
                std::swap(*l, *r);
                if (k == kb) break;
                do { ++l; } while (!(cmp_(*l, pv) > 0));
                do { --r; } while (!(cmp_(*r, pv) <= 0));
            }
            moved[(size_t)t] = mv; last_l[(size_t)t] = l; last_r[(size_t)t] = r;      // (once: neighbours in these arrays are other threads')
        });
        for (int t = nthreads - 1; t >= 0; t--) if (last_l[(size_t)t]) { pl = last_l[(size_t)t]; ph = last_r[(size_t)t]; break; }
        for (int t = 0; t < nthreads; t++) if (moved[(size_t)t]) mid = moved[(size_t)t];
    }
    hoare_run(lo, hi, pl, ph, mid, cmp);
    out_pl = pl;
    out_ph = ph;
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
            hoare_run(lo, hi, pl, ph, mid, cmp);
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
    // the first levels, where there are fewer spans than threads: the largest span is partitioned by all threads together
    const size_t par_min = (size_t)1 << 20;
    // (the first two levels only: on a host that gives the process a CPU quota the parallel form's second pass over a span costs more than
    // the idle threads it replaces are worth from the third level on - measured, profiles/NOTES.md round 4)
    while (queue.size() < 4) {
        size_t big = 0;
        for (size_t i = 1; i < queue.size(); i++) if (queue[i].r - queue[i].l > queue[big].r - queue[big].l) big = i;
        const Span sp = queue[big];
        if ((size_t)(sp.r - sp.l) + 1 < par_min) break;
        queue.erase(queue.begin() + (ptrdiff_t)big);
        T *pl = nullptr, *ph = nullptr;
        par_partition(sp.l, sp.r, cmp, nthreads, pl, ph);
        if (sp.l < ph) queue.push_back({sp.l, ph});
        if (pl < sp.r) queue.push_back({pl, sp.r});
        if (queue.empty()) return;
    }
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
