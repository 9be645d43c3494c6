// bk_prim.h - the device-wide primitives this library uses (prefix scans, radix sort of pairs, selection), as rocPRIM provides them.
// Every call follows rocPRIM's temporary-storage protocol: with tmp == nullptr it only reports the bytes it needs in `bytes`.
#pragma once
#include <cstring>
#include <iterator>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

namespace bk {
namespace prim {

template <class In, class Out>
inline hipError_t exclusive_sum(void *tmp, size_t &bytes, In in, Out out, size_t n, hipStream_t s)
{
    using T = typename std::iterator_traits<Out>::value_type;
    return rocprim::exclusive_scan(tmp, bytes, in, out, T(0), n, rocprim::plus<T>(), s);
}

template <class In, class Out>
inline hipError_t inclusive_sum(void *tmp, size_t &bytes, In in, Out out, size_t n, hipStream_t s)
{
    using T = typename std::iterator_traits<Out>::value_type;
    return rocprim::inclusive_scan(tmp, bytes, in, out, n, rocprim::plus<T>(), s);
}

template <class In, class Out>
inline hipError_t inclusive_max(void *tmp, size_t &bytes, In in, Out out, size_t n, hipStream_t s)
{
    using T = typename std::iterator_traits<Out>::value_type;
    return rocprim::inclusive_scan(tmp, bytes, in, out, n, rocprim::maximum<T>(), s);
}

// bits [begin_bit, end_bit) of the keys decide
template <class K, class V>
inline hipError_t sort_pairs(void *tmp, size_t &bytes, const K *keys_in, K *keys_out, const V *vals_in, V *vals_out, size_t n, int begin_bit, int end_bit, hipStream_t s)
{
    return rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, (unsigned)begin_bit, (unsigned)end_bit, s);
}

// the items of `in` that `pred` takes, in order; their number goes to *n_selected (device memory)
template <class In, class Out, class Count, class Pred>
inline hipError_t select_if(void *tmp, size_t &bytes, In in, Out out, Count *n_selected, size_t n, Pred pred, hipStream_t s)
{
    return rocprim::select(tmp, bytes, in, out, n_selected, n, pred, s);
}

}  // namespace prim
}  // namespace bk
