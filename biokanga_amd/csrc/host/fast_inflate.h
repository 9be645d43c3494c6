// fast_inflate.h - a DEFLATE (RFC 1951) decoder for read files that are held whole in memory and inflated into one buffer.
//
// zlib's inflate() is written for streams of any size through small windows; on read files - text that is mostly literals - it
// emits one symbol per table look-up and refills its bit buffer a byte at a time: 180-200 MB of text per second and thread here.
// This decoder reads the compressed bytes eight at a time, resolves up to three literals per refill from an 11-bit table and copies
// matches eight bytes at a time, because the whole input and the whole output are in memory and the history never wraps.
//
// It is an accelerator, not an authority: it says "no" (-1) to anything it does not take - codes that are not complete, symbols
// that are not defined, distances beyond the output so far, output that does not fit - and the caller hands the file to zlib
// (gzread), which decides what the file is and what is said about it.  What it does accept is checked against the container's
// CRC-32 and length by the callers in fasta.cpp.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>

namespace bk {

// One raw deflate stream, from its first block to its final block's end.  `out` .. `out + out_cap` receives the text; matches may
// reach back to `hist` (<= out: the start of what earlier streams of the same file left in the same buffer).  Returns the number of
// bytes written, or -1.  *in_used = compressed bytes consumed (the stream's last byte included, even when only part of it was code).
// `progress`, when given, is advanced to the number of bytes that are final in `out` after every block (for readers that follow).
long inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, const uint8_t *hist, size_t *in_used,
                 std::atomic<size_t> *progress = nullptr);

// The same stream by up to `nthreads` threads, when it is text: threads start at guessed block starts and decode with the 32 KB in
// front of them unknown (see the .cpp); a stream of less than 16 MB, or one where a guess does not hold, is decoded by one thread.
// `out` is also the history's start.  *pieces_used = the number of threads whose work made the result (1: the plain way).
long inflate_raw_parallel(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *in_used, int nthreads, int *pieces_used = nullptr);

}  // namespace bk
