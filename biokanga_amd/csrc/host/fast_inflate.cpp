// fast_inflate.cpp - see fast_inflate.h.  The format is RFC 1951's; the tables and loops are this file's own.
#include "fast_inflate.h"
#include "../bk_env.h"

#include <sys/mman.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <thread>
#include <vector>

namespace bk {
namespace {

// A table entry:  bits 0-4   code bits this entry consumes (for a pointer: the root's bits); bit 5 stays clear, so that the low six
//                            bits are the shift count as the CPU takes it
//                 bit 6      literal/length table: a literal, in bits 8-15
//                 otherwise: bits 8-12  extra bits that follow a length / distance code (for a pointer: the subtable's index bits)
//                            bits 13-15 kind
//                            bits 16-31 the base length or distance, the subtable's first index, a code-length symbol
// (Entries that hold two or three short literals were tried: on read files the inner loop gained 7-10 %, and making the entries for
// every block of 32 KB gave it back.)
constexpr uint32_t kLit = 0u << 13, kBase = 1u << 13, kSub = 2u << 13, kEnd = 3u << 13, kBad = 4u << 13, kKind = 7u << 13;
constexpr uint32_t kIsLit = 1u << 6;
constexpr int kLitRoot = 11, kDistRoot = 8, kClRoot = 7;
constexpr size_t kLitCap = (1u << kLitRoot) + 288 * 16, kDistCap = (1u << kDistRoot) + 32 * 128;

struct Tables {
    uint32_t lit[kLitCap];
    uint32_t dist[kDistCap];
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t lit_symbol(int s)
{
    if (s < 256) return kIsLit | ((uint32_t)s << 8);
    if (s == 256) return kEnd;
    if (s < 286) return kBase | ((uint32_t)kLenExtra[s - 257] << 8) | ((uint32_t)kLenBase[s - 257] << 16);
    return kBad;                                        // 286, 287: in the fixed code, never in data
}
inline uint32_t dist_symbol(int s) { return s < 30 ? kBase | ((uint32_t)kDistExtra[s] << 8) | ((uint32_t)kDistBase[s] << 16) : kBad; }
inline uint32_t cl_symbol(int s) { return kLit | ((uint32_t)s << 16); }

// Canonical code -> look-up table with a root of `root` bits and one subtable per root prefix that longer codes share.  A code that
// is over-subscribed, or incomplete in any way but the one distance code of one bit that encoders emit, is not taken.
template <class Sym>
bool build_table(const uint8_t *lens, int n, int root, uint32_t *tab, size_t cap, Sym sym, bool dist_rules)
{
    int count[16] = {0};
    for (int i = 0; i < n; i++) count[lens[i]]++;
    const uint32_t nroot = 1u << root;
    for (uint32_t i = 0; i < nroot; i++) tab[i] = kBad;
    if (count[0] == n) return dist_rules;               // no distance codes at all: a block of literals only
    long left = 1;
    for (int l = 1; l <= 15; l++) {
        left = (left << 1) - count[l];
        if (left < 0) return false;
    }
    if (left > 0 && !(dist_rules && n - count[0] == 1 && count[1] == 1)) return false;
    uint32_t next[16];
    uint32_t code = 0;
    count[0] = 0;
    for (int l = 1; l <= 15; l++) {
        code = (code + (uint32_t)count[l - 1]) << 1;
        next[l] = code;
    }
    uint16_t rev[320];
    uint8_t longest[1u << kLitRoot];
    memset(longest, 0, nroot);
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (!l) continue;
        uint32_t c = next[l]++, r = 0;
        for (int k = 0; k < l; k++) { r = (r << 1) | (c & 1); c >>= 1; }
        rev[s] = (uint16_t)r;
        if (l <= root)
            for (uint32_t i = r; i < nroot; i += 1u << l) tab[i] = sym(s) | (uint32_t)l;
        else if (l > longest[r & (nroot - 1)])
            longest[r & (nroot - 1)] = (uint8_t)l;
    }
    size_t free_at = nroot;
    for (uint32_t p = 0; p < nroot; p++)
        if (longest[p]) {
            const uint32_t bits = (uint32_t)longest[p] - (uint32_t)root;
            if (free_at + (1u << bits) > cap) return false;
            tab[p] = kSub | (bits << 8) | (uint32_t)root | ((uint32_t)free_at << 16);
            for (uint32_t i = 0; i < (1u << bits); i++) tab[free_at + i] = kBad;
            free_at += 1u << bits;
        }
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (l <= root) continue;
        const uint32_t e = tab[rev[s] & (nroot - 1)], off = e >> 16, bits = (e >> 8) & 31;
        for (uint32_t i = (uint32_t)rev[s] >> root; i < (1u << bits); i += 1u << (l - root)) tab[off + i] = sym(s) | (uint32_t)(l - root);
    }
    return true;
}

// the reader of everything outside the inner loop: a byte at a time, never past the end, no bits in `bb` above `bc`
struct Bits {
    const uint8_t *in, *end;
    uint64_t bb;
    uint32_t bc;
    void fill() { while (bc <= 56 && in < end) { bb |= (uint64_t)*in++ << bc; bc += 8; } }
    bool need(uint32_t n) { fill(); return bc >= n; }
    uint32_t take(uint32_t n) { const uint32_t v = (uint32_t)(bb & ((1ull << n) - 1)); bb >>= n; bc -= n; return v; }
};

const Tables *fixed_tables()
{
    static const Tables *t = []() {
        Tables *x = new Tables;
        uint8_t lens[288 + 32];
        for (int s = 0; s < 288; s++) lens[s] = s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8));
        for (int s = 0; s < 32; s++) lens[288 + s] = 5;
        build_table(lens, 288, kLitRoot, x->lit, kLitCap, lit_symbol, false);
        build_table(lens + 288, 32, kDistRoot, x->dist, kDistCap, dist_symbol, true);
        return x;
    }();
    return t;
}

bool read_dynamic(Bits &b, Tables &t)
{
    if (!b.need(14)) return false;
    const uint32_t hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    for (uint32_t i = 0; i < hclen; i++) {
        if (!b.need(3)) return false;
        cl[order[i]] = (uint8_t)b.take(3);
    }
    uint32_t cltab[1u << kClRoot];
    if (!build_table(cl, 19, kClRoot, cltab, 1u << kClRoot, cl_symbol, false)) return false;
    uint8_t lens[320];
    const uint32_t n = hlit + hdist;
    for (uint32_t i = 0; i < n;) {
        b.fill();
        const uint32_t e = cltab[b.bb & ((1u << kClRoot) - 1)];
        if ((e & kKind) != kLit || (e & 31) > b.bc) return false;
        b.take(e & 31);
        const uint32_t s = e >> 16;
        if (s < 16) { lens[i++] = (uint8_t)s; continue; }
        uint32_t rep;
        uint8_t what = 0;
        if (s == 16) {
            if (i == 0 || !b.need(2)) return false;
            rep = 3 + b.take(2);
            what = lens[i - 1];
        } else if (s == 17) {
            if (!b.need(3)) return false;
            rep = 3 + b.take(3);
        } else {
            if (!b.need(7)) return false;
            rep = 11 + b.take(7);
        }
        if (i + rep > n) return false;
        memset(lens + i, what, rep);
        i += rep;
    }
    if (lens[256] == 0) return false;                   // a block that cannot end
    return build_table(lens, (int)hlit, kLitRoot, t.lit, kLitCap, lit_symbol, false) &&
           build_table(lens + hlit, (int)hdist, kDistRoot, t.dist, kDistCap, dist_symbol, true);
}

inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }      // (x86-64: little-endian, unaligned is fine)

// The symbols of one block, up to and including its end-of-block.  The first loop runs while eight input bytes and the longest
// match plus the copies' overshoot fit; the second, careful one takes the block's - or the stream's - last stretch.
bool decode_block(Bits &b, const Tables &t, uint8_t *&out, uint8_t *const oend, const uint8_t *hist)
{
    const uint8_t *in = b.in, *const iend = b.end;
    uint64_t bb = b.bb;
    uint32_t bc = b.bc;
    uint8_t *o = out;
    bool ended = false;
    constexpr uint32_t lmask = (1u << kLitRoot) - 1, dmask = (1u << kDistRoot) - 1;
    while (iend - in >= 8 && oend - o >= 68 + 258 + 16) {
        bb |= load64(in) << bc;                         // at least 56 bits from here on: a length with its extra bits and a distance
        in += (63 - bc) >> 3;                           // with its own are 48 at most
        bc |= 56;
        uint32_t e = t.lit[bb & lmask];
        if (!(e & kIsLit) && (e & kKind) == kSub) {
            bb >>= kLitRoot; bc -= kLitRoot;
            e = t.lit[(e >> 16) + (uint32_t)(bb & ((1u << ((e >> 8) & 31)) - 1))];
        }
        bb >>= (e & 63); bc -= (e & 31);
        if (e & kIsLit) {
            // more literals out of the bits at hand (read files are literals nine symbols in ten, two to six bits each): a root entry
            // whose code is no longer than the bits left was found by real bits only, whatever lies above them
            do {
                *o++ = (uint8_t)(e >> 8);
                e = t.lit[bb & lmask];
                if (!(e & kIsLit) || (e & 31) > bc) break;
                bb >>= (e & 63); bc -= (e & 31);
            } while (true);
            continue;
        }
        if ((e & kKind) != kBase) {
            if ((e & kKind) != kEnd) return false;
            ended = true;
            break;
        }
        uint32_t x = (e >> 8) & 31;
        const uint32_t len = (e >> 16) + (uint32_t)(bb & ((1u << x) - 1));
        bb >>= x; bc -= x;
        uint32_t d = t.dist[bb & dmask];
        if ((d & kKind) == kSub) {
            bb >>= kDistRoot; bc -= kDistRoot;
            d = t.dist[(d >> 16) + (uint32_t)(bb & ((1u << ((d >> 8) & 31)) - 1))];
        }
        bb >>= (d & 63); bc -= (d & 31);
        if ((d & kKind) != kBase) return false;
        x = (d >> 8) & 31;
        const uint32_t dist = (d >> 16) + (uint32_t)(bb & ((1u << x) - 1));
        bb >>= x; bc -= x;
        if ((size_t)dist > (size_t)(o - hist)) return false;
        const uint8_t *s = o - dist;
        uint8_t *w = o;
        o += len;
        if (dist >= 8) {
            do { memcpy(w, s, 8); w += 8; s += 8; } while (w < o);
        } else if (dist == 1) {
            const uint64_t v = 0x0101010101010101ull * (uint64_t)*s;
            do { memcpy(w, &v, 8); w += 8; } while (w < o);
        } else {
            do { *w++ = *s++; } while (w < o);
        }
    }
    // whole bytes that were read ahead go back; what stays are the bits of the byte in front of `in`
    in -= bc >> 3;
    bc &= 7;
    bb &= (1ull << bc) - 1;
    Bits c{in, iend, bb, bc};
    while (!ended) {
        c.fill();
        uint32_t e = t.lit[c.bb & lmask];
        if (!(e & kIsLit) && (e & kKind) == kSub) {
            if (c.bc < (uint32_t)kLitRoot) return false;
            c.take(kLitRoot);
            e = t.lit[(e >> 16) + (uint32_t)(c.bb & ((1u << ((e >> 8) & 31)) - 1))];
        }
        if ((e & 31) > c.bc) return false;
        c.take(e & 31);
        if (e & kIsLit) {
            if (o >= oend) return false;
            *o++ = (uint8_t)(e >> 8);
            continue;
        }
        if ((e & kKind) == kEnd) break;
        if ((e & kKind) != kBase) return false;
        uint32_t x = (e >> 8) & 31;
        if (x > c.bc) return false;
        const uint32_t len = (e >> 16) + c.take(x);
        uint32_t d = t.dist[c.bb & dmask];
        if ((d & kKind) == kSub) {
            if (c.bc < (uint32_t)kDistRoot) return false;
            c.take(kDistRoot);
            d = t.dist[(d >> 16) + (uint32_t)(c.bb & ((1u << ((d >> 8) & 31)) - 1))];
        }
        if ((d & kKind) != kBase || (d & 31) > c.bc) return false;
        c.take(d & 31);
        x = (d >> 8) & 31;
        if (x > c.bc) return false;
        const uint32_t dist = (d >> 16) + c.take(x);
        if ((size_t)dist > (size_t)(o - hist) || (size_t)len > (size_t)(oend - o)) return false;
        const uint8_t *s = o - dist;
        for (uint32_t i = 0; i < len; i++) o[i] = s[i];
        o += len;
    }
    b = c;
    out = o;
    return true;
}

}  // namespace

namespace {

constexpr uint64_t kNoStop = ~0ull;

inline uint64_t bit_pos(const Bits &b, const uint8_t *in) { return (uint64_t)(b.in - in) * 8 - b.bc; }

inline bool bits_at(Bits &b, const uint8_t *in, size_t in_len, uint64_t bit)
{
    if (bit / 8 > in_len) return false;
    b = Bits{in + bit / 8, in + in_len, 0, 0};
    if (bit & 7) {
        if (!b.need((uint32_t)(bit & 7))) return false;
        b.take((uint32_t)(bit & 7));
    }
    return true;
}

// a stored block's length words, from the byte boundary: its payload's length, or -1
inline long stored_header(Bits &b)
{
    b.in -= b.bc >> 3;                                  // the rest of this byte is padding; whole bytes go back
    b.bb = 0;
    b.bc = 0;
    if (b.end - b.in < 4) return -1;
    const uint32_t len = (uint32_t)b.in[0] | ((uint32_t)b.in[1] << 8), nlen = (uint32_t)b.in[2] | ((uint32_t)b.in[3] << 8);
    if ((len ^ 0xffffu) != nlen) return -1;
    b.in += 4;
    if ((size_t)(b.end - b.in) < len) return -1;
    return (long)len;
}

// Blocks from bit `start` of the stream until its last block, or until a block ends at bit `stop` exactly (to pass it is an error).
long inflate_span(const uint8_t *in, size_t in_len, uint64_t start, uint64_t stop, uint8_t *out, size_t out_cap, const uint8_t *hist,
                  uint64_t *end_bit, bool *was_last, std::atomic<size_t> *progress)
{
    Bits b;
    if (!bits_at(b, in, in_len, start)) return -1;
    uint8_t *o = out, *const oend = out + out_cap;
    Tables dyn;
    bool last = false;
    for (;;) {
        if (!b.need(3)) return -1;
        last = b.take(1) != 0;
        const uint32_t type = b.take(2);
        if (type == 0) {
            const long len = stored_header(b);
            if (len < 0 || (size_t)(oend - o) < (size_t)len) return -1;
            memcpy(o, b.in, (size_t)len);
            o += len;
            b.in += len;
        } else if (type == 1) {
            if (!decode_block(b, *fixed_tables(), o, oend, hist)) return -1;
        } else if (type == 2) {
            if (!read_dynamic(b, dyn) || !decode_block(b, dyn, o, oend, hist)) return -1;
        } else
            return -1;
        if (progress) progress->store((size_t)(o - out), std::memory_order_release);
        if (last) break;
        if (stop != kNoStop) {
            const uint64_t at = bit_pos(b, in);
            if (at == stop) break;
            if (at > stop) return -1;
        }
    }
    *end_bit = bit_pos(b, in);
    *was_last = last;
    return (long)(o - out);
}

// --- one stream by several threads ---------------------------------------------------------------------------------------------
// A deflate stream can only be decoded from its start: a block says where the next one begins only by ending, and a match may reach
// 32 KB back into text that a decoder starting in the middle has not seen.  Both can be worked around for text (the idea is that of
// Kerbiriou and Chikhi's pugz; the code is this file's):
//  * a block start can be guessed: try every bit position, keep the first where a dynamic-code block header is well-formed, the
//    block decodes to text characters only and another well-formed header follows;
//  * a thread that starts there decodes into 16-bit symbols: a byte, or "the byte at place i of the 32 KB in front of my start".
//    Matches copy symbols like bytes.  Once the last 32 KB it made hold bytes only, nothing unknown can be reached any more and it goes
//    on with the byte decoder.  When the thread in front has finished, the places get their bytes.
// A wrong guess shows when the thread in front does not arrive exactly at the guessed bit (then the whole stream is decoded again by
// one thread) - and the caller checks the container's CRC-32 over the result in any case.

constexpr size_t kWin = 32768;

// (bytes of UTF-8 names count as text; what rules a trial block out is a control character - one literal in nine of anything binary)
inline bool text_byte(uint32_t c) { return (c >= 0x20 && c != 0x7f) || c == '\n' || c == '\r' || c == '\t'; }

// one block's symbols as 16-bit values behind `o`; `base` is the start of the buffer (the unknown window included)
bool decode_block_sym(Bits &c, const Tables &t, uint16_t *&out, uint16_t *const oend, const uint16_t *base, bool text_only)
{
    constexpr uint32_t lmask = (1u << kLitRoot) - 1, dmask = (1u << kDistRoot) - 1;
    uint16_t *o = out;
    if (!text_only) {
        // decode_block's first loop, on 16-bit places (the guesses' trial blocks take the careful loop below, which looks at every literal)
        const uint8_t *in = c.in, *const iend = c.end;
        uint64_t bb = c.bb;
        uint32_t bc = c.bc;
        bool ended = false;
        while (iend - in >= 8 && oend - o >= 68 + 258 + 16) {
            bb |= load64(in) << bc;
            in += (63 - bc) >> 3;
            bc |= 56;
            uint32_t e = t.lit[bb & lmask];
            if (!(e & kIsLit) && (e & kKind) == kSub) {
                bb >>= kLitRoot; bc -= kLitRoot;
                e = t.lit[(e >> 16) + (uint32_t)(bb & ((1u << ((e >> 8) & 31)) - 1))];
            }
            bb >>= (e & 63); bc -= (e & 31);
            if (e & kIsLit) {
                do {
                    *o++ = (uint16_t)((e >> 8) & 0xff);
                    e = t.lit[bb & lmask];
                    if (!(e & kIsLit) || (e & 31) > bc) break;
                    bb >>= (e & 63); bc -= (e & 31);
                } while (true);
                continue;
            }
            if ((e & kKind) != kBase) {
                if ((e & kKind) != kEnd) return false;
                ended = true;
                break;
            }
            uint32_t x = (e >> 8) & 31;
            const uint32_t len = (e >> 16) + (uint32_t)(bb & ((1u << x) - 1));
            bb >>= x; bc -= x;
            uint32_t d = t.dist[bb & dmask];
            if ((d & kKind) == kSub) {
                bb >>= kDistRoot; bc -= kDistRoot;
                d = t.dist[(d >> 16) + (uint32_t)(bb & ((1u << ((d >> 8) & 31)) - 1))];
            }
            bb >>= (d & 63); bc -= (d & 31);
            if ((d & kKind) != kBase) return false;
            x = (d >> 8) & 31;
            const uint32_t dist = (d >> 16) + (uint32_t)(bb & ((1u << x) - 1));
            bb >>= x; bc -= x;
            if ((size_t)dist > (size_t)(o - base)) return false;
            const uint16_t *s = o - dist;
            uint16_t *w = o;
            o += len;
            if (dist >= 8) {
                do { memcpy(w, s, 16); w += 8; s += 8; } while (w < o);
            } else if (dist >= 4) {
                do { memcpy(w, s, 8); w += 4; s += 4; } while (w < o);
            } else {
                do { *w++ = *s++; } while (w < o);
            }
        }
        in -= bc >> 3;
        bc &= 7;
        bb &= (1ull << bc) - 1;
        c = Bits{in, iend, bb, bc};
        if (ended) { out = o; return true; }
    }
    for (;;) {
        c.fill();
        uint32_t e = t.lit[c.bb & lmask];
        if (!(e & kIsLit) && (e & kKind) == kSub) {
            if (c.bc < (uint32_t)kLitRoot) return false;
            c.take(kLitRoot);
            e = t.lit[(e >> 16) + (uint32_t)(c.bb & ((1u << ((e >> 8) & 31)) - 1))];
        }
        if ((e & 31) > c.bc) return false;
        c.take(e & 31);
        if (e & kIsLit) {
            const uint32_t v = (e >> 8) & 0xff;
            if (o >= oend || (text_only && !text_byte(v))) return false;
            *o++ = (uint16_t)v;
            continue;
        }
        if ((e & kKind) == kEnd) break;
        if ((e & kKind) != kBase) return false;
        uint32_t x = (e >> 8) & 31;
        if (x > c.bc) return false;
        const uint32_t len = (e >> 16) + c.take(x);
        uint32_t d = t.dist[c.bb & dmask];
        if ((d & kKind) == kSub) {
            if (c.bc < (uint32_t)kDistRoot) return false;
            c.take(kDistRoot);
            d = t.dist[(d >> 16) + (uint32_t)(c.bb & ((1u << ((d >> 8) & 31)) - 1))];
        }
        if ((d & kKind) != kBase || (d & 31) > c.bc) return false;
        c.take(d & 31);
        x = (d >> 8) & 31;
        if (x > c.bc) return false;
        const uint32_t dist = (d >> 16) + c.take(x);
        if ((size_t)dist > (size_t)(o - base) || (size_t)len > (size_t)(oend - o)) return false;
        const uint16_t *s = o - dist;
        for (uint32_t i = 0; i < len; i++) o[i] = s[i];
        o += len;
    }
    out = o;
    return true;
}

// does a well-formed block header stand at the reader's place?
bool header_plausible(Bits b, Tables &scratch)
{
    if (!b.need(3)) return false;
    b.take(1);
    const uint32_t type = b.take(2);
    if (type == 0) return stored_header(b) >= 0;
    if (type == 1) return true;
    if (type == 2) return read_dynamic(b, scratch);
    return false;
}

// the first bit position in [from, until) where a block as described above starts, or -1
int64_t find_block_start(const uint8_t *in, size_t in_len, uint64_t from, uint64_t until, uint16_t *scratch16, size_t scratch_n)
{
    Tables t, t2;
    for (size_t i = 0; i < kWin; i++) scratch16[i] = 0x20;                  // (stands for the unknown text in front)
    for (uint64_t p = from; p < until && p / 8 + 8 <= in_len; p++) {
        const uint64_t w = load64(in + p / 8) >> (p & 7);                    // 56 bits and more from p on
        if ((w & 7) != 4) continue;                                          // not the last block, dynamic codes
        if (((w >> 3) & 31) > 29 || ((w >> 8) & 31) > 29) continue;          // more length or distance codes than there are
        Bits b;
        if (!bits_at(b, in, in_len, p)) return -1;
        b.need(3);
        b.take(3);
        if (!read_dynamic(b, t)) continue;
        uint16_t *o = scratch16 + kWin;
        if (!decode_block_sym(b, t, o, scratch16 + scratch_n, scratch16, true)) continue;
        if (o - (scratch16 + kWin) < 1024) continue;                         // (real blocks of a large file are not this small)
        if (!header_plausible(b, t2)) continue;
        return (int64_t)p;
    }
    return -1;
}

const bool g_debug = bk::env::inflate_debug();      // pieces, sizes and stage times on stderr (tools/inflate_bench.sh)

inline double now_s() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

struct Piece {
    uint64_t start = 0, stop = kNoStop, end_bit = 0;
    uint16_t *sym = nullptr;            // kWin places of the unknown window, then n_sym decoded symbols
    size_t sym_cap = 0, n_sym = 0;
    uint8_t *tail = nullptr;            // kWin bytes of history (pieces after the first), then n_tail decoded bytes
    size_t tail_cap = 0, n_tail = 0;
    bool ok = false, last = false;
    std::vector<uint8_t> window;        // what the kWin places stand for, once the piece in front is known
};

void *lazy_pages(size_t bytes)
{
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    return p == MAP_FAILED ? nullptr : p;
}

void decode_piece(const uint8_t *in, size_t in_len, Piece &pc, bool first)
{
    const double t_start = g_debug ? now_s() : 0;
    struct Say {
        const Piece &p; double t0;
        ~Say() { if (g_debug) fprintf(stderr, "  piece at bit %llu took %.3f s\n", (unsigned long long)p.start, now_s() - t0); }
    } say{pc, t_start};
    const uint64_t span_bits = (pc.stop == kNoStop ? (uint64_t)in_len * 8 : pc.stop) - pc.start;
    const size_t room = (size_t)(span_bits / 8) * 8 + (4u << 20);          // text of eight times the compressed bytes, as for the whole
    pc.tail_cap = kWin + room;
    pc.tail = (uint8_t *)lazy_pages(pc.tail_cap);
    if (!pc.tail) return;
    uint64_t at = pc.start;
    if (!first) {
        pc.sym_cap = kWin + room;
        pc.sym = (uint16_t *)lazy_pages(pc.sym_cap * 2);
        if (!pc.sym) return;
        for (size_t i = 0; i < kWin; i++) pc.sym[i] = (uint16_t)(256 + i);
        uint16_t *o = pc.sym + kWin, *const oend = pc.sym + pc.sym_cap;
        size_t last_unknown = kWin - 1;                                      // place of the last symbol that is not a byte
        Bits b;
        if (!bits_at(b, in, in_len, at)) return;
        Tables dyn;
        bool clean = false;
        for (;;) {
            if (!b.need(3)) return;
            const bool last = b.take(1) != 0;
            const uint32_t type = b.take(2);
            uint16_t *const o0 = o;
            if (type == 0) {
                const long len = stored_header(b);
                if (len < 0 || (size_t)(oend - o) < (size_t)len) return;
                for (long i = 0; i < len; i++) o[i] = b.in[i];
                o += len;
                b.in += len;
            } else if (type == 1) {
                if (!decode_block_sym(b, *fixed_tables(), o, oend, pc.sym, false)) return;
            } else if (type == 2) {
                if (!read_dynamic(b, dyn) || !decode_block_sym(b, dyn, o, oend, pc.sym, false)) return;
            } else
                return;
            for (uint16_t *q = o; q-- > o0;)
                if (*q >= 256) { last_unknown = (size_t)(q - pc.sym); break; }
            at = bit_pos(b, in);
            if (last) { pc.last = true; break; }
            if (pc.stop != kNoStop && at >= pc.stop) { if (at > pc.stop) return; break; }
            if ((size_t)(o - pc.sym) >= 2 * kWin && last_unknown + kWin < (size_t)(o - pc.sym)) { clean = true; break; }
        }
        pc.n_sym = (size_t)(o - pc.sym) - kWin;
        pc.end_bit = at;
        if (!clean) { pc.ok = true; return; }
        for (size_t i = 0; i < kWin; i++) pc.tail[i] = (uint8_t)o[(long)i - (long)kWin];
    }
    const long n = inflate_span(in, in_len, at, pc.stop, pc.tail + kWin, pc.tail_cap - kWin, first ? pc.tail + kWin : pc.tail, &pc.end_bit, &pc.last, nullptr);
    if (n < 0) return;
    pc.n_tail = (size_t)n;
    pc.ok = true;
}

}  // namespace

long inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, const uint8_t *hist, size_t *in_used, std::atomic<size_t> *progress)
{
    uint64_t end_bit = 0;
    bool last = false;
    const long n = inflate_span(in, in_len, 0, kNoStop, out, out_cap, hist, &end_bit, &last, progress);
    if (n < 0 || !last) return -1;
    if (in_used) *in_used = (size_t)((end_bit + 7) / 8);
    return n;
}

long inflate_raw_parallel(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *in_used, int nthreads, int *pieces_used)
{
    if (pieces_used) *pieces_used = 1;
    // compressed bytes from which another thread pays: 8 MB (BK_INFLATE_PIECE_MIN: the tests' small streams)
    const size_t piece_min = std::max<size_t>(4096, (size_t)bk::env::inflate_piece_min(8ULL << 20));
    size_t want = std::min<size_t>((size_t)std::max(1, nthreads), in_len / piece_min);
    if (want < 2) return inflate_raw(in, in_len, out, out_cap, out, in_used);
    // where the pieces start
    const double Tstart = g_debug ? now_s() : 0;
    std::vector<int64_t> found(want, -1);
    found[0] = 0;
    {
        std::vector<std::thread> th;
        for (size_t k = 1; k < want; k++)
            th.emplace_back([&, k]() {
                const size_t n16 = kWin + (4u << 20);
                uint16_t *scratch = (uint16_t *)lazy_pages(n16 * 2);
                if (!scratch) return;
                found[k] = find_block_start(in, in_len, (uint64_t)(in_len / want * k) * 8, (uint64_t)(in_len / want * (k + 1)) * 8, scratch, n16);
                munmap(scratch, n16 * 2);
            });
        for (auto &t : th) t.join();
    }
    const bool dbg = g_debug;
    const double T0 = dbg ? now_s() : 0;
    std::vector<Piece> pcs;
    for (size_t k = 0; k < want; k++)
        if (found[k] >= 0) { Piece p; p.start = (uint64_t)found[k]; pcs.push_back(p); }
    for (size_t j = 0; j + 1 < pcs.size(); j++) pcs[j].stop = pcs[j + 1].start;
    auto release = [&]() {
        for (auto &p : pcs) {
            if (p.sym) munmap(p.sym, p.sym_cap * 2);
            if (p.tail) munmap(p.tail, p.tail_cap);
        }
    };
    if (pcs.size() < 2) { release(); return inflate_raw(in, in_len, out, out_cap, out, in_used); }
    {
        std::vector<std::thread> th;
        for (size_t j = 1; j < pcs.size(); j++) th.emplace_back([&, j]() { decode_piece(in, in_len, pcs[j], false); });
        decode_piece(in, in_len, pcs[0], true);
        for (auto &t : th) t.join();
    }
    const double T1 = dbg ? now_s() : 0;
    bool ok = true;
    size_t total = 0;
    for (size_t j = 0; j < pcs.size() && ok; j++) {
        const Piece &p = pcs[j];
        ok = p.ok && (j + 1 < pcs.size() ? (!p.last && p.end_bit == p.stop) : p.last);
        total += p.n_sym + p.n_tail;
    }
    if (!ok || total > out_cap) {                       // a guess that did not hold (or text that does not fit): the plain way
        release();
        return ok ? -1 : inflate_raw(in, in_len, out, out_cap, out, in_used);
    }
    // Every piece's window is the last 32 KB in front of it: of the first piece's bytes for the second, and from then on of a piece's own
    // bytes or - where those are fewer - of its last symbols, read through its own window.  With the windows known, every piece's
    // thread puts its symbols' bytes and its own bytes in place.
    std::vector<size_t> at(pcs.size() + 1, 0);
    for (size_t j = 0; j < pcs.size(); j++) at[j + 1] = at[j] + pcs[j].n_sym + pcs[j].n_tail;
    auto byte_of = [](const Piece &p, uint32_t v) { return v < 256 ? (uint8_t)v : p.window[v - 256]; };
    if (pcs[0].n_tail < kWin) ok = false;               // (a first piece this short: not worth the case)
    for (size_t j = 1; j < pcs.size() && ok; j++) {
        Piece &p = pcs[j];
        const Piece &q = pcs[j - 1];
        p.window.resize(kWin);
        const size_t from_tail = std::min(kWin, q.n_tail), from_sym = kWin - from_tail;
        if (from_sym > q.n_sym) { ok = false; break; }
        for (size_t i = 0; i < from_sym; i++) p.window[i] = byte_of(q, q.sym[kWin + q.n_sym - from_sym + i]);
        memcpy(p.window.data() + from_sym, q.tail + kWin + q.n_tail - from_tail, from_tail);
    }
    if (ok) {
        std::vector<std::thread> th;
        auto place = [&](size_t j) {
            const Piece &p = pcs[j];
            uint8_t *dst = out + at[j];
            if (p.n_sym) {
                std::vector<uint8_t> map(256 + kWin);
                for (size_t v = 0; v < 256; v++) map[v] = (uint8_t)v;
                memcpy(map.data() + 256, p.window.data(), kWin);
                const uint16_t *sy = p.sym + kWin;
                for (size_t i = 0; i < p.n_sym; i++) dst[i] = map[sy[i]];
            }
            memcpy(dst + p.n_sym, p.tail + kWin, p.n_tail);
        };
        for (size_t j = 1; j < pcs.size(); j++) th.emplace_back(place, j);
        place(0);
        for (auto &t : th) t.join();
    }
    if (dbg) {
        fprintf(stderr, "inflate: %zu pieces; starts found %.3f s, decoded %.3f s, joined %.3f s\n", pcs.size(), T0 - Tstart, T1 - T0, now_s() - T1);
        for (auto &p : pcs) fprintf(stderr, "  piece at bit %llu: %zu symbols, %zu bytes\n", (unsigned long long)p.start, p.n_sym, p.n_tail);
    }
    const uint64_t end_bit = pcs.back().end_bit;
    const size_t n_pieces = pcs.size();
    release();
    if (!ok) return inflate_raw(in, in_len, out, out_cap, out, in_used);
    if (in_used) *in_used = (size_t)((end_bit + 7) / 8);
    if (pieces_used) *pieces_used = (int)n_pieces;
    return (long)total;
}

}  // namespace bk
