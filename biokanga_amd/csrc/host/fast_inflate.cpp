// fast_inflate.cpp - see fast_inflate.h.  The format is RFC 1951's; the tables and loops are this file's own.
#include "fast_inflate.h"

#include <cstring>

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

long inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, const uint8_t *hist, size_t *in_used, std::atomic<size_t> *progress)
{
    Bits b{in, in + in_len, 0, 0};
    uint8_t *o = out, *const oend = out + out_cap;
    Tables dyn;
    for (;;) {
        if (!b.need(3)) return -1;
        const bool last = b.take(1) != 0;
        const uint32_t type = b.take(2);
        if (type == 0) {
            b.in -= b.bc >> 3;                          // the rest of this byte is padding; whole bytes go back
            b.bb = 0;
            b.bc = 0;
            if (b.end - b.in < 4) return -1;
            const uint32_t len = (uint32_t)b.in[0] | ((uint32_t)b.in[1] << 8), nlen = (uint32_t)b.in[2] | ((uint32_t)b.in[3] << 8);
            if ((len ^ 0xffffu) != nlen) return -1;
            b.in += 4;
            if ((size_t)(b.end - b.in) < len || (size_t)(oend - o) < len) return -1;
            memcpy(o, b.in, len);
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
    }
    if (in_used) *in_used = (size_t)(b.in - in) - (b.bc >> 3);
    return (long)(o - out);
}

}  // namespace bk
