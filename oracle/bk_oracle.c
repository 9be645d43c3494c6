/*
 * bk_oracle.c - CPU ORACLE (test infrastructure only, see bk_oracle.h) for the `biokanga align`
 * hot path.  Plain C restatement; every function cites the reference file:line it follows.
 * Deliberately simple and scalar: 1 byte/base target exactly as stored in the .sfx, the suffix
 * array read element by element, the bisections restated probe for probe.
 */
#define _GNU_SOURCE
#include "bk_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

/* libbiokanga/commdefs.h:108-121 */
enum { B_A = 0, B_C = 1, B_G = 2, B_T = 3, B_N = 4, B_UNDEF = 5, B_INDEL = 6, B_EOS = 7 };
#define RPT_MSK 0x08

/* libbiokanga/SfxArrayV2.h:68-74 (tHRslt) */
enum { HR_NONE = 0, HR_HITS = 1, HR_MMDELTA = 2, HR_HITINSTS = 3, HR_RMMDELTA = 4 };
/* biokanga/Aligner.h:106-128 (teNAR) */
enum { NAR_UNALIGNED = 0, NAR_ACCEPTED = 1, NAR_NS = 2, NAR_NOHIT = 3, NAR_MMDELTA = 4, NAR_MULTIALIGN = 5 };
/* libbiokanga/SfxArrayV2.h:61-66 (eALStrand) */
enum { ALS_BOTH = 0, ALS_WATSON = 1, ALS_CRICK = 2, ALS_NONE = 3 };

#define HASH_MASK 0x3fff            /* cHashEntries, SfxArrayV2.h:16 */
#define MAX_IDENT_NODES 1024000     /* cMaxNumIdentNodes, SfxArrayV2.h:15 */
#define MAX_TOT_SUBS 63             /* cMaxTotAllowedSubs, Aligner.h:23 */

/* ------------------------------------------------------------------------------------------- */
/* .sfx reader: header tsSfxHeaderV3 pack(4) 1224 B (SfxArrayV2.h:174-187), block header pack(1)
 * 20 B + bases + SA (SfxArrayV2.h:97-104), entries block 8 B + 111 B/entry (SfxArrayV2.h:79-95). */

static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }

int ora_sfx_load(const char *path, ora_sfx **out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long long flen = ftello(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *img = (uint8_t *)malloc((size_t)flen);
    if (!img) { fclose(f); return -2; }
    size_t got = 0;
    while (got < (size_t)flen) {
        size_t n = fread(img + got, 1, (size_t)flen - got, f);
        if (n == 0) break;
        got += n;
    }
    fclose(f);
    if (got != (size_t)flen || flen < 1224) { free(img); return -3; }
    /* Disk2Hdr, SfxArrayV2.cpp:551-631: magic "sfx5" (3..5 accepted), version 3..5; only v4/v5
     * (81-char names) header layout is handled here. */
    if (img[0] != 's' || img[1] != 'f' || img[2] != 'x' || img[3] < '3' || img[3] > '5') { free(img); return -4; }
    uint32_t version = rd32(img + 4);
    if (version < 4 || version > 5) { free(img); return -5; }
    /* pack(4): Magic 0, Version 4, Attributes 8, FileLen 12, EntriesOfs 20, EntriesSize 28,
     * NumSfxBlocks 32, SfxBlockSize 36, SfxBlockOfs 44, szDatasetName 52 (81), szDescription 133
     * (1024), szTitle 1157 (64) -> 1221, rounded to 1224 */
    uint64_t entries_ofs = rd64(img + 20);
    uint32_t entries_size = rd32(img + 28);
    uint32_t num_blocks = rd32(img + 32);
    uint64_t block_ofs = rd64(img + 44);
    if (num_blocks != 1 || entries_ofs == 0 || entries_size < 8) { free(img); return -6; }

    ora_sfx *s = (ora_sfx *)calloc(1, sizeof(ora_sfx));
    s->file_image = img;
    memcpy(s->dataset, img + 52, 81);
    s->dataset[80] = 0;
    const uint8_t *blk = img + block_ofs;
    s->block_id = rd32(blk);
    s->concat_len = rd64(blk + 8);
    s->el_size = rd32(blk + 16);
    s->seq = (uint8_t *)blk + 20;
    s->sa = s->seq + s->concat_len;
    if (s->el_size != 4 && s->el_size != 5) { free(img); free(s); return -7; }
    const uint8_t *eb = img + entries_ofs;
    s->num_entries = rd32(eb);
    s->entries = (ora_entry *)calloc(s->num_entries ? s->num_entries : 1, sizeof(ora_entry));
    const uint8_t *e = eb + 8;
    for (uint32_t i = 0; i < s->num_entries; i++, e += 111) {
        ora_entry *d = &s->entries[i];
        d->entry_id = rd32(e);
        d->fblock_id = rd32(e + 4);
        memcpy(d->name, e + 8, 81);
        d->name[80] = 0;
        d->name_hash = rd16(e + 89);
        d->seq_len = rd32(e + 91);
        d->start_ofs = rd64(e + 95);
        d->end_ofs = rd64(e + 103);
        s->tot_seq_len += d->seq_len;     /* GetTotSeqsLen, SfxArrayV2.cpp:2070-2082 */
    }
    *out = s;
    return 0;
}

/* index image already in host memory (bench.py: copied back from the GPU-built synthetic index).
 * seq / sa are borrowed, not copied; entries: n_entries x {entry_id, seq_len, start_ofs, end_ofs}. */
int ora_sfx_from_memory(const uint8_t *seq, const uint8_t *sa, uint64_t concat_len, uint32_t el_size,
                        const uint64_t *entries4, uint32_t n_entries, ora_sfx **out)
{
    if (!seq || !sa || !entries4 || !n_entries || (el_size != 4 && el_size != 5)) return -1;
    ora_sfx *s = (ora_sfx *)calloc(1, sizeof(ora_sfx));
    s->seq = (uint8_t *)seq;
    s->sa = (uint8_t *)sa;
    s->concat_len = concat_len;
    s->el_size = el_size;
    s->block_id = 1;
    s->num_entries = n_entries;
    s->entries = (ora_entry *)calloc(n_entries, sizeof(ora_entry));
    for (uint32_t i = 0; i < n_entries; i++) {
        ora_entry *d = &s->entries[i];
        d->entry_id = (uint32_t)entries4[4 * i];
        d->fblock_id = 1;
        d->seq_len = (uint32_t)entries4[4 * i + 1];
        d->start_ofs = entries4[4 * i + 2];
        d->end_ofs = entries4[4 * i + 3];
        snprintf(d->name, sizeof(d->name), "seq%u", d->entry_id);
        s->tot_seq_len += d->seq_len;
    }
    *out = s;
    return 0;
}

void ora_sfx_free(ora_sfx *s)
{
    if (!s) return;
    free(s->entries);
    free(s->file_image);
    free(s);
}

/* ------------------------------------------------------------------------------------------- */
/* SfxOfsToLoci, SfxArrayV2.cpp:33-44 */
static inline int64_t sa_at(const ora_sfx *s, int64_t idx)
{
    const uint8_t *p = s->sa + idx * (int64_t)s->el_size;
    uint64_t v = rd32(p);
    if (s->el_size == 5) v |= ((uint64_t)p[4]) << 32;
    return (int64_t)v;
}

/* probe-vs-suffix comparison shared by the bisections and the candidate walk
 * (SfxArrayV2.cpp:7791-7812, 5884-5910): low nibbles; a target EOS makes the probe sort lower. */
static inline int cmp_probe(const uint8_t *probe, const uint8_t *targ, int len)
{
    for (int i = 0; i < len; i++) {
        uint8_t t = targ[i] & 0x0f;
        if (t == B_EOS) return -1;
        uint8_t p = probe[i] & 0x0f;
        if (p > t) return 1;
        if (p < t) return -1;
    }
    return 0;
}

/* LocateFirstExact, SfxArrayV2.cpp:7765-7876 (TargStart is always 0 on this path) */
int64_t ora_locate_first_exact(const ora_sfx *s, const uint8_t *probe, int plen,
                               int64_t lo, int64_t hi, ora_counters *ctr)
{
    if (ctr) ctr->n_search++;
    do {
        int64_t mid = (lo + hi) / 2;
        int c = cmp_probe(probe, s->seq + sa_at(s, mid), plen);
        if (ctr) ctr->n_probe++;
        if (c == 0) {
            if (mid == 0 || lo == mid) return mid + 1;
            int64_t mark = 0;
            for (;;) {                       /* walk down to the lowest matching index */
                if (c == 0) {
                    mark = mid;
                    if (mark == 0) return mark + 1;
                    hi = mid - 1;
                }
                mid = (lo + hi) / 2;
                c = cmp_probe(probe, s->seq + sa_at(s, mid), plen);
                if (ctr) ctr->n_probe++;
                if (c == 0) continue;
                lo = mid + 1;
                if (lo == mark) return mark + 1;
            }
        }
        if (c < 0) {
            if (mid == 0) break;
            hi = mid - 1;
        } else
            lo = mid + 1;
    } while (hi >= lo);
    return 0;
}

/* LocateLastExact, SfxArrayV2.cpp:7914-8027 */
int64_t ora_locate_last_exact(const ora_sfx *s, const uint8_t *probe, int plen,
                              int64_t lo, int64_t hi, ora_counters *ctr)
{
    int64_t hi_max = hi;
    if (ctr) ctr->n_last_search++;
    do {
        int64_t mid = (lo + hi) / 2;
        int c = cmp_probe(probe, s->seq + sa_at(s, mid), plen);
        if (ctr) ctr->n_probe++;
        if (c == 0) {
            if (mid == hi_max || hi == mid) return mid + 1;
            int64_t mark = 0;
            for (;;) {                       /* walk up to the highest matching index */
                if (c == 0) {
                    mark = mid;
                    if (mark == hi) return mark + 1;
                    lo = mid + 1;
                }
                mid = (lo + hi) / 2;
                c = cmp_probe(probe, s->seq + sa_at(s, mid), plen);
                if (ctr) ctr->n_probe++;
                if (c == 0) continue;
                hi = mid - 1;
                if (hi == mark) return mark + 1;
            }
        }
        if (c < 0) {
            if (mid == 0) break;
            hi = mid - 1;
        } else
            lo = mid + 1;
    } while (hi >= lo);
    return 0;
}

/* MapChunkHit2Entry, SfxArrayV2.cpp:2530-2575 */
static const ora_entry *map_entry(const ora_sfx *s, uint64_t ofs)
{
    int64_t lo = 0, hi = (int64_t)s->num_entries - 1;
    while (hi >= lo) {
        int64_t mid = (hi + lo) / 2;
        const ora_entry *e = &s->entries[mid];
        uint32_t b = e->fblock_id & 0xff;
        if (b > s->block_id) { hi = mid - 1; continue; }
        if (b < s->block_id) { lo = mid + 1; continue; }
        if (e->start_ofs <= ofs && e->end_ofs >= ofs) return e;
        if (e->start_ofs > ofs) { hi = mid - 1; continue; }
        if (e->end_ofs < ofs) { lo = mid + 1; continue; }
    }
    return NULL;
}

/* CSeqTrans::ReverseComplement, SeqTrans.cpp:458-512: complement A<->T, C<->G keeping the mask
 * bits; N/InDel/Undef unchanged; complementing stops at anything else; then reverse. */
static void revcomp(uint8_t *seq, int len)
{
    for (int i = 0; i < len; i++) {
        uint8_t flg = seq[i] & (RPT_MSK | 0x10);
        uint8_t b = seq[i] & ~(RPT_MSK | 0x10);
        if (b <= B_T) seq[i] = (uint8_t)((3 - b) | flg);
        else if (b == B_N || b == B_INDEL || b == B_UNDEF) continue;
        else break;
    }
    for (int i = 0, j = len - 1; i < j; i++, j--) { uint8_t t = seq[i]; seq[i] = seq[j]; seq[j] = t; }
}

/* ------------------------------------------------------------------------------------------- */
typedef struct ident_node { uint32_t id; int32_t next; } ident_node;

typedef struct scratch {
    int32_t     heads[HASH_MASK + 1];
    ident_node *nodes;          /* MAX_IDENT_NODES */
} scratch;

typedef struct hit_rec {
    uint8_t  strand;
    uint32_t chrom_id;
    uint32_t match_loci;
    uint16_t match_len;
    uint8_t  mismatches;
} hit_rec;

/* LocateCoreMultiples, SfxArrayV2.cpp:5693-6262 - standard (non chimeric, basespace, non
 * bisulfite) path.  probe is modified in place while the '-' strand is processed and restored. */
static int locate_core_multiples(const ora_sfx *s, int max_tot_mm, int core_len, int core_delta,
                                 int max_slides, int mm_delta, int align2strand,
                                 int *p_low_inst, int *p_low_mm, int *p_nxt_low_mm,
                                 uint8_t *probe, int plen, int max_hits, hit_rec *hits,
                                 int max_iter, scratch *sc, ora_counters *ctr)
{
    int low_inst, low_mm, nxt_low_mm;
    char cur_strand;
    int64_t sfx_len = (int64_t)s->concat_len;
    int cur_hit = -1;                              /* index into hits[] or -1 (pCurHit == NULL) */

    if (ctr) ctr->n_lcm_calls++;
    if (s->concat_len == 0) return -1;
    if (*p_low_inst > max_hits && *p_low_mm == 0) return HR_HITINSTS;                 /* :5778 */
    if (*p_low_inst >= 1 && *p_low_mm == 0 && (*p_nxt_low_mm - *p_low_mm) < mm_delta)  /* :5782 */
        return HR_MMDELTA;

    if (*p_low_inst <= 0 || *p_low_mm < 0 || *p_nxt_low_mm < 0) {                     /* :5790 */
        low_inst = *p_low_inst = 0;
        low_mm = *p_low_mm = max_tot_mm + mm_delta + 1;
        nxt_low_mm = *p_nxt_low_mm = low_mm;
    } else {
        low_inst = *p_low_inst; low_mm = *p_low_mm; nxt_low_mm = *p_nxt_low_mm;
    }
    if (low_inst < max_hits) cur_hit = low_inst;

    if (align2strand == ALS_CRICK) { revcomp(probe, plen); cur_strand = '-'; }
    else cur_strand = '+';

    do {
        int cur_delta = core_delta;
        int n_slides = 0;
        int n_nodes = 0;
        memset(sc->heads, 0xff, sizeof(sc->heads));                                   /* :5834 */
        for (int core_ofs = 0;
             n_slides < max_slides && core_ofs <= (plen - core_len) &&
             cur_delta > core_len / 3 && n_nodes < MAX_IDENT_NODES;
             n_slides++, core_ofs += cur_delta) {
            if ((core_ofs + core_len + cur_delta) > plen)                              /* :5846 */
                cur_delta = plen - (core_ofs + core_len);

            int64_t targ_idx = ora_locate_first_exact(s, probe + core_ofs, core_len, 0, sfx_len - 1, ctr);
            if (targ_idx == 0) continue;
            targ_idx -= 1;
            int iter_cnt = 0;
            uint32_t num_copies = 0;
            int first_iter = 1;
            while (!max_iter || iter_cnt < max_iter) {
                if (n_nodes >= MAX_IDENT_NODES) break;
                if (!first_iter) {
                    if ((targ_idx + 1) >= sfx_len ||
                        (sa_at(s, targ_idx + 1) + core_len) > sfx_len) break;         /* :5865 */
                    if (iter_cnt == 100 && !num_copies) {                             /* :5868 */
                        int64_t last = ora_locate_last_exact(s, probe + core_ofs, core_len,
                                                             targ_idx - 1, sfx_len - 1, ctr);
                        num_copies = last > 0 ? (uint32_t)(1 + last - targ_idx) : 0;
                        if (max_iter && num_copies > (uint32_t)max_iter) break;
                    }
                    if (cmp_probe(probe + core_ofs, s->seq + sa_at(s, targ_idx + 1), core_len) != 0)
                        break;
                    targ_idx += 1;
                }
                first_iter = 0;
                if (ctr) ctr->n_cand_seen++;
                int64_t loci = sa_at(s, targ_idx);
                if (loci < (int64_t)(uint32_t)core_ofs) continue;                      /* :5918 */
                int64_t left = loci - core_ofs;
                const ora_entry *ent = map_entry(s, (uint64_t)left);
                if (ent == NULL || !plen || ((uint64_t)left + (uint32_t)plen - 1) > ent->end_ofs)
                    continue;                                                          /* :5928 */

                /* dedupe on the (32 bit truncated) target start, :5932-5950 */
                uint32_t targ_id = (uint32_t)(1 + loci - (uint32_t)core_ofs);
                int h = (int)(targ_id & HASH_MASK);
                int32_t n = sc->heads[h];
                int dup = 0;
                while (n >= 0) {
                    if (sc->nodes[n].id == targ_id) { dup = 1; break; }
                    n = sc->nodes[n].next;
                }
                if (dup) continue;
                sc->nodes[n_nodes].id = targ_id;
                sc->nodes[n_nodes].next = sc->heads[h];
                sc->heads[h] = n_nodes++;
                iter_cnt++;
                if (ctr) ctr->n_cand++;

                /* Hamming extension over the whole read with the reference's early exits,
                 * :6085-6154 */
                const uint8_t *t = s->seq + left;
                int mm = 0, i;
                for (i = 0; i < plen; i++) {
                    uint8_t tb = t[i] & 0x0f, pb = probe[i] & 0x0f;
                    if (tb == B_EOS) break;
                    if (pb == tb) continue;
                    if (++mm > max_tot_mm) break;
                    if (mm >= nxt_low_mm) break;
                }
                if (i != plen) continue;

                if (mm < low_mm) {                                                     /* :6157 */
                    cur_hit = 0;
                    low_inst = 1;
                    nxt_low_mm = low_mm;
                    low_mm = mm;
                    hits[0].strand = (uint8_t)cur_strand;
                    hits[0].chrom_id = ent->entry_id;
                    hits[0].match_loci = (uint32_t)((uint64_t)left - ent->start_ofs);
                    hits[0].match_len = (uint16_t)plen;
                    hits[0].mismatches = (uint8_t)mm;
                } else if (mm == low_mm) {                                             /* :6179 */
                    low_inst += 1;
                    if (cur_hit >= 0 && low_inst <= max_hits) {
                        cur_hit += 1;
                        hits[cur_hit].strand = (uint8_t)cur_strand;
                        hits[cur_hit].chrom_id = ent->entry_id;
                        hits[cur_hit].match_loci = (uint32_t)((uint64_t)left - ent->start_ofs);
                        hits[cur_hit].match_len = (uint16_t)plen;
                        hits[cur_hit].mismatches = (uint8_t)mm;
                    }
                } else if (mm < nxt_low_mm)
                    nxt_low_mm = mm;
                if (low_inst > max_hits && low_mm == 0) break;                         /* :6206 */
            }
            if (low_inst > max_hits && low_mm == 0) { align2strand = ALS_NONE; break; } /* :6210 */
        }
        if (cur_strand == '+' && align2strand == ALS_BOTH) {                           /* :6216 */
            revcomp(probe, plen);
            cur_strand = '-';
            align2strand = ALS_CRICK;
        } else
            align2strand = ALS_NONE;
    } while (!(low_inst > max_hits && low_mm == 0) && align2strand != ALS_NONE);

    if (cur_strand == '-') revcomp(probe, plen);                                       /* :6231 */

    if (*p_low_mm == low_mm && *p_low_inst == low_inst) {                              /* :6238 */
        if (*p_nxt_low_mm > nxt_low_mm) {
            *p_nxt_low_mm = nxt_low_mm;
            if ((nxt_low_mm - *p_low_mm) < mm_delta) return HR_MMDELTA;
            return HR_RMMDELTA;
        }
        return HR_NONE;
    }
    *p_low_mm = low_mm;
    *p_low_inst = low_inst;
    *p_nxt_low_mm = nxt_low_mm;
    if (*p_low_inst >= 1 && (*p_nxt_low_mm - *p_low_mm) < mm_delta) return HR_MMDELTA;
    if (*p_low_inst > max_hits) return HR_HITINSTS;
    return HR_HITS;
}

/* AlignReads, SfxArrayV2.cpp:7666-7760 with microInDelLen = MaxSpliceJunctLen = MinChimericLen = 0 */
static int align_reads(const ora_sfx *s, int max_tot_mm, int core_len, int core_delta, int max_slides,
                       int mm_delta, int align2strand, int *p_low_inst, int *p_low_mm, int *p_nxt,
                       uint8_t *probe, int plen, int max_hits, hit_rec *hits, int max_iter,
                       scratch *sc, ora_counters *ctr)
{
    int rslt, allow_mm;
    if (max_tot_mm > 0) {
        for (allow_mm = 0; allow_mm <= max_tot_mm; allow_mm++) {
            int cl = plen / (allow_mm + mm_delta);
            if (cl <= core_len) break;
            rslt = locate_core_multiples(s, allow_mm, cl, cl, max_slides, mm_delta, align2strand,
                                         p_low_inst, p_low_mm, p_nxt, probe, plen, max_hits, hits,
                                         max_iter, sc, ctr);
            if (rslt != 0) return rslt;
        }
    } else
        allow_mm = 0;
    if (allow_mm <= max_tot_mm) {
        rslt = locate_core_multiples(s, max_tot_mm, core_len, core_delta, max_slides, mm_delta,
                                     align2strand, p_low_inst, p_low_mm, p_nxt, probe, plen,
                                     max_hits, hits, max_iter, sc, ctr);
        if (rslt != 0) return rslt;
    }
    return 0;
}

/* CAligner::LocateCoredApprox, Aligner.cpp:8725-8761 */
int ora_min_core_len(const ora_sfx *s, int pmode)
{
    int m;
    uint64_t t = s->tot_seq_len;
    if (t <= 500000ULL) m = 4;
    else if (t <= 20000000ULL) m = 4 + 3;
    else if (t <= 250000000ULL) m = 4 + 7;
    else if (t <= 3500000000ULL) m = 4 + 8;
    else m = 4 + 11;
    switch (pmode) {
    case 2: break;            /* ePMUltraSens */
    case 1: m += 1; break;    /* ePMMoreSens  */
    case 0: m += 2; break;    /* ePMdefault   */
    default: m += 4; break;   /* less sensitive */
    }
    return m;
}
int ora_max_num_slides(int pmode)
{
    switch (pmode) { case 2: return 9; case 1: return 8; case 0: return 8; default: return 6; }
}
/* CAligner::Align, Aligner.cpp:341-356 + Aligner.h:38-41 */
int ora_max_iter(int pmode)
{
    switch (pmode) { case 0: return 5000; case 1: return 10000; case 2: return 20000; default: return 2500; }
}

static int imax(int a, int b) { return a > b ? a : b; }

static int align_read_sc(const ora_sfx *s, const ora_params *p, const uint8_t *bases, int len,
                         ora_hit *out, ora_counters *ctr, scratch *sc, uint8_t *seqbuf)
{
    memset(out, 0, sizeof(*out));
    out->nar = NAR_NOHIT;                                      /* Aligner.cpp:9030 */
    out->strand = '?';                                         /* CAligner::AddEntry :10652 */
    if (ctr) ctr->n_reads++;

    /* strip quality, N policy: Aligner.cpp:9041-9063 */
    int max_ns_seq = 0, num_ns = 0, i;
    if (p->max_ns) max_ns_seq = imax((len * p->max_ns) / 100, p->max_ns);
    for (i = 0; i < len; i++) {
        uint8_t b = bases[i] & 0x07;
        if (b > B_N) break;
        seqbuf[i] = b;
        if (b == B_N && ++num_ns > max_ns_seq) break;
    }
    if (i != len) { out->nar = NAR_NS; return 0; }

    int match_len = len;
    /* Aligner.cpp:9085-9095 */
    int max_tot_mm = p->max_subs == 0 ? 0 : imax(1, (int)(0.5 + (match_len * p->max_subs) / 100.0));
    if (max_tot_mm > MAX_TOT_SUBS) max_tot_mm = MAX_TOT_SUBS;
    int min_core = ora_min_core_len(s, p->pmode);
    int core_len = imax(min_core, match_len / (p->min_edit_dist == 1 ? max_tot_mm + 1 : max_tot_mm + 2));
    int max_slides = imax(1, ((ora_max_num_slides(p->pmode) * len) + 99) / 100);
    int core_delta = imax(len / max_slides - 1, core_len);
    int align2strand = p->align_strand == 0 ? ALS_BOTH : (p->align_strand == 1 ? ALS_WATSON : ALS_CRICK);

    int low_inst = 0, low_mm = 0, nxt = 0;
    int max_ml = p->max_ml > 0 ? p->max_ml : 1;
    hit_rec hits[8];
    memset(hits, 0, sizeof(hits));
    if (max_ml > 7) max_ml = 7;
    int rslt = align_reads(s, max_tot_mm, core_len, core_delta, max_slides, p->min_edit_dist, align2strand,
                           &low_inst, &low_mm, &nxt, seqbuf, match_len, max_ml, hits,
                           ora_max_iter(p->pmode), sc, ctr);
    if (rslt < 0) return rslt;
    if (low_inst > max_ml) low_inst = max_ml + 1;                                       /* :9241 */
    out->rslt = (uint8_t)rslt;

    switch (rslt) {                                                                     /* :9311-9479 */
    case HR_NONE:
        out->nar = NAR_NOHIT;
        break;
    case HR_HITS:
        /* default MLMode (eMLdefault): unique -> accepted, else multialign */
        if (low_inst == 1) {
            out->nar = NAR_ACCEPTED;
            out->num_hits = 1;
            out->strand = hits[0].strand;
            out->chrom_id = hits[0].chrom_id;
            out->match_loci = hits[0].match_loci;
            out->match_len = hits[0].match_len;
            out->mismatches = hits[0].mismatches;
        } else {
            out->nar = NAR_MULTIALIGN;
            out->num_hits = 0;
            /* HitLoci.Hit is left as it was (the '?' strand of AddEntry) */
        }
        out->low_hit_instances = (int16_t)low_inst;
        out->low_mm = (int8_t)low_mm;
        out->nxt_low_mm = (int8_t)nxt;
        break;
    case HR_MMDELTA:
        out->nar = NAR_MMDELTA;
        out->strand = '?';
        out->match_len = (uint16_t)len;
        out->low_hit_instances = (int16_t)low_inst;
        out->low_mm = (int8_t)low_mm;
        out->nxt_low_mm = (int8_t)nxt;
        break;
    case HR_HITINSTS:
        out->nar = NAR_MULTIALIGN;
        out->strand = '?';
        out->match_len = (uint16_t)len;
        out->low_hit_instances = (int16_t)low_inst;
        out->low_mm = (int8_t)low_mm;
        out->nxt_low_mm = (int8_t)nxt;
        break;
    case HR_RMMDELTA:
        out->nxt_low_mm = (int8_t)nxt;
        break;
    }
    return 0;
}

static scratch *scratch_new(void)
{
    scratch *sc = (scratch *)malloc(sizeof(scratch));
    sc->nodes = (ident_node *)malloc(sizeof(ident_node) * MAX_IDENT_NODES);
    return sc;
}
static void scratch_free(scratch *sc) { free(sc->nodes); free(sc); }

int ora_align_read(const ora_sfx *s, const ora_params *p, const uint8_t *bases, int len,
                   ora_hit *out, ora_counters *ctr)
{
    scratch *sc = scratch_new();
    uint8_t *buf = (uint8_t *)malloc((size_t)len + 16);
    int r = align_read_sc(s, p, bases, len, out, ctr, sc, buf);
    free(buf);
    scratch_free(sc);
    return r;
}

typedef struct worker {
    const ora_sfx *s; const ora_params *p; const uint8_t *bases; const uint64_t *offs;
    const uint32_t *lens; uint32_t lo, hi; ora_hit *out; ora_counters ctr; int rslt;
} worker;

static void *worker_main(void *arg)
{
    worker *w = (worker *)arg;
    scratch *sc = scratch_new();
    uint32_t maxlen = 0;
    for (uint32_t i = w->lo; i < w->hi; i++) if (w->lens[i] > maxlen) maxlen = w->lens[i];
    uint8_t *buf = (uint8_t *)malloc((size_t)maxlen + 16);
    for (uint32_t i = w->lo; i < w->hi; i++) {
        int r = align_read_sc(w->s, w->p, w->bases + w->offs[i], (int)w->lens[i], &w->out[i], &w->ctr, sc, buf);
        if (r < 0) { w->rslt = r; break; }
    }
    free(buf);
    scratch_free(sc);
    return NULL;
}

int ora_align_batch(const ora_sfx *s, const ora_params *p, const uint8_t *bases,
                    const uint64_t *offs, const uint32_t *lens, uint32_t nreads,
                    ora_hit *out, ora_counters *ctr, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if ((uint32_t)nthreads > nreads) nthreads = nreads ? (int)nreads : 1;
    worker *w = (worker *)calloc((size_t)nthreads, sizeof(worker));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    /* contiguous blocks of <= 4096 reads are what the reference threads pull (Aligner.cpp:9636);
     * reads are independent so a static split gives the same answers */
    uint64_t per = ((uint64_t)nreads + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; t++) {
        w[t].s = s; w[t].p = p; w[t].bases = bases; w[t].offs = offs; w[t].lens = lens; w[t].out = out;
        uint64_t lo = per * t, hi = lo + per;
        if (lo > nreads) lo = nreads;
        if (hi > nreads) hi = nreads;
        w[t].lo = (uint32_t)lo; w[t].hi = (uint32_t)hi;
        pthread_create(&th[t], NULL, worker_main, &w[t]);
    }
    int rslt = 0;
    ora_counters tot;
    memset(&tot, 0, sizeof(tot));
    for (int t = 0; t < nthreads; t++) {
        pthread_join(th[t], NULL);
        if (w[t].rslt < 0) rslt = w[t].rslt;
        tot.n_reads += w[t].ctr.n_reads; tot.n_search += w[t].ctr.n_search; tot.n_probe += w[t].ctr.n_probe;
        tot.n_last_search += w[t].ctr.n_last_search; tot.n_cand += w[t].ctr.n_cand;
        tot.n_cand_seen += w[t].ctr.n_cand_seen; tot.n_lcm_calls += w[t].ctr.n_lcm_calls;
    }
    if (ctr) *ctr = tot;
    free(w); free(th);
    return rslt;
}
