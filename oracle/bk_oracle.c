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
    uint8_t  chimeric;          /* FlgChimeric */
    uint16_t trim_left, trim_right;   /* Seg[0].TrimLeft / TrimRight (read orientation) */
} hit_rec;

static int adaptive_trim_ex(uint32_t seq_len, const uint8_t *probe, const uint8_t *targ, uint32_t min_trim_len, uint32_t max_mm,
                            uint32_t min_flank, uint32_t *p_trim_mms, uint32_t *p_trim_start, uint32_t *p_trim_end);

/* LocateCoreMultiples, SfxArrayV2.cpp:5693-6262 - basespace, non bisulfite.  min_chimeric_pct 50..99 selects the chimeric
 * form (:5959-6080): every candidate is end-trimmed adaptively and the longest (then cleanest) trimmed placement wins.
 * probe is modified in place while the '-' strand is processed and restored. */
static int locate_core_multiples_c(const ora_sfx *s, int min_chimeric_pct, int max_tot_mm, int core_len, int core_delta,
                                   int max_slides, int mm_delta, int align2strand,
                                   int *p_low_inst, int *p_low_mm, int *p_nxt_low_mm,
                                   uint8_t *probe, int plen, int max_hits, hit_rec *hits,
                                   int max_iter, scratch *sc, ora_counters *ctr);
static int locate_core_multiples(const ora_sfx *s, int max_tot_mm, int core_len, int core_delta,
                                 int max_slides, int mm_delta, int align2strand,
                                 int *p_low_inst, int *p_low_mm, int *p_nxt_low_mm,
                                 uint8_t *probe, int plen, int max_hits, hit_rec *hits,
                                 int max_iter, scratch *sc, ora_counters *ctr)
{
    return locate_core_multiples_c(s, 0, max_tot_mm, core_len, core_delta, max_slides, mm_delta, align2strand, p_low_inst, p_low_mm,
                                   p_nxt_low_mm, probe, plen, max_hits, hits, max_iter, sc, ctr);
}
static int locate_core_multiples_c(const ora_sfx *s, int min_chimeric_pct, int max_tot_mm, int core_len, int core_delta,
                                   int max_slides, int mm_delta, int align2strand,
                                   int *p_low_inst, int *p_low_mm, int *p_nxt_low_mm,
                                   uint8_t *probe, int plen, int max_hits, hit_rec *hits,
                                   int max_iter, scratch *sc, ora_counters *ctr)
{
    const uint32_t min_probe_chimeric = (min_chimeric_pct >= 50 && min_chimeric_pct <= 99) ? (uint32_t)((min_chimeric_pct * plen) / 100) : 0;
    int best_chim_len = 0, best_chim_mm = 0;
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

                if (min_probe_chimeric > 0) {                                          /* :5959-6080 */
                    uint32_t tmm = 0, t5 = 0, t3 = 0;
                    int clen = adaptive_trim_ex((uint32_t)plen, probe, s->seq + left, min_probe_chimeric, (uint32_t)max_tot_mm, 3, &tmm, &t5, &t3);
                    if (clen < 0) clen = 0;
                    if (clen < (int)min_probe_chimeric) continue;
                    const int cmm = (int)tmm;
                    hit_rec nh;
                    memset(&nh, 0, sizeof(nh));
                    if (clen > best_chim_len || (clen == best_chim_len && cmm < best_chim_mm) ||
                        (clen == best_chim_len && cmm == best_chim_mm)) {
                        const ora_entry *e2 = map_entry(s, (uint64_t)left + t5);
                        if (e2 == NULL) e2 = ent;
                        nh.chimeric = 1; nh.strand = (uint8_t)cur_strand;
                        nh.trim_left = (uint16_t)(cur_strand == '+' ? t5 : t3); nh.trim_right = (uint16_t)(cur_strand == '+' ? t3 : t5);
                        nh.chrom_id = e2->entry_id; nh.match_loci = (uint32_t)((uint64_t)left - e2->start_ofs);
                        nh.match_len = (uint16_t)plen; nh.mismatches = (uint8_t)cmm;
                    }
                    if (clen > best_chim_len || (clen == best_chim_len && cmm < best_chim_mm)) {
                        if (best_chim_len > 0 && clen > best_chim_len) low_mm = cmm + mm_delta + 1;
                        best_chim_len = clen; best_chim_mm = cmm;
                        cur_hit = 0;
                        low_inst = 1;
                        nxt_low_mm = low_mm;
                        low_mm = cmm;
                        hits[0] = nh;
                    } else if (clen == best_chim_len && cmm == best_chim_mm) {
                        low_inst += 1;
                        if (cur_hit >= 0 && low_inst <= max_hits) hits[++cur_hit] = nh;
                    } else if (clen == best_chim_len && cmm < nxt_low_mm)
                        nxt_low_mm = cmm;
                    if (clen == plen && low_inst > max_hits && low_mm == 0) break;
                    continue;
                }
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

/* LocateBestMatches, SfxArrayV2.cpp:6654-7019 (`-N`: at most max_hits alignments with the fewest mismatches, none
 * above max_tot_mm; no phase schedule, no Hamming-delta rule).  Differences from LocateCoreMultiples that matter:
 * a candidate is hashed (and counted towards the iteration limits) before anything is known about the entry it
 * lies in - only the concatenation end is checked (:6816) - and entry boundaries are caught by the EOS test of
 * the Hamming loop; hits[] is kept ordered by mismatches, a new hit going in front of the first one with MORE
 * mismatches (:6917-6935), and once full the mismatch limit drops to that of its last element (:6957-6961).
 * hits[] must have room for max_hits + 1 entries (the reference's memmove moves one past the limit when full). */
static int locate_best_matches(const ora_sfx *s, int max_tot_mm, int core_len, int core_delta, int max_slides,
                               int align2strand, uint8_t *probe, int plen, int max_hits, hit_rec *hits,
                               int max_iter, scratch *sc, ora_counters *ctr)
{
    int64_t sfx_len = (int64_t)s->concat_len;
    int low_inst = 0, sloughed = 0;
    char cur_strand;
    if (ctr) ctr->n_lcm_calls++;
    if (s->concat_len == 0) return -1;
    if (align2strand == ALS_CRICK) { revcomp(probe, plen); cur_strand = '-'; }
    else cur_strand = '+';
    do {
        int cur_delta = core_delta, n_slides = 0, n_nodes = 0;
        memset(sc->heads, 0xff, sizeof(sc->heads));
        for (int core_ofs = 0;
             n_slides < max_slides && core_ofs <= (plen - core_len) && cur_delta > core_len / 3 && n_nodes < MAX_IDENT_NODES;
             n_slides++, core_ofs += cur_delta) {
            if ((core_ofs + core_len + cur_delta) > plen) cur_delta = plen - (core_ofs + core_len);
            int64_t targ_idx = ora_locate_first_exact(s, probe + core_ofs, core_len, 0, sfx_len - 1, ctr);
            if (targ_idx == 0) continue;
            targ_idx -= 1;
            int iter_cnt = 0, first_iter = 1;
            uint32_t num_copies = 0;
            while (!max_iter || iter_cnt < max_iter) {
                if (n_nodes >= MAX_IDENT_NODES) break;
                if (!first_iter) {
                    if ((targ_idx + 1) >= sfx_len || (sa_at(s, targ_idx + 1) + core_len) > sfx_len) break;
                    if (iter_cnt == 100 && !num_copies) {
                        int64_t last = ora_locate_last_exact(s, probe + core_ofs, core_len, targ_idx - 1, sfx_len - 1, ctr);
                        num_copies = last > 0 ? (uint32_t)(1 + last - targ_idx) : 0;
                        if (max_iter && num_copies > (uint32_t)max_iter) break;
                    }
                    if (cmp_probe(probe + core_ofs, s->seq + sa_at(s, targ_idx + 1), core_len) != 0) break;
                    targ_idx += 1;
                }
                first_iter = 0;
                if (ctr) ctr->n_cand_seen++;
                int64_t loci = sa_at(s, targ_idx);
                if (loci < (int64_t)(uint32_t)core_ofs) continue;                      /* :6809 */
                int64_t left = loci - core_ofs;
                if (!plen || ((uint64_t)left + (uint32_t)plen) > s->concat_len) continue;  /* :6816 */
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
                const uint8_t *t = s->seq + left;
                int mm = 0, i;
                for (i = 0; i < plen; i++) {
                    uint8_t tb = t[i] & 0x0f, pb = probe[i] & 0x0f;
                    if (tb == B_EOS) break;
                    if (pb == tb) continue;
                    if (++mm > max_tot_mm) break;
                }
                if (i != plen) continue;
                int at = -1;
                if (low_inst) {
                    if (low_inst == max_hits) sloughed = 1;
                    int k;
                    for (k = 0; k < low_inst; k++)
                        if (hits[k].mismatches > mm) {
                            at = k;
                            if (k + 1 < max_hits) memmove(&hits[k + 1], &hits[k], sizeof(hit_rec) * (size_t)(low_inst - k));
                            break;
                        }
                    if (k == low_inst && low_inst < max_hits) at = k;
                } else
                    at = 0;
                if (at >= 0) {
                    const ora_entry *ent = map_entry(s, (uint64_t)left);
                    if (ent == NULL) return -1;
                    hits[at].strand = (uint8_t)cur_strand;
                    hits[at].chrom_id = ent->entry_id;
                    hits[at].match_loci = (uint32_t)((uint64_t)left - ent->start_ofs);
                    hits[at].match_len = (uint16_t)plen;
                    hits[at].mismatches = (uint8_t)mm;
                    if (low_inst < max_hits) low_inst++;
                    else max_tot_mm = hits[low_inst - 1].mismatches;
                }
            }
            if (low_inst == max_hits && max_tot_mm == 0 && !sloughed) { align2strand = ALS_NONE; break; }
        }
        if (cur_strand == '+' && align2strand == ALS_BOTH) {
            revcomp(probe, plen);
            cur_strand = '-';
            align2strand = ALS_CRICK;
        } else
            align2strand = ALS_NONE;
    } while (!(low_inst == max_hits && max_tot_mm == 0 && !sloughed) && align2strand != ALS_NONE);
    if (cur_strand == '-') revcomp(probe, plen);
    return low_inst;
}

/* ---- microInDels (`-a`): LocateInDels SfxArrayV2.cpp:7348-7660, ExploreInDelMatchRight :8943-9168,
 * ExploreInDelMatchLeft :9172-9405 ------------------------------------------------------------- */
enum { MAX_MM_EXPLORE_INDEL = 7, MIN_INDEL_SEQ_LEN = 7, MAX_MICRO_INDEL_MM = 2, MAX_PUT_INDEL_OFSS = 80,
       BASE_SCORE = 500, MAX_SCORE = 1000, SCORE_MATCH = 3, SCORE_MISMATCH = 5, SCORE_INDEL_OPN = 20, SCORE_INDEL_EXTN = 1 };

typedef struct indel_hit {
    int score, is_indel, is_insert;
    uint64_t s0_loci, s1_loci;          /* suffix offsets until locate_indels makes them entry relative */
    int s0_len, s0_mm, s1_len, s1_mm, s1_read_ofs;
    uint32_t s0_chrom, s1_chrom;
    char strand;
} indel_hit;

static int base_mismatch(uint8_t pb, uint8_t tb) { return !(pb == tb && pb <= B_T); }

/* anchored at the 5' end of the probe: everything left of a mismatch position matches as is, the rest after
 * skipping 1..max_len probe bases (insertion in the read) or target bases (deletion from the read) */
static int explore_indel_right(char strand, int max_len, int max_mm, int plen, const uint8_t *probe, const ora_entry *ent,
                               int64_t targ_ofs, const uint8_t *targ, indel_hit *hit)
{
    int mm_ofs[MAX_PUT_INDEL_OFSS + 8];
    indel_hit ins, del;
    memset(hit, 0, sizeof(*hit));
    if (targ_ofs < (int64_t)ent->start_ofs || (targ_ofs + plen - 1) > (int64_t)ent->end_ofs) return 0;
    uint32_t targ_seq_len = (uint32_t)(ent->seq_len - (targ_ofs - (int64_t)ent->start_ofs));
    memset(&ins, 0, sizeof(ins));
    memset(&del, 0, sizeof(del));
    if (max_mm > MAX_PUT_INDEL_OFSS) max_mm = MAX_PUT_INDEL_OFSS;
    int n_mm = 0;
    int lim = max_mm > MAX_MM_EXPLORE_INDEL ? max_mm : MAX_MM_EXPLORE_INDEL;
    for (int i = 0; i < plen && n_mm <= lim; i++) {
        uint8_t pb = probe[i] & 7, tb = targ[i] & 7;
        if (tb > B_N || pb > B_N) return 0;
        if (!base_mismatch(pb, tb)) continue;
        mm_ofs[n_mm++] = i;
    }
    if (n_mm < MAX_MM_EXPLORE_INDEL || MIN_INDEL_SEQ_LEN > (plen - mm_ofs[0])) {
        if (n_mm > max_mm) return 0;
        hit->s0_len = plen; hit->s0_loci = (uint64_t)targ_ofs; hit->s0_mm = n_mm; hit->strand = strand;
        hit->score = BASE_SCORE + plen * SCORE_MATCH - n_mm * SCORE_MISMATCH;
        return 1;
    }
    int tot = max_mm < n_mm ? max_mm : n_mm;
    for (int k = 0; k <= tot && MIN_INDEL_SEQ_LEN < (plen - mm_ofs[k]); k++)
        for (int l = 1; l <= max_len; l++) {
            int score = BASE_SCORE + plen * SCORE_MATCH - ((l - 1) * SCORE_INDEL_EXTN + SCORE_INDEL_OPN);
            const uint8_t *pp = probe + mm_ofs[k] + l, *pt = targ + mm_ofs[k];
            uint32_t rest = (uint32_t)(plen - (mm_ofs[k] + l));
            if (rest < MIN_INDEL_SEQ_LEN) break;
            uint32_t trest = targ_seq_len - (uint32_t)mm_ofs[k];
            if (trest < rest) break;
            int imm = 0;
            uint32_t i;
            for (i = 0; i < rest && (k + imm) <= max_mm; i++, pt++, pp++) {
                uint8_t pb = *pp & 7, tb = *pt & 7;
                if (pb > B_N || tb > B_N) break;
                if (!base_mismatch(pb, tb)) continue;
                imm++;
                score -= SCORE_MISMATCH;
                if (rest < (uint32_t)(MIN_INDEL_SEQ_LEN * imm)) break;
            }
            if (i != rest) continue;
            if (score > ins.score) {
                memset(&ins, 0, sizeof(ins));
                ins.s0_len = mm_ofs[k]; ins.s0_loci = (uint64_t)targ_ofs; ins.s0_mm = k; ins.strand = strand;
                ins.s1_len = (int)rest; ins.s1_loci = ins.s0_loci + (uint64_t)ins.s0_len; ins.s1_mm = imm;
                ins.s1_read_ofs = ins.s0_len + l;
                ins.score = score; ins.is_indel = 1; ins.is_insert = 1;
            }
        }
    for (int k = 0; k <= tot && MIN_INDEL_SEQ_LEN < (plen - mm_ofs[k]); k++)
        for (int l = 1; l <= max_len; l++) {
            int score = BASE_SCORE + plen * SCORE_MATCH - ((l - 1) * SCORE_INDEL_EXTN + SCORE_INDEL_OPN);
            const uint8_t *pp = probe + mm_ofs[k], *pt = targ + mm_ofs[k] + l;
            uint32_t rest = (uint32_t)(plen - mm_ofs[k]);
            if (rest < MIN_INDEL_SEQ_LEN) break;
            uint32_t trest = targ_seq_len - (uint32_t)(mm_ofs[k] + l);
            if (trest < rest) break;
            int imm = 0;
            uint32_t i;
            for (i = 0; i < rest && (k + imm) <= max_mm; i++, pt++, pp++) {
                uint8_t pb = *pp & 7, tb = *pt & 7;
                if (pb > B_N || tb > B_N) break;
                if (!base_mismatch(pb, tb)) continue;
                imm++;
                score -= SCORE_MISMATCH;
                if (rest < (uint32_t)(MIN_INDEL_SEQ_LEN * imm)) break;
            }
            if (i != rest) continue;
            if (score > del.score) {
                memset(&del, 0, sizeof(del));
                del.s0_len = mm_ofs[k]; del.s0_loci = (uint64_t)targ_ofs; del.s0_mm = k; del.strand = strand;
                del.s1_len = (int)rest; del.s1_loci = (uint64_t)targ_ofs + (uint64_t)del.s0_len + (uint64_t)l; del.s1_mm = imm;
                del.s1_read_ofs = del.s0_len;
                del.score = score; del.is_indel = 1; del.is_insert = 0;
            }
        }
    if (del.score == 0 && ins.score == 0) return 0;
    if (del.score > ins.score) { *hit = del; return 3; }
    *hit = ins;
    return 2;
}

/* anchored at the 3' end of the probe, scanning right to left */
static int explore_indel_left(char strand, int max_len, int max_mm, int plen, const uint8_t *probe, const ora_entry *ent,
                              int64_t targ_ofs, const uint8_t *targ, indel_hit *hit)
{
    int mm_ofs[MAX_PUT_INDEL_OFSS + 8];
    indel_hit ins, del;
    memset(hit, 0, sizeof(*hit));
    if (targ_ofs < (int64_t)ent->start_ofs || (targ_ofs + plen - 1) > (int64_t)ent->end_ofs) return 0;
    memset(&ins, 0, sizeof(ins));
    memset(&del, 0, sizeof(del));
    if (max_mm > MAX_PUT_INDEL_OFSS) max_mm = MAX_PUT_INDEL_OFSS;
    int n_mm = 0;
    int lim = max_mm > MAX_MM_EXPLORE_INDEL ? max_mm : MAX_MM_EXPLORE_INDEL;
    for (int i = plen - 1; i >= 0 && n_mm <= lim; i--) {
        uint8_t pb = probe[i] & 7, tb = targ[i] & 7;
        if (tb > B_N || pb > B_N) return 0;
        if (!base_mismatch(pb, tb)) continue;
        mm_ofs[n_mm++] = i;
    }
    if (n_mm < MAX_MM_EXPLORE_INDEL || MIN_INDEL_SEQ_LEN > mm_ofs[0]) {
        if (n_mm > max_mm) return 0;
        hit->s0_len = plen; hit->s0_loci = (uint64_t)targ_ofs; hit->s0_mm = n_mm; hit->strand = strand;
        hit->score = BASE_SCORE + plen * SCORE_MATCH - n_mm * SCORE_MISMATCH;
        return 1;
    }
    int tot = max_mm < n_mm ? max_mm : n_mm;
    for (int k = 0; k <= tot && MIN_INDEL_SEQ_LEN < mm_ofs[k]; k++)
        for (int l = 1; l <= max_len; l++) {
            int score = BASE_SCORE + plen * SCORE_MATCH - ((l - 1) * SCORE_INDEL_EXTN + SCORE_INDEL_OPN);
            score -= k * SCORE_MISMATCH;
            if (score < ins.score) break;
            const uint8_t *pp = probe + mm_ofs[k] - l, *pt = targ + mm_ofs[k];
            uint32_t rest = (uint32_t)(mm_ofs[k] - (l - 1));
            if (rest < MIN_INDEL_SEQ_LEN) break;
            int imm = 0, i;
            for (i = 0; i < (int)rest && (k + imm) <= max_mm; i++, pt--, pp--) {
                uint8_t pb = *pp & 7, tb = *pt & 7;
                if (pb > B_N || tb > B_N) break;
                if (!base_mismatch(pb, tb)) continue;
                imm++;
                score -= SCORE_MISMATCH;
                if (rest < (uint32_t)(MIN_INDEL_SEQ_LEN * imm)) break;
            }
            if (i != (int)rest) continue;
            if (score > ins.score) {
                memset(&ins, 0, sizeof(ins));
                ins.s0_len = (int)rest; ins.s0_loci = (uint64_t)(uint32_t)(targ_ofs + l); ins.s0_mm = imm; ins.strand = strand;
                ins.s1_len = plen - ((int)rest + l); ins.s1_loci = ins.s0_loci + (uint64_t)ins.s0_len; ins.s1_mm = k;
                ins.s1_read_ofs = ins.s0_len + l;
                ins.score = score; ins.is_indel = 1; ins.is_insert = 1;
            }
        }
    for (int k = 0; k <= tot && MIN_INDEL_SEQ_LEN < mm_ofs[k]; k++)
        for (int l = 1; l <= max_len; l++) {
            int score = BASE_SCORE + plen * SCORE_MATCH - ((l - 1) * SCORE_INDEL_EXTN + SCORE_INDEL_OPN);
            score -= k * SCORE_MISMATCH;
            if (score < del.score) break;
            const uint8_t *pp = probe + mm_ofs[k], *pt = targ + mm_ofs[k] - l;
            uint32_t rest = (uint32_t)(mm_ofs[k] + 1);
            if (rest < MIN_INDEL_SEQ_LEN) break;
            if ((uint64_t)(uint32_t)l > (uint64_t)targ_ofs) break;
            int imm = 0, i;
            for (i = 0; i < (int)rest && (k + imm) <= max_mm; i++, pt--, pp--) {
                uint8_t pb = *pp & 7, tb = *pt & 7;
                if (pb > B_N || tb > B_N) break;
                if (!base_mismatch(pb, tb)) continue;
                imm++;
                score -= SCORE_MISMATCH;
                if (rest < (uint32_t)(MIN_INDEL_SEQ_LEN * imm)) break;
            }
            if (i != (int)rest) continue;
            if (score > del.score) {
                memset(&del, 0, sizeof(del));
                del.s0_len = (int)rest; del.s0_loci = (uint64_t)(uint32_t)(targ_ofs - l); del.s0_mm = imm; del.strand = strand;
                del.s1_len = plen - (int)rest; del.s1_loci = del.s0_loci + (uint64_t)rest + (uint64_t)l; del.s1_mm = k;
                del.s1_read_ofs = (int)rest;
                del.score = score; del.is_indel = 1; del.is_insert = 0;
            }
        }
    if (del.score == 0 && ins.score == 0) return 0;
    if (del.score > ins.score) { *hit = del; return 3; }
    *hit = ins;
    return 2;
}

/* LocateInDels with MaxHits = 1 (the only value AlignReads passes, :7731): two anchor cores per strand, the 5' one explored
 * rightwards and the 3' one leftwards; the best scoring placement wins, further placements with the same score at a
 * different Seg[0] start make the read ambiguous (returns eHRnone) */
static int locate_indels(const ora_sfx *s, int max_len, int max_tot_mm, int core_len, int align2strand,
                         int *p_low_inst, int *p_low_mm, int *p_nxt, uint8_t *probe, int plen, indel_hit *best,
                         int max_iter, ora_counters *ctr)
{
    int64_t sfx_len = (int64_t)s->concat_len;
    char cur_strand;
    int best_instances = 0;
    const int max_hits = 1;
    if (s->concat_len == 0) return -1;
    *p_low_inst = 0; *p_low_mm = 0; *p_nxt = 0;
    memset(best, 0, sizeof(*best));
    if (align2strand == ALS_CRICK) { revcomp(probe, plen); cur_strand = '-'; }
    else cur_strand = '+';
    do {
        for (int phase = 0; phase < 2; phase++) {
            int core_ofs = phase == 0 ? 0 : plen - core_len;
            int64_t targ_idx = ora_locate_first_exact(s, probe + core_ofs, core_len, 0, sfx_len - 1, ctr);
            if (targ_idx == 0) continue;
            targ_idx -= 1;
            int iter_cnt = 0, first_iter = 1;
            uint32_t num_copies = 0;
            while (!max_iter || iter_cnt < max_iter) {
                if (!first_iter) {
                    if ((targ_idx + 1) >= sfx_len || (sa_at(s, targ_idx + 1) + core_len) > sfx_len) break;
                    if (iter_cnt == 100 && !num_copies) {
                        int64_t last = ora_locate_last_exact(s, probe + core_ofs, core_len, targ_idx - 1, sfx_len - 1, ctr);
                        num_copies = last > 0 ? (uint32_t)(1 + last - targ_idx) : 0;
                        if (max_iter && num_copies > (uint32_t)max_iter) break;
                    }
                    if (cmp_probe(probe + core_ofs, s->seq + sa_at(s, targ_idx + 1), core_len) != 0) break;
                    targ_idx += 1;
                }
                first_iter = 0;
                int64_t loci = sa_at(s, targ_idx);
                if (loci < (int64_t)(uint32_t)core_ofs) continue;
                int64_t left = loci - core_ofs;
                const ora_entry *ent = map_entry(s, (uint64_t)loci);
                if (ent == NULL) continue;
                if (left < (int64_t)ent->start_ofs || (left + plen - 1) > (int64_t)ent->end_ofs) continue;
                if ((left + plen) > sfx_len) continue;
                iter_cnt++;
                indel_hit h;
                int r = phase == 0 ? explore_indel_right(cur_strand, max_len, max_tot_mm, plen, probe, ent, left, s->seq + left, &h)
                                   : explore_indel_left(cur_strand, max_len, max_tot_mm, plen, probe, ent, left, s->seq + left, &h);
                if (r > 0 && h.score >= best->score) {
                    if (h.score == best->score) {
                        if (best->s0_loci == h.s0_loci) continue;
                        if (++best_instances > max_hits) continue;
                    } else
                        best_instances = 0;
                    *best = h;                       /* pHits[BestScoreInstances++] with MaxHits 1: always slot 0 */
                    best_instances++;
                }
            }
            if (best_instances >= 1 && best->score >= MAX_SCORE) { align2strand = ALS_NONE; break; }
        }
        if (cur_strand == '+' && align2strand == ALS_BOTH) {
            revcomp(probe, plen);
            cur_strand = '-';
            align2strand = ALS_CRICK;
        } else
            align2strand = ALS_NONE;
    } while (!(best_instances >= 1 && best->score >= MAX_SCORE) && align2strand != ALS_NONE);
    if (cur_strand == '-') revcomp(probe, plen);
    if (best_instances == 0) return HR_NONE;
    if (best->score > MAX_SCORE) best->score = MAX_SCORE;
    {
        const ora_entry *e0 = map_entry(s, best->s0_loci), *e1 = map_entry(s, best->s1_loci);
        if (e0 == NULL || e1 == NULL) return HR_NONE;
        if (e0->entry_id != e1->entry_id) return HR_NONE;      /* also drops placements without a second segment outside the first entry */
        best->s0_chrom = e0->entry_id;
        best->s0_loci -= e0->start_ofs;
        if (best->s1_loci > 0) { best->s1_chrom = e1->entry_id; best->s1_loci -= e1->start_ofs; }
    }
    *p_low_inst = best_instances < max_hits ? best_instances : max_hits;
    *p_low_mm = best->s0_mm + best->s1_mm;
    *p_nxt = *p_low_mm + 2;
    return best_instances <= max_hits ? HR_HITS : HR_NONE;
}

/* ---- splice junctions (`-A`): LocateSpliceJuncts SfxArrayV2.cpp:7022-7345, ExploreSpliceRight :8437-8685,
 * ExploreSpliceLeft :8688-8940.  Restated as written, quirks included: the mismatch budget test of the segment compare is
 * strict (`<`), the rolling base-sum filter, the 35 bases probed for an EOS first.  One deliberate difference: with a
 * mismatch budget of 0 (-s0) the reference indexes its offset table at -1 (ExploreSpliceRight :8521); here such a call
 * reports nothing. */
enum { MIN_JUNCT_ALIGN_SEP = 25, MAX_JUNCT_ALIGN_MM = 2, MIN_JUNCT_SEG_LEN = 10, SPLICE_DONOR_ACCEPT = 50, SPLICE_LEN = 10 };

static int splice_bonus(char strand, uint8_t d0, uint8_t d1, uint8_t a0, uint8_t a1)
{
    int gt_ag = d0 == B_G && d1 == B_T && a0 == B_G && a1 == B_A, ct_ac = d0 == B_C && d1 == B_T && a0 == B_C && a1 == B_A;
    if (strand == '+') return gt_ag ? SPLICE_DONOR_ACCEPT : (ct_ac ? SPLICE_DONOR_ACCEPT / 2 : 0);
    return ct_ac ? SPLICE_DONOR_ACCEPT : (gt_ag ? SPLICE_DONOR_ACCEPT / 2 : 0);
}

static int explore_splice_right(char strand, int max_junct, int max_mm, int core_len, int plen, const uint8_t *probe, int64_t targ_ofs,
                                int64_t targ_len, const uint8_t *targ, indel_hit *hit)
{
    int mm_ofs[MAX_PUT_INDEL_OFSS + 8];
    indel_hit cur;
    uint8_t pb = 0, tb = 0;
    memset(hit, 0, sizeof(*hit));
    if ((targ_ofs + plen + MIN_JUNCT_ALIGN_SEP) > targ_len) return 0;
    memset(&cur, 0, sizeof(cur));
    if (max_mm > MAX_JUNCT_ALIGN_MM) max_mm = MAX_JUNCT_ALIGN_MM;
    int n_mm = 0;
    const int lim = max_mm > MAX_JUNCT_ALIGN_MM * 5 ? max_mm : MAX_JUNCT_ALIGN_MM * 5;
    for (int i = core_len; i < plen && n_mm <= lim; i++) {
        pb = probe[i] & 7; tb = targ[i] & 7;
        if (tb > B_N || pb > B_N) return 0;
        if (!base_mismatch(pb, tb)) continue;
        mm_ofs[n_mm++] = i;
    }
    if (n_mm < MAX_JUNCT_ALIGN_MM * 4 || MIN_JUNCT_SEG_LEN > (plen - mm_ofs[0])) {
        if (n_mm > max_mm) return 0;
        hit->s0_len = plen; hit->s0_loci = (uint64_t)targ_ofs; hit->s0_mm = n_mm; hit->strand = strand;
        hit->score = BASE_SCORE + plen * SCORE_MATCH - n_mm * SCORE_MISMATCH;
        return 1;
    }
    const int tot = n_mm < max_mm ? n_mm : max_mm;
    if (tot < 1) return 0;
    {
        const uint8_t *pt = targ + mm_ofs[tot - 1];
        for (int i = 0; i < MIN_JUNCT_ALIGN_SEP + MIN_JUNCT_SEG_LEN; i++, pt++)
            if ((*pt & 7) > B_N) return 0;
    }
    for (int k = 0; k <= tot && MIN_JUNCT_SEG_LEN < (plen - mm_ofs[k]); k++) {
        if (cur.score >= MAX_SCORE) break;
        const uint32_t seg_len = (uint32_t)(plen - mm_ofs[k]);
        const uint8_t *cur_p = probe + mm_ofs[k], *donor = targ + mm_ofs[k];
        const uint8_t *t_start = donor + MIN_JUNCT_ALIGN_SEP;
        const int hash_diff = 4 * (max_mm - k);
        int probe_hash = 100000;
        for (uint32_t i = 0; i < seg_len; i++) probe_hash += cur_p[i] & 7;
        const int min_hash = probe_hash - hash_diff, max_hash = probe_hash + hash_diff;
        int targ_hash = 100000;
        const uint8_t *t_end = t_start;
        uint32_t i;
        for (i = 0; i < seg_len - 1; i++) {
            if ((*t_end & 7) > B_N) break;
            targ_hash += *t_end++ & 7;
        }
        if (i < seg_len - 1) break;
        for (int gap = MIN_JUNCT_ALIGN_SEP; gap < max_junct - (int)seg_len; gap++, t_start++, t_end++) {
            if ((tb = *t_end & 7) > B_N) break;
            targ_hash += tb;
            if (targ_hash < min_hash || targ_hash > max_hash) { targ_hash -= (*t_start & 7); continue; }
            targ_hash -= (*t_start & 7);
            const uint8_t *pt = t_start, *pp = cur_p;
            const uint32_t tmp_len = (uint32_t)(targ_len - (targ_ofs + mm_ofs[k] + gap + 1));
            if (tmp_len < seg_len) break;
            int cmm = 0;
            for (i = 0; i < seg_len && (k + cmm) < max_mm; i++, pt++, pp++) {
                pb = *pp & 7; tb = *pt & 7;
                if (pb > B_N || tb > B_N) break;
                if (!base_mismatch(pb, tb)) continue;
                cmm++;
            }
            if (i != seg_len) {
                if (pb > B_N || tb > B_N) break;
                continue;
            }
            int score = BASE_SCORE + plen * SCORE_MATCH - ((k + cmm) * SCORE_MISMATCH + (gap / 1000) * SPLICE_LEN);
            score += splice_bonus(strand, donor[0] & 7, donor[1] & 7, t_start[-1] & 7, t_start[-2] & 7);
            if (score > cur.score) {
                memset(&cur, 0, sizeof(cur));
                cur.s0_len = mm_ofs[k]; cur.s0_loci = (uint64_t)targ_ofs; cur.s0_mm = k; cur.strand = strand;
                cur.s1_len = plen - mm_ofs[k]; cur.s1_loci = (uint64_t)(targ_ofs + mm_ofs[k] + gap); cur.s1_mm = cmm;
                cur.s1_read_ofs = mm_ofs[k]; cur.score = score; cur.is_indel = 2;      /* here: 2 = FlgSplice */
            }
        }
    }
    if (cur.score == 0) return 0;
    *hit = cur;
    return 3;
}

static int explore_splice_left(char strand, int max_junct, int max_mm, int core_len, int plen, const uint8_t *probe0, int64_t targ_ofs,
                               int64_t targ_len, const uint8_t *targ0, indel_hit *hit)
{
    int mm_ofs[MAX_PUT_INDEL_OFSS + 8];
    indel_hit cur;
    uint8_t pb = 0, tb = 0;
    (void)targ_len;
    memset(hit, 0, sizeof(*hit));
    if (targ_ofs < (int64_t)(MIN_JUNCT_ALIGN_SEP + MIN_JUNCT_SEG_LEN)) return 0;
    memset(&cur, 0, sizeof(cur));
    if (max_mm > MAX_JUNCT_ALIGN_MM) max_mm = MAX_JUNCT_ALIGN_MM;
    const uint8_t *probe = probe0 + plen - 1, *targ = targ0 + plen - 1;          /* last base; offsets below count leftwards */
    int n_mm = 0;
    const int lim = max_mm > MAX_JUNCT_ALIGN_MM * 5 ? max_mm : MAX_JUNCT_ALIGN_MM * 5;
    for (int i = core_len; i < plen && n_mm <= lim; i++) {
        pb = probe[-i] & 7; tb = targ[-i] & 7;
        if (tb > B_N || pb > B_N) return 0;
        if (!base_mismatch(pb, tb)) continue;
        mm_ofs[n_mm++] = i;
    }
    if (n_mm < MAX_JUNCT_ALIGN_MM * 4 || MIN_JUNCT_SEG_LEN > (plen - mm_ofs[0])) {
        if (n_mm > max_mm) return 0;
        hit->s0_len = plen; hit->s0_loci = (uint64_t)targ_ofs; hit->s0_mm = n_mm; hit->strand = strand;
        hit->score = BASE_SCORE + plen * SCORE_MATCH - n_mm * SCORE_MISMATCH;
        return 1;
    }
    const int tot = n_mm < max_mm ? n_mm : max_mm;
    {
        const uint8_t *pt = targ - mm_ofs[tot];
        for (int i = 0; i < MIN_JUNCT_ALIGN_SEP + MIN_JUNCT_SEG_LEN; i++, pt--)
            if ((*pt & 7) > B_N) return 0;
    }
    for (int k = 0; k <= tot && MIN_JUNCT_SEG_LEN < (plen - mm_ofs[k]); k++) {
        if (cur.score >= MAX_SCORE) break;
        const uint32_t seg_len = (uint32_t)(plen - mm_ofs[k]);
        const uint8_t *cur_p = probe - mm_ofs[k], *donor = targ - mm_ofs[k];
        const uint8_t *t_start = donor - MIN_JUNCT_ALIGN_SEP;
        const int hash_diff = 4 * (max_mm - k);
        int probe_hash = 100000;
        for (uint32_t i = 0; i < seg_len; i++) probe_hash += *(cur_p - i) & 7;
        const int min_hash = probe_hash - hash_diff, max_hash = probe_hash + hash_diff;
        int targ_hash = 100000;
        const uint8_t *t_end = t_start;
        uint32_t i;
        for (i = 0; i < seg_len - 1; i++) {
            if ((*t_end & 7) > B_N) break;
            targ_hash += *t_end-- & 7;
        }
        if (i < seg_len - 1) break;
        for (int gap = MIN_JUNCT_ALIGN_SEP; gap < max_junct - (int)seg_len; gap++, t_start--, t_end--) {
            if ((tb = *t_end & 7) > B_N) break;
            targ_hash += tb;
            if (targ_hash < min_hash || targ_hash > max_hash) { targ_hash -= (*t_start & 7); continue; }
            targ_hash -= (*t_start & 7);
            const uint8_t *pt = t_start, *pp = cur_p;
            const uint32_t tmp_len = (uint32_t)(targ_ofs - gap);
            if (tmp_len < 1) break;
            int cmm = 0;
            for (i = 0; i < seg_len && (k + cmm) < max_mm; i++, pt--, pp--) {
                pb = *pp & 7; tb = *pt & 7;
                if (pb > B_N || tb > B_N) break;
                if (!base_mismatch(pb, tb)) continue;
                cmm++;
            }
            if (i != seg_len) {
                if (pb > B_N || tb > B_N) break;
                continue;
            }
            int score = BASE_SCORE + plen * SCORE_MATCH - ((k + cmm) * SCORE_MISMATCH + (gap / 1000) * SPLICE_LEN);
            score += splice_bonus(strand, t_start[1] & 7, t_start[2] & 7, donor[0] & 7, donor[-1] & 7);
            if (score > cur.score) {
                memset(&cur, 0, sizeof(cur));
                cur.s0_len = plen - mm_ofs[k]; cur.s0_loci = (uint64_t)(targ_ofs - gap); cur.s0_mm = cmm; cur.strand = strand;
                cur.s1_len = mm_ofs[k]; cur.s1_loci = cur.s0_loci + (uint64_t)cur.s0_len + (uint64_t)gap; cur.s1_mm = k;
                cur.s1_read_ofs = cur.s0_len; cur.score = score; cur.is_indel = 2;
            }
        }
    }
    if (cur.score == 0) return 0;
    *hit = cur;
    return 3;
}

static int locate_splice_juncts(const ora_sfx *s, int max_junct, int max_tot_mm, int core_len, int align2strand,
                                int *p_low_inst, int *p_low_mm, int *p_nxt, uint8_t *probe, int plen, indel_hit *best,
                                int max_iter, ora_counters *ctr)
{
    int64_t sfx_len = (int64_t)s->concat_len;
    char cur_strand;
    int best_instances = 0;
    const int max_hits = 1;
    if (s->concat_len == 0) return -1;
    if (max_tot_mm > MAX_JUNCT_ALIGN_MM) max_tot_mm = MAX_JUNCT_ALIGN_MM;
    *p_low_inst = 0; *p_low_mm = 0; *p_nxt = 0;
    memset(best, 0, sizeof(*best));
    if (align2strand == ALS_CRICK) { revcomp(probe, plen); cur_strand = '-'; }
    else cur_strand = '+';
    do {
        for (int phase = 0; phase < 2; phase++) {
            int core_ofs = phase == 0 ? 0 : plen - core_len;
            int64_t targ_idx = ora_locate_first_exact(s, probe + core_ofs, core_len, 0, sfx_len - 1, ctr);
            if (targ_idx == 0) continue;
            targ_idx -= 1;
            int iter_cnt = 0, first_iter = 1;
            uint32_t num_copies = 0;
            while (!max_iter || iter_cnt < max_iter) {
                if (!first_iter) {
                    if ((targ_idx + 1) >= sfx_len || (sa_at(s, targ_idx + 1) + (phase == 0 ? plen : core_len)) >= sfx_len) break;
                    if (iter_cnt == 100 && !num_copies) {
                        int64_t last = ora_locate_last_exact(s, probe + core_ofs, core_len, targ_idx - 1, sfx_len - 1, ctr);
                        num_copies = last > 0 ? (uint32_t)(1 + last - targ_idx) : 0;
                        if (max_iter && num_copies > (uint32_t)max_iter) break;
                    }
                    if (cmp_probe(probe + core_ofs, s->seq + sa_at(s, targ_idx + 1), core_len) != 0) break;
                    targ_idx += 1;
                }
                first_iter = 0;
                int64_t loci = sa_at(s, targ_idx);
                if (loci < (int64_t)(uint32_t)core_ofs) continue;
                int64_t left = loci - core_ofs;
                if ((left + plen) >= sfx_len) continue;
                const ora_entry *ent = map_entry(s, (uint64_t)loci);
                if (ent == NULL) continue;
                if (left < (int64_t)ent->start_ofs || (left + plen) > (int64_t)ent->end_ofs) continue;
                iter_cnt++;
                indel_hit h;
                int r = 0;
                if (phase == 0) {
                    int limit = (int)(sfx_len - left);
                    if (limit > (MIN_JUNCT_ALIGN_SEP + MIN_JUNCT_SEG_LEN)) {
                        limit -= (MIN_JUNCT_ALIGN_SEP + MIN_JUNCT_SEG_LEN);
                        if (limit > max_junct) limit = max_junct;
                        r = explore_splice_right(cur_strand, limit, max_tot_mm, core_len, plen, probe, left, sfx_len, s->seq + left, &h);
                    }
                } else if ((uint64_t)left >= (uint64_t)(uint32_t)(core_ofs + MIN_JUNCT_SEG_LEN)) {
                    int limit = (int)((uint64_t)left < (uint64_t)(uint32_t)max_junct ? (uint64_t)left : (uint64_t)(uint32_t)max_junct);
                    if (limit >= (MIN_JUNCT_ALIGN_SEP + MIN_JUNCT_SEG_LEN)) {
                        limit -= MIN_JUNCT_SEG_LEN;
                        r = explore_splice_left(cur_strand, limit, max_tot_mm, core_len, plen, probe, left, sfx_len, s->seq + left, &h);
                    }
                }
                if (r > 0 && h.score >= best->score) {
                    if (h.score == best->score) {
                        if (best->s0_loci == h.s0_loci) continue;
                        if (++best_instances > max_hits) continue;
                    } else
                        best_instances = 0;
                    *best = h;
                    best_instances++;
                }
            }
            if (best_instances >= 1 && best->score >= MAX_SCORE) { align2strand = ALS_NONE; break; }
        }
        if (cur_strand == '+' && align2strand == ALS_BOTH) {
            revcomp(probe, plen);
            cur_strand = '-';
            align2strand = ALS_CRICK;
        } else
            align2strand = ALS_NONE;
    } while (!(best_instances >= 1 && best->score >= MAX_SCORE) && align2strand != ALS_NONE);
    if (cur_strand == '-') revcomp(probe, plen);
    if (best_instances == 0) return HR_NONE;
    if (best->score > MAX_SCORE) best->score = MAX_SCORE;
    {
        const ora_entry *e0 = map_entry(s, best->s0_loci);
        if (e0 == NULL) return HR_NONE;
        best->s0_chrom = e0->entry_id;
        best->s0_loci -= e0->start_ofs;
        if (best->s1_loci > 0) {
            const ora_entry *e1 = map_entry(s, best->s1_loci);
            if (e1 == NULL) return HR_NONE;
            best->s1_chrom = e1->entry_id;
            best->s1_loci -= e1->start_ofs;
        }
    }
    *p_low_inst = best_instances;
    *p_low_mm = best->s0_mm + best->s1_mm;
    *p_nxt = *p_low_mm + 2;
    return best_instances <= max_hits ? HR_HITS : HR_NONE;
}

/* AlignReads, SfxArrayV2.cpp:7666-7760 with microInDelLen = MaxSpliceJunctLen = MinChimericLen = 0 */
static int align_reads(const ora_sfx *s, int max_tot_mm, int core_len, int core_delta, int max_slides,
                       int mm_delta, int align2strand, int *p_low_inst, int *p_low_mm, int *p_nxt,
                       uint8_t *probe, int plen, int max_hits, hit_rec *hits, int max_iter,
                       scratch *sc, ora_counters *ctr, int micro_indel_len, int splice_junct_len, indel_hit *ih, int *got_indel,
                       int min_chimeric_pct, int min_core_len)
{
    int rslt = 0, allow_mm;
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
    if (rslt == 0 && micro_indel_len > 0) {                                            /* :7722-7734 */
        int splice_core = core_len * 2 < (plen - 1) / 2 ? core_len * 2 : (plen - 1) / 2;
        rslt = locate_indels(s, micro_indel_len, max_tot_mm > MAX_MICRO_INDEL_MM ? MAX_MICRO_INDEL_MM : max_tot_mm, splice_core,
                             align2strand, p_low_inst, p_low_mm, p_nxt, probe, plen, ih, max_iter, ctr);
        if (rslt != 0) { *got_indel = 1; return rslt; }
    }
    if (rslt == 0 && splice_junct_len > 0) {                                           /* :7736-7748 */
        int splice_core = core_len * 2 < (plen - 1) / 2 ? core_len * 2 : (plen - 1) / 2;
        rslt = locate_splice_juncts(s, splice_junct_len, max_tot_mm > MAX_JUNCT_ALIGN_MM ? MAX_JUNCT_ALIGN_MM : max_tot_mm, splice_core,
                                    align2strand, p_low_inst, p_low_mm, p_nxt, probe, plen, ih, max_iter, ctr);
        if (rslt != 0) { *got_indel = 1; return rslt; }
    }
    if (min_chimeric_pct > 0) {                                                        /* :7750-7757 */
        int cl = min_core_len > plen / (max_tot_mm + 4) ? min_core_len : plen / (max_tot_mm + 4);
        int cd = max_slides > 1 ? plen / (max_slides - 1) : plen;
        if (cd < cl) cd = cl;
        return locate_core_multiples_c(s, min_chimeric_pct, max_tot_mm, cl, cd, max_slides, mm_delta, align2strand, p_low_inst, p_low_mm, p_nxt,
                                       probe, plen, max_hits, hits, max_iter, sc, ctr);
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

/* loci_out (optional): room for p->max_ml entries - the pHits[] contents of a read whose AlignReads returned
 * eHRhits, in the order LocateCoreMultiples left them (the multi-loci modes -r1..-r5 consume these) */
static int align_read_sc(const ora_sfx *s, const ora_params *p, const uint8_t *bases, int len,
                         ora_hit *out, ora_counters *ctr, scratch *sc, uint8_t *seqbuf, ora_loci *loci_out, ora_seg2 *seg2_out, ora_trims *trims_out)
{
    if (seg2_out) memset(seg2_out, 0, sizeof(*seg2_out));
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
    hit_rec hits_small[8];
    hit_rec *hits = hits_small;
    if (max_ml > 7) hits = (hit_rec *)malloc(sizeof(hit_rec) * ((size_t)max_ml + 1));
    memset(hits, 0, sizeof(hit_rec) * ((size_t)max_ml + 1));
    int rslt, got_indel = 0;
    indel_hit ih;
    memset(&ih, 0, sizeof(ih));
    if (p->best_matches) {                                                             /* Aligner.cpp:9197-9218 */
        rslt = locate_best_matches(s, max_tot_mm, core_len, core_delta, max_slides, align2strand, seqbuf, match_len, max_ml, hits,
                                   ora_max_iter(p->pmode), sc, ctr);
        if (rslt >= 0) { low_inst = rslt; rslt = rslt ? HR_HITS : HR_NONE; }           /* LowMMCnt / NxtLowMMCnt stay 0 */
    } else
        rslt = align_reads(s, max_tot_mm, core_len, core_delta, max_slides, p->min_edit_dist, align2strand,
                           &low_inst, &low_mm, &nxt, seqbuf, match_len, max_ml, hits,
                           ora_max_iter(p->pmode), sc, ctr, p->micro_indel_len, p->splice_junct_len, &ih, &got_indel,
                           p->min_chimeric_len, min_core);
    if (got_indel && rslt == HR_HITS) {            /* the one tsHitLoci LocateInDels returned -> pMultiHits[0] */
        hits[0].strand = (uint8_t)ih.strand; hits[0].chrom_id = ih.s0_chrom; hits[0].match_loci = (uint32_t)ih.s0_loci;
        hits[0].match_len = (uint16_t)ih.s0_len; hits[0].mismatches = (uint8_t)ih.s0_mm;
        if (seg2_out) {
            seg2_out->match_loci = (uint32_t)ih.s1_loci; seg2_out->match_len = (uint16_t)ih.s1_len;
            seg2_out->read_ofs = (uint16_t)ih.s1_read_ofs; seg2_out->mismatches = (uint8_t)ih.s1_mm;
            seg2_out->flags = (uint8_t)((ih.is_indel == 1 ? 1 : 0) | (ih.is_insert ? 2 : 0) | (ih.is_indel == 2 ? 4 : 0));
            seg2_out->score = (uint16_t)ih.score;
        }
    }
    if (rslt < 0) { if (hits != hits_small) free(hits); return rslt; }
    if (!got_indel && rslt == HR_HITS && low_inst == 1 && hits[0].chimeric && seg2_out) {   /* trims of a unique chimeric placement travel in the seg2 record */
        seg2_out->flags = 8; seg2_out->match_len = hits[0].trim_left; seg2_out->read_ofs = hits[0].trim_right;
    }
    if (low_inst > max_ml) low_inst = max_ml + 1;                                       /* :9241 */
    out->rslt = (uint8_t)rslt;
    if (loci_out && (rslt == HR_HITS || (rslt == HR_HITINSTS && p->clamp_ml)))
        for (int k = 0; k < low_inst && k < max_ml; k++) {
            loci_out[k].chrom_id = hits[k].chrom_id; loci_out[k].match_loci = hits[k].match_loci;
            loci_out[k].match_len = hits[k].match_len; loci_out[k].strand = hits[k].strand;
            loci_out[k].mismatches = hits[k].mismatches;
            if (trims_out) {                       /* chimeric placements (FlgChimeric) carry their end trims, every locus its own */
                trims_out[k].left = hits[k].chimeric ? hits[k].trim_left : 0;
                trims_out[k].right = hits[k].chimeric ? hits[k].trim_right : 0;
                trims_out[k].chimeric = hits[k].chimeric ? 1 : 0;
                trims_out[k].reserved = 0;
            }
        }

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
    if (hits != hits_small) free(hits);
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
    int r = align_read_sc(s, p, bases, len, out, ctr, sc, buf, NULL, NULL, NULL);
    free(buf);
    scratch_free(sc);
    return r;
}

typedef struct worker {
    const ora_sfx *s; const ora_params *p; const uint8_t *bases; const uint64_t *offs;
    const uint32_t *lens; uint32_t lo, hi; ora_hit *out; ora_counters ctr; int rslt; ora_loci *loci; ora_seg2 *seg2; ora_trims *trims;
} worker;

static void *worker_main(void *arg)
{
    worker *w = (worker *)arg;
    scratch *sc = scratch_new();
    uint32_t maxlen = 0;
    for (uint32_t i = w->lo; i < w->hi; i++) if (w->lens[i] > maxlen) maxlen = w->lens[i];
    uint8_t *buf = (uint8_t *)malloc((size_t)maxlen + 16);
    for (uint32_t i = w->lo; i < w->hi; i++) {
        ora_loci *lo = w->loci ? w->loci + (size_t)i * (size_t)(w->p->max_ml > 0 ? w->p->max_ml : 1) : NULL;
        ora_trims *tr = w->trims ? w->trims + (size_t)i * (size_t)(w->p->max_ml > 0 ? w->p->max_ml : 1) : NULL;
        int r = align_read_sc(w->s, w->p, w->bases + w->offs[i], (int)w->lens[i], &w->out[i], &w->ctr, sc, buf, lo,
                              w->seg2 ? w->seg2 + i : NULL, tr);
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
    return ora_align_batch_ex(s, p, bases, offs, lens, nreads, out, NULL, NULL, ctr, nthreads);
}

int ora_align_batch_multi(const ora_sfx *s, const ora_params *p, const uint8_t *bases,
                          const uint64_t *offs, const uint32_t *lens, uint32_t nreads,
                          ora_hit *out, ora_loci *loci, ora_counters *ctr, int nthreads)
{
    return ora_align_batch_ex(s, p, bases, offs, lens, nreads, out, loci, NULL, ctr, nthreads);
}

int ora_align_batch_ex(const ora_sfx *s, const ora_params *p, const uint8_t *bases,
                       const uint64_t *offs, const uint32_t *lens, uint32_t nreads,
                       ora_hit *out, ora_loci *loci, ora_seg2 *seg2, ora_counters *ctr, int nthreads)
{
    return ora_align_batch_ex2(s, p, bases, offs, lens, nreads, out, loci, seg2, NULL, ctr, nthreads);
}

int ora_align_batch_ex2(const ora_sfx *s, const ora_params *p, const uint8_t *bases,
                        const uint64_t *offs, const uint32_t *lens, uint32_t nreads,
                        ora_hit *out, ora_loci *loci, ora_seg2 *seg2, ora_trims *trims, ora_counters *ctr, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if ((uint32_t)nthreads > nreads) nthreads = nreads ? (int)nreads : 1;
    worker *w = (worker *)calloc((size_t)nthreads, sizeof(worker));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    /* contiguous blocks of <= 4096 reads are what the reference threads pull (Aligner.cpp:9636);
     * reads are independent so a static split gives the same answers */
    uint64_t per = ((uint64_t)nreads + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; t++) {
        w[t].s = s; w[t].p = p; w[t].bases = bases; w[t].offs = offs; w[t].lens = lens; w[t].out = out; w[t].loci = loci; w[t].seg2 = seg2; w[t].trims = trims;
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

/* =========================================================================================== */
/* Paired-end association: restatement of CAligner::ProcessPairedEnds (biokanga/Aligner.cpp:
 * 3055-3489), AcceptProvPE / PEInsertSize (:2726-2850), CSfxArrayV3::AlignPairedRead
 * (libbiokanga/SfxArrayV2.cpp:8247-8433), AdaptiveTrim (:5482-5682), IterateExactsRange (:3382-3474).
 * Chromosome filters, circularised PE and chimeric trimming are off (their defaults). */

enum { NAR_CHROMFILT = 11, NAR_PEINSERTMIN = 13, NAR_PEINSERTMAX = 14, NAR_PENOHIT = 15, NAR_PESTRAND = 16,
       NAR_PECHROM = 17, NAR_PEUNALIGN = 18 };
enum { PE_ORPHAN = 1, PE_UNIQUE = 2, PE_ORPHAN_SE = 3, PE_UNIQUE_SE = 4 };

/* PEInsertSize, Aligner.cpp:2802-2850 */
static int pe_insert_size(int min_len, int max_len, int pair_strand, uint8_t s1, uint32_t st1, uint32_t en1,
                          uint8_t s2, uint32_t st2, uint32_t en2)
{
    int frag;
    if ((pair_strand && s1 != s2) || (!pair_strand && s1 == s2)) return -1;
    if (s1 == '+') frag = 1 + (int)en2 - (int)st1;
    else frag = 1 + (int)en1 - (int)st2;
    if (frag < 0) return -1;
    if (frag < min_len) return -6;
    if (frag > max_len) return -7;
    return frag;
}

typedef struct at_region { uint16_t ofs, len; uint8_t mm, trim5, trim3; } at_region;

/* AdaptiveTrim, SfxArrayV2.cpp:5482-5682 */
static int adaptive_trim_ex(uint32_t seq_len, const uint8_t *probe, const uint8_t *targ, uint32_t min_trim_len, uint32_t max_mm,
                            uint32_t min_flank, uint32_t *p_trim_mms, uint32_t *p_trim_start, uint32_t *p_trim_end);
static int adaptive_trim_ex(uint32_t seq_len, const uint8_t *probe, const uint8_t *targ, uint32_t min_trim_len, uint32_t max_mm,
                            uint32_t min_flank, uint32_t *p_trim_mms, uint32_t *p_trim_start, uint32_t *p_trim_end)
{
    static __thread at_region regs[2048];
    uint32_t best_start = 0, best_end = 0;
    *p_trim_mms = 0; *p_trim_start = 0; *p_trim_end = 0;
    if (seq_len < 25 || seq_len > 2048 || min_trim_len < 15 || min_trim_len > seq_len || max_mm > 15 || min_flank > 10)
        return -100;
    if (min_flank == 0) min_flank = 1;
    uint32_t n_regs = 0, n_min_exact = 0;
    at_region *cur = NULL;
    for (uint32_t o = 0; o < seq_len; o++) {
        int mm = (probe[o] & 0x0f) != (targ[o] & 0x0f);
        if (cur == NULL || mm != cur->mm) {
            cur = &regs[n_regs++];
            cur->mm = (uint8_t)mm; cur->len = 1; cur->ofs = (uint16_t)o;
        } else {
            cur->len++;
            if (cur->len == 8 && !mm) n_min_exact++;
        }
    }
    if (!n_min_exact) return 0;
    uint32_t first_start = 0, last_start = 0, first_end = 0, last_end = 0;
    for (uint32_t i = 0; i < n_regs; i++) {
        at_region *r = &regs[i];
        r->trim5 = r->trim3 = 0;
        if (r->mm == 0 && r->len >= min_flank) {
            if (r->ofs <= seq_len - min_trim_len) { r->trim5 = 1; last_start = i + 1; if (!first_start) first_start = last_start; }
            if ((uint32_t)r->ofs + r->len >= (uint16_t)min_trim_len) { r->trim3 = 1; last_end = i + 1; if (!first_end) first_end = last_end; }
        }
    }
    if (!first_start || !first_end) return 0;
    uint32_t best_len = 0, best_mm = 0;
    for (uint32_t si = first_start - 1; si < last_start; si++) {
        at_region *sr = &regs[si];
        if (!sr->trim5) continue;
        uint32_t cur_len = 0, cur_mm = 0, idx = si;
        at_region *r = sr;
        while (idx++ < last_end) {
            cur_len += r->len;
            if (r->mm) {
                if (max_mm == 0) break;
                cur_mm += r->len;
                if ((max_mm + 1.0) / 100.0 <= (double)cur_mm / (seq_len - sr->ofs)) break;
            } else if (best_len == 0) {
                best_start = sr->ofs; best_end = seq_len - (best_start + cur_len);
                best_len = cur_len; best_mm = 0;
                r++;
                continue;
            }
            if (cur_len < min_trim_len || !r->trim3) { r++; continue; }
            r++;
            if ((max_mm + 1.0) / 100.0 <= (double)cur_mm / cur_len) continue;
            if (best_len < cur_len || (best_len == cur_len && (best_mm == 0 || cur_mm < best_mm))) {
                best_start = sr->ofs; best_end = seq_len - (best_start + cur_len);
                best_len = cur_len; best_mm = cur_mm;
            }
        }
    }
    if (best_len >= min_trim_len) { *p_trim_mms = best_mm; *p_trim_start = best_start; *p_trim_end = best_end; return (int)best_len; }
    return 0;
}

/* AlignPairedRead, SfxArrayV2.cpp:8247-8433 (MinChimericLen = 0) */
static int align_paired_read(const ora_sfx *s, int b3prime, int antisense, uint32_t chrom_id, uint32_t start_loci, uint32_t end_loci,
                             int min_dist, int max_dist, int max_allowed_mm, int read_len, int min_chimeric_pct, int core_len, int core_delta,
                             const uint8_t *read, hit_rec *out)
{
    if (min_dist < read_len || min_dist > max_dist) return 0;
    if (chrom_id < 1 || chrom_id > s->num_entries) return 0;
    const ora_entry *ent = &s->entries[chrom_id - 1];
    uint32_t targ_len = ent->seq_len;
    if (!targ_len) return 0;
    int targ_loci;
    if (b3prime) { targ_loci = (int)start_loci; if ((uint32_t)(targ_loci + min_dist) > targ_len) return 0; }
    else { targ_loci = (int)end_loci; if (targ_loci < min_dist || (uint32_t)targ_loci >= targ_len) return 0; }
    const uint8_t *chrom = s->seq + ent->start_ofs;
    uint8_t rs[2100];
    memcpy(rs, read, (size_t)read_len);
    rs[read_len] = B_EOS;
    if (antisense) revcomp(rs, read_len);
    /* MinChimericLen > 0: the partner may be end-trimmed down to that percentage of its length (:8327-8330) */
    uint32_t min_put_len = min_chimeric_pct > 0 ? (uint32_t)((read_len * min_chimeric_pct + 50) / 100) : (uint32_t)read_len, start_put, end_put;
    uint32_t t5, t3;
    if (b3prime) {
        start_put = (uint32_t)(targ_loci + min_dist);
        if (start_put + min_put_len >= targ_len) return 0;
        end_put = (uint32_t)(targ_loci + max_dist);
    } else {
        start_put = end_loci < (uint32_t)max_dist ? 0 : end_loci - (uint32_t)max_dist;
        end_put = end_loci - (uint32_t)min_dist;
    }
    uint32_t prev_best = (uint32_t)max_allowed_mm + 1, mms;
    memset(out, 0, sizeof(*out));
    if (end_put - start_put >= 1000) {
        for (uint32_t core_ofs = 0; (int)core_ofs + core_len <= read_len; core_ofs += (uint32_t)core_delta) {
            /* IterateExactsRange: every suffix matching the core, in SA order, restricted to the window */
            int64_t idx = ora_locate_first_exact(s, rs + core_ofs, core_len, 0, (int64_t)s->concat_len - 1, NULL);
            if (idx == 0) continue;
            for (idx -= 1; idx < (int64_t)s->concat_len; idx++) {
                int64_t pos = sa_at(s, idx);
                if (cmp_probe(rs + core_ofs, s->seq + pos, core_len) != 0) break;
                const ora_entry *e = map_entry(s, (uint64_t)pos);
                if (e == NULL || e->entry_id != chrom_id) continue;
                uint32_t hit = (uint32_t)((uint64_t)pos - e->start_ofs);
                if (hit < start_put || hit > end_put) continue;
                if (core_ofs > hit || (hit + (uint32_t)read_len - core_ofs) >= targ_len) continue;
                int r = adaptive_trim_ex((uint32_t)read_len, rs, chrom + (hit - core_ofs), min_put_len, (uint32_t)max_allowed_mm, 3, &mms, &t5, &t3);
                if (r > (int)min_put_len || (r == (int)min_put_len && mms < prev_best)) {
                    prev_best = mms; min_put_len = (uint32_t)r;
                    out->strand = antisense ? '-' : '+'; out->chrom_id = chrom_id; out->match_loci = hit - core_ofs;
                    out->match_len = (uint16_t)read_len; out->mismatches = (uint8_t)mms;
                    out->chimeric = (int)min_put_len == read_len ? 0 : 1;
                    out->trim_left = (uint16_t)(antisense ? t3 : t5); out->trim_right = (uint16_t)(antisense ? t5 : t3);
                }
            }
        }
    } else {
        for (uint32_t hit = start_put; hit <= end_put; hit++) {
            const uint8_t *win = chrom + hit;
            uint8_t tmpw[2100];
            if (ent->start_ofs + hit + (uint64_t)read_len > s->concat_len) {      /* window runs off the concatenation: */
                for (int k = 0; k < read_len; k++) {                              /* (the reference reads whatever follows) */
                    uint64_t q = ent->start_ofs + hit + (uint64_t)k;
                    tmpw[k] = q < s->concat_len ? s->seq[q] : B_EOS;
                }
                win = tmpw;
            }
            int r = adaptive_trim_ex((uint32_t)read_len, rs, win, min_put_len, (uint32_t)max_allowed_mm, 3, &mms, &t5, &t3);
            if (r > (int)min_put_len || (r == (int)min_put_len && mms < prev_best)) {
                prev_best = mms; min_put_len = (uint32_t)r;
                out->strand = antisense ? '-' : '+'; out->chrom_id = chrom_id; out->match_loci = hit;
                out->match_len = (uint16_t)read_len; out->mismatches = (uint8_t)mms;
                out->chimeric = (int)min_put_len == read_len ? 0 : 1;
                out->trim_left = (uint16_t)(antisense ? t3 : t5); out->trim_right = (uint16_t)(antisense ? t5 : t3);
            }
        }
    }
    return prev_best <= (uint32_t)max_allowed_mm ? 1 : 0;
}

static void pe_clear(ora_hit *h) { h->num_hits = 0; h->low_hit_instances = 0; }

/* hits[2i] = PE1, hits[2i+1] = PE2 as left by the SE alignment; updated in place.  ora_hit.flags
 * bit 7 = FlgPEAligned.  Returns 0. */
int ora_process_paired_ends(const ora_sfx *s, const ora_params *p, int pe_mode, int pair_min_len, int pair_max_len,
                            int pair_strand, const uint8_t *bases, const uint64_t *offs, const uint32_t *lens,
                            uint32_t n_pairs, ora_hit *hits)
{
    return ora_process_paired_ends_ex(s, p, pe_mode, pair_min_len, pair_max_len, pair_strand, bases, offs, lens, n_pairs, hits, NULL);
}

/* AdjStartLoci / AdjEndLoci (Aligner.cpp:1528-1544) of a read's Seg[0]: the end trims of a chimeric placement (seg2 flags bit 3:
 * match_len = TrimLeft, read_ofs = TrimRight) move them inwards */
static void adj_loci(const ora_hit *h, const ora_seg2 *g, uint32_t *start, uint32_t *end)
{
    uint32_t tl = 0, tr = 0;
    if (g != NULL && (g->flags & 8)) { tl = g->match_len; tr = g->read_ofs; }
    if (h->strand == '+') { *start = h->match_loci + tl; *end = h->match_loci + (h->match_len - tr - 1); }
    else { *start = h->match_loci + tr; *end = h->match_loci + (h->match_len - tl - 1); }
}

/* with seg2 != NULL (one entry per read, as ora_align_batch_ex left them): p->min_chimeric_len takes part - the pair rules look at the
 * trimmed loci and the orphan recovery may place the partner end-trimmed, whose trims then replace its seg2 entry */
int ora_process_paired_ends_ex(const ora_sfx *s, const ora_params *p, int pe_mode, int pair_min_len, int pair_max_len,
                               int pair_strand, const uint8_t *bases, const uint64_t *offs, const uint32_t *lens,
                               uint32_t n_pairs, ora_hit *hits, ora_seg2 *seg2)
{
    return ora_process_paired_ends_filt(s, p, pe_mode, pair_min_len, pair_max_len, pair_strand, bases, offs, lens, n_pairs, hits, seg2, NULL, 0);
}

/* .. with the chromosome filters (-Z / -z) the reference consults INSIDE its pair rules: accept[id] != 0 <=> CAligner::AcceptThisChromID(id)
 * (Aligner.cpp:2651-2715); ids beyond n_accept - 1 and a NULL table accept everything.  AcceptProvPE's results -3 / -4 / -5 (:2771-2786), their
 * handling (:3166-3187), the anchors of the orphan recovery (:3224,3316-3322,3323,3416-3421 - the second of which changes the FIRST read's
 * record, as it stands there) and the single-end acceptance at the end (:3442-3473). */
static int chrom_accepted(const uint8_t *accept, uint32_t n_accept, uint32_t id) { return accept == NULL || id >= n_accept || accept[id] != 0; }

int ora_process_paired_ends_filt(const ora_sfx *s, const ora_params *p, int pe_mode, int pair_min_len, int pair_max_len,
                                 int pair_strand, const uint8_t *bases, const uint64_t *offs, const uint32_t *lens,
                                 uint32_t n_pairs, ora_hit *hits, ora_seg2 *seg2, const uint8_t *accept, uint32_t n_accept)
{
    const int min_chim = seg2 != NULL ? p->min_chimeric_len : 0;
    int min_core = ora_min_core_len(s, p->pmode);
    int slides = ora_max_num_slides(p->pmode);
    for (uint32_t i = 0; i < n_pairs; i++) {
        ora_hit *f = &hits[2 * i], *r = &hits[2 * i + 1];
        f->flags &= 0x7f; r->flags &= 0x7f;
        int f_un = (f->nar == NAR_NS || f->nar == NAR_NOHIT || f->nar == NAR_UNALIGNED);
        int r_un = (r->nar == NAR_NS || r->nar == NAR_NOHIT || r->nar == NAR_UNALIGNED);
        if (!(f->nar == NAR_ACCEPTED || r->nar == NAR_ACCEPTED)) continue;
        if (pe_mode == PE_UNIQUE && (f_un || r_un)) {
            pe_clear(f); pe_clear(r);
            if (f->nar == NAR_ACCEPTED) f->nar = NAR_PENOHIT;
            if (r->nar == NAR_ACCEPTED) r->nar = NAR_PENOHIT;
            continue;
        }
        if (f->nar == NAR_ACCEPTED && r->nar == NAR_ACCEPTED) {
            int frag = 0;
            if (f->num_hits == 1 && r->num_hits == 1) {                 /* AcceptProvPE */
                const int f_ok = chrom_accepted(accept, n_accept, f->chrom_id);
                if (f->chrom_id != r->chrom_id) {
                    const int r_ok = chrom_accepted(accept, n_accept, r->chrom_id);
                    frag = (f_ok && r_ok) ? -2 : ((!f_ok && !r_ok) ? -3 : (!f_ok ? -4 : -5));
                } else if (!f_ok) frag = -3;
                else {
                    uint32_t fs, fe, rs_, re;
                    adj_loci(f, seg2 ? &seg2[2 * i] : NULL, &fs, &fe);
                    adj_loci(r, seg2 ? &seg2[2 * i + 1] : NULL, &rs_, &re);
                    frag = pe_insert_size(pair_min_len, pair_max_len, pair_strand, f->strand, fs, fe, r->strand, rs_, re);
                }
            }
            if (frag > 0) { f->flags |= 0x80; r->flags |= 0x80; continue; }
            switch (frag) {
            case -1: f->nar = r->nar = NAR_PESTRAND; break;
            case -2: f->nar = r->nar = NAR_PECHROM; break;
            case -3: pe_clear(f); pe_clear(r); f->nar = r->nar = NAR_CHROMFILT; break;      /* both ends on filtered sequences: next pair */
            case -4: f->nar = NAR_CHROMFILT; pe_clear(f); break;
            case -5: r->nar = NAR_CHROMFILT; pe_clear(r); break;
            case -6: f->nar = r->nar = NAR_PEINSERTMIN; break;
            case -7: f->nar = r->nar = NAR_PEINSERTMAX; break;
            }
            if (frag == -3) continue;
            if (pe_mode == PE_UNIQUE) {
                pe_clear(f); pe_clear(r);
                if (f->nar == NAR_ACCEPTED) f->nar = NAR_PENOHIT;
                if (r->nar == NAR_ACCEPTED) r->nar = NAR_PENOHIT;
                continue;
            }
        }
        if (pe_mode == PE_ORPHAN || pe_mode == PE_ORPHAN_SE) {
            int done = 0;
            for (int anchor = 0; anchor < 2 && !done; anchor++) {
                ora_hit *a = anchor == 0 ? f : r, *o = anchor == 0 ? r : f;
                int o_un = anchor == 0 ? r_un : f_un;
                if (!(a->num_hits == 1 && !o_un)) continue;
                if (!chrom_accepted(accept, n_accept, a->chrom_id)) {
                    /* an anchor on a filtered sequence is not used; :3316-3322 marks the first read, and so does :3416-3421 for the second */
                    if (a->nar == NAR_ACCEPTED) { pe_clear(f); f->nar = NAR_CHROMFILT; }
                    continue;
                }
                uint32_t oi = 2 * i + (anchor == 0 ? 1 : 0);
                int b3, anti;
                if (anchor == 0) {
                    b3 = a->strand == '+';
                    anti = pair_strand ? (a->strand != '+') : (a->strand == '+');
                } else {
                    b3 = a->strand == '+'; anti = a->strand == '+';
                    if (pair_strand) { b3 = !b3; anti = !anti; }
                }
                uint32_t a_start, a_end;
                adj_loci(a, seg2 ? &seg2[2 * i + (anchor == 0 ? 0 : 1)] : NULL, &a_start, &a_end);
                int probe_len = (int)lens[oi], match_len = probe_len - 1;
                int max_tot_mm = p->max_subs == 0 ? 0 : imax(1, (int)(0.5 + (match_len * p->max_subs) / 100.0));
                if (max_tot_mm > MAX_TOT_SUBS) max_tot_mm = MAX_TOT_SUBS;
                int core_len = imax(min_core, probe_len / (p->min_edit_dist == 1 ? max_tot_mm + 1 : max_tot_mm + 2));
                int core_delta = imax(probe_len / slides - 1, core_len);
                uint8_t rs[2100];
                for (int k = 0; k < probe_len; k++) rs[k] = bases[offs[oi] + k] & 0x07;
                hit_rec h;
                int rc = align_paired_read(s, b3, anti, a->chrom_id, a_start, a_end, pair_min_len, pair_max_len, p->max_subs,
                                           probe_len, min_chim, core_len, core_delta, rs, &h);
                if (rc == 1) {
                    int frag;
                    /* AdjStartLoci / AdjEndLoci of the placement just found */
                    uint32_t h_start = h.match_loci + (h.strand == '+' ? h.trim_left : h.trim_right);
                    uint32_t h_end = h.match_loci + (h.match_len - (h.strand == '+' ? h.trim_right : h.trim_left) - 1);
                    if (anchor == 0) frag = pe_insert_size(pair_min_len, pair_max_len, pair_strand, a->strand, a_start, a_end, h.strand, h_start, h_end);
                    else frag = pe_insert_size(pair_min_len, pair_max_len, pair_strand, h.strand, h_start, h_end, a->strand, a_start, a_end);
                    if (frag <= 0) rc = 0;
                }
                if (rc == 1) {
                    o->chrom_id = h.chrom_id; o->match_loci = h.match_loci; o->match_len = h.match_len; o->strand = h.strand;
                    o->mismatches = h.mismatches; o->num_hits = 1; o->low_mm = (int8_t)h.mismatches; o->low_hit_instances = 1;
                    if (seg2 != NULL) {                          /* pXReadHit->HitLoci.Hit = HitLoci: the whole record is replaced */
                        ora_seg2 *g = &seg2[oi];
                        memset(g, 0, sizeof(*g));
                        if (h.chimeric) { g->flags = 8; g->match_len = h.trim_left; g->read_ofs = h.trim_right; }
                    }
                    f->flags |= 0x80; r->flags |= 0x80;
                    f->nar = r->nar = NAR_ACCEPTED;
                    done = 1;
                }
            }
            if (done) continue;
        }
        if (!(pe_mode == PE_ORPHAN_SE || pe_mode == PE_UNIQUE_SE)) {
            pe_clear(f); pe_clear(r);
            if (f->nar == NAR_ACCEPTED) f->nar = NAR_PENOHIT;
            if (r->nar == NAR_ACCEPTED) r->nar = NAR_PENOHIT;
            continue;
        }
        for (int k = 0; k < 2; k++) {                    /* accept as SE what aligned uniquely */
            ora_hit *h = k == 0 ? f : r;
            int chrom_ok = h->num_hits == 1 ? chrom_accepted(accept, n_accept, h->chrom_id) : 0;
            if (h->num_hits != 1 || !chrom_ok) {
                pe_clear(h);
                if (h->nar == NAR_ACCEPTED) h->nar = chrom_ok ? NAR_CHROMFILT : NAR_PEUNALIGN;
            } else
                h->nar = NAR_ACCEPTED;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * SNP pile-up and screening.  CAligner::ProcessSNPs walks the sorted accepted reads of one sequence and counts, per
 * locus, read bases equal to / different from the target (Aligner.cpp:7737-7960, basespace branch :7937-7957);
 * CAligner::OutputSNPs then slides a 51 base window over the counts for the local background rate and screens every
 * locus (:6880-7110).  P-values and everything after them are host code of the product, pinned on reference output. */
typedef struct snp_cnts { uint8_t ref_base; uint32_t num_ref, num_nonref, nonref[5]; } snp_cnts;   /* tsSNPcnts, Aligner.h:280 */

int64_t ora_snp_chrom_sites(const ora_sfx *s, const uint8_t *bases, const uint64_t *offs, const ora_snp_aln *alns, uint64_t n_alns,
                            uint32_t chrom_id, int min_reads, double min_nonref_prop, ora_snp_site *sites_out, uint64_t max_sites,
                            ora_snp_chrom *totals)
{
    if (!s || !totals || chrom_id < 1 || chrom_id > s->num_entries) return -1;
    const ora_entry *ent = &s->entries[chrom_id - 1];
    const uint32_t chrom_len = ent->seq_len;
    snp_cnts *cnts = (snp_cnts *)calloc((size_t)chrom_len + 16, sizeof(snp_cnts));
    if (!cnts) return -2;
    uint64_t tot_match = 0, tot_mismatch = 0;
    for (uint64_t a = 0; a < n_alns; a++) {
        const ora_snp_aln *al = &alns[a];
        if (al->chrom_id != chrom_id) continue;
        uint32_t match_len = al->len, hit_loci = al->loci;
        uint8_t rd[4096], tg[4096];
        if (match_len > sizeof(rd)) { free(cnts); return -1; }
        for (uint32_t i = 0; i < match_len; i++) tg[i] = s->seq[ent->start_ofs + hit_loci + i] & 0x07;
        for (uint32_t i = 0; i < match_len; i++) rd[i] = bases[offs[al->read_idx] + al->read_ofs + i] & 0x07;
        if (al->strand == '-') {                                              /* CSeqTrans::ReverseComplement */
            for (uint32_t i = 0, j = match_len - 1; i < j; i++, j--) { uint8_t t = rd[i]; rd[i] = rd[j]; rd[j] = t; }
            for (uint32_t i = 0; i < match_len; i++) if (rd[i] < 4) rd[i] = (uint8_t)(3 - rd[i]);
        }
        if ((uint64_t)hit_loci + match_len > chrom_len) {                     /* :7826-7830 */
            if ((int)chrom_len - (int)hit_loci < 10) continue;
            match_len = chrom_len - hit_loci;
        }
        snp_cnts *p = &cnts[hit_loci];
        for (uint32_t i = 0; i < match_len; i++, p++) {
            if (tg[i] >= 4 || rd[i] > 4) continue;                            /* :7877 */
            p->ref_base = tg[i];
            if (tg[i] == rd[i]) { p->num_ref++; tot_match++; }
            else { p->nonref[rd[i]]++; p->num_nonref++; tot_mismatch++; }
        }
    }
    /* OutputSNPs: window of cSNPBkgndRateWindow = 51 loci */
    const uint32_t flank = 51 / 2, window = flank * 2 + 1;
    uint32_t local_mm = 0, local_m = 0;
    const snp_cnts *win_r = cnts, *win_l = cnts, *p = cnts;
    for (uint32_t l = 0; l < (window < chrom_len ? window : chrom_len); l++, win_r++) { local_mm += win_r->num_nonref; local_m += win_r->num_ref; }
    uint64_t n_sites = 0, covered = 0, coverage = 0;
    for (uint32_t l = 0; l < chrom_len; l++, p++) {
        if (l > flank && (l + flank) < chrom_len) {
            local_mm = local_mm >= win_l->num_nonref ? local_mm - win_l->num_nonref : 0;
            local_m = local_m >= win_l->num_ref ? local_m - win_l->num_ref : 0;
            local_mm += win_r->num_nonref;
            local_m += win_r->num_ref;
            win_l++; win_r++;
        }
        int tot = (int)(p->num_nonref + p->num_ref);
        if (tot > 0) { covered++; coverage += (uint64_t)tot; }
        if (tot < min_reads) continue;
        if (p->num_nonref < 1) continue;                                      /* cMinSNPreads */
        if ((double)p->num_nonref / tot < min_nonref_prop) continue;
        if (n_sites < max_sites && sites_out) {
            ora_snp_site *o = &sites_out[n_sites];
            o->loci = l; o->num_ref = p->num_ref;
            for (int k = 0; k < 5; k++) o->non_ref[k] = p->nonref[k];
            o->win_mismatches = local_mm; o->win_matches = local_m; o->ref_base = p->ref_base;
        }
        n_sites++;
    }
    totals->tot_match = tot_match; totals->tot_mismatch = tot_mismatch; totals->loci_covered = covered; totals->bases_coverage = coverage;
    free(cnts);
    return (int64_t)n_sites;
}

/* Test helper for the stage-level search parity (tests/test_gpu_search_stage.py): LocateFirstExact and LocateLastExact over the whole
 * suffix array for n probes cut from `bases` (probe i = bases[probe_ofs[i] .. + probe_len[i])), by nthreads threads.  first / last:
 * index + 1 of the lowest / highest matching suffix array element, 0 = no match - what the two reference functions return. */
typedef struct { const ora_sfx *s; const uint8_t *bases; const uint64_t *ofs; const int32_t *len; int64_t *first, *last; uint64_t lo, hi; } locate_job;
static void *locate_main(void *arg)
{
    locate_job *j = (locate_job *)arg;
    const int64_t top = (int64_t)j->s->concat_len - 1;
    for (uint64_t i = j->lo; i < j->hi; i++) {
        j->first[i] = ora_locate_first_exact(j->s, j->bases + j->ofs[i], j->len[i], 0, top, NULL);
        j->last[i] = j->first[i] ? ora_locate_last_exact(j->s, j->bases + j->ofs[i], j->len[i], 0, top, NULL) : 0;
    }
    return NULL;
}
void ora_locate_cores(const ora_sfx *s, const uint8_t *bases, const uint64_t *probe_ofs, const int32_t *probe_len, uint64_t n,
                      int64_t *first, int64_t *last, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    locate_job job[64];
    pthread_t th[64];
    const uint64_t per = (n + (uint64_t)nthreads - 1) / (uint64_t)nthreads;
    for (int t = 0; t < nthreads; t++) {
        uint64_t lo = per * (uint64_t)t, hi = lo + per;
        if (lo > n) lo = n;
        if (hi > n) hi = n;
        job[t] = (locate_job){s, bases, probe_ofs, probe_len, first, last, lo, hi};
        pthread_create(&th[t], NULL, locate_main, &job[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
}

/* Test helper: element idx of the suffix array (the stage-level search test maps a handed-on suffix back to its index) */
int64_t ora_sa_element(const ora_sfx *s, int64_t idx)
{
    return (idx < 0 || (uint64_t)idx >= s->concat_len) ? -1 : (int64_t)sa_at(s, idx);
}
