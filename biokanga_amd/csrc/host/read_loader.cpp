// read_loader.cpp - see read_loader.h
#include "read_loader.h"

#include <atomic>
#include <cctype>
#include <cstring>
#include <thread>

#include "fasta.h"

namespace bkcli {

namespace {

// The acceptance rules of load_reads() applied to a file that was parsed whole (fasta.h, ParsedChunk): every
// chunk is filtered and measured by its own thread, a prefix sum gives each chunk its place in the read
// store, and the threads copy their accepted records there.  Same records, same order, same log lines.
// An empty read store ADOPTS the parser's bases buffer (fasta.h, ParsedFile): the bases stay where the pieces wrote them and the store's
// offsets point there - no placement pass over the 5 GB of a 50 M-read file; only the names are laid back to back.  A second input
// file is appended to the store by copy.
int accept_chunks(bk::ParsedFile &file, const std::string &fn, int trim5, int trim3, int min_len, int max_len,
                  int nthreads, ReadStore &rs)
{
    std::vector<bk::ParsedChunk> &chunks = file.chunks;
    const size_t nc = chunks.size();
    const bool adopt = rs.lens.empty() && rs.bases.empty();
    bool sim = false;
    for (const auto &c : chunks)
        if (!c.lens.empty()) {
            size_t dl = std::min<size_t>(c.descr_lens[0], 127);
            sim = dl >= 14 && (!strncmp(c.descr, "lcl|usimreads|", 14) || !strncmp(c.descr, "lcr|usimreads|", 14));
            break;
        }
    struct Tot { uint64_t n_acc = 0, n_bases = 0, n_names = 0, n_under = 0, n_over = 0, n_rec = 0; long bad_at = -1; };
    std::vector<Tot> tot(nc);
    std::vector<std::vector<uint32_t>> keep_name_len(nc);       // per record: accepted name length + 1, or 0 when sloughed
    auto name_len = [](const char *d, size_t dl, bool sim_) {
        if (dl > 127) dl = 127;
        if (sim_) return dl;
        size_t k = 0;
        while (k < 79 && k < dl && !isspace((unsigned char)d[k])) k++;
        return k;
    };
    auto run = [&](auto fn_) {
        std::vector<std::thread> th;
        for (int w = 1; w < nthreads; w++) th.emplace_back([&, w]() { for (size_t c = (size_t)w; c < nc; c += (size_t)nthreads) fn_(c); });
        for (size_t c = 0; c < nc; c += (size_t)nthreads) fn_(c);
        for (auto &t : th) t.join();
    };
    run([&](size_t ci) {
        // Everything the record loop reads is a copy of this call's own: the function's parameters and flags live in the frame of the thread
        // that also works through chunks here - its stores to that frame would take the cache line away from every other thread once per
        // record (seen: this pass between 0.1 and 3.7 s for the same input, depending on how a build laid the frame out).  The counts go
        // to `tot` once, for the same reason: neighbours there belong to other threads' chunks.
        const bk::ParsedChunk &c = chunks[ci];
        const int t5 = trim5, t3 = trim3, mn = min_len, mx = max_len;
        const bool sim_ = sim;
        const size_t n_rec = c.lens.size();
        const uint32_t *lens_ = c.lens.data(), *dlens_ = c.descr_lens.data();
        const char *descr_ = c.descr;
        Tot t;
        auto &kn_v = keep_name_len[ci];
        kn_v.assign(n_rec, 0);
        uint32_t *kn = kn_v.data();
        size_t dofs = 0;
        for (size_t i = 0; i < n_rec; i++) {
            const int len = (int)lens_[i];
            const size_t dl = dlens_[i];
            t.n_rec++;
            if (len < 1 || len > 0x30000) { if (t.bad_at < 0) t.bad_at = (long)i; }
            else if (t5 + t3 + mn > len) t.n_under++;
            else if (t5 + t3 + mx < len) t.n_over++;
            else {
                size_t nl = name_len(descr_ + dofs, dl, sim_);
                kn[i] = (uint32_t)nl + 1;
                t.n_acc++;
                t.n_bases += (uint64_t)(len - t5 - t3);
                t.n_names += nl + 1;
            }
            dofs += dl;
        }
        tot[ci] = t;
    });
    // log lines in file order, as the serial loader prints them
    uint64_t n_descr = 0, n_under = 0, n_over = 0, n_acc = 0;
    for (size_t ci = 0; ci < nc; ci++) {
        const bk::ParsedChunk &c = chunks[ci];
        // (a chunk with an unusable record: the lines of the records in front of it first, as the serial loader would have printed them)
        const size_t upto = tot[ci].bad_at >= 0 ? (size_t)tot[ci].bad_at : c.lens.size();
        if (tot[ci].bad_at >= 0 || (n_under < 10 && tot[ci].n_under) || (n_over < 10 && tot[ci].n_over))
            for (size_t i = 0; i < upto; i++) {
                const int len = (int)c.lens[i];
                if (trim5 + trim3 + min_len > len) { if (++n_under <= 10) diag("Load: under length (%d) sequence in '%s' after end trims has been sloughed..", len, fn.c_str()); }
                else if (trim5 + trim3 + max_len < len) { if (++n_over <= 10) diag("Load: over length (%d) sequence in '%s' after end trims has been sloughed..", len, fn.c_str()); }
            }
        else { n_under += tot[ci].n_under; n_over += tot[ci].n_over; }
        if (tot[ci].bad_at >= 0) { diag("Problem parsing sequence after %llu reads parsed", (unsigned long long)(n_descr + tot[ci].bad_at + 1)); return -63; }
        n_descr += tot[ci].n_rec;
        n_acc += tot[ci].n_acc;
    }
    // placement
    std::vector<uint64_t> r0(nc + 1), b0(nc + 1), m0(nc + 1);
    r0[0] = rs.lens.size(); b0[0] = rs.bases.size(); m0[0] = rs.names.size();
    for (size_t ci = 0; ci < nc; ci++) {
        r0[ci + 1] = r0[ci] + tot[ci].n_acc;
        b0[ci + 1] = b0[ci] + tot[ci].n_bases;
        m0[ci + 1] = m0[ci] + tot[ci].n_names;
    }
    rs.lens.resize(r0[nc]); rs.offs.resize(r0[nc]); rs.name_ofs.resize(r0[nc]);
    if (!adopt) rs.bases.resize(b0[nc]);
    rs.names.resize(m0[nc]);
    run([&](size_t ci) {
        // (copies again: see the first pass)
        bk::ParsedChunk &c = chunks[ci];
        const uint32_t *kn = keep_name_len[ci].data(), *lens_ = c.lens.data(), *dlens_ = c.descr_lens.data();
        const size_t n_rec = c.lens.size();
        const uint32_t t5 = (uint32_t)trim5, t3 = (uint32_t)trim3;
        const bool adopt_ = adopt;
        const uint8_t *cbases = c.bases;
        const char *descr_ = c.descr;
        const uint64_t chunk_at = adopt_ ? (uint64_t)(c.bases - file.bases.data()) : 0;
        uint32_t *o_lens = rs.lens.data();
        uint64_t *o_offs = rs.offs.data(), *o_nofs = rs.name_ofs.data();
        uint8_t *o_bases = adopt_ ? nullptr : rs.bases.data();
        char *o_names = rs.names.data();
        uint64_t r = r0[ci], bo = b0[ci], mo = m0[ci];
        size_t dofs = 0, sofs = 0;
        for (size_t i = 0; i < n_rec; i++) {
            const uint32_t len = lens_[i];
            if (kn[i]) {
                const uint32_t keep = len - t5 - t3, nl = kn[i] - 1;
                o_lens[r] = keep;
                if (adopt_) o_offs[r] = chunk_at + sofs + (uint64_t)t5;
                else {
                    o_offs[r] = bo;
                    memcpy(o_bases + bo, cbases + sofs + t5, keep);
                }
                o_nofs[r] = mo;                                // (the names - a tenth of the bases - are laid back to back)
                memcpy(o_names + mo, descr_ + dofs, nl);
                o_names[mo + nl] = '\0';
                r++; bo += keep; mo += nl + 1;
            }
            dofs += dlens_[i];
            sofs += len;
        }
    });
    if (adopt) {
        rs.bases.swap(file.bases);
        rs.used_bases = b0[nc];
    } else
        rs.used_bases = 0;
    diag("Load: %llu reads parsed, %llu accepted, %llu under length, %llu over length from '%s'", (unsigned long long)n_descr,
         (unsigned long long)n_acc, (unsigned long long)n_under, (unsigned long long)n_over, fn.c_str());
    g_whole_file_loads++;
    return 0;
}

}  // namespace

// CAligner::LoadRawReads (Aligner.cpp:10724-11427): descriptor rule, -y/-Y trims, -l/-L acceptance
int g_qual_mode = 3;
int g_sample_nth = 1;
int g_whole_file_loads = 0;

int load_reads(const std::vector<std::string> &files, int trim5, int trim3, int min_len, int max_len, int nthreads, ReadStore &rs)
{
    for (const std::string &fn : files) {
        bk::RecordStream rd;
        std::string err;
        rd.set_quality_mode(g_qual_mode);
        int rc = rd.open(fn, nthreads, &err);
        if (rc) { diag("Load: %s", err.c_str()); return rc; }
        diag("Loading reads from '%s'", fn.c_str());
        if (rd.parsed() && g_sample_nth <= 1) {
            rc = accept_chunks(rd.file(), fn, trim5, trim3, min_len, max_len, nthreads, rs);
            if (rc) return rc;
            continue;
        }
        const char *d;
        const uint8_t *b;
        size_t dl, bl;
        bool sim = false;
        uint32_t n_descr = 0, n_under = 0, n_over = 0, n_acc = 0;
        int nxt_sample = g_sample_nth;
        while ((rc = rd.next(d, dl, b, bl)) > 0) {
            n_descr++;
            if (dl > 127) dl = 127;                                       // cMaxDescrLen-1
            if (n_descr == 1) sim = dl >= 14 && (!strncmp(d, "lcl|usimreads|", 14) || !strncmp(d, "lcr|usimreads|", 14));
            if (g_sample_nth > 1) {
                nxt_sample++;
                if (g_sample_nth > nxt_sample) continue;
                nxt_sample = 0;
            }
            int len = (int)bl;
            if (bl < 1 || bl > 0x30000) { diag("Problem parsing sequence after %u reads parsed", n_descr); return -63; }
            if (trim5 + trim3 + min_len > len) {
                if (++n_under <= 10) diag("Load: under length (%d) sequence in '%s' after end trims has been sloughed..", len, fn.c_str());
                continue;
            }
            if (trim5 + trim3 + max_len < len) {
                if (++n_over <= 10) diag("Load: over length (%d) sequence in '%s' after end trims has been sloughed..", len, fn.c_str());
                continue;
            }
            if (!sim) {                                                   // cut at first whitespace, < cMaxDescrIDLen
                size_t k = 0;
                while (k < 79 && k < dl && !isspace((unsigned char)d[k])) k++;
                dl = k;
            }
            int keep = len - trim5 - trim3;
            rs.offs.push_back(rs.bases.size());
            rs.lens.push_back((uint32_t)keep);
            rs.bases.insert(rs.bases.end(), b + trim5, b + trim5 + keep);
            rs.name_ofs.push_back(rs.names.size());
            rs.names.insert(rs.names.end(), d, d + dl);
            rs.names.push_back('\0');
            n_acc++;
        }
        if (rc < 0) { diag("Load: errors whilst parsing '%s'", fn.c_str()); return rc; }
        diag("Load: %u reads parsed, %u accepted, %u under length, %u over length from '%s'", n_descr, n_acc, n_under, n_over, fn.c_str());
    }
    return 0;
}

namespace {

// paired end loading: PE1/PE2 records in lockstep, both ends must pass the length acceptance
// (Aligner.cpp:11080-11130); stored interleaved PE1, PE2
// The paired loader's loop over two whole-file parses, by all threads: records i of the two files are the mates of pair i, a pair is
// kept when both mates pass the length rules (the first failing mate decides which counter it goes to), kept pairs are laid out PE1, PE2,
// PE1, PE2 .. with their bases and names back to back - what the serial loop below produces record by record at 60 ns each (2.5 s of a
// 20 M-pair run).  Prefix sums over blocks of pairs give every piece of either file its place; each piece's thread places its own records.
// Returns 0, a negative code after the message the serial loop would have printed, or 1 when the files are not eligible (then the
// serial loop runs).
int accept_pairs(bk::ParsedFile &fa, bk::ParsedFile &fb, const std::string &na, const std::string &nb, int trim5, int trim3, int min_len, int max_len,
                 int nthreads, ReadStore &rs)
{
    bk::ParsedFile *F[2] = {&fa, &fb};
    size_t nc[2], N[2];
    std::vector<uint64_t> r0[2];                   // first record number of every piece
    for (int e = 0; e < 2; e++) {
        nc[e] = F[e]->chunks.size();
        r0[e].assign(nc[e] + 1, 0);
        for (size_t c = 0; c < nc[e]; c++) r0[e][c + 1] = r0[e][c] + F[e]->chunks[c].lens.size();
        N[e] = r0[e][nc[e]];
    }
    if (N[0] == 0 || N[0] >= 0xFFFFFFFFull) return 1;
    if (N[1] < N[0]) return 1;                     // (the serial loop says where the second file ends)
    const size_t P = N[0];                         // pairs (surplus records of the second file are never looked at)
    bool sim[2] = {false, false};
    for (int e = 0; e < 2; e++)
        for (const auto &c : F[e]->chunks)
            if (!c.lens.empty()) {
                const size_t dl = std::min<size_t>(c.descr_lens[0], 127);
                sim[e] = dl >= 14 && (!strncmp(c.descr, "lcl|usimreads|", 14) || !strncmp(c.descr, "lcr|usimreads|", 14));
                break;
            }
    // per record: status (0 kept so far, 1 under length, 2 over length, 3 unusable), kept bases, name length + 1
    bk::RawVec<uint8_t> st[2], nl[2];
    bk::RawVec<uint32_t> kl[2];
    for (int e = 0; e < 2; e++) { st[e].resize(P); nl[e].resize(P); kl[e].resize(P); }
    auto run = [&](size_t n_items, auto fn_) {
        std::vector<std::thread> th;
        std::atomic<size_t> next{0};
        auto work = [&]() { for (size_t i; (i = next.fetch_add(1)) < n_items;) fn_(i); };
        for (int w = 1; w < nthreads; w++) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
    };
    run(nc[0] + nc[1], [&](size_t item) {
        // (copies of what the record loop reads: see accept_chunks)
        const int e = item < nc[0] ? 0 : 1;
        const bk::ParsedChunk &c = F[e]->chunks[item - (e ? nc[0] : 0)];
        const uint64_t g0 = r0[e][item - (e ? nc[0] : 0)];
        const int t5 = trim5, t3 = trim3, mn = min_len, mx = max_len;
        const bool sim_ = sim[e];
        const uint32_t *lens_ = c.lens.data(), *dlens_ = c.descr_lens.data();
        const char *descr_ = c.descr;
        uint8_t *st_ = st[e].data(), *nl_ = nl[e].data();
        uint32_t *kl_ = kl[e].data();
        const size_t n_rec = c.lens.size(), pairs = P;
        size_t dofs = 0;
        for (size_t i = 0; i < n_rec && g0 + i < pairs; i++) {
            const int len = (int)lens_[i];
            size_t dl = dlens_[i];
            uint8_t s = 0;
            if (len < 1 || len > 0x30000) s = 3;
            else if (t5 + t3 + mn > len) s = 1;
            else if (t5 + t3 + mx < len) s = 2;
            if (dl > 127) dl = 127;
            size_t q = dl;
            if (!sim_) { q = 0; while (q < 79 && q < dl && !isspace((unsigned char)descr_[dofs + q])) q++; }
            st_[g0 + i] = s;
            nl_[g0 + i] = (uint8_t)(q + 1);
            kl_[g0 + i] = s == 0 ? (uint32_t)(len - t5 - t3) : 0u;
            dofs += dlens_[i];
        }
    });
    // pairs in blocks: kept pairs, their bases and name bytes, the counters, the first unusable record
    const size_t kBlock = 65536, nblk = (P + kBlock - 1) / kBlock;
    struct Blk { uint64_t kept = 0, bases = 0, names = 0, under = 0, over = 0; long bad = -1; };
    std::vector<Blk> blk(nblk);
    run(nblk, [&](size_t b) {
        const uint8_t *sa = st[0].data(), *sb = st[1].data(), *la = nl[0].data(), *lb = nl[1].data();
        const uint32_t *ka = kl[0].data(), *kb = kl[1].data();
        Blk t;
        const size_t lo = b * kBlock, hi = std::min(P, lo + kBlock);
        for (size_t i = lo; i < hi; i++) {
            const uint8_t a = sa[i], z = sb[i];
            if (a == 3 || z == 3) { if (t.bad < 0) t.bad = (long)i; continue; }
            if (a == 1) t.under++;
            else if (a == 2) t.over++;
            else if (z == 1) t.under++;
            else if (z == 2) t.over++;
            else { t.kept++; t.bases += (uint64_t)ka[i] + kb[i]; t.names += (uint64_t)la[i] + lb[i]; }
        }
        blk[b] = t;
    });
    uint64_t n_under = 0, n_over = 0;
    std::vector<uint64_t> k0(nblk + 1, 0), b0(nblk + 1, 0), m0(nblk + 1, 0);
    for (size_t b = 0; b < nblk; b++) {
        if (blk[b].bad >= 0) { diag("Problem parsing sequence after %u reads parsed", (uint32_t)(blk[b].bad + 1)); return -63; }
        k0[b + 1] = k0[b] + blk[b].kept; b0[b + 1] = b0[b] + blk[b].bases; m0[b + 1] = m0[b] + blk[b].names;
        n_under += blk[b].under; n_over += blk[b].over;
    }
    const uint64_t K = k0[nblk];
    const uint64_t rd_at = rs.lens.size(), bs_at = rs.bases.size(), nm_at = rs.names.size();
    rs.lens.resize(rd_at + 2 * K); rs.offs.resize(rd_at + 2 * K); rs.name_ofs.resize(rd_at + 2 * K);
    rs.bases.resize(bs_at + b0[nblk]);
    rs.names.resize(nm_at + m0[nblk]);
    rs.used_bases = 0;
    // placement: every piece of either file walks its records with the running place of their pairs
    run(nc[0] + nc[1], [&](size_t item) {
        const int e = item < nc[0] ? 0 : 1;
        const bk::ParsedChunk &c = F[e]->chunks[item - (e ? nc[0] : 0)];
        const uint64_t g0 = r0[e][item - (e ? nc[0] : 0)];
        if (g0 >= P) return;
        const uint8_t *sa = st[0].data(), *sb = st[1].data(), *la = nl[0].data(), *lb = nl[1].data();
        const uint32_t *ka = kl[0].data(), *kb = kl[1].data();
        const uint32_t *lens_ = c.lens.data(), *dlens_ = c.descr_lens.data();
        const uint8_t *cbases = c.bases;
        const char *descr_ = c.descr;
        const uint32_t t5 = (uint32_t)trim5;
        uint32_t *o_lens = rs.lens.data() + rd_at;
        uint64_t *o_offs = rs.offs.data() + rd_at, *o_nofs = rs.name_ofs.data() + rd_at;
        uint8_t *o_bases = rs.bases.data();
        char *o_names = rs.names.data();
        // place of pair g0: its block's, plus the kept pairs of the block in front of it
        const size_t b = g0 / kBlock;
        uint64_t k = k0[b], bo = bs_at + b0[b], mo = nm_at + m0[b];
        for (size_t i = b * kBlock; i < g0; i++)
            if (sa[i] == 0 && sb[i] == 0) { k++; bo += (uint64_t)ka[i] + kb[i]; mo += (uint64_t)la[i] + lb[i]; }
        const size_t n_rec = c.lens.size();
        size_t dofs = 0, sofs = 0;
        for (size_t i = 0; i < n_rec && g0 + i < P; i++) {
            const size_t g = g0 + i;
            if (sa[g] == 0 && sb[g] == 0) {
                const uint64_t r = 2 * k + (uint64_t)e;
                const uint64_t my_b = bo + (e ? ka[g] : 0), my_m = mo + (e ? la[g] : 0);
                const uint32_t keep = e ? kb[g] : ka[g], nlen = (uint32_t)(e ? lb[g] : la[g]) - 1;
                o_lens[r] = keep;
                o_offs[r] = my_b;
                memcpy(o_bases + my_b, cbases + sofs + t5, keep);
                o_nofs[r] = my_m;
                memcpy(o_names + my_m, descr_ + dofs, nlen);
                o_names[my_m + nlen] = '\0';
                k++; bo += (uint64_t)ka[g] + kb[g]; mo += (uint64_t)la[g] + lb[g];
            }
            dofs += dlens_[i];
            sofs += lens_[i];
        }
    });
    diag("Load: %u pairs parsed, %u accepted, %u under length, %u over length", (uint32_t)P, (uint32_t)K, (uint32_t)n_under, (uint32_t)n_over);
    (void)na; (void)nb;
    g_whole_file_loads += 2;
    return 0;
}

}  // namespace

int load_reads_pe(const std::vector<std::string> &f1, const std::vector<std::string> &f2, int trim5, int trim3, int min_len, int max_len,
                  int nthreads, ReadStore &rs)
{
    for (size_t k = 0; k < f1.size(); k++) {
        bk::RecordStream rd[2];
        std::string err;
        rd[0].set_quality_mode(g_qual_mode);
        rd[1].set_quality_mode(g_qual_mode);
        // the mate files are opened side by side, half of the threads each: a file that is one gzip member is inflated by one thread
        // however many there are, and two such files should not wait for each other
        int rc, rc2 = 0;
        if (nthreads >= 4) {
            std::string err2;
            std::thread second([&]() { rc2 = rd[1].open(f2[k], nthreads - nthreads / 2, &err2); });
            rc = rd[0].open(f1[k], nthreads / 2, &err);
            second.join();
            if (!rc && rc2) err = err2;
        } else {
            rc = rd[0].open(f1[k], nthreads, &err);
            if (!rc) rc2 = rd[1].open(f2[k], nthreads, &err);
        }
        if (rc || rc2) { diag("Load: %s", err.c_str()); return rc ? rc : rc2; }
        diag("Loading paired end reads from '%s' and '%s'", f1[k].c_str(), f2[k].c_str());
        if (rd[0].parsed() && rd[1].parsed() && g_sample_nth <= 1) {
            rc = accept_pairs(rd[0].file(), rd[1].file(), f1[k], f2[k], trim5, trim3, min_len, max_len, nthreads, rs);
            if (rc < 0) return rc;
            if (rc == 0) continue;
        }
        const char *d[2];
        const uint8_t *b[2];
        size_t dl[2], bl[2];
        bool sim[2] = {false, false};
        uint32_t n_descr = 0, n_under = 0, n_over = 0, n_acc = 0;
        int nxt_sample = g_sample_nth;
        for (;;) {
            int rc1 = rd[0].next(d[0], dl[0], b[0], bl[0]);
            if (rc1 < 0) { diag("Load: errors whilst parsing '%s'", f1[k].c_str()); return rc1; }
            if (rc1 == 0) break;
            int rc2 = rd[1].next(d[1], dl[1], b[1], bl[1]);
            if (rc2 <= 0) { diag("Load: '%s' has fewer reads than '%s'", f2[k].c_str(), f1[k].c_str()); return -63; }
            n_descr++;
            bool skip = false;
            for (int e = 0; e < 2; e++) {
                if (dl[e] > 127) dl[e] = 127;
                if (n_descr == 1) sim[e] = dl[e] >= 14 && (!strncmp(d[e], "lcl|usimreads|", 14) || !strncmp(d[e], "lcr|usimreads|", 14));
                if (bl[e] < 1 || bl[e] > 0x30000) { diag("Problem parsing sequence after %u reads parsed", n_descr); return -63; }
            }
            if (g_sample_nth > 1) {
                nxt_sample++;
                if (g_sample_nth > nxt_sample) continue;
                nxt_sample = 0;
            }
            for (int e = 0; e < 2 && !skip; e++) {
                int len = (int)bl[e];
                if (trim5 + trim3 + min_len > len) { n_under++; skip = true; }
                else if (trim5 + trim3 + max_len < len) { n_over++; skip = true; }
            }
            if (skip) continue;
            for (int e = 0; e < 2; e++) {
                if (!sim[e]) {
                    size_t q = 0;
                    while (q < 79 && q < dl[e] && !isspace((unsigned char)d[e][q])) q++;
                    dl[e] = q;
                }
                int keep = (int)bl[e] - trim5 - trim3;
                rs.offs.push_back(rs.bases.size());
                rs.lens.push_back((uint32_t)keep);
                rs.bases.insert(rs.bases.end(), b[e] + trim5, b[e] + trim5 + keep);
                rs.name_ofs.push_back(rs.names.size());
                rs.names.insert(rs.names.end(), d[e], d[e] + dl[e]);
                rs.names.push_back('\0');
            }
            n_acc++;
        }
        diag("Load: %u pairs parsed, %u accepted, %u under length, %u over length", n_descr, n_acc, n_under, n_over);
    }
    return 0;
}

}  // namespace bkcli
