// bk_sam.hip - SAM records formatted on the device (bk_sam_format of include/biokanga_amd.h).
//
// CAligner::ReportBAMread (biokanga/Aligner.cpp:5768-6126) prints one text line per read with sprintf, ~4.5 us each on one
// thread; formatting 50 M records is byte-parallel work: a lane per record measures its line, a prefix sum places it, the lane
// writes it.  The host keeps what is serial by nature - the reference's output order (its quicksort replica) and the file itself.
// The text leaves the device in slices through a pair of pinned buffers; the caller's sink receives them in file order.
#include <hip/hip_runtime.h>
#include "bk_prim.h"

#include <algorithm>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "bk_ctx_int.h"
#include "bk_env.h"

namespace {

struct SamDev {
    const uint8_t *bases;
    const uint64_t *offs;
    const uint32_t *lens;
    const char *names;
    const uint64_t *name_ofs;        // n_reads + 1 entries: name i = [name_ofs[i], name_ofs[i + 1] - 1)
    const bk_hit *hits;
    const uint32_t *order;
    const char *ent_names;           // n_ent x 81 bytes
    uint32_t n_ent;
    int fmt6, pe_mode;
    const uint32_t *any_qual;        // one word: non-zero when some base of the read store carries a score (else every QUAL is '*' and nobody looks)
    // packed reads (bases == nullptr): 16 bases per word, first base in the top bits; word offset of every read; the exceptions (non-acgt
    // runs) sorted by (read, pos) and, per read, the index of its first one (kNoExc = none)
    const uint32_t *pk_words;
    const uint64_t *pk_wofs;
    const bk_nbase *pk_exc;
    const uint32_t *pk_efirst;
    uint64_t n_pk_exc;
};
constexpr uint32_t kNoExc = 0xFFFFFFFFu;

__device__ __forceinline__ int n_digits(unsigned long v)
{
    int n = 1;
    while (v >= 10) { v /= 10; n++; }
    return n;
}

__device__ __forceinline__ char *put_num(char *w, long v)
{
    char t[24];
    int n = 0;
    const bool neg = v < 0;
    unsigned long u = neg ? (unsigned long)(-v) : (unsigned long)v;
    do { t[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (neg) *w++ = '-';
    while (n) *w++ = t[--n];
    return w;
}

__device__ __forceinline__ char *put_str(char *w, const char *z)
{
    while (*z) *w++ = *z++;
    return w;
}

__device__ const char kSamNarTag[20][3] = {"NA", "AA", "EN", "NL", "MH", "ML", "ET", "OJ", "OM", "DP", "DS", "FC", "PR", "UI", "OI", "UP", "IS", "IT", "NP", "LC"};

// the fields of record k that both passes need
struct SamRec {
    uint32_t i, len, nml;
    bool acc, reported, has_qual;
    int flag, tlen;
    long pnext;
};

__device__ __forceinline__ SamRec sam_rec(const SamDev &d, uint64_t k)
{
    SamRec r;
    r.i = d.order[k];
    const bk_hit h = d.hits[r.i];
    r.acc = h.nar == BK_NAR_ACCEPTED;
    r.reported = r.acc || d.fmt6;
    r.len = d.lens[r.i];
    r.nml = (uint32_t)(d.name_ofs[r.i + 1] - d.name_ofs[r.i] - 1);
    r.flag = 0; r.tlen = 0; r.pnext = -1; r.has_qual = false;
    if (!r.reported) return r;
    if (!d.pe_mode) r.flag = r.acc ? (h.strand == '+' ? 0 : 16) : 4;
    else {
        // flags of CAligner::ReportBAMread for paired ends (Aligner.cpp:5850-5924)
        const bool first_of_pair = (r.i & 1) == 0;
        const bk_hit m = d.hits[first_of_pair ? r.i + 1 : r.i - 1];
        r.flag = 0x1 | 0x2 | (first_of_pair ? 0x40 : 0x80);
        r.flag |= r.acc ? (h.strand == '+' ? 0 : 0x10) : 0x4;
        if ((h.flags & 0x80) && (m.flags & 0x80) && m.nar == BK_NAR_ACCEPTED) {
            r.flag |= m.strand == '+' ? 0 : 0x20;
            if (r.acc) {
                r.pnext = (long)m.match_loci;
                const long s0 = (long)h.match_loci, s1 = (long)m.match_loci;
                r.tlen = (int)(s0 <= s1 ? (s1 - s0) + (long)m.match_len : (s0 - s1) + (long)h.match_len);
            }
        } else
            r.flag |= 0x8;
    }
    return r;
}

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src)
{
    return ((uint64_t)(uint32_t)__shfl((int)(v >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)v, src);
}

// QUAL is '*' unless a base of the read carries a score (bits 4..7 of its byte).  The wave looks at its 64 records together: the lanes
// read a record's bases side by side (one or two cache lines per step) instead of each lane walking its own read byte by byte.
// Called by all lanes of a wave; len = 0 for a lane without a record.
__device__ __forceinline__ bool wave_has_qual(const SamDev &d, uint64_t off, uint32_t len)
{
    if (d.bases == nullptr || *d.any_qual == 0) return false;
    const int lane = (int)(threadIdx.x & 63);
    bool mine = false;
    for (int rr = 0; rr < 64; rr++) {
        const uint32_t l = (uint32_t)__shfl((int)len, rr);
        if (!l) continue;
        const uint8_t *s = d.bases + shfl64(off, rr);
        uint32_t acc = 0;
        for (uint32_t q = (uint32_t)lane; q < l; q += 64) acc |= s[q] & 0xf0u;
        const bool any = __ballot(acc != 0) != 0;
        if (lane == rr) mine = any;
    }
    return mine;
}

// non-zero when any base of the store carries a score
__global__ void __launch_bounds__(256) k_sam_any_qual(const uint8_t *__restrict__ bases, uint64_t n, uint32_t *__restrict__ flag)
{
    uint32_t acc = 0;
    const uint64_t n16 = n / 16;
    const uint4 *b16 = reinterpret_cast<const uint4 *>(bases);
    if ((reinterpret_cast<uintptr_t>(bases) & 15) == 0)
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
            const uint4 v = b16[i];
            acc |= (v.x | v.y | v.z | v.w) & 0xf0f0f0f0u;
        }
    else
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16 * 16; i += (uint64_t)gridDim.x * blockDim.x) acc |= bases[i] & 0xf0u;
    for (uint64_t i = n16 * 16 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) acc |= bases[i] & 0xf0u;
    if (__ballot(acc != 0) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

__global__ void __launch_bounds__(256) k_sam_measure(SamDev d, uint64_t k0, uint32_t n, unsigned long long *__restrict__ bytes, uint32_t *__restrict__ n_rep)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t rep = 0;
    SamRec r;
    r.reported = false; r.len = 0; r.i = 0;
    if (j < n) r = sam_rec(d, k0 + j);
    r.has_qual = wave_has_qual(d, (r.reported && d.bases) ? d.offs[r.i] : 0, (r.reported && d.bases) ? r.len : 0);
    if (j < n) {
        unsigned long long b = 0;
        if (r.reported) {
            rep = 1;
            const bk_hit h = d.hits[r.i];
            b = r.nml + 1 + n_digits((unsigned long)r.flag);
            const unsigned qual = r.has_qual ? r.len : 1u;
            if (r.acc) {
                const char *nm = d.ent_names + (size_t)(h.chrom_id - 1) * 81;
                uint32_t cl = 0;
                while (nm[cl]) cl++;
                // \t RNAME \t POS \t255\t <len>M \t [*=] \t PNEXT \t TLEN \t SEQ \t QUAL \n
                b += 1 + cl + 1 + n_digits((unsigned long)h.match_loci + 1) + 5 + n_digits(h.match_len) + 1 + 1 + 1 + 1 +
                     n_digits((unsigned long)(r.pnext < 0 ? 0 : r.pnext + 1)) + 1 + n_digits((unsigned long)r.tlen) + 1 + r.len + 1 + qual + 1;
            } else
                // \t*\t0\t255\t <len>M\t*\t0\t0\t SEQ \t QUAL \t\tYU:Z:xx \n
                b += 9 + n_digits(r.len) + 8 + r.len + 1 + qual + 7 + 2 + 1;
        }
        bytes[j] = b;
    }
    const uint64_t m = __ballot(rep != 0);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_rep, (uint32_t)__popcll(m));
}

// One bulk piece of a line - name, sequence, scores - copied by the whole wave: lane q moves byte q, q + 64, .. of the piece of record rr,
// for rr = 0 .. 63 in turn (coalesced loads and stores; the lanes' own small fields went out before).  mode 0: bytes as they are;
// 1: base letters; 2: complement letters, read backwards; 3: scores; 4: scores, read backwards; 5 / 6: as 1 / 2 from packed words
// (src = the read's first word; the bases that are not a,c,g,t are written over afterwards by the record's own lane).
__device__ __forceinline__ void wave_copy(const uint8_t *src, char *dst, uint32_t len, int mode)
{
    const int lane = (int)(threadIdx.x & 63);
    const uint64_t fwd = 0x4E4E4E4E54474341ULL, comp = 0x4E4E4E4E41434754ULL;      // "ACGTNNNN", "TGCANNNN": letter of code c = byte c
    for (int rr = 0; rr < 64; rr++) {
        const uint32_t l = (uint32_t)__shfl((int)len, rr);
        if (!l) continue;
        const uint8_t *s = reinterpret_cast<const uint8_t *>(shfl64(reinterpret_cast<uint64_t>(src), rr));
        char *w = reinterpret_cast<char *>(shfl64(reinterpret_cast<uint64_t>(dst), rr));
        const int m = __shfl(mode, rr);
        if (m >= 5) {
            const uint32_t *w32 = reinterpret_cast<const uint32_t *>(s);
            for (uint32_t q = (uint32_t)lane; q < l; q += 64) {
                const uint32_t qq = m == 6 ? l - 1 - q : q;
                const uint32_t code = (w32[qq >> 4] >> (30 - 2 * (qq & 15))) & 3;
                w[q] = (char)(((m == 5 ? fwd : comp) >> (8 * code)) & 0xff);
            }
            continue;
        }
        for (uint32_t q = (uint32_t)lane; q < l; q += 64) {
            const uint8_t c = (m == 2 || m == 4) ? s[l - 1 - q] : s[q];
            w[q] = m == 0 ? (char)c : (m <= 2 ? (char)(((m == 1 ? fwd : comp) >> (8 * (c & 7))) & 0xff) : (char)(33 + (((c >> 4) & 15) * 40) / 15));
        }
    }
}

__global__ void __launch_bounds__(256) k_sam_write(SamDev d, uint64_t k0, uint32_t n, const unsigned long long *__restrict__ at, char *__restrict__ out)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    SamRec r;
    r.reported = false; r.len = 0; r.i = 0; r.nml = 0; r.acc = false;
    if (j < n) r = sam_rec(d, k0 + j);
    const bool packed = d.bases == nullptr;
    const uint8_t *s = !r.reported ? nullptr : (packed ? reinterpret_cast<const uint8_t *>(d.pk_words + d.pk_wofs[r.i]) : d.bases + d.offs[r.i]);
    r.has_qual = wave_has_qual(d, (r.reported && !packed) ? d.offs[r.i] : 0, (r.reported && !packed) ? r.len : 0);
    // the lane writes its record's small fields and notes where the three bulk pieces go
    char *w_name = nullptr, *w_seq = nullptr, *w_qual = nullptr;
    int seq_mode = 1, qual_mode = 3;
    if (r.reported) {
        const bk_hit h = d.hits[r.i];
        char *w = out + at[j];
        w_name = w;
        w += r.nml;
        *w++ = '\t';
        w = put_num(w, r.flag);
        if (r.acc) {
            *w++ = '\t';
            w = put_str(w, d.ent_names + (size_t)(h.chrom_id - 1) * 81);
            *w++ = '\t';
            w = put_num(w, (long)h.match_loci + 1);
            *w++ = '\t'; *w++ = '2'; *w++ = '5'; *w++ = '5'; *w++ = '\t';
            w = put_num(w, h.match_len);
            *w++ = 'M';
            *w++ = '\t';
            *w++ = r.pnext < 0 ? '*' : '=';
            *w++ = '\t';
            w = put_num(w, r.pnext < 0 ? 0L : r.pnext + 1);
            *w++ = '\t';
            w = put_num(w, r.tlen);
            *w++ = '\t';
            w_seq = w;
            seq_mode = h.strand == '+' ? 1 : 2;
            w += r.len;
            *w++ = '\t';
            if (!r.has_qual) *w++ = '*';
            else { w_qual = w; qual_mode = h.strand == '+' ? 3 : 4; w += r.len; }
            *w++ = '\n';
        } else {
            w = put_str(w, "\t*\t0\t255\t");
            w = put_num(w, r.len);
            w = put_str(w, "M\t*\t0\t0\t");
            w_seq = w;
            w += r.len;
            *w++ = '\t';
            if (!r.has_qual) *w++ = '*';
            else { w_qual = w; w += r.len; }
            w = put_str(w, "\t\tYU:Z:");                           // the doubled TAB is what the reference writes
            w = put_str(w, kSamNarTag[h.nar < 20 ? h.nar : 0]);
            *w++ = '\n';
        }
    }
    wave_copy(r.reported ? reinterpret_cast<const uint8_t *>(d.names + d.name_ofs[r.i]) : nullptr, w_name, r.reported ? r.nml : 0, 0);
    wave_copy(s, w_seq, r.reported ? r.len : 0, packed ? seq_mode + 4 : seq_mode);
    wave_copy(s, w_qual, w_qual ? r.len : 0, qual_mode);
    if (packed && r.reported) {
        // the record's bases that are not a,c,g,t: every code prints as N (the few there are: the lane walks its own runs)
        uint32_t e = d.pk_efirst[r.i];
        if (e != kNoExc) {
            __threadfence_block();                         // (after the wave's letters, whichever lanes stored them)
            for (; e < d.n_pk_exc && d.pk_exc[e].read == r.i; e++) {
                const bk_nbase x = d.pk_exc[e];
                for (uint32_t t = 0; t <= x.run; t++) {
                    const uint32_t pp = (uint32_t)x.pos + t;
                    if (pp < r.len) w_seq[seq_mode == 2 ? r.len - 1 - pp : pp] = 'N';
                }
            }
        }
    }
}

// packed reads: lengths as 32-bit words, words per read (for the offsets' prefix sum), first exception per read
__global__ void __launch_bounds__(256) k_sam_unpack_lens(const uint16_t *__restrict__ lens16, uint64_t n, uint32_t *__restrict__ lens, unsigned long long *__restrict__ nwords)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t l = lens16[i];
        lens[i] = l;
        nwords[i] = (l + 15) / 16;
    }
}

__global__ void __launch_bounds__(256) k_sam_first_exc(const bk_nbase *__restrict__ exc, uint64_t n_exc, uint32_t *__restrict__ first)
{
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_exc; e += (uint64_t)gridDim.x * blockDim.x)
        if (e == 0 || exc[e - 1].read != exc[e].read) first[exc[e].read] = (uint32_t)e;
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

}  // namespace

// Records per slice.  A slice's text (150 bytes a record for 100-base reads) passes through two page-locked buffers, and page-locked memory
// costs 0.18 s per GB to make and about as much to give back: 2 M records keep the pair at 0.6 GB and still fill the device.
constexpr uint64_t kSamSliceRecords = 2u << 20;

// the read side of a job on the device: 1 byte/base reads (bases, offs) or the packed form (words, their offsets, exceptions), lengths, names
struct SamReads {
    DevBuf d_bases, d_offs, d_lens, d_names, d_nofs;
    DevBuf d_words, d_wofs, d_exc, d_efirst;
};

// what bk_sam_prepare() starts: the job's read-side arrays on their way to the device and the pinned text buffers
struct bk_sam_prep {
    bk_ctx *ctx = nullptr;
    SamReads rd;
    void *h_text[2] = {nullptr, nullptr};
    uint64_t cap_text = 0;
    uint64_t n_reads = 0, n_bases = 0, n_name_bytes = 0;
    const uint8_t *bases = nullptr;
    const uint32_t *pk_words = nullptr;
    int rc = BK_OK;
    std::thread worker;
    std::mutex join_mu;
    ~bk_sam_prep()
    {
        { std::lock_guard<std::mutex> lk(join_mu); if (worker.joinable()) worker.join(); }
        for (void *&p : h_text) if (p) { (void)hipHostFree(p); p = nullptr; }
    }
};

static bool sam_job_reads_ok(const bk_sam_job *job)
{
    if (!job->n_reads || !job->names || !job->name_ofs) return false;
    if (job->pk_words) return job->pk_lens16 != nullptr && (job->n_pk_exc == 0 || job->pk_exc != nullptr);
    return job->bases && job->offs && job->lens;
}

static int sam_upload_reads(bk_ctx *c, const bk_sam_job *job, SamReads &rd)
{
    const uint64_t nr = job->n_reads;
    if (rd.d_lens.alloc(nr * 4) != hipSuccess || rd.d_names.alloc(job->n_name_bytes + 16) != hipSuccess || rd.d_nofs.alloc((nr + 1) * 8) != hipSuccess) {
        (void)hipGetLastError();
        return BK_ERR_MEM;
    }
    if (bk::upload_host(rd.d_names.p, job->names, job->n_name_bytes, c->device) || bk::upload_host(rd.d_nofs.p, job->name_ofs, nr * 8, c->device)) return BK_ERR_INTERNAL;
    if (hipMemcpy((char *)rd.d_nofs.p + nr * 8, &job->n_name_bytes, 8, hipMemcpyHostToDevice) != hipSuccess) return BK_ERR_INTERNAL;
    if (!job->pk_words) {
        if (rd.d_bases.alloc(job->n_bases + 16) != hipSuccess || rd.d_offs.alloc(nr * 8) != hipSuccess) { (void)hipGetLastError(); return BK_ERR_MEM; }
        if (bk::upload_host(rd.d_bases.p, job->bases, job->n_bases, c->device) || bk::upload_host(rd.d_offs.p, job->offs, nr * 8, c->device) ||
            bk::upload_host(rd.d_lens.p, job->lens, nr * 4, c->device))
            return BK_ERR_INTERNAL;
        return BK_OK;
    }
    // (the exceptions are checked here, while the caller's arrays are certainly there: ascending (read, pos), inside their reads)
    for (uint64_t e = 0; e < job->n_pk_exc; e++) {
        const bk_nbase &x = job->pk_exc[e];
        if (x.read >= nr || (uint32_t)x.pos + x.run >= job->pk_lens16[x.read]) return BK_ERR_PARAMS;
        if (e && (job->pk_exc[e - 1].read > x.read || (job->pk_exc[e - 1].read == x.read && (uint32_t)job->pk_exc[e - 1].pos + job->pk_exc[e - 1].run >= x.pos))) return BK_ERR_PARAMS;
    }
    // packed reads: words, 16-bit lengths and exceptions travel (page-locked buffers: plain DMA); lengths, word offsets and every read's
    // first exception are made here, on a stream of this call's own
    DevBuf d_l16, d_nw, d_tmp;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return BK_ERR_INTERNAL; }
    struct Guard { hipStream_t s; ~Guard() { (void)hipStreamDestroy(s); } } guard{s};
    if (rd.d_words.alloc((job->n_pk_words + 16) * 4) != hipSuccess || rd.d_wofs.alloc((nr + 1) * 8) != hipSuccess || d_l16.alloc(nr * 2) != hipSuccess ||
        d_nw.alloc((nr + 1) * 8) != hipSuccess || rd.d_efirst.alloc(nr * 4) != hipSuccess || rd.d_exc.alloc((job->n_pk_exc + 1) * sizeof(bk_nbase)) != hipSuccess) {
        (void)hipGetLastError();
        return BK_ERR_MEM;
    }
    if (bk::upload_host(rd.d_words.p, job->pk_words, job->n_pk_words * 4, c->device) || bk::upload_host(d_l16.p, job->pk_lens16, nr * 2, c->device) ||
        (job->n_pk_exc && bk::upload_host(rd.d_exc.p, job->pk_exc, job->n_pk_exc * sizeof(bk_nbase), c->device)))
        return BK_ERR_INTERNAL;
    hipError_t e = hipMemsetAsync((char *)d_nw.p + nr * 8, 0, 8, s);
    if (e == hipSuccess) e = hipMemsetAsync(rd.d_efirst.p, 0xff, nr * 4, s);
    if (e != hipSuccess) return BK_ERR_INTERNAL;
    hipLaunchKernelGGL(k_sam_unpack_lens, dim3(4096), dim3(256), 0, s, d_l16.as<uint16_t>(), nr, rd.d_lens.as<uint32_t>(), d_nw.as<unsigned long long>());
    if (job->n_pk_exc) hipLaunchKernelGGL(k_sam_first_exc, dim3(1024), dim3(256), 0, s, rd.d_exc.as<bk_nbase>(), job->n_pk_exc, rd.d_efirst.as<uint32_t>());
    size_t tb = 0;
    e = bk::prim::exclusive_sum(nullptr, tb, d_nw.as<unsigned long long>(), rd.d_wofs.as<unsigned long long>(), (size_t)nr + 1, s);
    if (e == hipSuccess) e = d_tmp.alloc(tb + 256);
    if (e == hipSuccess) e = bk::prim::exclusive_sum(d_tmp.p, tb, d_nw.as<unsigned long long>(), rd.d_wofs.as<unsigned long long>(), (size_t)nr + 1, s);
    unsigned long long total = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&total, rd.d_wofs.as<unsigned long long>() + nr, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { (void)hipGetLastError(); return BK_ERR_INTERNAL; }
    if (total != job->n_pk_words) return BK_ERR_PARAMS;           // (the lengths do not add up to the words given)
    return BK_OK;
}

extern "C" int bk_sam_prepare(bk_ctx *c, const bk_sam_job *job, uint32_t text_bytes_per_record, bk_sam_prep **out)
{
    if (!c || !job || !out || !sam_job_reads_ok(job)) return BK_ERR_PARAMS;
    bk_sam_prep *p = new bk_sam_prep();
    p->ctx = c;
    p->n_reads = job->n_reads; p->n_bases = job->n_bases; p->n_name_bytes = job->n_name_bytes; p->bases = job->bases; p->pk_words = job->pk_words;
    const bk_sam_job j = *job;
    p->worker = std::thread([p, j, text_bytes_per_record]() {
        if (hipSetDevice(p->ctx->device) != hipSuccess) { p->rc = BK_ERR_INTERNAL; return; }
        p->rc = sam_upload_reads(p->ctx, &j, p->rd);
        if (p->rc == BK_OK && text_bytes_per_record) {
            const uint64_t slice = std::min<uint64_t>(j.n_reads, kSamSliceRecords);
            const uint64_t cap = slice * (uint64_t)text_bytes_per_record + (1u << 20);
            bool ok = true;
            for (int q = 0; q < 2 && ok; q++) ok = hipHostMalloc(&p->h_text[q], cap, hipHostMallocDefault) == hipSuccess;
            if (ok) p->cap_text = cap;
            else { (void)hipGetLastError(); for (void *&h : p->h_text) if (h) { (void)hipHostFree(h); h = nullptr; } }      // (the first slice will size them)
        }
    });
    *out = p;
    return BK_OK;
}

extern "C" void bk_sam_prep_free(bk_sam_prep *prep) { delete prep; }

extern "C" int bk_sam_prep_wait(bk_sam_prep *prep)
{
    if (!prep) return BK_ERR_PARAMS;
    std::lock_guard<std::mutex> lk(prep->join_mu);             // (the format call may be joining too)
    if (prep->worker.joinable()) prep->worker.join();
    return prep->rc;
}

extern "C" int bk_sam_format(bk_ctx *c, const bk_sam_job *job, bk_sam_sink sink, void *user, uint64_t *n_reported, uint64_t *n_bytes)
{
    if (!job) return BK_ERR_PARAMS;
    std::unique_ptr<bk_sam_prep> prep(job->prep);          // consumed whatever happens
    if (!c || !sink || !n_reported || !n_bytes) return BK_ERR_PARAMS;
    *n_reported = 0;
    *n_bytes = 0;
    if (!job->n_order) return BK_OK;
    if (!sam_job_reads_ok(job) || !job->hits || !job->order) return BK_ERR_PARAMS;
    if (job->pe_mode && (job->n_reads & 1)) return BK_ERR_PARAMS;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const uint64_t nr = job->n_reads;
    const uint32_t n_ent = (uint32_t)c->entries.size();
    const bool timing = bk::env::timing();
    auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
    double t_mark = now(), t_dev = 0, t_settle = 0, t_pin = 0;
    auto lap = [&](const char *what) { const double t = now(); if (timing) fprintf(stderr, "bk timing: bk_sam_format %-28s %7.1f ms\n", what, 1e3 * (t - t_mark)); t_mark = t; };
    // the read store, names, records and order travel to the device once (pageable memory: staged by a few threads)
    SamReads own;
    DevBuf d_hits, d_order, d_ent, d_bytes, d_at, d_tmp, d_cnt;
#define SAM_TRY(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); return BK_ERR_MEM; } } while (0)
    // the read-side arrays: already on the device when bk_sam_prepare() was given these reads, else they travel now
    if (prep) {
        { std::lock_guard<std::mutex> lk(prep->join_mu); if (prep->worker.joinable()) prep->worker.join(); }
        if (prep->ctx != c || prep->n_reads != nr || prep->n_bases != job->n_bases || prep->n_name_bytes != job->n_name_bytes || prep->bases != job->bases ||
            prep->pk_words != job->pk_words)
            return BK_ERR_PARAMS;
        if (prep->rc != BK_OK) prep.reset();
    }
    lap(prep ? "head start taken, waited" : "no head start");
    if (!prep) { int ru = sam_upload_reads(c, job, own); if (ru) return ru; }
    SamReads &rd = prep ? prep->rd : own;
    const bool packed = job->pk_words != nullptr;
    SAM_TRY(d_hits.alloc(nr * sizeof(bk_hit)));
    SAM_TRY(d_order.alloc(job->n_order * 4));
    SAM_TRY(d_ent.alloc((size_t)n_ent * 81));
    const uint32_t slice = (uint32_t)std::min<uint64_t>(job->n_order, kSamSliceRecords);
    SAM_TRY(d_bytes.alloc(((size_t)slice + 1) * 8));
    SAM_TRY(d_at.alloc(((size_t)slice + 1) * 8));
    SAM_TRY(d_cnt.alloc(16));
    size_t tb = 0;
    SAM_TRY(bk::prim::exclusive_sum(nullptr, tb, d_bytes.as<unsigned long long>(), d_at.as<unsigned long long>(), (size_t)slice + 1, s));
    SAM_TRY(d_tmp.alloc(tb + 256));
    if (bk::upload_host(d_hits.p, job->hits, nr * sizeof(bk_hit), c->device) || bk::upload_host(d_order.p, job->order, job->n_order * 4, c->device))
        return BK_ERR_INTERNAL;
    {
        std::vector<char> en((size_t)n_ent * 81);
        for (uint32_t e = 0; e < n_ent; e++) memcpy(&en[(size_t)e * 81], c->entries[e].name, 81);
        HIP_TRY(hipMemcpy(d_ent.p, en.data(), en.size(), hipMemcpyHostToDevice));
    }
    lap("uploads");
    SamDev d{};
    d.bases = packed ? nullptr : rd.d_bases.as<uint8_t>(); d.offs = rd.d_offs.as<uint64_t>(); d.lens = rd.d_lens.as<uint32_t>(); d.names = rd.d_names.as<char>();
    d.name_ofs = rd.d_nofs.as<uint64_t>(); d.hits = d_hits.as<bk_hit>(); d.order = d_order.as<uint32_t>(); d.ent_names = d_ent.as<char>();
    d.pk_words = rd.d_words.as<uint32_t>(); d.pk_wofs = rd.d_wofs.as<uint64_t>(); d.pk_exc = rd.d_exc.as<bk_nbase>(); d.pk_efirst = rd.d_efirst.as<uint32_t>();
    d.n_pk_exc = packed ? job->n_pk_exc : 0;
    d.n_ent = n_ent; d.fmt6 = job->report_unaligned ? 1 : 0; d.pe_mode = job->pe_mode;
    // does any base carry a score at all?  (one streaming pass; without scores - FASTA input, or -g3 - no record is scanned for them)
    DevBuf d_anyq;
    SAM_TRY(d_anyq.alloc(16));
    SAM_TRY(hipMemsetAsync(d_anyq.p, 0, 16, s));
    if (!packed) hipLaunchKernelGGL(k_sam_any_qual, dim3(4096), dim3(256), 0, s, d.bases, job->n_bases, d_anyq.as<uint32_t>());
    SAM_TRY(hipGetLastError());
    d.any_qual = d_anyq.as<uint32_t>();
    // Everything the device indexes with is checked here first (the command line passes consistent arrays; another caller of the ABI
    // must get BK_ERR_PARAMS, not an out-of-bounds device access): chrom ids name entries 1..n, order[] names reads, every read lies
    // inside the bases, names are '\0'-terminated stretches in ascending order inside the name bytes.  A few host threads, slices each.
    {
        const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(8, std::max(nr, job->n_order) >> 16));
        std::vector<int> bad(nt, 0);
        auto check = [&](unsigned t) {
            const uint64_t r0 = nr * t / nt, r1 = nr * (t + 1) / nt;
            for (uint64_t i = r0; i < r1; i++) {
                const bk_hit &h = job->hits[i];
                if (h.nar == BK_NAR_ACCEPTED && (h.chrom_id < 1 || h.chrom_id > n_ent)) bad[t] = 1;
                if (!packed) {
                    const uint64_t o = job->offs[i], l = job->lens[i];
                    if (o > job->n_bases || l > job->n_bases - o) bad[t] = 1;
                }                                          // (packed reads: the lengths were checked against the word count when they were uploaded)
                const uint64_t a = job->name_ofs[i], z = i + 1 < nr ? job->name_ofs[i + 1] : job->n_name_bytes;
                if (z > job->n_name_bytes || a >= z) bad[t] = 1;
            }
            const uint64_t k0 = job->n_order * t / nt, k1 = job->n_order * (t + 1) / nt;
            for (uint64_t k = k0; k < k1; k++)
                if (job->order[k] >= nr) bad[t] = 1;
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(check, t);
        check(0);
        for (auto &x : th) x.join();
        for (int b : bad) if (b) return BK_ERR_PARAMS;
    }
    lap("record check");
    void *h_text[2] = {nullptr, nullptr};
    DevBuf d_text[2];
    uint64_t cap_text = 0;
    if (prep && prep->cap_text) {                      // (the pinned text buffers came with the head start; a slice that outgrows them replaces them)
        for (int q = 0; q < 2; q++) { h_text[q] = prep->h_text[q]; prep->h_text[q] = nullptr; }
        cap_text = prep->cap_text;
    } else if (c->sam_text_cap) {                      // (.. or were left by the previous call)
        for (int q = 0; q < 2; q++) { h_text[q] = c->sam_text[q]; c->sam_text[q] = nullptr; }
        cap_text = c->sam_text_cap;
        c->sam_text_cap = 0;
    }
    if (cap_text && (d_text[0].alloc(cap_text) != hipSuccess || d_text[1].alloc(cap_text) != hipSuccess)) {
        (void)hipGetLastError();
        for (int q = 0; q < 2; q++) if (d_text[q].p) { (void)hipFree(d_text[q].p); d_text[q].p = nullptr; }
        for (void *&p : h_text) if (p) { (void)hipHostFree(p); p = nullptr; }
        cap_text = 0;
    }
    auto free_host = [&]() { for (void *&p : h_text) if (p) { (void)hipHostFree(p); p = nullptr; } };
    int rc = BK_OK;
    uint64_t total_rep = 0, total_bytes = 0;
    // A slice's text is written into one of two device buffers, copied back by the copy stream into the pinned buffer of the same number
    // and handed to the sink by a thread of its own (which waits for the copy): the device formats slice i + 1 while slice i crosses
    // PCIe and slice i - 1 is still being stored.  The sink is told where in the text a slice starts, so two of its calls may overlap.
    hipStream_t cs = nullptr;
    hipEvent_t ev_written[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
    bool ev_ok = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess;
    for (int q = 0; q < 2 && ev_ok; q++)
        ev_ok = hipEventCreateWithFlags(&ev_written[q], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ev_copied[q], hipEventDisableTiming) == hipSuccess;
    auto free_events = [&]() {
        for (int q = 0; q < 2; q++) { if (ev_written[q]) (void)hipEventDestroy(ev_written[q]); if (ev_copied[q]) (void)hipEventDestroy(ev_copied[q]); }
        if (cs) (void)hipStreamDestroy(cs);
    };
    if (!ev_ok) { (void)hipGetLastError(); free_events(); free_host(); return BK_ERR_INTERNAL; }
    std::thread sinker[2];
    int sink_rc[2] = {0, 0};
    auto settle = [&](int q) { if (sinker[q].joinable()) { sinker[q].join(); if (sink_rc[q] && rc == BK_OK) rc = sink_rc[q] < 0 ? BK_ERR_INTERNAL : BK_ERR_FILEACCESS; } };
    lap("text buffers, streams");
    for (uint64_t k0 = 0, si = 0; k0 < job->n_order && rc == BK_OK; k0 += slice, si++) {
        const uint32_t n = (uint32_t)std::min<uint64_t>(slice, job->n_order - k0);
        double t0 = now();
        hipError_t e = hipMemsetAsync(d_cnt.p, 0, 16, s);
        if (e == hipSuccess) e = hipMemsetAsync((char *)d_bytes.p + (size_t)n * 8, 0, 8, s);
        hipLaunchKernelGGL(k_sam_measure, dim3((n + 255) / 256), dim3(256), 0, s, d, k0, n, d_bytes.as<unsigned long long>(), d_cnt.as<uint32_t>());
        size_t t2 = tb + 256;
        if (e == hipSuccess) e = bk::prim::exclusive_sum(d_tmp.p, t2, d_bytes.as<unsigned long long>(), d_at.as<unsigned long long>(), (size_t)n + 1, s);
        unsigned long long bytes = 0;
        uint32_t rep = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&bytes, d_at.as<unsigned long long>() + n, 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(&rep, d_cnt.p, 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { rc = BK_ERR_INTERNAL; break; }
        t_dev += now() - t0;
        if (bytes > cap_text) {
            // (the text buffers follow the largest slice seen; the first slice sizes them for the run)
            const double tp = now();
            settle(0); settle(1);
            for (int q = 0; q < 2; q++) if (d_text[q].p) { (void)hipFree(d_text[q].p); d_text[q].p = nullptr; }
            free_host();
            cap_text = bytes + bytes / 8 + (1u << 20);
            bool ok = d_text[0].alloc(cap_text) == hipSuccess && d_text[1].alloc(cap_text) == hipSuccess;
            for (int q = 0; q < 2 && ok; q++) ok = hipHostMalloc(&h_text[q], cap_text, hipHostMallocDefault) == hipSuccess;
            if (!ok) { (void)hipGetLastError(); rc = BK_ERR_MEM; break; }
            t_pin += now() - tp;
        }
        const int q = (int)(si & 1);
        t0 = now();
        settle(q);                                      // (slice i - 2 is stored: both buffers of this number are free)
        t_settle += now() - t0;
        if (rc != BK_OK) break;
        t0 = now();
        hipLaunchKernelGGL(k_sam_write, dim3((n + 255) / 256), dim3(256), 0, s, d, k0, n, d_at.as<unsigned long long>(), d_text[q].as<char>());
        e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(ev_written[q], s);
        if (e == hipSuccess) e = hipStreamWaitEvent(cs, ev_written[q], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(h_text[q], d_text[q].p, bytes, hipMemcpyDeviceToHost, cs);
        if (e == hipSuccess) e = hipEventRecord(ev_copied[q], cs);
        if (e != hipSuccess) { rc = BK_ERR_INTERNAL; break; }
        t_dev += now() - t0;
        const uint64_t at = total_bytes;
        const int dev = c->device;
        sinker[q] = std::thread([&, q, bytes, at, dev]() {
            if (hipSetDevice(dev) != hipSuccess || hipEventSynchronize(ev_copied[q]) != hipSuccess) { (void)hipGetLastError(); sink_rc[q] = -1; return; }
            sink_rc[q] = bytes ? sink(user, reinterpret_cast<const char *>(h_text[q]), bytes, at) : 0;
        });
        total_rep += rep;
        total_bytes += bytes;
    }
    { const double t0 = now(); settle(0); settle(1); t_settle += now() - t0; }
    (void)hipStreamSynchronize(cs);
    lap("slices");
    free_events();
    if (rc == BK_OK && h_text[0] && h_text[1] && !c->sam_text_cap) {       // kept with the context for its next call
        for (int q = 0; q < 2; q++) { c->sam_text[q] = h_text[q]; h_text[q] = nullptr; }
        c->sam_text_cap = cap_text;
    }
    free_host();
    lap("text buffers put away");
    if (timing) fprintf(stderr, "bk timing: bk_sam_format slices: device (measure, scan, write enqueued) %.1f ms, text buffers %.1f ms, waiting for the sink %.1f ms\n",
                        1e3 * t_dev, 1e3 * t_pin, 1e3 * t_settle);
    *n_reported = total_rep;
    *n_bytes = total_bytes;
    return rc;
#undef SAM_TRY
}
