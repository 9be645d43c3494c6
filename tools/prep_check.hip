// debug harness: k_prep_fused<16, true> (packed input) on n reads of 150 bases, rows checked against host code
#include "../biokanga_amd/csrc/bk_prep.hip"
#include <vector>
#include <random>
#include <cstdio>
using namespace bk;
int main(int argc, char **argv)
{
    const uint32_t n = argc > 1 ? atoi(argv[1]) : 100000;
    const int len = 150, NW = 16, wpr = 12;
    std::mt19937_64 rng(5);
    std::vector<uint32_t> words((size_t)n * 10 + 16), lens(n, len);
    std::vector<uint64_t> offs(n);
    std::vector<uint8_t> bases((size_t)n * len);
    for (auto &x : bases) x = rng() & 3;
    for (uint32_t r = 0; r < n; r++) {
        offs[r] = (uint64_t)r * 10;
        for (int w = 0; w < 10; w++) { uint32_t v = 0; for (int k = 0; k < 16 && 16 * w + k < len; k++) v |= (uint32_t)bases[(size_t)r * len + 16 * w + k] << (30 - 2 * k); words[(size_t)r * 10 + w] = v; }
    }
    DevBatch b{};
    uint32_t *d_words, *d_lens, *d_rmeta, *d_cnt, *d_stage; uint64_t *d_offs, *d_rd2, *d_rd4; bk_hit *d_out;
    hipMalloc(&d_words, words.size() * 4); hipMalloc(&d_lens, n * 4); hipMalloc(&d_rmeta, n * 4 + 8); hipMalloc(&d_offs, n * 8);
    hipMalloc(&d_rd2, (size_t)n * 2 * 8 * 8 + 64); hipMalloc(&d_rd4, (size_t)n * 2 * wpr * 8); hipMalloc(&d_out, (size_t)n * sizeof(bk_hit));
    hipMalloc(&d_cnt, 2 * kListStripes * 16 * 4); hipMalloc(&d_stage, ((size_t)n + 66 * 1024) * 4);
    hipMemcpy(d_words, words.data(), words.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_lens, lens.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_offs, offs.data(), n * 8, hipMemcpyHostToDevice); hipMemset(d_rmeta, 0, n * 4 + 8); hipMemset(d_cnt, 0, 2 * kListStripes * 16 * 4);
    hipMemset(d_rd2, 0xAB, (size_t)n * 2 * 8 * 8);
    b.offs = d_offs; b.lens = d_lens; b.pk_words = d_words; b.rd2 = d_rd2; b.rd4 = d_rd4; b.rmeta = d_rmeta; b.out = d_out; b.wpr = wpr; b.n_reads = n; b.nw = NW; b.iv_cores = 12;
    DevAlignCfg cfg{}; cfg.max_subs = 5; cfg.mm_delta = 1; cfg.max_ns = 1; cfg.max_hits = 1; cfg.min_core_len = 14; cfg.slides_per100 = 8; cfg.max_iter = 5000; cfg.heavy_thresh = 64;
    StripeSet out; out.cnt = d_cnt; out.stage[0] = out.stage[1] = out.stage[2] = d_stage; const unsigned blocks = (n + 255) / 256; out.cap = ((blocks + kListStripes - 1) / kListStripes) * 256;
    hipLaunchKernelGGL((k_prep_fused<16, true>), dim3(blocks), dim3(256), 0, 0, cfg, b, out);
    hipDeviceSynchronize();
    std::vector<uint64_t> rd2((size_t)n * 16);
    hipMemcpy(rd2.data(), d_rd2, rd2.size() * 8, hipMemcpyDeviceToHost);
    uint64_t badf = 0, badr = 0; uint32_t firstbad = ~0u;
    for (uint32_t r = 0; r < n; r++) {
        uint64_t f[8] = {0}, rc[8] = {0};
        for (int j = 0; j < len; j++) { f[j / 32] |= (uint64_t)bases[(size_t)r * len + j] << (62 - 2 * (j % 32)); rc[j / 32] |= (uint64_t)(3 - bases[(size_t)r * len + len - 1 - j]) << (62 - 2 * (j % 32)); }
        bool bf = false, br = false;
        for (int k = 0; k < 8; k++) { bf |= rd2[(size_t)r * 16 + k] != f[k]; br |= rd2[(size_t)r * 16 + 8 + k] != rc[k]; }
        badf += bf; badr += br;
        if ((bf || br) && firstbad == ~0u) { firstbad = r; printf("first bad read %u: fwd %d rc %d; rc words got %016llx %016llx want %016llx %016llx\n", r, bf, br, (unsigned long long)rd2[(size_t)r * 16 + 8], (unsigned long long)rd2[(size_t)r * 16 + 9], (unsigned long long)rc[0], (unsigned long long)rc[1]); }
    }
    printf("n %u: forward rows wrong %llu, reverse complement rows wrong %llu\n", n, (unsigned long long)badf, (unsigned long long)badr);
    return 0;
}
