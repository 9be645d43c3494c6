// lds_dma_probe - where global_load_lds_dword / _dwordx4 (gfx950) put each lane's data in LDS, with every lane active and with some masked
// off: LDS address = the (wave-uniform) base + lane * size, a masked lane's words stay as they were.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_probe(const uint32_t *__restrict__ src, uint32_t *__restrict__ out, unsigned long long mask)
{
    __shared__ uint32_t s4[64];
    __shared__ uint4 s16[64];
    const int lane = threadIdx.x & 63;
    s4[lane] = 0xAAAA0000u + lane;
    s16[lane] = make_uint4(0xBBBB0000u + lane, 0, 0, 0);
    __builtin_amdgcn_wave_barrier();
    if ((mask >> lane) & 1) {
        // every lane its own address: lane l reads word 1000 - 3 l, and the 16 bytes at 16 (200 - l)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (1000 - 3 * lane)), (__attribute__((address_space(3))) void *)s4, 4, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 4 * (200 - lane)), (__attribute__((address_space(3))) void *)s16, 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
    __builtin_amdgcn_wave_barrier();
    out[lane] = s4[lane];
    const uint4 v = s16[lane];
    out[64 + 4 * lane] = v.x; out[64 + 4 * lane + 1] = v.y; out[64 + 4 * lane + 2] = v.z; out[64 + 4 * lane + 3] = v.w;
}

int main()
{
    std::vector<uint32_t> h(4096);
    for (int i = 0; i < 4096; i++) h[i] = 0x10000u + i;
    uint32_t *d_src, *d_out;
    hipMalloc(&d_src, 4096 * 4); hipMalloc(&d_out, 320 * 4);
    hipMemcpy(d_src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    int bad = 0;
    for (unsigned long long mask : {~0ULL, 0x00000000FFFF00FFULL, 0x8000000000000001ULL}) {
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d_src, d_out, mask);
        std::vector<uint32_t> o(320);
        hipMemcpy(o.data(), d_out, 320 * 4, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l++) {
            const bool on = (mask >> l) & 1;
            const uint32_t e4 = on ? 0x10000u + (1000 - 3 * l) : 0xAAAA0000u + l;
            if (o[l] != e4) { if (bad < 8) printf("mask %llx lane %d dword: got %x expected %x\n", mask, l, o[l], e4); bad++; }
            for (int q = 0; q < 4; q++) {
                const uint32_t e = on ? 0x10000u + 4 * (200 - l) + q : (q == 0 ? 0xBBBB0000u + l : 0u);
                if (o[64 + 4 * l + q] != e) { if (bad < 8) printf("mask %llx lane %d dwordx4 word %d: got %x expected %x\n", mask, l, q, o[64 + 4 * l + q], e); bad++; }
            }
        }
    }
    printf("lds_dma_probe: %s (%d mismatches)\n", bad ? "FAILED" : "ok: base + lane * size, masked lanes untouched", bad);
    return bad != 0;
}
