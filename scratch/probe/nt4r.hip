// PROBE: main loop of a 256x256 NT GEMM with FOUR waves (one per SIMD, 128x128 outputs each) whose operand stages reach
// LDS through REGISTERS (global_load_dwordx4 -> VGPR -> ds_write_b128) instead of LDS-DMA.  nt4w.hip measured that the
// LDS-DMA refills cost a quarter of the main loop wherever they are issued ("no DMA" 1574 vs 1190 TF/s-equivalent); the
// question here: is that the price of moving 64 KiB per stage into LDS at all, or of the DMA path in particular?
//   two stage buffers of 64 KiB; during stage j (64 MFMAs per wave) a wave writes its share of stage j+1 (16 ds_write_b128
//   per lane, loaded during stage j-1) and re-loads the same registers with stage j+2; one barrier per stage.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cmath>
#include "../../maest_amd/csrc/common.h"
using namespace maest;

constexpr int ROWB = 128, UNIT = 256 * ROWB, STAGE = 2 * UNIT, SMEM = 2 * STAGE;

// ABL bit 0: no global loads in the loop, bit 1: no fragment reads, bit 2: no ds_writes
template <int VARIANT, int ABL = 0>
__global__ __launch_bounds__(256) void nt4r_kernel(const char* __restrict__ A, const char* __restrict__ B, float* __restrict__ C,
                                                   int M, int N, int K, int do_store, unsigned long long* clk = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, h = lane >> 5;
    const int tiles_n = N / 256, nwg = (M / 256) * tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int tile_m = wg / tiles_n, tile_n = wg - tile_m * tiles_n;
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const int nstages = K / 64;
    const int64_t lda = (int64_t)K * 2, ldb = (int64_t)K * 2;

    // staging map: slot q < 8 -> A rows q*32 + tid/8, slot q >= 8 -> B rows (q-8)*32 + tid/8; chunk tid%8 of the 128-byte row
    const int srow = tid >> 3, schunk = tid & 7;
    const char* a_g = A + (int64_t)(m0 + srow) * lda + schunk * 16;
    const char* b_g = B + (int64_t)(n0 + srow) * ldb + schunk * 16;
    const int st_off = srow * ROWB + ((schunk ^ ((srow >> 1) & 7)) << 4);       // + q*4096 (+ UNIT for B)
    chunk16 sreg[16];
    auto gload = [&](int q, int stage) __attribute__((always_inline)) {
        const int sc = stage < nstages ? stage : nstages - 1;
        const char* src = (q < 8 ? a_g + (int64_t)q * 32 * lda : b_g + (int64_t)(q - 8) * 32 * ldb) + (int64_t)sc * ROWB;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sreg[q]) : "v"(src) : "memory");
    };
    auto swrite = [&](int q, int buf) __attribute__((always_inline)) {
        const uint32_t ad = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + buf * STAGE + (q < 8 ? 0 : UNIT) + (q & 7) * 4096 + st_off);
        asm volatile("ds_write_b128 %0, %1" :: "v"(ad), "v"(sreg[q]) : "memory");
    };

    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    int a_off[4], b_off[4], a_swz[4], b_swz[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ra = wm * 128 + t * 32 + (lane & 31), rb = wn * 128 + t * 32 + (lane & 31);
        a_off[t] = ra * ROWB; a_swz[t] = (ra >> 1) & 7;
        b_off[t] = UNIT + rb * ROWB; b_swz[t] = (rb >> 1) & 7;
    }
    chunk16 fa[2][4], fb[2][4];
    bool in_loop = false;
    auto load_frags = [&](int set, int buf, int ks) __attribute__((always_inline)) {
        if ((ABL & 2) && in_loop) return;
        const char* st = smem + buf * STAGE;
        const int kc = 2 * ks + h;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t ad = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(st + b_off[t] + ((kc ^ b_swz[t]) << 4));
            asm volatile("ds_read_b128 %0, %1" : "=v"(fb[set][t]) : "v"(ad));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t ad = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(st + a_off[t] + ((kc ^ a_swz[t]) << 4));
            asm volatile("ds_read_b128 %0, %1" : "=v"(fa[set][t]) : "v"(ad));
        }
    };
    auto pin = [&](int set) __attribute__((always_inline)) {
        asm volatile("" : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fa[set][2]), "+v"(fa[set][3]),
                          "+v"(fb[set][0]), "+v"(fb[set][1]), "+v"(fb[set][2]), "+v"(fb[set][3]));
    };

    // prologue: stage 0 -> buffer 0 through the registers, stage 1 -> registers
#pragma unroll
    for (int q = 0; q < 16; ++q) gload(q, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 16; ++q) swrite(q, 0);
#pragma unroll
    for (int q = 0; q < 16; ++q) gload(q, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, 0);
    in_loop = true;
    for (int j = 0; j < nstages; ++j) {
        const int buf = j & 1, nbuf = buf ^ 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks < 3) {
                load_frags(nxt, buf, ks + 1);
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");      // the set in use (issued a k-step ago) and the older writes
                pin(cur);
            } else {
                // everybody has read stage j and written its share of stage j+1
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                pin(cur);
                load_frags(nxt, nbuf, 0);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    mma_chunk<bf16_t>(acc[nt][mt], fb[cur][nt], fa[cur][mt]);
                    if (mt == 3) {
                        // slot q of this stage: write register q (stage j+1, loaded a stage ago: 15 younger loads may fly),
                        // then re-load it with stage j+2.  The writes of k-step 3 would land AFTER the barrier above, so the
                        // slots are shifted: k-steps 3(prev iteration's tail is not available) -> use ks 0..2 for 16 slots:
                        // 6 + 5 + 5.
                    }
                    if (VARIANT == 0) {
                        // 16 slots over k-steps 0..2 (48 MFMAs): one slot per 3 MFMAs
                        const int m = ks * 16 + nt * 4 + mt;
                        if (ks < 3 && m % 3 == 2) {
                            const int q = m / 3;
                            if (!(ABL & 4)) {
                                if (!(ABL & 1)) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                                swrite(q, nbuf);
                            }
                            if (!(ABL & 1)) gload(q, j + 2);
                        }
                    }
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 16; ++q) asm volatile("" :: "v"(sreg[q]));
    if (clk != nullptr && threadIdx.x == 0) {
        clk[2 * blockIdx.x] = clock64() - c0;
        clk[2 * blockIdx.x + 1] = wall_clock64() - w0;
    }
    if (do_store) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = m0 + wm * 128 + mt * 32 + (lane & 31);
                    const int col = n0 + wn * 128 + nt * 32 + 8 * g + 4 * h;
                    *reinterpret_cast<float4*>(C + (int64_t)row * N + col) =
                        make_float4(acc[nt][mt][4 * g], acc[nt][mt][4 * g + 1], acc[nt][mt][4 * g + 2], acc[nt][mt][4 * g + 3]);
                }
    }
}

static void fill(std::vector<uint16_t>& h) {
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < h.size(); ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const float f = ((float)(s >> 40) / 8388608.0f) - 1.0f;
        uint32_t u; memcpy(&u, &f, 4);
        h[i] = (uint16_t)((u + 0x8000u) >> 16);
    }
}
static float bf(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }

template <int VARIANT, int ABL = 0>
static void run(const char* name, const void* A, const void* B, float* C, int M, int N, int K) {
    hipFuncSetAttribute((const void*)&nt4r_kernel<VARIANT, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    const int grid = (M / 256) * (N / 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) nt4r_kernel<VARIANT, ABL><<<grid, 256, SMEM>>>((const char*)A, (const char*)B, C, M, N, K, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) nt4r_kernel<VARIANT, ABL><<<grid, 256, SMEM>>>((const char*)A, (const char*)B, C, M, N, K, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    static unsigned long long* dclk = nullptr;
    if (!dclk) hipMalloc(&dclk, (size_t)2 * 8192 * 8);
    nt4r_kernel<VARIANT, ABL><<<grid, 256, SMEM>>>((const char*)A, (const char*)B, C, M, N, K, 0, dclk);
    std::vector<unsigned long long> hc(2 * grid);
    hipMemcpy(hc.data(), dclk, hc.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < grid; ++i) { cyc += (double)hc[2 * i]; wall += (double)hc[2 * i + 1]; }
    printf("%-18s M=%6d N=%5d K=%5d: %8.3f ms  %7.1f TF/s (main loop only, no C write)   shader clock %.0f MHz\n", name, M, N, K, ms,
           2.0 * M * N * K / ms / 1e9, cyc / wall * 100.0);
    fflush(stdout);
}

int main() {
    const int Mmax = 65536, Nmax = 4096, Kmax = 4096;
    std::vector<uint16_t> hA((size_t)Mmax * Kmax), hB((size_t)Nmax * Kmax);
    fill(hA); fill(hB);
    void *A, *B; float* C;
    hipMalloc(&A, hA.size() * 2); hipMalloc(&B, hB.size() * 2); hipMalloc(&C, (size_t)2048 * 1024 * 4);
    hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {   // correctness on a small problem (operands = the leading rows, pitch K), repeated: race screen
        const int M = 512, N = 512, K = 512;
        hipFuncSetAttribute((const void*)&nt4r_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        hipMemset(C, 0, (size_t)M * N * 4);
        nt4r_kernel<0><<<4, 256, SMEM>>>((const char*)A, (const char*)B, C, M, N, K, 1);
        std::vector<float> hC((size_t)M * N);
        hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int t = 0; t < 4000; ++t) {
            const int r = (t * 7919) % M, c = (t * 104729) % N;
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)bf(hA[(size_t)r * K + k]) * bf(hB[(size_t)c * K + k]);
            worst = fmax(worst, fabs(ref - hC[(size_t)r * N + c]));
        }
        printf("check 512^3: max |err| over 4000 samples = %.3e %s\n", worst, worst < 1e-3 ? "OK" : "WRONG");
    }
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("4r full", A, B, C, 65536, 4096, 4096);
        run<0, 1>("  no gload", A, B, C, 65536, 4096, 4096);
        run<0, 4>("  no dswrite", A, B, C, 65536, 4096, 4096);
        run<0, 5>("  no gload+dswrite", A, B, C, 65536, 4096, 4096);
        run<0, 2>("  no dsrd", A, B, C, 65536, 4096, 4096);
        run<0, 7>("  mfma only", A, B, C, 65536, 4096, 4096);
        run<0>("4r full K=768", A, B, C, 65536, 3072, 768);
    }
    return 0;
}
