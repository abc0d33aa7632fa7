// PROBE: main loop of a 256x256 NT GEMM with FOUR waves (one per SIMD), each owning 128x128 outputs (256 accumulator
// registers), software-pipelined inside the wave: fragments of k-step s+1 are read while the 16 MFMAs of k-step s
// run, one barrier per K stage.  Against the product kernel (8 waves, 128x64 per wave, LOAD / COMPUTE phases of two
// wave groups, 4 barriers per stage): 8 instead of 12 fragment reads per 16 MFMAs, no phase hand-over.
// Question: does the main loop run faster than gemm_nt256w's (~1190 TFLOP/s on random data at K = 4096)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cmath>
#include "../../maest_amd/csrc/common.h"
using namespace maest;

constexpr int ROWB = 128, UNIT = 256 * ROWB, NBUF = 5, SMEM = NBUF * UNIT;
#define WAIT_VMCNT(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (((N) >> 4) << 14) | 0x0F70)

template <int VARIANT, int ABL = 0>
__global__ __launch_bounds__(256) void nt4w_kernel(const char* __restrict__ A, const char* __restrict__ B, float* __restrict__ C,
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

    const char* a_src[8];
    const char* b_src[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = (wave * 8 + i) * 8 + (lane >> 3);
        const int csrc = (lane & 7) ^ ((r >> 1) & 7);
        a_src[i] = A + (int64_t)(m0 + r) * lda + csrc * 16;
        b_src[i] = B + (int64_t)(n0 + r) * ldb + csrc * 16;
    }
    const int dma_off = wave * 8 * 1024;
    bool in_loop = false;
    auto issue = [&](int stage, bool is_b, int buf, int i) __attribute__((always_inline)) {
        if ((ABL & 1) && in_loop) return;
        const int sc = stage < nstages ? stage : nstages - 1;
        const char* src = (is_b ? b_src[i] : a_src[i]) + (int64_t)sc * ROWB;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + buf * UNIT + dma_off + i * 1024), 16, 0, 0);
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
        b_off[t] = rb * ROWB; b_swz[t] = (rb >> 1) & 7;
    }
    chunk16 fa[2][4], fb[2][4];
    auto load_frags = [&](int set, int abuf, int bbuf, int ks) __attribute__((always_inline)) {
        if ((ABL & 2) && in_loop) return;
        const char* la = smem + abuf * UNIT;
        const char* lb = smem + bbuf * UNIT;
        const int kc = 2 * ks + h;
        if (VARIANT >= 2) {     // asm-issued: the compiler's waitcnt pass does not see them; waits are placed by hand
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t ad = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(lb + b_off[t] + ((kc ^ b_swz[t]) << 4));
                asm volatile("ds_read_b128 %0, %1" : "=v"(fb[set][t]) : "v"(ad));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t ad = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(la + a_off[t] + ((kc ^ a_swz[t]) << 4));
                asm volatile("ds_read_b128 %0, %1" : "=v"(fa[set][t]) : "v"(ad));
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) fb[set][t] = *reinterpret_cast<const chunk16*>(lb + b_off[t] + ((kc ^ b_swz[t]) << 4));
#pragma unroll
        for (int t = 0; t < 4; ++t) fa[set][t] = *reinterpret_cast<const chunk16*>(la + a_off[t] + ((kc ^ a_swz[t]) << 4));
    };
    auto pin = [&](int set) __attribute__((always_inline)) {
        asm volatile("" : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fa[set][2]), "+v"(fa[set][3]),
                          "+v"(fb[set][0]), "+v"(fb[set][1]), "+v"(fb[set][2]), "+v"(fb[set][3]));
    };
    auto next = [](int b, int by) { b += by; return b >= NBUF ? b - NBUF : b; };

    // prologue: units A0 B0 A1 B1 A2 -> buffers 0..4
#pragma unroll
    for (int i = 0; i < 8; ++i) issue(0, false, 0, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) issue(0, true, 1, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) issue(1, false, 2, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) issue(1, true, 3, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) issue(2, false, 4, i);
    WAIT_VMCNT(24);
    __builtin_amdgcn_s_barrier();
    int abuf = 0, bbuf = 1;
    load_frags(0, abuf, bbuf, 0);
    in_loop = true;
    for (int j = 0; j < nstages; ++j) {
        const int abuf_n = next(abuf, 2), bbuf_n = next(bbuf, 2);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks < 3) {
                load_frags(nxt, abuf, bbuf, ks + 1);
                if (VARIANT >= 2) {
                    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");     // the set in use (8 reads, issued a k-step ago) is complete
                    pin(cur);
                }
            } else {
                // every wave has read stage j (its last fragment reads were issued a k-step ago): stage j's buffers are
                // free, and this wave's share of stage j + 1 has landed once only its newest unit (8 loads) is in flight
                __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
                WAIT_VMCNT(8);
                __builtin_amdgcn_s_barrier();
                if (VARIANT >= 3 && VARIANT < 10) {      // re-skew the four waves (SIMDs) by (VARIANT - 2) x 16 cycles each after every barrier
                    const int wv_s = __builtin_amdgcn_readfirstlane(wave);
                    for (int i = 0; i < wv_s * (VARIANT - 2); ++i) asm volatile("s_nop 15");
                }
                if (VARIANT >= 2) pin(cur);
                load_frags(nxt, abuf_n, bbuf_n, 0);
                if (VARIANT == 10) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) issue(j + 2, true, abuf, q);
#pragma unroll
                    for (int q = 0; q < 8; ++q) issue(j + 3, false, bbuf, q);
                }
                if (VARIANT == 11) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) issue(j + 2, true, abuf, q);
                }
            }
            if (VARIANT == 11 && ks == 1 && j > 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) issue(j + 2, false, next(bbuf, 3), q);
            }
            if (VARIANT == 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    mma_chunk<bf16_t>(acc[nt][mt], fb[cur][nt], fa[cur][mt]);
                    // refills: after the barrier of stage j (ks == 3) B_{j+2} -> A_j's buffer, then during ks 0..2 of
                    // stage j+1 the rest; 16 per stage and wave = 4 per k-step, one per four MFMAs
                    if (mt == 3 && VARIANT < 10) {
                        const int q = ((ks + 1) & 3) * 4 + nt;          // 0..15 in issue order starting at ks == 3
                        if (ks == 3) issue(j + 2, true, abuf, q);                 // q = 0..3  : B_{j+2} part 1
                        else if (ks == 0) { if (j > 0) issue(j + 1, true, next(abuf, 3), q); }   // q = 4..7: B_{(j-1)+2} part 2 -> A_{j-1}'s buffer
                        else if (ks == 1) { if (j > 0) issue(j + 2, false, next(bbuf, 3), q - 8); }   // A_{(j-1)+3} part 1 -> B_{j-1}'s buffer
                        else { if (j > 0) issue(j + 2, false, next(bbuf, 3), q - 8); }               // part 2
                    }
                }
            if (VARIANT == 0) __builtin_amdgcn_s_setprio(0);
        }
        abuf = abuf_n; bbuf = bbuf_n;
    }
    WAIT_VMCNT(0);
    if (clk != nullptr && threadIdx.x == 0) {      // shader cycles and 100 MHz wall ticks this workgroup lived
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
    hipFuncSetAttribute((const void*)&nt4w_kernel<VARIANT, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    const int grid = (M / 256) * (N / 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) nt4w_kernel<VARIANT, ABL><<<grid, 256, SMEM>>>((const char*)A, (const char*)B, C, M, N, K, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) nt4w_kernel<VARIANT, ABL><<<grid, 256, SMEM>>>((const char*)A, (const char*)B, C, M, N, K, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    static unsigned long long* dclk = nullptr;
    if (!dclk) hipMalloc(&dclk, (size_t)2 * 8192 * 8);
    nt4w_kernel<VARIANT, ABL><<<grid, 256, SMEM>>>((const char*)A, (const char*)B, C, M, N, K, 0, dclk);
    std::vector<unsigned long long> hc(2 * grid);
    hipMemcpy(hc.data(), dclk, hc.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < grid; ++i) { cyc += (double)hc[2 * i]; wall += (double)hc[2 * i + 1]; }
    printf("%-10s M=%6d N=%5d K=%5d: %8.3f ms  %7.1f TF/s (main loop only, no C write)   shader clock %.0f MHz\n", name, M, N, K, ms,
           2.0 * M * N * K / ms / 1e9, cyc / wall * 100.0);
}

#define CHECK(V) { const int M = 512, N = 512, K = 512; \
        hipFuncSetAttribute((const void*)&nt4w_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM); \
        hipMemset(C, 0, (size_t)M * N * 4); \
        nt4w_kernel<V><<<4, 256, SMEM>>>((const char*)A, (const char*)B, C, M, N, K, 1); \
        std::vector<float> hC((size_t)M * N); \
        hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost); \
        double worst = 0; \
        for (int t = 0; t < 2000; ++t) { \
            const int r = (t * 7919) % M, c = (t * 104729) % N; \
            double ref = 0; \
            for (int k = 0; k < K; ++k) ref += (double)bf(hA[(size_t)r * K + k]) * bf(hB[(size_t)c * K + k]); \
            worst = fmax(worst, fabs(ref - hC[(size_t)r * N + c])); \
        } \
        printf("check variant %d: max |err| = %.3e %s\n", V, worst, worst < 1e-3 ? "OK" : "WRONG"); }
int main() {
    const int Mmax = 65536, Nmax = 4096, Kmax = 4096;
    std::vector<uint16_t> hA((size_t)Mmax * Kmax), hB((size_t)Nmax * Kmax);
    fill(hA); fill(hB);
    void *A, *B; float* C;
    hipMalloc(&A, hA.size() * 2); hipMalloc(&B, hB.size() * 2); hipMalloc(&C, (size_t)2048 * 1024 * 4);
    hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    {   // correctness on a small problem: M = 512, N = 512, K = 512 (operands = the leading rows, pitch K)
        const int M = 512, N = 512, K = 512;
        hipFuncSetAttribute((const void*)&nt4w_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        nt4w_kernel<0><<<4, 256, SMEM>>>((const char*)A, (const char*)B, C, M, N, K, 1);
        std::vector<float> hC((size_t)M * N);
        hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int t = 0; t < 2000; ++t) {
            const int r = (t * 7919) % M, c = (t * 104729) % N;
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)bf(hA[(size_t)r * K + k]) * bf(hB[(size_t)c * K + k]);
            worst = fmax(worst, fabs(ref - hC[(size_t)r * N + c]));
        }
        printf("check 512^3: max |err| over 2000 samples = %.3e %s\n", worst, worst < 1e-3 ? "OK" : "WRONG");
    }
    {
        const int M = 512, N = 512, K = 512;
        hipFuncSetAttribute((const void*)&nt4w_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        hipMemset(C, 0, (size_t)M * N * 4);
        nt4w_kernel<2><<<4, 256, SMEM>>>((const char*)A, (const char*)B, C, M, N, K, 1);
        std::vector<float> hC((size_t)M * N);
        hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int t = 0; t < 2000; ++t) {
            const int r = (t * 7919) % M, c = (t * 104729) % N;
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)bf(hA[(size_t)r * K + k]) * bf(hB[(size_t)c * K + k]);
            worst = fmax(worst, fabs(ref - hC[(size_t)r * N + c]));
        }
        printf("check 512^3 (asm reads): max |err| = %.3e %s\n", worst, worst < 1e-3 ? "OK" : "WRONG");
    }
    run<2>("4w full", A, B, C, 65536, 4096, 4096);
    run<2, 1>("  no DMA", A, B, C, 65536, 4096, 4096);
    run<2, 2>("  no dsrd", A, B, C, 65536, 4096, 4096);
    run<2, 3>("  mfma only", A, B, C, 65536, 4096, 4096);
    hipMemset(A, 0, (size_t)65536 * 4096 * 2); hipMemset(B, 0, (size_t)4096 * 4096 * 2);
    run<2>("4w zeros", A, B, C, 65536, 4096, 4096);
    run<2, 3>("  mfma only zeros", A, B, C, 65536, 4096, 4096);
    return 0;
}
