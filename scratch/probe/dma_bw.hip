// L2 -> LDS (LDS-DMA) and L2 -> VGPR bandwidth per CU by access pattern.  One 512-thread WG per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define WAIT_VMCNT(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (((N) >> 4) << 14) | 0x0F70)

// SEG = contiguous bytes per row segment fetched by one instruction (64, 128, 512, 1024); rows are `pitch` apart
template <int SEG, bool TO_LDS>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, size_t region, int pitch, int iters, u4* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int LPS = SEG / 16;            // lanes per segment
    constexpr int ROWS = 64 / LPS;           // rows per instruction
    const char* base = src + (size_t)blockIdx.x * region;
    // this wave's instruction j covers rows (j*8 + wave)*ROWS .. ; wraps inside the region
    const int rows_in_region = (int)(region / pitch);
    u4 acc = {0, 0, 0, 0};
    char* lds = smem + wave * 4 * 1024;
    int row = wave * ROWS + lane / LPS;
    const int col = (lane % LPS) * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const char* p = base + (size_t)row * pitch + col;
            if (TO_LDS)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(lds + j * 1024), 16, 0, 0);
            else
                acc += *(const u4*)p;
            row += 8 * ROWS;
            if (row >= rows_in_region) row -= rows_in_region;
        }
        if (TO_LDS) WAIT_VMCNT(8);
    }
    WAIT_VMCNT(0);
    if (acc.x == 0x12345u) sink[0] = acc;
}
template <int SEG, bool TO_LDS>
void run(const char* name, const char* src, size_t region, int pitch, u4* sink) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)&k<SEG, TO_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    k<SEG, TO_LDS><<<256, 512, 65536>>>(src, region, pitch, 100, sink);
    hipEventRecord(e0);
    k<SEG, TO_LDS><<<256, 512, 65536>>>(src, region, pitch, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bytes = 256.0 * 8 * iters * 4 * 1024;
    printf("%-44s %7.2f TB/s  %6.1f B/clk/CU (2.4 GHz)\n", name, bytes / (ms * 1e-3) / 1e12, bytes / 256 / (ms * 1e-3 * 2.4e9));
}
int main() {
    size_t total = (size_t)1 << 30;
    char* src; hipMalloc(&src, total); hipMemset(src, 1, total);
    u4* sink; hipMalloc(&sink, 64);
    for (size_t region : {(size_t)64 << 10, (size_t)1 << 20, (size_t)4 << 20}) {
        printf("-- region per WG %zu KB (%s)\n", region >> 10, region <= (64 << 10) ? "L2 resident" : region <= (1 << 20) ? "256 MB total: MALL" : "1 GB total: HBM");
        run<64, true>("LDS-DMA 16 rows x 64 B, pitch 1536", src, region, 1536, sink);
        run<128, true>("LDS-DMA 8 rows x 128 B, pitch 1536", src, region, 1536, sink);
        run<512, true>("LDS-DMA 2 rows x 512 B, pitch 1536", src, region, 1536, sink);
        run<1024, true>("LDS-DMA 1 KB contiguous, pitch 1024", src, region, 1024, sink);
        run<64, false>("VGPR load 16 rows x 64 B, pitch 1536", src, region, 1536, sink);
        run<128, false>("VGPR load 8 rows x 128 B, pitch 1536", src, region, 1536, sink);
        run<1024, false>("VGPR load 1 KB contiguous", src, region, 1024, sink);
    }
    return 0;
}
