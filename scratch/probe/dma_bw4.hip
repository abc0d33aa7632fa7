// L2 -> LDS (LDS-DMA) rate per CU as a function of the row PITCH and of the contiguous bytes per row segment.
// Question: is the ~34 B/clk/CU of "8 rows x 128 B" a property of the instruction shape, or of the pitch (rows of a
// K = 768 bf16 matrix are 1536 B = 12 lines apart: only 4 of 16 L2 channels if channels interleave by line)?
// One 512-thread WG per CU, each looping over its own 512-row region (L2 resident: 512 x SEG bytes per WG).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define WAIT_VMCNT(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (((N) >> 4) << 14) | 0x0F70)
template <int SEG, int SWZ>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, size_t wg_stride, int pitch, int iters, int shared) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int LPS = SEG / 16, ROWS = 64 / LPS;
    const char* base = src + (size_t)(shared ? (blockIdx.x / shared) : blockIdx.x) * wg_stride;
    char* lds = smem + wave * 4 * 1024;
    int row = wave * ROWS + lane / LPS;
    int col = (lane % LPS) * 16;
    if (SWZ == 1) col = ((lane % LPS) ^ ((row >> 1) & 7 & (LPS - 1))) * 16;      // the GEMM's source-side bank swizzle
    if (SWZ == 2) col = ((LPS - 1) - (lane % LPS)) * 16;                          // lanes run backwards through the segment
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const char* p = base + (size_t)row * pitch + col;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                             (__attribute__((address_space(3))) void*)(lds + j * 1024), 16, 0, 0);
            row += 8 * ROWS;
            if (row >= 512) row -= 512;
        }
        WAIT_VMCNT(8);
    }
    WAIT_VMCNT(0);
}
template <int SEG, int SWZ = 0>
void run(const char* src, int pitch, int shared) {
    const int iters = 4000;
    const size_t wg_stride = (size_t)512 * pitch + 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)&k<SEG, SWZ>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    k<SEG, SWZ><<<256, 512, 65536>>>(src, wg_stride, pitch, 200, shared);
    hipEventRecord(e0);
    k<SEG, SWZ><<<256, 512, 65536>>>(src, wg_stride, pitch, iters, shared);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bytes = 256.0 * 8 * iters * 4 * 1024;
    printf("swz %d seg %4d B x %2d rows/instr  pitch %5d  share %d : %6.2f TB/s  %5.1f B/clk/CU @2.4GHz  %5.1f clk/instr/CU\n", SWZ, SEG, 1024 / SEG, pitch,
           shared, bytes / (ms * 1e-3) / 1e12, bytes / 256 / (ms * 1e-3 * 2.4e9), (ms * 1e-3 * 2.4e9) / (8.0 * iters * 4));
}
int main() {
    size_t total = (size_t)1 << 31;
    char* src; hipMalloc(&src, total); hipMemset(src, 1, total);
    for (int shared : {0, 8}) {
        for (int pitch : {1536, 6144}) {
            run<128, 0>(src, pitch, shared);
            run<128, 1>(src, pitch, shared);
            run<128, 2>(src, pitch, shared);
        }
    }
    return 0;
}
