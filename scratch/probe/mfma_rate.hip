// issue rate of the legacy K=8 bf16 MFMA (v_mfma_f32_32x32x8_bf16_1k: 4 bf16 per lane) on gfx950, next to the
// K=16 form and the exact-f32 form.  One wave per SIMD, 4 independent accumulators, N back-to-back MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    bf16x8 a8, b8; s16x4 a4, b4; float af = threadIdx.x * 1e-3f, bf = 1.0f;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(threadIdx.x * 0.001f + i); b8[i] = (__bf16)(1.0f); }
    for (int i = 0; i < 4; ++i) { a4[i] = (short)(0x3f80 + threadIdx.x + i); b4[i] = 0x3f80; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[i], 0, 0, 0);
            else if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
static void run(const char* name, double flop_per_mfma) {
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = 4.0 * iters;                      // MFMAs per wave
    const double tf = n * flop_per_mfma * 1024 / (ms * 1e-3) / 1e12;   // 1024 waves
    printf("%-28s %8.3f ms  %7.1f ns/MFMA/wave  -> %7.1f TFLOP/s chip (%.1f cyc @2.4GHz)\n", name, ms, ms * 1e6 / n, tf,
           ms * 1e-3 / n * 2.4e9);
    hipFree(out);
}
int main() {
    run<0>("mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16);
    run<1>("mfma_f32_32x32x8bf16_1k", 2.0 * 32 * 32 * 8);
    run<2>("mfma_f32_32x32x2f32", 2.0 * 32 * 32 * 2);
    return 0;
}
