// Does the MFMA shape change what the chip's power limit lets the matrix pipe sustain?  Back-to-back bf16 MFMAs on random register operands,
// one wave per SIMD (256-thread workgroups, one per CU, 512 registers budget not needed here), no memory traffic in the loop:
//   32x32x16: 16 independent accumulator blocks (256 fp32 per lane), 16 MFMAs per k-step, fragments A[4] B[4]
//   16x16x32: 64 independent accumulator blocks (256 fp32 per lane), 64 MFMAs per step, fragments A[8] B[8]
// Both = 128 x 128 outputs per wave and the same flops per step pair.  Reports sustained TFLOP/s over ~0.3 s per shape, alternating.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <utility>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// fixed registers, every instruction an asm statement (hipcc otherwise parks part of the 16x16 accumulators in VGPRs and shuffles them in the loop):
// accumulators a0 .. a255, fragments v[100 + 4 i ..+3]
template <int I> __device__ __forceinline__ void zero1() { asm volatile("v_accvgpr_write_b32 a%c0, 0" : : "i"(I)); }
template <int... I> __device__ __forceinline__ void zero_all(std::integer_sequence<int, I...>) { (zero1<I>(), ...); }
template <int I> __device__ __forceinline__ void ldfrag(const bf16x8* p) { asm volatile("global_load_dwordx4 v[%c1:%c2], %0, off" : : "v"(p), "i"(100 + 4 * I), "i"(103 + 4 * I) : "memory"); }
template <int Q> __device__ __forceinline__ void m32() {
    constexpr int N = Q >> 2, M = Q & 3;
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], v[%c2:%c3], v[%c4:%c5], a[%c0:%c1]" : : "i"(16 * Q), "i"(16 * Q + 15), "i"(116 + 4 * N), "i"(119 + 4 * N), "i"(100 + 4 * M), "i"(103 + 4 * M));
}
template <int Q> __device__ __forceinline__ void m16() {
    constexpr int N = Q >> 3, M = Q & 7;
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c0:%c1], v[%c2:%c3], v[%c4:%c5], a[%c0:%c1]" : : "i"(4 * Q), "i"(4 * Q + 3), "i"(132 + 4 * N), "i"(135 + 4 * N), "i"(100 + 4 * M), "i"(103 + 4 * M));
}
template <int... Q> __device__ __forceinline__ void step32(std::integer_sequence<int, Q...>) { (m32<Q>(), ...); }
template <int... Q> __device__ __forceinline__ void step16(std::integer_sequence<int, Q...>) { (m16<Q>(), ...); }
template <int SHAPE>
__global__ __launch_bounds__(256, 1) void kk(const bf16x8* src, float* out, int iters) {
    asm volatile("" : : : "a0", "a255", "v100", "v163");
    zero_all(std::make_integer_sequence<int, 256>{});
    const bf16x8* p = src + threadIdx.x;
    ldfrag<0>(p); ldfrag<1>(p + 256); ldfrag<2>(p + 512); ldfrag<3>(p + 768); ldfrag<4>(p + 1024); ldfrag<5>(p + 1280); ldfrag<6>(p + 1536); ldfrag<7>(p + 1792);
    ldfrag<8>(p + 2048); ldfrag<9>(p + 2304); ldfrag<10>(p + 2560); ldfrag<11>(p + 2816); ldfrag<12>(p + 3072); ldfrag<13>(p + 3328); ldfrag<14>(p + 3584); ldfrag<15>(p + 3840);
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 32) step32(std::make_integer_sequence<int, 16>{});
        else step16(std::make_integer_sequence<int, 64>{});
    }
    float x;
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_accvgpr_read_b32 %0, a0" : "=v"(x));
    out[blockIdx.x * 256 + threadIdx.x] = x;
}
int main(int argc, char** argv) {
    const int zero = argc > 1 && atoi(argv[1]) == 1;
    std::vector<unsigned short> h(4096 * 8);
    srand(1);
    for (auto& x : h) { float f = zero ? 0.f : ((rand() / (float)RAND_MAX) - 0.5f) * 0.02f; unsigned u; memcpy(&u, &f, 4); x = (unsigned short)(u >> 16); }
    bf16x8* d; float* o;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, 256 * 256 * 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // per iteration per wave: 32x32x16: 16 MFMAs x 32768 flops; 16x16x32: 64 MFMAs x 16384 flops -> 16x16 does 2x the flops per iteration
    const int it32 = 400000, it16 = 200000;
    for (int rep = 0; rep < 3; ++rep) {
        for (int shape = 0; shape < 2; ++shape) {
            hipEventRecord(e0);
            if (shape == 0) hipLaunchKernelGGL(kk<32>, dim3(256), dim3(256), 0, 0, d, o, it32);
            else hipLaunchKernelGGL(kk<16>, dim3(256), dim3(256), 0, 0, d, o, it16);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fl = 256.0 * 4 * (shape == 0 ? (double)it32 * 16 * 32768 : (double)it16 * 64 * 16384);
            printf("%s %s: %.1f ms  %.1f TFLOP/s\n", zero ? "zeros " : "random", shape == 0 ? "32x32x16" : "16x16x32", ms, fl / ms / 1e9);
        }
    }
    return 0;
}
