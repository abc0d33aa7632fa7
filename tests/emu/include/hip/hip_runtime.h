// SIMT lockstep emulator -- TEST INFRASTRUCTURE ONLY (never shipped, never loaded by maest_amd).
//
// This header shadows <hip/hip_runtime.h> when the kernel sources under maest_amd/csrc are
// compiled for the HOST (x86, clang) by tests/emu/build_emu.py.  It lets the `-m "not gpu"`
// tests execute the *same* kernel source text on host threads at tiny shapes to check index
// arithmetic, LDS layouts, MFMA fragment bookkeeping and barrier placement without a GPU.
// It models: a grid of blocks executed one after another; one OS thread per work-item;
// __syncthreads as a block barrier; a single global `smem` array as the block's LDS;
// wave64 collectives (MFMA 32x32x16 bf16, 32x32x2 f32, 16x16x32 bf16, 16x16x4 f32, shuffles)
// implemented through a per-wave staging area with the documented gfx950 lane<->element maps.
// It does NOT model timing, bank conflicts, caches, or memory-ordering hazards.
#pragma once
#include <pthread.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { unsigned x, y; };
struct __attribute__((aligned(16))) uint4 { unsigned x, y, z, w; };
struct __attribute__((aligned(8))) float2 { float x, y; };
struct __attribute__((aligned(16))) float4 { float x, y, z, w; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }

typedef int hipError_t;
typedef void* hipStream_t;
#define hipSuccess 0
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
static inline hipError_t hipPeekAtLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
template <typename F>
static inline hipError_t hipFuncSetAttribute(F, int, int) { return 0; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3

// ---------------------------------------------------------------------------------------
// execution context
// ---------------------------------------------------------------------------------------
extern thread_local dim3 threadIdx;
extern thread_local dim3 blockIdx;
extern dim3 blockDim;
extern dim3 gridDim;

namespace emu {
struct Wave {
    pthread_barrier_t bar;
    float A[64][8];
    float B[64][8];
    uint32_t X[64];
};
extern pthread_barrier_t block_bar;
extern Wave* waves;
extern thread_local int tid_linear;
inline Wave& wave() { return waves[tid_linear >> 6]; }
inline int lane() { return tid_linear & 63; }
inline void wave_sync() { pthread_barrier_wait(&wave().bar); }
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
}  // namespace emu

static inline void __syncthreads() { pthread_barrier_wait(&emu::block_bar); }

#define hipLaunchKernelGGL(kernel, grid, block, smem_bytes, stream, ...) \
    emu::launch(dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); })

// ---------------------------------------------------------------------------------------
// wave collectives
// ---------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 emu_bf16x8;
typedef __attribute__((ext_vector_type(16))) float emu_f32x16;
typedef __attribute__((ext_vector_type(4))) float emu_f32x4;

static inline emu_f32x16 emu_mfma_f32_32x32x16_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x16 c, int, int, int) {
    emu::Wave& W = emu::wave();
    const int l = emu::lane();
    for (int j = 0; j < 8; ++j) { W.A[l][j] = (float)a[j]; W.B[l][j] = (float)b[j]; }
    emu::wave_sync();
    const int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float s = c[r];
        for (int k = 0; k < 16; ++k)  // A[row][k] lives in lane row+32*(k/8), element k%8
            s += W.A[row + 32 * (k >> 3)][k & 7] * W.B[col + 32 * (k >> 3)][k & 7];
        c[r] = s;
    }
    emu::wave_sync();
    return c;
}
static inline emu_f32x16 emu_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c, int, int, int) {
    emu::Wave& W = emu::wave();
    const int l = emu::lane();
    W.A[l][0] = a; W.B[l][0] = b;
    emu::wave_sync();
    const int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float s = c[r];
        for (int k = 0; k < 2; ++k) s = fmaf(W.A[row + 32 * k][0], W.B[col + 32 * k][0], s);
        c[r] = s;
    }
    emu::wave_sync();
    return c;
}
static inline emu_f32x4 emu_mfma_f32_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c, int, int, int) {
    emu::Wave& W = emu::wave();
    const int l = emu::lane();
    for (int j = 0; j < 8; ++j) { W.A[l][j] = (float)a[j]; W.B[l][j] = (float)b[j]; }
    emu::wave_sync();
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) * 4 + r;
        float s = c[r];
        for (int k = 0; k < 32; ++k)  // A[row][k] lives in lane row+16*(k/8), element k%8
            s += W.A[row + 16 * (k >> 3)][k & 7] * W.B[col + 16 * (k >> 3)][k & 7];
        c[r] = s;
    }
    emu::wave_sync();
    return c;
}
static inline emu_f32x4 emu_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
    emu::Wave& W = emu::wave();
    const int l = emu::lane();
    W.A[l][0] = a; W.B[l][0] = b;
    emu::wave_sync();
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) * 4 + r;
        float s = c[r];
        for (int k = 0; k < 4; ++k) s = fmaf(W.A[row + 16 * k][0], W.B[col + 16 * k][0], s);
        c[r] = s;
    }
    emu::wave_sync();
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 emu_mfma_f32_32x32x16_bf16
#define __builtin_amdgcn_mfma_f32_32x32x2f32 emu_mfma_f32_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 emu_mfma_f32_16x16x32_bf16
#define __builtin_amdgcn_mfma_f32_16x16x4f32 emu_mfma_f32_16x16x4f32

// ds_read_b64_tr_b16 (gfx950), semantics probed on hardware (scratch/probe/tr_probe.hip): within each
// 16-lane group, lane q supplies the address of 4 contiguous b16 of row q/4 (column chunk q%4) of a
// 4x16 block (any row stride); it receives column q of rows 0..3: out[j] = loaded[lane 4j + q/4][q%4].
typedef short emu_v4i16 __attribute__((ext_vector_type(4)));
static inline emu_v4i16 emu_ds_read_tr16_b64(const void* p) {
    emu::Wave& W = emu::wave();
    const int l = emu::lane();
    const uint16_t* src = reinterpret_cast<const uint16_t*>(p);
    for (int e = 0; e < 4; ++e) W.A[l][e] = (float)src[e];   // b16 payload fits a float exactly
    emu::wave_sync();
    const int g = l & ~15, q = l & 15;
    emu_v4i16 r;
    for (int j = 0; j < 4; ++j) r[j] = (short)(uint16_t)W.A[g + 4 * j + (q >> 2)][q & 3];
    emu::wave_sync();
    return r;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) emu_ds_read_tr16_b64((const void*)(uintptr_t)(p))

// global_load_lds_dwordx4 (LDS-DMA): LDS destination = wave-uniform base + lane * size.  The emulator
// performs the copy immediately (it cannot model the asynchrony; races are a GPU-test concern).
static inline void emu_global_load_lds(const void* g, void* l, unsigned size, int offset, unsigned) {
    memcpy(reinterpret_cast<char*>(l) + offset + emu::lane() * size, reinterpret_cast<const char*>(g) + offset, size);
}
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) \
    emu_global_load_lds((const void*)(uintptr_t)(g), (void*)(uintptr_t)(l), size, off, aux)

template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    static_assert(sizeof(T) == 4, "emu shuffle: 32-bit types only");
    emu::Wave& W = emu::wave();
    const int l = emu::lane();
    uint32_t u;
    memcpy(&u, &v, 4);
    W.X[l] = u;
    emu::wave_sync();
    uint32_t r = W.X[(l ^ mask) & 63];
    emu::wave_sync();
    T out;
    memcpy(&out, &r, 4);
    return out;
}
template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
    static_assert(sizeof(T) == 4, "emu shuffle: 32-bit types only");
    emu::Wave& W = emu::wave();
    const int l = emu::lane();
    uint32_t u;
    memcpy(&u, &v, 4);
    W.X[l] = u;
    emu::wave_sync();
    uint32_t r = W.X[src & 63];
    emu::wave_sync();
    T out;
    memcpy(&out, &r, 4);
    return out;
}

// v_permlane32_swap_b32 vdst, src: lanes 32-63 of vdst swap with lanes 0-31 of src; returns {new vdst, new src}
struct emu_u32x2 {
    uint32_t v[2];
    uint32_t operator[](int i) const { return v[i]; }
};
static inline emu_u32x2 __builtin_amdgcn_permlane32_swap(uint32_t vdst, uint32_t src, bool, bool) {
    emu::Wave& W = emu::wave();
    const int l = emu::lane();
    W.X[l] = vdst;
    emu::wave_sync();
    const uint32_t upper_vdst = W.X[(l & 31) + 32];
    emu::wave_sync();
    W.X[l] = src;
    emu::wave_sync();
    const uint32_t lower_src = W.X[l & 31];
    emu::wave_sync();
    emu_u32x2 r;
    if (l < 32) { r.v[0] = vdst; r.v[1] = upper_vdst; }
    else { r.v[0] = lower_src; r.v[1] = src; }
    return r;
}

static inline float atomicAdd(float* p, float v) {
    std::atomic_ref<float> a(*p);
    float old = a.load();
    while (!a.compare_exchange_weak(old, old + v)) {}
    return old;
}
static inline float unsafeAtomicAdd(float* p, float v) { return atomicAdd(p, v); }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }

#define __expf expf
#define __logf logf
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline float __builtin_amdgcn_logf(float x) { return log2f(x); }
static inline int __builtin_amdgcn_readfirstlane(int x) { return x; }
static inline uint64_t __builtin_amdgcn_s_memtime() { static thread_local uint64_t t = 0; return t += 1000; }
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline void __builtin_amdgcn_s_setprio(int) {}
// wave-local LDS hand-off (csrc/mel.hip: wave_lds_sync): one host thread per lane, so it needs a real wave barrier
static inline void __builtin_amdgcn_wave_barrier() { emu::wave_sync(); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
static inline void __builtin_amdgcn_s_waitcnt(int) {}
static inline void __builtin_amdgcn_s_barrier() { __syncthreads(); }
static inline void __builtin_amdgcn_sched_barrier(int) {}
