// maest_amd device-side common definitions (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/maest_hip.h"

namespace maest {

// ------------------------------------------------------------------ element types
typedef uint16_t bf16_t;  // raw 16-bit operand bits; all conversions are explicit

// THE 16-BIT FLAVOUR.  The library is built twice from these sources (maest_amd/build.py): libmaest_hip.so, in which the 16-bit operand type
// (dtype code MAEST_BF16, `bf16_t` here) is bfloat16 -- the training mode --, and libmaest_hip_f16.so (-DMAEST_16BIT_F16), in which the same code
// paths run on IEEE half: v_mfma_*_f16, v_cvt_pk_f16_f32, the same layouts and schedules.  fp16 has 11 significand bits against 8: logits
// 6e-4 .. 8e-4 from fp32 where bf16 is at 5e-3 .. 8e-3 (scratch/fp16_operand_sim.py, confirmed on the GPU), i.e. INSIDE north_star's 1e-3, at the
// bf16 kernels' speed -- and it is the reference's own GPU arithmetic (precision="16-mixed", ex_maest.py:51).  precision="fp16": evaluation
// forwards, and training steps under a scaled loss (gradients in half underflow without one).  Everything that interprets the 16 bits goes through
// the few definitions below (and MAEST_T16 in the asm mnemonics of gemm_nt_ow.h / attn_fwd_pw.hip).
#ifdef MAEST_16BIT_F16
#define MAEST_T16 "f16"
#define MAEST_ONE16X2 0x3c003c00u          // two 16-bit ones (colsum / row-sum operands)
#define MAEST_ONE16X2_STR "0x3c003c00"
typedef _Float16 native16_t;
#define MAEST_MFMA_32X32X16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define MAEST_MFMA_16X16X32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#else
#define MAEST_T16 "bf16"
#define MAEST_ONE16X2 0x3f803f80u
#define MAEST_ONE16X2_STR "0x3f803f80"
typedef __bf16 native16_t;
#define MAEST_MFMA_32X32X16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define MAEST_MFMA_16X16X32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#endif

typedef __attribute__((ext_vector_type(8))) native16_t bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ float bf2f(bf16_t h) {
#ifdef MAEST_16BIT_F16
    return (float)__builtin_bit_cast(_Float16, h);
#else
    uint32_t u = ((uint32_t)h) << 16;
    return __builtin_bit_cast(float, u);
#endif
}
// the low / high 16-bit element of a packed pair as fp32
__device__ __forceinline__ float lo16f(uint32_t w) { return bf2f((bf16_t)(w & 0xffffu)); }
__device__ __forceinline__ float hi16f(uint32_t w) { return bf2f((bf16_t)(w >> 16)); }
// fp32 -> bf16, round-to-nearest-even: the gfx950 hardware conversion (v_cvt_pk_bf16_f32).  The integer
// emulation this replaces compiled to ~8 VALU ops and an exec-mask branch PER VALUE, which made every
// bf16 epilogue and the attention probability tiles VALU-bound.
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) native16_t bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.0f) & 0xffffu); }

// 2^x: bare v_exp_f32 in bf16 mode (libm's exp2f adds a denormal-range rescale: cmp + cndmask + ldexp per
// value); the exact libm form in fp32 parity mode
template <typename T>
__device__ __forceinline__ float fast_exp2(float x) {
    if constexpr (sizeof(T) == 2) return __builtin_amdgcn_exp2f(x);
    else return exp2f(x);
}

// A 16-byte MFMA operand chunk: 8 bf16 or 4 fp32, as loaded from LDS / global.
// (native vector types, not structs-with-arrays: the latter were left in scratch memory by hipcc)
typedef __attribute__((ext_vector_type(4))) uint32_t chunk16;
typedef __attribute__((ext_vector_type(2))) uint32_t chunk8;
// bit casts always go through by-value scalars: __builtin_bit_cast applied directly to a
// vector-element lvalue miscompiles (observed with this clang on both host and gfx950).
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }

template <typename T>
struct elem_traits;
template <>
struct elem_traits<bf16_t> {
    static constexpr int kPerChunk = 8;  // elements per 16-byte chunk
    __device__ static __forceinline__ float to_f32(bf16_t v) { return bf2f(v); }
    __device__ static __forceinline__ bf16_t from_f32(float v) { return f2bf(v); }
};
template <>
struct elem_traits<float> {
    static constexpr int kPerChunk = 4;
    __device__ static __forceinline__ float to_f32(float v) { return v; }
    __device__ static __forceinline__ float from_f32(float v) { return v; }
};

// One "chunk step" of a 32x32 MFMA accumulation: every lane supplies a 16-byte piece of its
// A row (row = lane&31) and of its B row (col = lane&31); lanes 0-31 and 32-63 supply DIFFERENT
// k-slices.  bf16: one v_mfma_f32_32x32x16_bf16 (16 k per step).  fp32: four
// v_mfma_f32_32x32x2_f32 (8 k per step, exact fp32 fmaf chain).  The reduction is invariant to
// any permutation of k that A and B share, which is what every caller relies on.
template <typename T>
__device__ __forceinline__ void mma_chunk(f32x16_t& acc, const chunk16& a, const chunk16& b);

template <>
__device__ __forceinline__ void mma_chunk<bf16_t>(f32x16_t& acc, const chunk16& a, const chunk16& b) {
    acc = MAEST_MFMA_32X32X16(__builtin_bit_cast(bf16x8_t, a),
                                                  __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_chunk<float>(f32x16_t& acc, const chunk16& a, const chunk16& b) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(u2f(a[q]), u2f(b[q]), acc, 0, 0, 0);
}

// Two chunk steps at once, optionally in "bf16 x 3" split precision (X3; fp32 operands only): every fp32 operand value
// x is split as x = hi + lo + O(2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi), and a*b is accumulated (fp32) as
// hi_a*hi_b + hi_a*lo_b + lo_a*hi_b on the full-rate bf16 matrix pipe: 3 MFMAs of K = 16 for 16 k-values against 8
// exact-fp32 MFMAs at 1/16 of the rate (MI355X_MICROARCH.md: 32x32x16 bf16 = 32 cycles, 32x32x2 f32 = 64 cycles), i.e.
// 96 instead of 512 matrix-pipe cycles, with a per-product error of ~2^-16 (the dropped lo*lo term and the
// truncation of lo) instead of bf16's 2^-9.  SURVEY H1 "split-bf16"; the layouts are those of the fp32 mode.
// Without X3 this is exactly two mma_chunk steps in the callers' original order (bit-identical results).
__device__ __forceinline__ void split_bf16x8(const chunk16& c0, const chunk16& c1, bf16x8_t& hi, bf16x8_t& lo) {
    chunk16 h, l;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const chunk16& c = j == 0 ? c0 : c1;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float x0 = u2f(c[2 * e]), x1 = u2f(c[2 * e + 1]);
            const uint32_t hb = pack_bf2(x0, x1);
            const float r0 = x0 - lo16f(hb), r1 = x1 - hi16f(hb);   // exact in fp32
            h[2 * j + e] = hb;
            l[2 * j + e] = pack_bf2(r0, r1);
        }
    }
    hi = __builtin_bit_cast(bf16x8_t, h);
    lo = __builtin_bit_cast(bf16x8_t, l);
}
// (a, b) -> the packed bf16 pair of their hi parts and the pair of their lo parts (the split of split_bf16x8)
__device__ __forceinline__ void split_bf2(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = pack_bf2(a, b);
    lo = pack_bf2(a - lo16f(hi), b - hi16f(hi));
}
template <typename T, bool X3>
__device__ __forceinline__ void mma_chunk2(f32x16_t& acc, const chunk16& a0, const chunk16& a1, const chunk16& b0,
                                           const chunk16& b1) {
    if constexpr (X3) {
        static_assert(sizeof(T) == 4, "split precision applies to fp32 operands");
        bf16x8_t ah, al, bh, bl;
        split_bf16x8(a0, a1, ah, al);
        split_bf16x8(b0, b1, bh, bl);
        acc = MAEST_MFMA_32X32X16(al, bh, acc, 0, 0, 0);     // small terms first
        acc = MAEST_MFMA_32X32X16(ah, bl, acc, 0, 0, 0);
        acc = MAEST_MFMA_32X32X16(ah, bh, acc, 0, 0, 0);
    } else {
        mma_chunk<T>(acc, a0, b0);
        mma_chunk<T>(acc, a1, b1);
    }
}

// C/D fragment map of every 32x32 MFMA on gfx950: register r of lane l holds
//   row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5),   col = l & 31.
__device__ __forceinline__ int frag_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Convert `n` consecutive accumulator registers (rows of one D column) into the B/A operand
// chunk of a follow-up MFMA whose reduction index is the D row index.
//   bf16: regs [8s, 8s+8)  -> rows 16s + 8(j>>2) + 4h + (j&3)
//   fp32: regs [4s, 4s+4)  -> rows  8s + 4h + j
template <typename T>
__device__ __forceinline__ chunk16 acc_to_chunk(const f32x16_t& p, int s);
template <>
__device__ __forceinline__ chunk16 acc_to_chunk<bf16_t>(const f32x16_t& p, int s) {
    chunk16 c;
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = pack_bf2(p[8 * s + 2 * j], p[8 * s + 2 * j + 1]);
    return c;
}
template <>
__device__ __forceinline__ chunk16 acc_to_chunk<float>(const f32x16_t& p, int s) {
    chunk16 c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c[j] = f2u(p[4 * s + j]);
    }
    return c;
}

// Matching operand read from a TRANSPOSED LDS tile  At[i][rho]  (row pitch `pitch` bytes):
// lane (i = lane&31, h = lane>>5) fetches the rho values listed above for step s of the
// 32-row sub-tile starting at row `rho0`.
template <typename T>
__device__ __forceinline__ chunk16 read_transposed_chunk(const char* row_ptr, int rho0, int s, int h);
template <>
__device__ __forceinline__ chunk16 read_transposed_chunk<bf16_t>(const char* row_ptr, int rho0, int s, int h) {
    const chunk8 lo = *reinterpret_cast<const chunk8*>(row_ptr + (rho0 + 16 * s + 4 * h) * 2);
    const chunk8 hi = *reinterpret_cast<const chunk8*>(row_ptr + (rho0 + 16 * s + 8 + 4 * h) * 2);
    chunk16 c;
    c[0] = lo[0]; c[1] = lo[1]; c[2] = hi[0]; c[3] = hi[1];
    return c;
}
template <>
__device__ __forceinline__ chunk16 read_transposed_chunk<float>(const char* row_ptr, int rho0, int s, int h) {
    return *reinterpret_cast<const chunk16*>(row_ptr + (rho0 + 8 * s + 4 * h) * 4);
}
template <typename T>
struct acc_steps;  // chunk steps per 32-row accumulator tile
template <>
struct acc_steps<bf16_t> { static constexpr int value = 2; };
template <>
struct acc_steps<float> { static constexpr int value = 4; };

// ------------------------------------------------------------------ wave helpers (wave64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
// true on every lane when the predicate holds on any lane of the wave (one v_cmp into an SGPR pair; the branch on it is wave-uniform)
__device__ __forceinline__ bool wave_any(bool p) {
#if defined(__AMDGCN__)
    return __builtin_amdgcn_ballot_w64(p) != 0;
#else
    return wave_max(p ? 1.0f : 0.0f) > 0.0f;       // host emulator (tests/emu)
#endif
}

// GELU value and derivative together (one erf, one exp).  EXACT = true: libm erff (fp32 parity mode);
// EXACT = false: Abramowitz-Stegun 7.1.26 erf (|err| <= 1.5e-7, far below bf16 resolution; bf16 perf mode).
template <bool EXACT>
__device__ __forceinline__ void gelu_pair(float x, float& g, float& dg) {
    const float u = x * 0.70710678118654752f;
    const float e = __expf(-u * u);                       // = exp(-x^2 / 2)
    float erfu;
    if (EXACT) {
        erfu = erff(u);
    } else {
        const float au = fabsf(u);
        const float t = 1.0f / (1.0f + 0.3275911f * au);
        const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
        const float ea = 1.0f - poly * e;
        erfu = u < 0.0f ? -ea : ea;
    }
    const float cdf = 0.5f * (1.0f + erfu);
    g = x * cdf;
    dg = cdf + x * e * 0.39894228040143268f;
}

// The same on a pair of values with packed fp32 math (v_pk_mul_f32 / v_pk_fma_f32: two lanes of work per
// instruction) -- the GEMM epilogues spend their VALU time here.  EXACT falls back to the scalar form.
template <bool EXACT>
__device__ __forceinline__ void gelu_pair2(f32x2_t x, f32x2_t& g, f32x2_t& dg) {
    if constexpr (EXACT) {
        float g0, g1, d0, d1;
        gelu_pair<true>(x[0], g0, d0);
        gelu_pair<true>(x[1], g1, d1);
        g = f32x2_t{g0, g1};
        dg = f32x2_t{d0, d1};
    } else {
        // erf(u), u = |x| / sqrt(2), in the Abramowitz-Stegun 7.1.25 / 7.1.26 form  1 - (a1 t + .. + a4 t^4) exp(-u^2),
        // t = 1 / (1 + p u), with FOUR terms and (p, a) re-fitted for the minimax error: |err| <= 1.7e-6 (the 5-term
        // 7.1.26 it replaces: 1.5e-7; the 3-term 7.1.25: 2.5e-5 = half a bf16 ulp of small activations) -- 25 x below
        // a bf16 ulp wherever the result is stored (bf16 perf mode only).  The derivative's normal density
        // pdf = exp(-x^2/2) / sqrt(2 pi) comes out of the SAME exponential (log2 of the constant folded into its
        // argument, the constant divided out of the polynomial); |x| = 2 x copysign(0.5, x) re-uses the sign factor
        // of the cdf.  11 packed fp32 operations + 2 v_bfi + 4 transcendentals per pair of values (before: 13 + 4 + 4).
        const f32x2_t k1 = {-0.72134752044448170f, -0.72134752044448170f};       // -0.5 * log2(e)
        const f32x2_t lc = {-1.32574806473615900f, -1.32574806473615900f};       // log2(1 / sqrt(2 pi))
        const f32x2_t t2 = __builtin_elementwise_fma(x * x, k1, lc);
        const f32x2_t pdf = {__builtin_amdgcn_exp2f(t2[0]), __builtin_amdgcn_exp2f(t2[1])};
        const f32x2_t hs = {__builtin_copysignf(0.5f, x[0]), __builtin_copysignf(0.5f, x[1])};
        const f32x2_t one = {1.0f, 1.0f}, half = {0.5f, 0.5f};
        const f32x2_t kp = {0.54123076f, 0.54123076f};                           // p * sqrt(2), p = 0.382707944 (u = |x| / sqrt 2)
        const f32x2_t den = __builtin_elementwise_fma(x * hs, kp, one);          // 1 + p u
        const f32x2_t t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
        // a_i * sqrt(2 pi):  a = (0.147278317, 0.598270948, -0.587041756, 0.841494165)
        const f32x2_t a4 = {2.10931316f, 2.10931316f}, a3 = {-1.47149548f, -1.47149548f};
        const f32x2_t a2 = {1.49964288f, 1.49964288f}, a1 = {0.36917199f, 0.36917199f};
        f32x2_t poly = __builtin_elementwise_fma(t, a4, a3);
        poly = __builtin_elementwise_fma(t, poly, a2);
        poly = __builtin_elementwise_fma(t, poly, a1);
        poly = poly * t;
        const f32x2_t ea = __builtin_elementwise_fma(-poly, pdf, one);         // erf(|u|)
        const f32x2_t cdf = __builtin_elementwise_fma(ea, hs, half);           // 0.5 * (1 + erf(u))
        g = x * cdf;
        dg = __builtin_elementwise_fma(x, pdf, cdf);
    }
}

// exact-erf GELU (nn.GELU default) and its derivative
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// XCD-aware bijective remap of a 1-D block id: blocks dispatched round-robin over the 8 XCDs
// (block b -> XCD b % 8, observed) are renumbered so that each XCD owns a contiguous range of
// logical tile ids and therefore re-uses operand panels out of its private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ------------------------------------------------------------------ host-side error plumbing
void set_error(const char* fmt, ...);
int check_launch(const char* what);

// Process-wide tuning / test switches (include/maest_hip.h: maest_set_option).  Defaults come from the
// environment (MAEST_GEMM_MIN_M / MAEST_GEMM_VARIANT / MAEST_GEMM_EPILOGUE), read ONCE; lock-free reads.
int option(int opt);

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: raise it once per
// (kernel, device), not once per process -- a process that touches a second GPU would otherwise fail to
// launch every kernel that needs more than 64 KiB of LDS there.  One DeviceOnce per kernel instantiation;
// safe to call concurrently (forward thread + autograd thread): the worst case sets the attribute twice.
struct DeviceOnce {
    std::atomic<uint64_t> done{0};
};
template <typename K>
static inline void ensure_dynamic_lds(DeviceOnce& once, K kernel, int bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (dev < 64 && (once.done.load(std::memory_order_acquire) & bit)) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (dev < 64) once.done.fetch_or(bit, std::memory_order_release);
}

}  // namespace maest

#define MAEST_REQUIRE(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            maest::set_error(__VA_ARGS__);  \
            return MAEST_ERR_INVALID;       \
        }                                   \
    } while (0)
