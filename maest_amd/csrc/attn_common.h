// Definitions shared by the attention kernels (attention.hip: every numeric mode, forward and backward;
// attn_fwd_pw.hip: the persistent one-wave-per-SIMD bf16 forward).  Layout contract and MFMA scheme: attention.hip.
#pragma once
#include "common.h"

namespace maest {

constexpr int HD = 64;         // head dim
constexpr int NHEADS = 12;
constexpr int QKV_LD = 3 * NHEADS * HD;  // 2304
constexpr int OUT_LD = NHEADS * HD;      // 768
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float NEG_BIG = -1.0e30f;
// The three places the softmax scale enters an attention kernel.  Raw q (every dtype): the score product q k^T is multiplied by
// scale * log2(e) on its way into exp2, dQ = scale * dS K and dK = scale * dS^T q.  MAEST_BF16_QS (the bf16 training / inference
// mode of maest_amd/maest.py): the q columns of the qkv tensor hold q' = scale * log2(e) * q -- the factor is folded into the q rows
// of the qkv projection's operand copy (maest_cast_weights_multi), so q' is rounded to bf16 ONCE and forward and backward see the
// same operand --: the product q' k^T is the exponent as it stands, dQ keeps its factor (the gradient with respect to the TRUE q:
// dgrad and wgrad of the projection then run on the unscaled weights as before) and dK = ln 2 * dS^T q'.
struct AttnScale {
    float c2;    // exponent (log2 domain) per unit of the raw score product
    float dq;    // factor of dS K
    float dk;    // factor of dS^T q
};
inline AttnScale attn_scale(float scale, bool q_prescaled) {
    return q_prescaled ? AttnScale{1.0f, scale, LN2} : AttnScale{scale * LOG2E, scale, scale};
}
// drain this wave's vector-memory queue (LDS-DMA included) without touching the LDS / scalar counters
#define MAEST_ATTN_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)

typedef short v4i16a_t __attribute__((ext_vector_type(4)));

// A generic pointer into shared memory as an LDS pointer.  Device build: the low half of the flat address IS the LDS byte address; going
// through the integer avoids the null test hipcc puts around a flat -> LDS address-space cast (on `smem + offset` that test has come out
// as an illegal `v_cmp_ne_u32 0, src_shared_base` -- "Operand has incorrect register class" -- depending on unrelated code in the kernel).
// Make the compiler finish a register-fragment load HERE (an empty asm statement that reads and "writes" the registers): hipcc otherwise sinks
// loads whose first use sits inside the tile loop below the hand-placed `s_waitcnt vmcnt(0)` in front of the loop, and then keeps counted
// vmcnt waits for them INSIDE the loop body -- which, on every later iteration, wait for the LDS-DMA pieces of the NEXT tile instead (the
// DMA is hidden from the compiler but not from the counter): the prefetch then overlaps with a quarter of an iteration instead of a whole one.
__device__ __forceinline__ void pin_loaded(chunk16& c) {
#if defined(__AMDGCN__)
    asm volatile("" : "+v"(c));
#endif
}
__device__ __forceinline__ void pin_loaded(float& x) {
#if defined(__AMDGCN__)
    asm volatile("" : "+v"(x));
#endif
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
template <typename P>
__device__ __forceinline__ __attribute__((address_space(3))) P* lds_cast(const void* p) {
#if defined(__AMDGCN__)
    return (__attribute__((address_space(3))) P*)(uint32_t)(uintptr_t)p;
#else
    return (__attribute__((address_space(3))) P*)(p);
#endif
}
#pragma clang diagnostic pop

__device__ __forceinline__ int swz128(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }
// One LDS-DMA instruction (global_load_lds_dwordx4: 64 lanes x 16 B -> 1 KiB at the wave-uniform LDS address `dst`),
// issued as inline asm so that hipcc does not know about it: through the builtin the compiler treats the DMA as a
// store to LDS that may alias every later LDS read and puts `s_waitcnt vmcnt(0)` in front of the next ds_read (here:
// inside the dQ loop), i.e. it waits for the tile it has just requested.  The waits are placed by hand
// (MAEST_ATTN_WAIT_VM0 one step later); a hidden DMA can only make the compiler's own counted waits longer, never
// shorter.  M0 (the DMA's LDS base) is saved and restored inside the statement.  (The host emulator build takes
// the builtin, which it executes synchronously.)
__device__ __forceinline__ void dma16(const void* gsrc, char* dst) {
#if defined(__AMDGCN__)
    const uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)dst);   // (low half of the flat address = the LDS address)
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds) : "memory");
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
#endif
}

// The same accumulator pair as 16-byte pieces (bf16): [64 d][32 rows] (lane = row, registers = d: d = 32 db + 8 g + 4 h + j) -> 16-byte
// pieces of the row -- half the store instructions of store_dT at the same bytes and addresses (a row-per-lane store tail is bound by
// store ISSUE, not bandwidth: cdna_hip_programming.md T21).  Call it from converged code (the exchange is a wave operation); `ok`
// predicates the store per lane: the two half-waves exchange one 8-byte quarter so that lane (key, h) owns d = 32 db + 8 (pair + 2 h) .. + 7
// (the exchange + store of four packed quarters `pk` of a lane's 32 d: lane (row, h) ends up with d = 8 (pair + 2 h) .. + 7; stored at
// row_ptr and, REP > 1, again at row_ptr + k * rep_stride)
template <int REP = 1>
__device__ __forceinline__ void store_32d_words16(const uint32_t (&pk)[4][2], bf16_t* row_ptr, int lane, bool ok, int rep_stride = 0) {
    const int h = lane >> 5;
#pragma unroll
    for (int pair = 0; pair < 2; ++pair) {
        // v_permlane32_swap(vdst, src): lanes 32-63 of vdst <-> lanes 0-31 of src.  vdst = group `pair`, src = group `pair + 2`:
        // afterwards the lower lane holds [own | upper's] quarter of group `pair`, the upper lane [lower's | own] of `pair + 2`
        const auto x = __builtin_amdgcn_permlane32_swap(pk[pair][0], pk[pair + 2][0], false, false);
        const auto y = __builtin_amdgcn_permlane32_swap(pk[pair][1], pk[pair + 2][1], false, false);
        chunk16 c;
        c[0] = x[0];
        c[1] = y[0];
        c[2] = x[1];
        c[3] = y[1];
        if (ok) {
#pragma unroll
            for (int k = 0; k < REP; ++k) *reinterpret_cast<chunk16*>(row_ptr + k * rep_stride + 8 * (pair + 2 * h)) = c;
        }
    }
}
__device__ __forceinline__ void store_32d_rows16(const f32x16_t& acc, bf16_t* row_ptr, int lane, float mul, bool ok) {
    uint32_t pk[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        pk[g][0] = pack_bf2(acc[4 * g] * mul, acc[4 * g + 1] * mul);
        pk[g][1] = pack_bf2(acc[4 * g + 2] * mul, acc[4 * g + 3] * mul);
    }
    store_32d_words16(pk, row_ptr, lane, ok);
}
// the same 32 d of an fp32 result as MAEST_SPLIT3_A thirds of a bf16 row of 3 * `third` columns: [ hi | hi | lo ], 16-byte pieces
__device__ __forceinline__ void store_32d_split3(const f32x16_t& acc, bf16_t* row_ptr, int lane, float mul, bool ok, int third) {
    uint32_t hi[4][2], lo[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        split_bf2(acc[4 * g] * mul, acc[4 * g + 1] * mul, hi[g][0], lo[g][0]);
        split_bf2(acc[4 * g + 2] * mul, acc[4 * g + 3] * mul, hi[g][1], lo[g][1]);
    }
    store_32d_words16<2>(hi, row_ptr, lane, ok, third);
    store_32d_words16(lo, row_ptr + 2 * third, lane, ok);
}

}  // namespace maest
