// Shared primitives of the one-wave-per-SIMD NT GEMM kernels (gemm_nt_ow.hip: gemm_nt256o_kernel; gemm_nt_owd.hip: gemm_nt256d_kernel, the
// form with the deferred C-tile store): the operand ring, the owned-register map, and every main-loop statement (MFMA, fragment read,
// LDS-DMA request, counted waits, barrier) as an `asm volatile` with a plain-C++ twin for the host emulator.  See gemm_nt_ow.hip's header
// for the design.  Include inside neither namespace; define OW_PROF_VAR before including to get the profiling variable's definition.
#pragma once
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"

namespace maest {

constexpr int OW_UNIT = 256 * 128;            // one operand unit: 256 rows x 128 B
constexpr int OW_NBUF = 5;
constexpr int OW_SMEM = OW_NBUF * OW_UNIT;    // 163840: the whole LDS, one workgroup per CU
constexpr int OW_EPI0 = 2 * OW_UNIT;          // the C staging area starts behind the ring's first two units
// register map (device build): accumulator tile (nt, mt) = a[16 (4 nt + mt) ..+15]; fragment set s (k-step parity):
// A[mt] = v[192 + 32 s + 4 mt ..+3], B[nt] = v[208 + 32 s + 4 nt ..+3]
constexpr int OW_V_F = 192;
constexpr int OW_V_BIAS = 188;                // this lane's four bias values of the tile (columns 4 lane ..+3), tile top -> epilogue
constexpr int OW_V_LO = 188, OW_V_HI = 255;   // (the audited range)
constexpr int OW_BIAS0 = 2 * 33792;           // the bias row's place in the C staging area: behind the largest staging buffer

#if defined(__AMDGCN__)
#define OW_DEV 1
#else
#define OW_DEV 0
#endif
#ifndef OW_ABLATE
#define OW_ABLATE 0       // timing experiments only (results wrong on purpose): bit 0 no LDS-DMA requests, 1 no barrier / vmcnt wait,
#endif                    // 2 no MFMAs, 3 no fragment reads, 4 no epilogue

// OW_PROF: timing instrumentation only (scratch/ow_prof.py builds a second library with it; never defined in the product build):
// shader-clock time the four waves of workgroup 5 spend in each part of a stage, summed over the tile.
#ifdef OW_PROF
#ifdef OW_PROF_VAR
__device__ unsigned long long* g_ow_prof = nullptr;
#endif
#define OW_TICK(slot) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); c.prof[slot] += (unsigned)(t_ - c.tprev); c.tprev = t_; } while (0)
#else
#define OW_TICK(slot) ((void)0)
#endif

// The fragment and bias registers (OW_FRAGS, on every main-loop statement) and the whole accumulator half (OW_ACCS, on the waits and
// barriers only: four statements per stage keep hipcc from parking a value there across the loop; on every statement they cost minutes
// of compile time), as clobber lists on every main-loop statement: hipcc may then use v192 .. v255 for values that do not live
// across the main loop (the epilogue, which needs them), and must keep everything else out of them.
#define OW_FRAGS "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
#ifndef OW_MORE_OWNED
#define OW_MORE_OWNED     // further registers an including file owns across its main loop: `, "v60", ...` appended to the clobber lists of the waits and barriers
#endif
#define OW_ACCS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"

struct OwCtx {
    uint32_t pa[3][4], pb[3][4]; // LDS addresses of this lane's A / B row chunk of k-step 0 .. 3 in ring buffers 0, 2, 4 (buffers 1, 3 and
                                 // the wave's m- / n-tile go into the read's immediate offset: nothing per stage is left to compute)
    uint32_t lds0;               // LDS address of the dynamic segment
    int wave;                    // (wave-uniform)
#ifdef OW_PROF
    unsigned prof[24];
    unsigned long long tprev;
#endif
#if !OW_DEV
    f32x16_t acc[4][4];          // (host emulator: the state the device keeps in owned registers)
    chunk16 fa[2][4], fb[2][4];
    f32x4_t bias;
    char* lds;
#endif
};

// fragment read: one ds_read_b128 = this lane's 16-byte chunk of row (tile T) of the A (ISB = false) or B operand
template <int SET, int T, bool ISB, int OFF = 0>
__device__ __forceinline__ void ow_read(OwCtx& c, uint32_t addr) {
#if OW_DEV
    constexpr int V = OW_V_F + 32 * SET + (ISB ? 16 : 0) + 4 * T;
    if (!(OW_ABLATE & 8))
        asm volatile("ds_read_b128 v[%c1:%c2], %0 offset:%c3" : : "v"(addr), "i"(V), "i"(V + 3), "i"(OFF + T * 4096) : OW_FRAGS);
#else
    const chunk16 v = *reinterpret_cast<const chunk16*>(c.lds + addr + OFF + T * 4096);
    if (ISB) c.fb[SET][T] = v;
    else c.fa[SET][T] = v;
#endif
}
// acc(nt, mt) (+)= B[nt] A[mt]^T : rows of the result tile = output columns n (4 consecutive per lane and register group), lane = row m
template <int SET, int NT, int MT, bool ZERO>
__device__ __forceinline__ void ow_mfma(OwCtx& c) {
#if OW_DEV
    constexpr int D = 16 * (4 * NT + MT), A = OW_V_F + 32 * SET + 4 * MT, B = OW_V_F + 32 * SET + 16 + 4 * NT;
    if (OW_ABLATE & 4) return;
    if constexpr (ZERO)
        asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], v[%c2:%c3], v[%c4:%c5], 0"
                     : : "i"(D), "i"(D + 15), "i"(B), "i"(B + 3), "i"(A), "i"(A + 3) : OW_FRAGS);
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], v[%c2:%c3], v[%c4:%c5], a[%c0:%c1]"
                     : : "i"(D), "i"(D + 15), "i"(B), "i"(B + 3), "i"(A), "i"(A + 3) : OW_FRAGS);
#else
    if (ZERO) {
#pragma unroll
        for (int r = 0; r < 16; ++r) c.acc[NT][MT][r] = 0.0f;
    }
    mma_chunk<bf16_t>(c.acc[NT][MT], c.fb[SET][NT], c.fa[SET][MT]);
#endif
}
// One LDS-DMA request (1 KiB = 8 rows x 128 B): lane l's 16 bytes come from base + voff (base wave-uniform, in SGPRs) and land at
// LDS address dst + 16 l; voff then moves on to the next K stage (+ 128 bytes), inside the same statement so that the add rides in
// the request's slot.  Inline asm so that hipcc does not count it (attn_common.h: dma16); M0 is left holding the address.
template <int DST>               // DST: byte offset of the piece from the wave's first piece of ring buffer 0 (`piece0`, an SGPR)
__device__ __forceinline__ void ow_dma(const char* base, uint32_t& voff, uint32_t piece0, OwCtx& c) {
#if OW_DEV
    if (OW_ABLATE & 1) return;
    const uint32_t lds = __builtin_amdgcn_readfirstlane(piece0);
    asm volatile("s_add_u32 m0, %2, %c3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tv_add_u32 %0, 0x80, %0"
                 : "+v"(voff) : "s"(base), "s"(lds), "i"(DST) : "memory", "scc", OW_FRAGS);
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + voff),
                                     (__attribute__((address_space(3))) void*)(c.lds + piece0 + DST), 16, 0, 0);
    voff += 128;
#endif
}
template <int N>
__device__ __forceinline__ void ow_wait_vm() {       // all but this wave's N newest LDS-DMA requests have landed
#if OW_DEV
    if (OW_ABLATE & 2) return;
    asm volatile("s_waitcnt vmcnt(%c0)" : : "i"(N) : "memory", OW_FRAGS);
#endif
}
__device__ __forceinline__ void ow_wait_lds() {      // every fragment read this wave has issued (hipcc does not count the asm ones)
#if OW_DEV
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory", OW_FRAGS, OW_ACCS OW_MORE_OWNED);
#endif
}
__device__ __forceinline__ void ow_barrier() {
#if OW_DEV
    if (OW_ABLATE & 2) return;
    asm volatile("s_barrier" : : : "memory", OW_FRAGS, OW_ACCS OW_MORE_OWNED);
#else
    __syncthreads();
#endif
}
// the epilogue's barrier: LDS traffic only, and no claim on the fragment registers (hipcc's values may sit in them there)
__device__ __forceinline__ void ow_sync_epilogue() {
#if OW_DEV
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");
#else
    __syncthreads();
#endif
}
// The tile's bias row: requested at the tile's top into owned registers (a value hipcc holds across the main loop ends up in the
// accumulator half under this kernel's register pressure), stored to LDS when the ring has been drained.
__device__ __forceinline__ void ow_bias_load(OwCtx& c, const float* src) {      // src == nullptr: zeros
#if OW_DEV
    if (src != nullptr)
        asm volatile("global_load_dwordx4 v[%c1:%c2], %0, off" : : "v"(src), "i"(OW_V_BIAS), "i"(OW_V_BIAS + 3) : "memory", OW_FRAGS);
    else
        asm volatile("v_mov_b32 v%c0, 0\n\tv_mov_b32 v%c1, 0\n\tv_mov_b32 v%c2, 0\n\tv_mov_b32 v%c3, 0"
                     : : "i"(OW_V_BIAS), "i"(OW_V_BIAS + 1), "i"(OW_V_BIAS + 2), "i"(OW_V_BIAS + 3) : OW_FRAGS);
#else
    c.bias = src != nullptr ? *reinterpret_cast<const f32x4_t*>(src) : f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#endif
}
__device__ __forceinline__ void ow_bias_store(OwCtx& c, uint32_t addr) {        // (behind a vmcnt(0))
#if OW_DEV
    asm volatile("ds_write_b128 %0, v[%c1:%c2]" : : "v"(addr), "i"(OW_V_BIAS), "i"(OW_V_BIAS + 3) : "memory");
#else
    *reinterpret_cast<f32x4_t*>(c.lds + addr) = c.bias;
#endif
}
#if OW_DEV
template <int A>
__device__ __forceinline__ float ow_acc_read1() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "i"(A));
    return x;
}
template <int A, int... R>
__device__ __forceinline__ void ow_acc_read16(f32x16_t& v, std::integer_sequence<int, R...>) {
    ((v[R] = ow_acc_read1<A + R>()), ...);
}
#endif
// accumulator tile (NT, MT) out of the owned registers (the caller has put the wait states behind the last MFMA)
template <int NT, int MT>
__device__ __forceinline__ f32x16_t ow_acc_read(OwCtx& c) {
#if OW_DEV
    f32x16_t v;
    ow_acc_read16<16 * (4 * NT + MT)>(v, std::make_integer_sequence<int, 16>{});
    return v;
#else
    return c.acc[NT][MT];
#endif
}

// One slot of a k-step: an MFMA and what rides in its shadow.  k-step S of a stage multiplies fragment set S & 1; slots 0 .. 7
// carry the fragment reads of the NEXT k-step (A tiles 0 .. 3, then B tiles 0 .. 3) into the other set, slots 8 .. 15 this
// wave's LDS-DMA requests into the unit buffer at LDS address dst: NDMA = 8 one per slot (pieces 0 .. 7), NDMA = 4 every other
// slot (pieces I0 .. I0 + 3).
template <int S, int Q, bool ZERO, int NDMA, int I0, int RA, int RB, int KS, int DBUF>
__device__ __forceinline__ void ow_slot(OwCtx& c, const char* base, uint32_t (&vo)[8], uint32_t piece0) {
    constexpr int SET = S & 1;
    ow_mfma<SET, (Q >> 2), (Q & 3), ZERO>(c);
    if constexpr (Q < 4) {
        ow_read<SET ^ 1, Q, false, (RA & 1) * OW_UNIT>(c, c.pa[RA >> 1][KS]);
    } else if constexpr (Q < 8) {
        ow_read<SET ^ 1, Q - 4, true, (RB & 1) * OW_UNIT>(c, c.pb[RB >> 1][KS]);
    } else if constexpr (NDMA == 4 && (Q & 1) == 0) {
        constexpr int I = I0 + ((Q - 8) >> 1);
        ow_dma<DBUF * OW_UNIT + I * 1024>(base, vo[I], piece0, c);
    }
}
template <int S, bool ZERO, int NDMA, int I0, int RA, int RB, int KS, int DBUF, int... Q>
__device__ __forceinline__ void ow_step_slots(OwCtx& c, const char* base, uint32_t (&vo)[8], uint32_t piece0,
                                              std::integer_sequence<int, Q...>) {
    (ow_slot<S, Q, ZERO, NDMA, I0, RA, RB, KS, DBUF>(c, base, vo, piece0), ...);
}
// k-step S: reads k-step KS of the operand units in ring buffers RA / RB, requests pieces I0 .. I0 + 3 (NDMA = 4) into buffer DBUF
template <int S, bool ZERO, int NDMA, int I0, int RA, int RB, int KS, int DBUF>
__device__ __forceinline__ void ow_step(OwCtx& c, const char* base, uint32_t (&vo)[8], uint32_t piece0) {
    ow_step_slots<S, ZERO, NDMA, I0, RA, RB, KS, DBUF>(c, base, vo, piece0, std::make_integer_sequence<int, 16>{});
}

}  // namespace maest
