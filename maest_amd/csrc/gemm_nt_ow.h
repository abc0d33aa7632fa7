// Primitives of the one-wave-per-SIMD NT GEMM kernel (gemm_nt_ow.hip: gemm_nt256o_kernel): the operand ring, the owned-register map, and every
// main-loop statement (MFMA with the fragment read or LDS-DMA request that rides in its shadow, counted waits, barrier) as an `asm volatile`
// with a plain-C++ twin for the host emulator.  See gemm_nt_ow.hip's header for the design.
//
// Round 6: the matrix instruction is v_mfma_f32_16x16x32_bf16 (was 32x32x16).  On random operands the chip's power limit leaves the 16 x 16 shape
// 14.5 % more throughput than the 32 x 32 shape (profiles/r06_mfma_shape_power.txt: 2090 against 1825 TFLOP/s for back-to-back MFMAs, 2450 both
// on zeros; the 32 x 32 shape moves 32 accumulator registers per 32768 flops, the 16 x 16 shape 8 per 16384) -- the vendor library's kernels at
// these shapes are 16 x 16 kernels and run at a 9 % higher clock than the 32 x 32 form of this kernel did (profiles/r06_gemm_vs_library.txt).
// A wave still owns 128 x 128 outputs in a0 .. a255: 64 blocks of 16 x 16, block (n16, m16) = a[4 (8 n16 + m16) ..+3]; lane l holds row
// 16 m16 + (l & 15) and columns 16 n16 + 4 (l >> 4) ..+3 of the wave's tile.  A K stage (64 deep) is two k32 halves of 64 MFMAs.
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
// register map (device build): fragment set s (= k32 half of a stage): A[t] = v[128 + 64 s + 4 t ..+3] (rows 16 t ..+15 of the wave's 128 A
// rows), B[t] = v[160 + 64 s + 4 t ..+3]; a lane holds the 16 bytes k = 8 (l >> 4) ..+7 of the half for row l & 15
constexpr int OW_V_F = 128;
constexpr int OW_V_BIAS = 124;                // this lane's four bias values of the tile (columns 4 lane ..+3), tile top -> epilogue
constexpr int OW_V_LO = 124, OW_V_HI = 255;   // (the audited range)
constexpr int OW_BIAS0 = 2 * 33792;           // the bias row's place in the C staging area: behind the largest staging buffer

#if defined(__AMDGCN__)
#define OW_DEV 1
#else
#define OW_DEV 0
#endif
#ifndef OW_ABLATE
#define OW_ABLATE 0       // timing experiments only (results wrong on purpose): bit 0 no LDS-DMA requests, 1 no barrier / vmcnt wait,
#endif                    // 2 no MFMAs, 3 no fragment reads, 4 no epilogue, 5 every workgroup loads tile 0, 7 plain instead of streaming C stores

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

// The fragment and bias registers (OW_FRAGS) and the whole accumulator half (OW_ACCS) as clobber lists: OW_FRAGS on every main-loop statement,
// OW_ACCS on the waits and barriers only (two or three statements per stage keep hipcc from parking a value there across the loop; on every
// statement they cost minutes of compile time).  hipcc may then use v124 .. v255 for values that do not live across the main loop (the epilogue,
// which needs them), and must keep everything else out of them; maest_amd/build.py audits the code object.
#define OW_FRAGS "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
#define OW_ACCS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"

struct OwCtx {
    uint32_t pa[3][2], pb[3][2]; // LDS addresses of this lane's A / B row chunk of k32 half 0 / 1 in ring buffers 0, 2, 4 (buffers 1, 3 and the
                                 // 16-row block go into the read's immediate offset: nothing per stage is left to compute)
    uint32_t lds0;               // LDS address of the dynamic segment
    int wave;                    // (wave-uniform)
#ifdef OW_PROF
    unsigned prof[24];
    unsigned long long tprev;
#endif
#if !OW_DEV
    f32x4_t acc[8][8];           // (host emulator: the state the device keeps in owned registers) [n16][m16]
    chunk16 fa[2][8], fb[2][8];
    f32x4_t bias;
    char* lds;
#endif
};

// ---- the pieces of a slot, as assembler text (device) ...
// MFMA of block (N16, M16) on fragment set SET: rows of the result block = output columns n (4 consecutive per lane), lane & 15 = row m
#define OW_MFMA_ACC "v_mfma_f32_16x16x32_" MAEST_T16 " a[%c0:%c0+3], v[%c1:%c1+3], v[%c2:%c2+3], a[%c0:%c0+3]"
#define OW_MFMA_ZERO "v_mfma_f32_16x16x32_" MAEST_T16 " a[%c0:%c0+3], v[%c1:%c1+3], v[%c2:%c2+3], 0"

// ... and as C++ (host emulator)
template <int SET, int N16, int M16, bool ZERO>
__device__ __forceinline__ void ow_mfma_twin(OwCtx& c) {
#if !OW_DEV
    if (ZERO) c.acc[N16][M16] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    c.acc[N16][M16] = MAEST_MFMA_16X16X32(__builtin_bit_cast(bf16x8_t, c.fb[SET][N16]), __builtin_bit_cast(bf16x8_t, c.fa[SET][M16]),
                                                              c.acc[N16][M16], 0, 0, 0);
#endif
}
template <int SET, int T, bool ISB>
__device__ __forceinline__ void ow_read_twin(OwCtx& c, uint32_t addr, int off) {
#if !OW_DEV
    const chunk16 v = *reinterpret_cast<const chunk16*>(c.lds + addr + off + T * 2048);
    if (ISB) c.fb[SET][T] = v;
    else c.fa[SET][T] = v;
#endif
}

// A slot = one MFMA and what rides in its shadow, ONE asm statement (hipcc puts a wait state behind every inline-asm statement: with the 16-cycle
// MFMA a statement per rider would fill the issue slots with them).
//   plain:  the MFMA alone
//   read:   + one ds_read_b128 = this lane's 16-byte chunk of block T of the A (ISB = false) or B operand into set RSET
//   dma:    + one LDS-DMA request (1 KiB = 8 rows x 128 B): lane l's 16 bytes come from base + voff (base wave-uniform, in SGPRs) and land at LDS
//           address piece0 + DST + 16 l; voff then moves on to the next K stage (+ 128 bytes) inside the same statement.  M0 is written in FRONT of
//           the MFMA (its issue is the wait state the request needs behind an SALU write of M0) and left holding the address.
template <int SET, int N16, int M16, bool ZERO>
__device__ __forceinline__ void ow_slot_plain(OwCtx& c) {
#if OW_DEV
    constexpr int D = 4 * (8 * N16 + M16), A = OW_V_F + 64 * SET + 4 * M16, B = OW_V_F + 64 * SET + 32 + 4 * N16;
    if (OW_ABLATE & 4) return;
    if constexpr (ZERO) asm volatile(OW_MFMA_ZERO : : "i"(D), "i"(B), "i"(A) : OW_FRAGS);
    else asm volatile(OW_MFMA_ACC : : "i"(D), "i"(B), "i"(A) : OW_FRAGS);
#else
    ow_mfma_twin<SET, N16, M16, ZERO>(c);
#endif
}
template <int SET, int N16, int M16, bool ZERO, int RSET, int T, bool ISB, int OFF>
__device__ __forceinline__ void ow_slot_read(OwCtx& c, uint32_t addr) {
#if OW_DEV
    constexpr int D = 4 * (8 * N16 + M16), A = OW_V_F + 64 * SET + 4 * M16, B = OW_V_F + 64 * SET + 32 + 4 * N16;
    constexpr int V = OW_V_F + 64 * RSET + (ISB ? 32 : 0) + 4 * T;
    if ((OW_ABLATE & 12) == 12) return;
    if (OW_ABLATE & 8) { ow_slot_plain<SET, N16, M16, ZERO>(c); return; }
    if (OW_ABLATE & 4) { asm volatile("ds_read_b128 v[%c1:%c1+3], %0 offset:%c2" : : "v"(addr), "i"(V), "i"(OFF + T * 2048) : OW_FRAGS); return; }
    if constexpr (ZERO)
        asm volatile(OW_MFMA_ZERO "\n\tds_read_b128 v[%c4:%c4+3], %3 offset:%c5" : : "i"(D), "i"(B), "i"(A), "v"(addr), "i"(V), "i"(OFF + T * 2048) : OW_FRAGS);
    else
        asm volatile(OW_MFMA_ACC "\n\tds_read_b128 v[%c4:%c4+3], %3 offset:%c5" : : "i"(D), "i"(B), "i"(A), "v"(addr), "i"(V), "i"(OFF + T * 2048) : OW_FRAGS);
#else
    ow_mfma_twin<SET, N16, M16, ZERO>(c);
    ow_read_twin<RSET, T, ISB>(c, addr, OFF);
#endif
}
template <int SET, int N16, int M16, bool ZERO, int DST>
__device__ __forceinline__ void ow_slot_dma(OwCtx& c, const char* base, uint32_t& voff, uint32_t piece0) {
#if OW_DEV
    constexpr int D = 4 * (8 * N16 + M16), A = OW_V_F + 64 * SET + 4 * M16, B = OW_V_F + 64 * SET + 32 + 4 * N16;
    if (OW_ABLATE & 1) { ow_slot_plain<SET, N16, M16, ZERO>(c); return; }
    const uint32_t lds = __builtin_amdgcn_readfirstlane(piece0);
    if (OW_ABLATE & 4) {
        asm volatile("s_add_u32 m0, %2, %c3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tv_add_u32 %0, 0x80, %0" : "+v"(voff) : "s"(base), "s"(lds), "i"(DST) : "memory", "scc", OW_FRAGS);
        return;
    }
    if constexpr (ZERO)
        asm volatile("s_add_u32 m0, %2, %c3\n\t" "v_mfma_f32_16x16x32_" MAEST_T16 " a[%c4:%c4+3], v[%c5:%c5+3], v[%c6:%c6+3], 0" "\n\tglobal_load_lds_dwordx4 %0, %1\n\tv_add_u32 %0, 0x80, %0"
                     : "+v"(voff) : "s"(base), "s"(lds), "i"(DST), "i"(D), "i"(B), "i"(A) : "memory", "scc", OW_FRAGS);
    else
        asm volatile("s_add_u32 m0, %2, %c3\n\t" "v_mfma_f32_16x16x32_" MAEST_T16 " a[%c4:%c4+3], v[%c5:%c5+3], v[%c6:%c6+3], a[%c4:%c4+3]" "\n\tglobal_load_lds_dwordx4 %0, %1\n\tv_add_u32 %0, 0x80, %0"
                     : "+v"(voff) : "s"(base), "s"(lds), "i"(DST), "i"(D), "i"(B), "i"(A) : "memory", "scc", OW_FRAGS);
#else
    ow_mfma_twin<SET, N16, M16, ZERO>(c);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + voff),
                                     (__attribute__((address_space(3))) void*)(c.lds + piece0 + DST), 16, 0, 0);
    voff += 128;
#endif
}
// One LDS-DMA request outside the main loop (prologue, next tile's first units)
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
// fragment read outside the main loop (the first half's fragments of a tile)
template <int SET, int T, bool ISB, int OFF = 0>
__device__ __forceinline__ void ow_read(OwCtx& c, uint32_t addr) {
#if OW_DEV
    constexpr int V = OW_V_F + 64 * SET + (ISB ? 32 : 0) + 4 * T;
    if (!(OW_ABLATE & 8))
        asm volatile("ds_read_b128 v[%c1:%c1+3], %0 offset:%c2" : : "v"(addr), "i"(V), "i"(OFF + T * 2048) : OW_FRAGS);
#else
    ow_read_twin<SET, T, ISB>(c, addr, OFF);
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
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory", OW_FRAGS, OW_ACCS);
#endif
}
__device__ __forceinline__ void ow_barrier() {
#if OW_DEV
    if (OW_ABLATE & 2) return;
    asm volatile("s_barrier" : : : "memory", OW_FRAGS, OW_ACCS);
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
        asm volatile("global_load_dwordx4 v[%c1:%c1+3], %0, off" : : "v"(src), "i"(OW_V_BIAS) : "memory", OW_FRAGS);
    else
        asm volatile("v_mov_b32 v%c0, 0\n\tv_mov_b32 v[%c0+1], 0\n\tv_mov_b32 v[%c0+2], 0\n\tv_mov_b32 v[%c0+3], 0" : : "i"(OW_V_BIAS) : OW_FRAGS);
#else
    c.bias = src != nullptr ? *reinterpret_cast<const f32x4_t*>(src) : f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#endif
}
__device__ __forceinline__ void ow_bias_store(OwCtx& c, uint32_t addr) {        // (behind a vmcnt(0))
#if OW_DEV
    asm volatile("ds_write_b128 %0, v[%c1:%c1+3]" : : "v"(addr), "i"(OW_V_BIAS) : "memory");
#else
    *reinterpret_cast<f32x4_t*>(c.lds + addr) = c.bias;
#endif
}
// accumulator block (N16, M16) out of the owned registers (the caller has put the wait states behind the last MFMA)
template <int N16, int M16>
__device__ __forceinline__ f32x4_t ow_acc_read(OwCtx& c) {
#if OW_DEV
    f32x4_t v;
    constexpr int A = 4 * (8 * N16 + M16);
    asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c4+1]\n\tv_accvgpr_read_b32 %2, a[%c4+2]\n\tv_accvgpr_read_b32 %3, a[%c4+3]"
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]) : "i"(A));
    return v;
#else
    return c.acc[N16][M16];
#endif
}

// ---- a k32 half of a stage: 64 slots, slot HS multiplies block (HS >> 3, HS & 7) of fragment set H.
//   reads: the 16 fragments of the OTHER set (A blocks 0 .. 7, then B blocks 0 .. 7) ride in slots 0, 3, .., 45 (one ds_read_b128 per 48 cycles and wave:
//          a third of the LDS's read rate over four waves) and have >= 16 slots to land; RA / RB: ring buffers they come from, KK: their k32 half
//   requests: four per 32-slot quarter in slots 8, 14, 20, 26 (+ 32): quarter Qa of the half requests pieces IA .. IA + 3 of the unit at `base_a` into
//          buffer DA, quarter Qb pieces IB ..+3 of `base_b` into DB (NDMA = 0: none)
template <int H, int HS, bool ZERO, int RA, int RB, int KK, int NDMA, int IA, int DA, int IB, int DB>
__device__ __forceinline__ void ow_half_slot(OwCtx& c, const char* base_a, uint32_t (&voa)[8], const char* base_b, uint32_t (&vob)[8], uint32_t piece0) {
    constexpr int N16 = HS >> 3, M16 = HS & 7, QS = HS & 31;
    if constexpr (HS < 48 && HS % 3 == 0) {
        constexpr int F = HS / 3;                                   // fragment 0 .. 15 of the other set
        if constexpr (F < 8) ow_slot_read<H, N16, M16, ZERO, H ^ 1, F, false, (RA & 1) * OW_UNIT>(c, c.pa[RA >> 1][KK]);
        else ow_slot_read<H, N16, M16, ZERO, H ^ 1, F - 8, true, (RB & 1) * OW_UNIT>(c, c.pb[RB >> 1][KK]);
    } else if constexpr (NDMA != 0 && (QS == 8 || QS == 14 || QS == 20 || QS == 26)) {
        constexpr int I = (QS - 8) / 6;
        if constexpr (HS < 32) ow_slot_dma<H, N16, M16, ZERO, DA * OW_UNIT + (IA + I) * 1024>(c, base_a, voa[IA + I], piece0);
        else ow_slot_dma<H, N16, M16, ZERO, DB * OW_UNIT + (IB + I) * 1024>(c, base_b, vob[IB + I], piece0);
    } else {
        ow_slot_plain<H, N16, M16, ZERO>(c);
    }
}
template <int H, bool ZERO, int RA, int RB, int KK, int NDMA, int IA, int DA, int IB, int DB, int... HS>
__device__ __forceinline__ void ow_half_slots(OwCtx& c, const char* base_a, uint32_t (&voa)[8], const char* base_b, uint32_t (&vob)[8], uint32_t piece0,
                                              std::integer_sequence<int, HS...>) {
    (ow_half_slot<H, HS, ZERO, RA, RB, KK, NDMA, IA, DA, IB, DB>(c, base_a, voa, base_b, vob, piece0), ...);
}
template <int H, bool ZERO, int RA, int RB, int KK, int NDMA, int IA, int DA, int IB, int DB>
__device__ __forceinline__ void ow_half(OwCtx& c, const char* base_a, uint32_t (&voa)[8], const char* base_b, uint32_t (&vob)[8], uint32_t piece0) {
    ow_half_slots<H, ZERO, RA, RB, KK, NDMA, IA, DA, IB, DB>(c, base_a, voa, base_b, vob, piece0, std::make_integer_sequence<int, 64>{});
}

}  // namespace maest
