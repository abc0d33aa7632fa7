// bf16 attention forward for the long token rows (N > 320: the 10 s inference and the 30 s shapes), in a persistent,
// one-wave-per-SIMD form (reference: Attention.forward, models/maest.py:371-375 -- softmax((q k^T) * scale) v).
// Same layout contract and MFMA scheme as attention.hip (S^T = K Q^T so that a lane owns a query and softmax statistics are
// per lane; the probability tile goes straight back as the B operand of O^T = V^T P^T; V^T gathered from the row-major
// tile by ds_read_b64_tr_b16).  What is different, and why:
//
//  * A wave owns THREE 32-row query blocks (96 rows) and the whole 512-register file of its SIMD (2 workgroups of 2 waves per
//    CU = one wave per SIMD).  A work item is 192 query rows of one (batch, head): N = 560 is 3 x 192 = 576 rows (97 % useful;
//    128-row workgroups: 640, 256-row: 768), N = 875 is 5 items, N = 1685 is 9.  Every K / V fragment read from LDS feeds
//    three MFMAs instead of one.
//  * The three query blocks of a wave form a software pipeline over "units" u = (key tile t, query block i): while the VALU
//    turns S'(u) into P(u), the matrix pipe runs P(u-1) V and S'(u+1) -- the softmax of one block hides under the products
//    of the other two, inside ONE wave (attn_fwd_dma_kernel relies on four co-resident workgroups for that and leaves the
//    matrix pipe 25 % busy).
//  * The softmax costs 2.5 VALU instructions per score instead of 5: Q is pre-scaled by scale * log2(e) when its
//    fragments are built (once per item), the running maximum enters as the C operand of the first MFMA of S' = K Q'^T - m
//    (a 16-register block per query block that only changes when the maximum does), and the maximum itself is DEFERRED:
//    P = 2^S' is computed against the maximum of the tiles seen so far, and only when a row sum says that a score came
//    out far above it (sum > 2^12, i.e. some S' > 5) does the wave take the slow path that finds the tile's maximum,
//    rescales O and l and recomputes P (first tile of an item: always).  P <= 2^12 keeps its bf16 relative precision
//    and fp32 range; the normaliser carries the same factor.
//  * Persistent: 512 workgroups walk the items; an XCD owns a contiguous range so the 3 ... 9 items of a (batch, head)
//    run side by side on one L2.  K / V tiles stream through a 2-deep LDS ring by LDS-DMA (scalar base + per-lane offset +
//    immediate: no vector address arithmetic), two tiles ahead, ACROSS item boundaries; the next item's Q rows arrive in LDS
//    during the current item's last tile.  One barrier per key tile.
//  * Registers are owned by hand (cdna_hip_programming.md 5.7, "asm-owned"): O (96), the Q' fragments (48), the K (32) and V^T
//    (32) fragments of the current tile live in FIXED accumulator registers a0 .. a239, and S' (64: two units in flight), P
//    (32) and -m (48) in FIXED arch VGPRs v96 .. v245; only this file's inline asm touches them -- every MFMA, the fragment reads
//    from LDS, the softmax slices, the rare rescale and the final read-out.  hipcc keeps what is left (addresses, row sums,
//    cursors: a few dozen registers, allocated from v0 upwards) and, having no reason to spill, stays out of both ranges;
//    build.py AUDITS the code object for exactly that (no scratch, no compiler-generated v_accvgpr_*, no v96 .. v245 outside
//    the asm blocks) and fails the build otherwise.  Why not leave it to hipcc: at 512 registers it puts every MFMA result in
//    the AGPR half (a v_accvgpr_read per score), copies O around the rescale branch, and its scheduler -- with no occupancy to
//    defend -- lets the pressure run past 256 and parks 16-register tuples in the accumulator half.
//    What the asm has to do itself: wait states around MFMA operands / results and behind v_exp (noted at each site), the LDS /
//    VMEM waits.  The statements execute in program order, so the pipeline below IS the instruction schedule.
//    The host emulator build (tests/emu) takes the plain C++ twin of every primitive.
#include <utility>

#include "attn_common.h"

#ifdef MAEST_OWNED_DISABLED
// maest_amd/build.py compiles this file with MAEST_OWNED_DISABLED when the audit of the code object fails (a hipcc that allocates
// registers differently from the validated one): the kernel is left out, attention.hip keeps the four-wave LDS-DMA form at every N.
namespace maest {
bool attn_fwd_pw_available() { return false; }
int attn_fwd_pw_launch(const void*, void*, float*, int, int, AttnScale, bool, hipStream_t) {
    set_error("maest_attn_fwd(persistent): the kernel was left out of this build (register audit failed)");
    return MAEST_ERR_INVALID;
}
}  // namespace maest
#else

namespace maest {

bool attn_fwd_pw_available() { return true; }

constexpr int PW_QB = 3;                      // 32-row query blocks per wave
constexpr int PW_NW = 2;                      // waves per workgroup
constexpr int PW_ROWS = 32 * PW_QB * PW_NW;   // 192 query rows per work item
constexpr int PW_TILE = 64 * 128;             // a K or a V tile: 64 keys x 128 B, unpadded, bank-swizzled (swz128)
constexpr int PW_RING = 3;                    // ring slots: a tile is requested three tiles (two barriers) before its first read
constexpr int PW_V0 = PW_RING * PW_TILE;      // LDS: K slots | V slots | Q rows of the next item (96 rows x 128 B per wave,
constexpr int PW_Q0 = 2 * PW_RING * PW_TILE;  // swizzled like a K tile): 72 KiB, two workgroups per CU
constexpr int PW_LDS = PW_Q0 + PW_ROWS * 128;
constexpr float PW_HOT = 4096.0f;             // a tile's (half-)row sum above this sends the wave to the rescale path
constexpr float PW_COLD = 1.0e-30f;           // ... and, on an item's first tile (scores taken against m = 0), one below this
#ifndef PW_SUM
#define PW_SUM 0          // row sums: 0 = two v_add_f32 per slice, 1 = one v_dot2c_f32_bf16 on the packed word (timing variants)
#endif
#ifndef PW_CHAIN
#define PW_CHAIN 0        // MFMA order inside a group: 0 = the two accumulators alternate, 1 = one accumulator's four MFMAs back to back
#endif
#ifndef PW_DMAPOS
#define PW_DMAPOS 0       // where a tile's eight LDS-DMA requests ride (timing variants)
#endif
#ifndef PW_ORDER
#define PW_ORDER 0        // MFMAs of a pipeline region: 0 = the eight P V products, then the eight S' products; 1 = alternating (P V in the even
#endif                    // slots, S' in the odd ones: four MFMAs between two on the same accumulator): equal time (r05_attn_fwd_pw_variants.txt)
#ifndef PW_TAILNOP
#define PW_TAILNOP 0      // 1: the 12 wait states behind a unit's last S' MFMA also inside the pipeline regions (timing variant; see pw_s_mfma)
#endif
#ifndef PW_ABLATE
#define PW_ABLATE 0       // timing experiments only (scratch/pw_ablate.sh; results wrong on purpose): bit 0 no LDS-DMA requests, 1 no
#endif                    // barrier / vmcnt wait, 2 no softmax slices, 3 no MFMAs, 4 no fragment reads, 5 no stores, 6 no Q take, 8 v_mov for v_exp

// register map (device build)
constexpr int PW_A_O = 0;                     // O^T[i][db]       16 registers each: a0   .. a95
constexpr int PW_A_Q = 96;                    // Q'[i][j]          4 registers each: a96  .. a143
constexpr int PW_A_K = 144;                   // K[kb][j]          4 registers each: a144 .. a175
constexpr int PW_A_V = 176;                   // V^T[kb][db][s2]   4 registers each: a176 .. a207
constexpr int PW_A_K2 = 208;                  // K of the other tile parity            : a208 .. a239
constexpr int PW_V_PK = 96;                   // P[buf][kb][s2]    4 registers each: v96  .. v127
constexpr int PW_V_S = 128;                   // S'[buf][kb]      16 registers each: v128 .. v191
constexpr int PW_V_NM = 192;                  // -m[i]            16 registers each: v192 .. v239
constexpr int PW_V_T = 240;                   // softmax slices: exponential pairs (v240, v241 | v244, v245), row sum v242, a pair of bf16 ones v243
constexpr int PW_V_LO = 96, PW_V_HI = 245;    // (the audited range)

#if defined(__AMDGCN__)
#define PW_DEV 1
#else
#define PW_DEV 0
#endif

struct PwCtx {
    float m[PW_QB], l[PW_QB];    // running maximum (log2 domain) and this half-wave's share of the row sum, per query block
    uint32_t kaddr[4];           // LDS byte addresses (ring slot included) of this lane's K row chunks
    uint32_t vaddr[2][2];        // ... of its transpose-read pieces: [d block][row / row + 8]
    int kslot, vslot;            // ring slot those addresses point into (wave-uniform)
#ifdef PW_PROF
    unsigned long long* pp;      // (timing build: stamp buffer in LDS, next index, this workgroup is the profiled one, lane)
    int pidx, plane;
    bool pon;
#endif
#if !PW_DEV
    f32x16_t o[PW_QB][2];        // (host emulator: the state the device keeps in owned registers)
    chunk16 qf[PW_QB][4];
    chunk16 kf[2][2][4];
    chunk16 vf[2][2][2];
    f32x16_t negm[PW_QB];
    f32x16_t s[2][2];
    chunk16 pk[2][2][2];
    float t[2][2], a0;
    char* lds;
#endif
};

// PW_PROF: timing instrumentation only (scratch/pw_prof.py builds a second library with it; never defined in the product
// build): shader-clock stamps of both waves of workgroup 5, one per pipeline region / barrier / item phase, parked in LDS behind
// the kernel's own 72 KiB (no vector-memory operation: the counted vmcnt waits stay exact) and copied out at the end.
#ifdef PW_PROF
__device__ unsigned long long* g_pw_prof = nullptr;
#define PW_STAMP() do { if (c.pon && c.pidx < 512) { if (c.plane == 0) c.pp[c.pidx] = __builtin_amdgcn_s_memtime(); ++c.pidx; } } while (0)
#ifdef PW_PROF_SLOTS
#define PW_STAMP_SLOT() PW_STAMP()
#else
#define PW_STAMP_SLOT() ((void)0)
#endif
#else
#define PW_STAMP() ((void)0)
#define PW_STAMP_SLOT() ((void)0)
#endif

// One LDS-DMA piece (1 KiB): lane l's 16 bytes come from (char*)sbase + voff + IMM (sbase wave-uniform, in SGPRs) and land at
// smem + dst + 16 l.  Inline asm for the reasons given at dma16 (attn_common.h); the waits are placed by hand.  M0 is left
// holding the LDS address: nothing else in this kernel uses it (no indirect register indexing, no builtin LDS-DMA).
template <int IMM>
__device__ __forceinline__ void pw_dma(const bf16_t* sbase, uint32_t voff, char* smem, uint32_t dst) {
#if PW_DEV
    if (PW_ABLATE & 1) return;
    // (the instruction's immediate offset is added to the LDS address as well as to the global one: take it out again)
    const uint32_t lds = __builtin_amdgcn_readfirstlane(dst - IMM + (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"
                 : : "v"(voff), "s"(sbase), "s"(lds), "i"(IMM) : "memory");
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)sbase + voff + IMM),
                                     (__attribute__((address_space(3))) void*)(smem + dst), 16, 0, 0);
#endif
}
// Waits.  This wave's LDS-DMA pieces complete in the order they were requested (loads among loads); behind every barrier it
// requests exactly 8 (one K / V tile), so "all but the 8 newest vector-memory operations" covers every piece requested before the
// previous barrier whatever the stores in between do: fewer than 8 left in flight means at most the newest tile's pieces.
__device__ __forceinline__ void pw_wait_all() {          // everything: kernel start
#if PW_DEV
    if (PW_ABLATE & 2) return;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : : : "memory");
#endif
}
__device__ __forceinline__ void pw_wait_tile() {         // in front of a tile's barrier: pieces up to the previous barrier's, LDS reads
#if PW_DEV
    if (PW_ABLATE & 2) return;
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" : : : "memory");
#endif
}
__device__ __forceinline__ void pw_wait_q() {            // the next item's Q rows (requested in front of the last tile's 8 pieces)
#if PW_DEV
    asm volatile("s_waitcnt vmcnt(8)" : : : "memory");
#endif
}
__device__ __forceinline__ void pw_wait_lds() {      // every LDS read this wave has issued (hipcc does not count the asm ones)
#if PW_DEV
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
#endif
}
__device__ __forceinline__ void pw_barrier() {
#if PW_DEV
    if (PW_ABLATE & 2) return;
    asm volatile("s_barrier" : : : "memory");
#else
    __syncthreads();
#endif
}
// nothing moves across a pipeline-region boundary: the producers of an asm MFMA's VGPR operands stay a region in front of it,
// the VALU readers of its result a region behind
__device__ __forceinline__ void pw_fence() { __builtin_amdgcn_sched_barrier(0); }

#if PW_DEV
template <int A>
__device__ __forceinline__ float pw_acc_read() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "i"(A));
    return x;
}
template <int A>
__device__ __forceinline__ void pw_acc_write(uint32_t x) {
    asm volatile("v_accvgpr_write_b32 a%c0, %1" : : "i"(A), "v"(x));
}
template <int A>
__device__ __forceinline__ void pw_acc_scale(float alpha) {
    float t;
    asm volatile("v_accvgpr_read_b32 %0, a%c1\n\tv_mul_f32 %0, %0, %2\n\tv_accvgpr_write_b32 a%c1, %0" : "=&v"(t) : "i"(A), "v"(alpha));
}
template <int A, int... R>
__device__ __forceinline__ void pw_acc_read16(f32x16_t& v, std::integer_sequence<int, R...>) {
    ((v[R] = pw_acc_read<A + R>()), ...);
}
template <int A, int... R>
__device__ __forceinline__ void pw_acc_scale16(float alpha, std::integer_sequence<int, R...>) {
    (pw_acc_scale<A + R>(alpha), ...);
}
template <int A, int... R>
__device__ __forceinline__ void pw_acc_zero16(std::integer_sequence<int, R...>) {
    (pw_acc_write<A + R>(0u), ...);
}
// owned arch VGPRs
template <int V>
__device__ __forceinline__ void pw_vmax(float& mx) { asm volatile("v_max_f32 %0, %0, v%c1" : "+v"(mx) : "i"(V)); }
template <int V>
__device__ __forceinline__ void pw_vset(float x) { asm volatile("v_mov_b32 v%c0, %1" : : "i"(V), "v"(x)); }
template <int V, int... R>
__device__ __forceinline__ void pw_vmax_n(float& mx, std::integer_sequence<int, R...>) { (pw_vmax<V + R>(mx), ...); }
template <int V, int... R>
__device__ __forceinline__ void pw_vset_n(float x, std::integer_sequence<int, R...>) { (pw_vset<V + R>(x), ...); }
#endif

// O^T[I][DB] out of the accumulator registers (the caller has put 16 wait states behind the item's last MFMAs)
template <int I, int DB>
__device__ __forceinline__ f32x16_t pw_o_read(PwCtx& c) {
#if PW_DEV
    f32x16_t v;
    pw_acc_read16<PW_A_O + (2 * I + DB) * 16>(v, std::make_integer_sequence<int, 16>{});
    return v;
#else
    return c.o[I][DB];
#endif
}
template <int I>
__device__ __forceinline__ void pw_o_zero(PwCtx& c) {
#if PW_DEV
    pw_acc_zero16<PW_A_O + 2 * I * 16>(std::make_integer_sequence<int, 32>{});
#else
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) c.o[I][db][r] = 0.0f;
#endif
}
// O^T[I] *= alpha (the rescale path; the last MFMAs into O[I] are a pipeline region behind, the nops make that a guarantee)
template <int I>
__device__ __forceinline__ void pw_o_scale(PwCtx& c, float alpha) {
#if PW_DEV
    asm volatile("s_nop 7\n\ts_nop 7");
    pw_acc_scale16<PW_A_O + 2 * I * 16>(alpha, std::make_integer_sequence<int, 32>{});
    asm volatile("s_nop 1");            // v_accvgpr_write -> MFMA reading it as C
#else
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) c.o[I][db][r] *= alpha;
#endif
}
// Q' fragment (I, J) <- a pre-scaled chunk
template <int I, int J>
__device__ __forceinline__ void pw_q_write(PwCtx& c, const chunk16& q) {
#if PW_DEV
    constexpr int A = PW_A_Q + (4 * I + J) * 4;
    pw_acc_write<A>(q[0]);
    pw_acc_write<A + 1>(q[1]);
    pw_acc_write<A + 2>(q[2]);
    pw_acc_write<A + 3>(q[3]);
#else
    c.qf[I][J] = q;
#endif
}

// Q' fragment (I, J) straight out of LDS (MAEST_BF16_QS: the rows hold q' = scale * log2(e) * q already); `addr`: this lane's LDS byte
// address of chunk J of its row in query block 0 of this wave's staging rows
template <int I, int J>
__device__ __forceinline__ void pw_q_read(PwCtx& c, uint32_t addr) {
#if PW_DEV
    constexpr int A = PW_A_Q + (4 * I + J) * 4;
    asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" : : "v"(addr), "i"(A), "i"(A + 3), "i"(I * 4096));
#else
    c.qf[I][J] = *reinterpret_cast<const chunk16*>(c.lds + addr + I * 4096);
#endif
}

__device__ __forceinline__ chunk16 pw_scale_chunk(const chunk16& raw, float c2) {
    chunk16 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = pack_bf2(lo16f(raw[e]) * c2, hi16f(raw[e]) * c2);
    return r;
}

// fragment reads of a tile, one instruction (K: a 16-byte row chunk; V^T: a pair of transpose reads) per call so that a
// pipeline region can deal them out between its MFMAs; behind a tile's last read the address registers flip to the other slot
template <int SET, int K>        // K = 0 .. 7: key block K & 1, d chunk K >> 1; SET: the fragment set of the tile's parity
__device__ __forceinline__ void pw_k_read(PwCtx& c) {
    constexpr int KB = K & 1, J = K >> 1;
#if PW_DEV
    constexpr int A = (SET ? PW_A_K2 : PW_A_K) + (4 * KB + J) * 4;
    if (!(PW_ABLATE & 16))
    asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" : : "v"(c.kaddr[J]), "i"(A), "i"(A + 3), "i"(KB * 4096));
#else
    c.kf[SET][KB][J] = *reinterpret_cast<const chunk16*>(c.lds + c.kaddr[J] + KB * 4096);
#endif
    if constexpr (K == 7) {
        const int step = c.kslot == PW_RING - 1 ? -(PW_RING - 1) * PW_TILE : PW_TILE;
        c.kslot = c.kslot == PW_RING - 1 ? 0 : c.kslot + 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) c.kaddr[j] += step;
    }
}
template <int K>        // K = 0 .. 7: key block K >> 2, k step (K >> 1) & 1, d block K & 1
__device__ __forceinline__ void pw_v_read(PwCtx& c) {
    constexpr int KB = K >> 2, S2 = (K >> 1) & 1, DB = K & 1;
#if PW_DEV
    constexpr int A = PW_A_V + ((2 * KB + DB) * 2 + S2) * 4;
    if (!(PW_ABLATE & 16))
    asm volatile("ds_read_b64_tr_b16 a[%c2:%c3], %0 offset:%c6\n\tds_read_b64_tr_b16 a[%c4:%c5], %1 offset:%c6"
                 : : "v"(c.vaddr[DB][0]), "v"(c.vaddr[DB][1]), "i"(A), "i"(A + 1), "i"(A + 2), "i"(A + 3), "i"(KB * 4096 + S2 * 2048));
#else
    const char* p0 = c.lds + c.vaddr[DB][0] + KB * 4096 + S2 * 2048;
    const char* p1 = c.lds + c.vaddr[DB][1] + KB * 4096 + S2 * 2048;
    const v4i16a_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16a_t*)(p0));
    const v4i16a_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16a_t*)(p1));
    const chunk8 l2 = __builtin_bit_cast(chunk8, lo), h2 = __builtin_bit_cast(chunk8, hi);
    chunk16 f;
    f[0] = l2[0]; f[1] = l2[1]; f[2] = h2[0]; f[3] = h2[1];
    c.vf[KB][DB][S2] = f;
#endif
    if constexpr (K == 7) {
        const int step = c.vslot == PW_RING - 1 ? -(PW_RING - 1) * PW_TILE : PW_TILE;
        c.vslot = c.vslot == PW_RING - 1 ? 0 : c.vslot + 1;
#pragma unroll
        for (int db = 0; db < 2; ++db) { c.vaddr[db][0] += step; c.vaddr[db][1] += step; }
    }
}
template <int... K>
__device__ __forceinline__ void pw_read_k_all(PwCtx& c, std::integer_sequence<int, K...>) { (pw_k_read<0, K>(c), ...); }
template <int... K>
__device__ __forceinline__ void pw_read_v_all(PwCtx& c, std::integer_sequence<int, K...>) { (pw_v_read<K>(c), ...); }
__device__ __forceinline__ void pw_read_k(PwCtx& c) { pw_read_k_all(c, std::make_integer_sequence<int, 8>{}); }
__device__ __forceinline__ void pw_read_v(PwCtx& c) { pw_read_v_all(c, std::make_integer_sequence<int, 8>{}); }

// One MFMA of S'^T[key][q] = K Q'^T - m for query block I into S' buffer BUF (K = 0 .. 7: d chunk K >> 1, key block K & 1): D and C
// (S', -m) are owned arch VGPRs -- the VALU reads S' in place --, A / B the fragment registers.  TILE0: the item's first key
// tile, m = 0: C is the inline constant.  Wait states hipcc does not insert for an asm MFMA: s_nop 1 in front of the group (a
// v_accvgpr_write / VALU result as operand), 12 states behind its last one before a VALU may read D (the readers are a
// pipeline region away in program order; the nops make that a guarantee).  NOPS = false (inside a pipeline region whose successor
// region opens with a P V MFMA): the first reader -- slice 0 of the next region, S' registers 0 / 1 of key block 0, last written by MFMA
// K = 6 -- then sits behind MFMA K = 7, the region's closing statements and that P V MFMA: two matrix-pipe occupancies (>= 64 cycles) and
// more than 20 issued instructions behind its producer, no nops needed; key block 1's registers are first read eight slots later.
// (-2.5 % on the kernel: profiles/r05_attn_fwd_pw_variants.txt)
template <int I, int BUF, int SET, int K, bool TILE0, bool NOPS = true>
__device__ __forceinline__ void pw_s_mfma(PwCtx& c) {
    constexpr int KB = PW_CHAIN ? K >> 2 : K & 1, J = PW_CHAIN ? K & 3 : K >> 1;
#if PW_DEV
    if (PW_ABLATE & 8) return;
    constexpr int KF = (SET ? PW_A_K2 : PW_A_K) + (4 * KB + J) * 4, Q = PW_A_Q + (4 * I + J) * 4, S = PW_V_S + (2 * BUF + KB) * 16, NM = PW_V_NM + 16 * I;
    if constexpr (J == 0 && TILE0) {
        if constexpr (K == 0)
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_" MAEST_T16 " v[%c0:%c1], a[%c2:%c3], a[%c4:%c5], 0" : : "i"(S), "i"(S + 15), "i"(KF), "i"(KF + 3), "i"(Q), "i"(Q + 3));
        else
            asm volatile("v_mfma_f32_32x32x16_" MAEST_T16 " v[%c0:%c1], a[%c2:%c3], a[%c4:%c5], 0" : : "i"(S), "i"(S + 15), "i"(KF), "i"(KF + 3), "i"(Q), "i"(Q + 3));
    } else if constexpr (J == 0) {
        if constexpr (K == 0)
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_" MAEST_T16 " v[%c0:%c1], a[%c2:%c3], a[%c4:%c5], v[%c6:%c7]"
                         : : "i"(S), "i"(S + 15), "i"(KF), "i"(KF + 3), "i"(Q), "i"(Q + 3), "i"(NM), "i"(NM + 15));
        else
            asm volatile("v_mfma_f32_32x32x16_" MAEST_T16 " v[%c0:%c1], a[%c2:%c3], a[%c4:%c5], v[%c6:%c7]"
                         : : "i"(S), "i"(S + 15), "i"(KF), "i"(KF + 3), "i"(Q), "i"(Q + 3), "i"(NM), "i"(NM + 15));
    } else if constexpr (K == 7 && (NOPS || PW_TAILNOP)) {
        asm volatile("v_mfma_f32_32x32x16_" MAEST_T16 " v[%c0:%c1], a[%c2:%c3], a[%c4:%c5], v[%c0:%c1]\n\ts_nop 7\n\ts_nop 3"
                     : : "i"(S), "i"(S + 15), "i"(KF), "i"(KF + 3), "i"(Q), "i"(Q + 3));
    } else {
        asm volatile("v_mfma_f32_32x32x16_" MAEST_T16 " v[%c0:%c1], a[%c2:%c3], a[%c4:%c5], v[%c0:%c1]"
                     : : "i"(S), "i"(S + 15), "i"(KF), "i"(KF + 3), "i"(Q), "i"(Q + 3));
    }
#else
    f32x16_t cin = c.s[BUF][KB];
    if (J == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) cin[r] = TILE0 ? 0.0f : c.negm[I][r];
    }
    c.s[BUF][KB] = MAEST_MFMA_32X32X16(__builtin_bit_cast(bf16x8_t, c.kf[SET][KB][J]), __builtin_bit_cast(bf16x8_t, c.qf[I][J]), cin, 0, 0, 0);
#endif
}
// One MFMA of O^T[d][q] += V^T[d][key] P^T[key][q] for query block I (K = 0 .. 7: key block K >> 2, k step (K >> 1) & 1, d block
// K & 1) into the O registers, B = a chunk of P buffer BUF.  TILE0: the first two take C = 0 (O[I] needs no clearing).
template <int I, int BUF, int K, bool TILE0>
__device__ __forceinline__ void pw_pv_mfma(PwCtx& c) {
    constexpr int KB = PW_CHAIN ? (K >> 1) & 1 : K >> 2, S2 = PW_CHAIN ? K & 1 : (K >> 1) & 1, DB = PW_CHAIN ? K >> 2 : K & 1;
    constexpr bool FIRSTOF = KB == 0 && S2 == 0;       // this accumulator's first MFMA of the group
#if PW_DEV
    if (PW_ABLATE & 8) return;
    constexpr int O = PW_A_O + (2 * I + DB) * 16, V = PW_A_V + ((2 * KB + DB) * 2 + S2) * 4, P = PW_V_PK + ((2 * BUF + KB) * 2 + S2) * 4;
    if constexpr (TILE0 && FIRSTOF && K == 0)
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_" MAEST_T16 " a[%c0:%c1], a[%c2:%c3], v[%c4:%c5], 0" : : "i"(O), "i"(O + 15), "i"(V), "i"(V + 3), "i"(P), "i"(P + 3));
    else if constexpr (TILE0 && FIRSTOF)
        asm volatile("v_mfma_f32_32x32x16_" MAEST_T16 " a[%c0:%c1], a[%c2:%c3], v[%c4:%c5], 0" : : "i"(O), "i"(O + 15), "i"(V), "i"(V + 3), "i"(P), "i"(P + 3));
    else if constexpr (K == 0)
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_" MAEST_T16 " a[%c0:%c1], a[%c2:%c3], v[%c4:%c5], a[%c0:%c1]" : : "i"(O), "i"(O + 15), "i"(V), "i"(V + 3), "i"(P), "i"(P + 3));
    else
        asm volatile("v_mfma_f32_32x32x16_" MAEST_T16 " a[%c0:%c1], a[%c2:%c3], v[%c4:%c5], a[%c0:%c1]" : : "i"(O), "i"(O + 15), "i"(V), "i"(V + 3), "i"(P), "i"(P + 3));
#else
    f32x16_t cin = c.o[I][DB];
    if (TILE0 && FIRSTOF) {
#pragma unroll
        for (int r = 0; r < 16; ++r) cin[r] = 0.0f;
    }
    c.o[I][DB] = MAEST_MFMA_32X32X16(__builtin_bit_cast(bf16x8_t, c.vf[KB][DB][S2]), __builtin_bit_cast(bf16x8_t, c.pk[BUF][KB][S2]), cin, 0, 0, 0);
#endif
}
template <int I, int BUF, bool TILE0, int... K>
__device__ __forceinline__ void pw_s_all(PwCtx& c, std::integer_sequence<int, K...>) { (pw_s_mfma<I, BUF, 0, K, TILE0>(c), ...); }
template <int I, int BUF, bool TILE0, int... K>
__device__ __forceinline__ void pw_pv_all(PwCtx& c, std::integer_sequence<int, K...>) { (pw_pv_mfma<I, BUF, K, TILE0>(c), ...); }
template <int I, int BUF, bool TILE0>
__device__ __forceinline__ void pw_s_products(PwCtx& c) { pw_s_all<I, BUF, TILE0>(c, std::make_integer_sequence<int, 8>{}); }
template <int I, int BUF, bool TILE0>
__device__ __forceinline__ void pw_pv_products(PwCtx& c) { pw_pv_all<I, BUF, TILE0>(c, std::make_integer_sequence<int, 8>{}); }

// Softmax of the unit in buffer BUF, two scores at a time (slice K = 0 .. 15: key block K >> 3, registers 2 (K & 7), + 1), in two
// halves so that a region can keep a slice's exponentials in flight under the next MFMA:
//   pw_sm_exp: [the key mask] P = 2^(S' - d) (SUB; the fast path has d = 0 and no subtraction) into exponential pair K & 1;
//   pw_sm_fin: the row-sum pair += P, and the bf16 B-operand chunk word (acc_to_chunk's register -> key map).
// S' stays intact (the rescale path needs it again).  MASK (the item's last key tile): keys at or beyond `klim` (tile-relative,
// this lane's half already taken out) are padding: their S' becomes -1e30 first.  gfx950: a VALU reading a v_exp result needs one
// wait state in between (hipcc pads this itself): the callers always put other instructions between the two halves of a slice.
// Device: exponentials and row sums live in owned registers (PW_V_T ..); pw_sum_begin / pw_sum_end bracket a unit.
// The row sum is taken over the ROUNDED probabilities -- the values the P V product sees -- by v_dot2c_f32_bf16 against a pair of
// ones: one instruction per slice instead of two adds.
__device__ __forceinline__ void pw_sum_begin(PwCtx& c) {
#if PW_DEV
    if constexpr (PW_SUM == 1) asm volatile("v_mov_b32 v%c0, 0\n\tv_mov_b32 v%c1, " MAEST_ONE16X2_STR : : "i"(PW_V_T + 2), "i"(PW_V_T + 3));
    else asm volatile("v_mov_b32 v%c0, 0\n\tv_mov_b32 v%c1, 0" : : "i"(PW_V_T + 2), "i"(PW_V_T + 3));
#else
    c.a0 = 0.0f;
#endif
}
__device__ __forceinline__ float pw_sum_end(PwCtx& c) {
#if PW_DEV
    float r;
    if constexpr (PW_SUM == 1) asm volatile("s_nop 2\n\tv_mov_b32 %0, v%c1" : "=v"(r) : "i"(PW_V_T + 2));     // (a dot result read by another VALU: 3 wait states)
    else asm volatile("v_add_f32 %0, v%c1, v%c2" : "=v"(r) : "i"(PW_V_T + 2), "i"(PW_V_T + 3));
    return r;
#else
    return c.a0;
#endif
}
template <int BUF, int K, bool MASK, bool SUB>
__device__ __forceinline__ void pw_sm_exp(PwCtx& c, int klim, float d) {
    constexpr int KB = K >> 3, R = 2 * (K & 7);
#if PW_DEV
    if (PW_ABLATE & 4) return;
    constexpr int S = PW_V_S + (2 * BUF + KB) * 16 + R, T0 = PW_V_T + 4 * (K & 1);
    constexpr int KEY0 = KB * 32 + (R & 3) + 8 * (R >> 2), KEY1 = KB * 32 + ((R + 1) & 3) + 8 * ((R + 1) >> 2);
    if constexpr (MASK)
        asm volatile("v_cmp_lt_i32 vcc, %c1, %0\n\tv_cndmask_b32 v%c3, %5, v%c3, vcc\n\t"
                     "v_cmp_lt_i32 vcc, %c2, %0\n\tv_cndmask_b32 v%c4, %5, v%c4, vcc"
                     : : "v"(klim), "i"(KEY0), "i"(KEY1), "i"(S), "i"(S + 1), "v"(NEG_BIG) : "vcc");
    if constexpr (SUB)
        asm volatile("v_sub_f32 v%c1, v%c3, %0\n\tv_sub_f32 v%c2, v%c4, %0\n\tv_exp_f32 v%c1, v%c1\n\tv_exp_f32 v%c2, v%c2"
                     : : "v"(d), "i"(T0), "i"(T0 + 1), "i"(S), "i"(S + 1));
    else
        if constexpr ((PW_ABLATE & 256) != 0) asm volatile("v_mov_b32 v%c0, v%c2\n\tv_mov_b32 v%c1, v%c3" : : "i"(T0), "i"(T0 + 1), "i"(S), "i"(S + 1));
        else asm volatile("v_exp_f32 v%c0, v%c2\n\tv_exp_f32 v%c1, v%c3" : : "i"(T0), "i"(T0 + 1), "i"(S), "i"(S + 1));
#else
    if (MASK) {
        if (KB * 32 + (R & 3) + 8 * (R >> 2) >= klim) c.s[BUF][KB][R] = NEG_BIG;
        if (KB * 32 + ((R + 1) & 3) + 8 * ((R + 1) >> 2) >= klim) c.s[BUF][KB][R + 1] = NEG_BIG;
    }
    c.t[K & 1][0] = __builtin_amdgcn_exp2f(SUB ? c.s[BUF][KB][R] - d : c.s[BUF][KB][R]);
    c.t[K & 1][1] = __builtin_amdgcn_exp2f(SUB ? c.s[BUF][KB][R + 1] - d : c.s[BUF][KB][R + 1]);
#endif
}
template <int BUF, int K>
__device__ __forceinline__ void pw_sm_fin(PwCtx& c) {
    constexpr int KB = K >> 3, R = 2 * (K & 7);
#if PW_DEV
    if (PW_ABLATE & (4 | 512)) return;
    constexpr int P = PW_V_PK + ((2 * BUF + KB) * 2 + (R >> 3)) * 4 + ((R & 7) >> 1), T0 = PW_V_T + 4 * (K & 1), A0 = PW_V_T + 2;
    if constexpr (PW_SUM == 1)
        asm volatile("v_cvt_pk_" MAEST_T16 "_f32 v%c4, v%c0, v%c1\n\tv_dot2c_f32_" MAEST_T16 " v%c2, v%c4, v%c3"
                     : : "i"(T0), "i"(T0 + 1), "i"(A0), "i"(A0 + 1), "i"(P));
    else
        asm volatile("v_add_f32 v%c2, v%c2, v%c0\n\tv_add_f32 v%c3, v%c3, v%c1\n\tv_cvt_pk_" MAEST_T16 "_f32 v%c4, v%c0, v%c1"
                     : : "i"(T0), "i"(T0 + 1), "i"(A0), "i"(A0 + 1), "i"(P));
#else
    const uint32_t w = pack_bf2(c.t[K & 1][0], c.t[K & 1][1]);
    c.pk[BUF][KB][R >> 3][(R & 7) >> 1] = w;
    c.a0 += PW_SUM == 1 ? lo16f(w) + hi16f(w) : c.t[K & 1][0] + c.t[K & 1][1];
#endif
}
// a whole unit, outside the pipeline (the rescale path): exp(0) | exp(1) fin(0) | ... | fin(15)
template <int BUF, bool MASK, bool SUB, int... K>
__device__ __forceinline__ void pw_sm_all(PwCtx& c, int klim, float d, std::integer_sequence<int, K...>) {
    ((pw_sm_exp<BUF, K, MASK, SUB>(c, klim, d), (K > 0 ? pw_sm_fin<BUF, (K > 0 ? K - 1 : 0)>(c) : (void)0)), ...);
    pw_sm_fin<BUF, 15>(c);
}
// The rescale path of the unit in buffer BUF: the tile's maximum, O and l brought to it, P again (the fast path's slices have applied
// the key mask already).  FIRST: on an item's first tile the maximum is SET (O and l are still untouched).
template <int I, int BUF, bool FIRST, bool MASK>
__device__ __forceinline__ void pw_softmax_slow(PwCtx& c, int klim) {
    float mx = NEG_BIG;
#if PW_DEV
    pw_vmax_n<PW_V_S + 2 * BUF * 16>(mx, std::make_integer_sequence<int, 32>{});
#else
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, c.s[BUF][kb][r]);
#endif
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));                 // both half-waves of a query agree on its maximum
    const float d = FIRST ? mx : fmaxf(mx, 0.0f);          // the maximum only moves up
    if constexpr (FIRST) {
        c.m[I] = d;
    } else {
        const float alpha = __builtin_amdgcn_exp2f(-d);     // 1 exactly for the lanes that stay
        pw_o_scale<I>(c, alpha);
        c.l[I] *= alpha;
        c.m[I] += d;
    }
    const float nm = -c.m[I];
#if PW_DEV
    pw_vset_n<PW_V_NM + 16 * I>(nm, std::make_integer_sequence<int, 16>{});
#else
#pragma unroll
    for (int r = 0; r < 16; ++r) c.negm[I][r] = nm;
#endif
    pw_sum_begin(c);
    pw_sm_all<BUF, false, true>(c, klim, d, std::make_integer_sequence<int, 16>{});
    c.l[I] += pw_sum_end(c);
}
// The request cursor of the K / V tile stream.
struct PwDma {
    const bf16_t* base;      // K rows of the (batch, head) of the tile being requested, at that tile's first row (wave-uniform)
    uint32_t voff[4];        // this lane's byte offsets of this wave's four pieces (rows 8 (4 wave + e) + lane / 8, swizzled 16-byte column)
    uint32_t voff_last[4];   // the same for an item's last tile: rows beyond N - 1 repeat row N - 1
    bool last;               // the tile being requested is its item's last
    int t, blk, slot;        // tile, item, ring slot
};
// One request: piece E of this wave's four, K (VSEL = 0) or V (1) of the tile under the cursor
template <int E, int VSEL>
__device__ __forceinline__ void pw_dma_piece(const PwDma& d, char* smem, int wave) {
    const uint32_t vo = d.last ? d.voff_last[E] : d.voff[E];
    const uint32_t dst = (uint32_t)(d.slot * PW_TILE + (4 * wave + E) * 1024);
    if constexpr (VSEL == 0) pw_dma<0>(d.base, vo, smem, dst);
    else pw_dma<NHEADS * HD * 2>(d.base, vo, smem, dst + PW_V0);      // V sits 768 columns to the right of K
}

// One slot of a pipeline region: an MFMA, and what rides in its shadow.
//   region I of tile t works around unit u = (t, I):  slots 0 .. 7: P(u - 1) V (query block I - 1; PV0: of the item's tile 0),
//   slots 8 .. 15: S'(u + 1) (query block I + 1; SN0: of the item's tile 0), all 16: a slice of the softmax of unit u (SLICE; an
//   item's first tile takes its maximum first and runs pw_softmax_slow behind the slots instead).
//   I == 0: the tile's V^T fragments are read in slots 8 .. 11 (behind the P V products that still use the previous tile's), well
//   in front of the barrier that waits for them;
//   I == 1 (behind the tile's barrier): the next tile's K fragments in slots 0 .. 7, into the fragment set of the other tile
//   parity (the S' products of this region still read this tile's);
//   I == 1 and I == 2 carry this wave's eight LDS-DMA requests for tile t + 3.
template <int I, int PAR, bool MASK, bool PV, bool PV0, bool SN, bool SN0, bool KN, int K>
__device__ __forceinline__ void pw_slot(PwCtx& c, int klim, const PwDma& dm, char* smem, int wave) {
    constexpr int BUF = (PAR + I) & 1;
    if constexpr (PW_ORDER == 1) {
        if constexpr ((K & 1) == 0) {
            if constexpr (PV) pw_pv_mfma<(I + 2) % 3, BUF ^ 1, K / 2, PV0>(c);
        } else {
            if constexpr (K == 1 && I == 2 && SN) pw_wait_lds();
            if constexpr (SN) pw_s_mfma<(I + 1) % 3, BUF ^ 1, (I == 2 ? PAR ^ 1 : PAR), K / 2, SN0, false>(c);
        }
    } else if constexpr (K < 8) {
        if constexpr (PV) pw_pv_mfma<(I + 2) % 3, BUF ^ 1, K, PV0>(c);
    } else {
        if constexpr (K == 8 && I == 2 && SN) pw_wait_lds();
        if constexpr (SN) pw_s_mfma<(I + 1) % 3, BUF ^ 1, (I == 2 ? PAR ^ 1 : PAR), K - 8, SN0, false>(c);
    }
    pw_sm_exp<BUF, K, MASK, false>(c, klim, 0.0f);
    if constexpr (I == 1 && KN && K < 8) pw_k_read<PAR ^ 1, K>(c);          // the next tile's K fragments, into the other set
    if constexpr (PW_ORDER == 1) {                                         // fragment K / 2 right behind the (previous tile's) product that read it
        if constexpr (I == 0 && (K & 1)) pw_v_read<K / 2>(c);
    } else if constexpr (I == 0 && K >= 8 && K < 12) {                     // this tile's V^T fragments, two pieces a slot
        pw_v_read<2 * (K - 8)>(c);
        pw_v_read<2 * (K - 8) + 1>(c);
    }
    if constexpr (PW_DMAPOS == 0) {          // region 1: slots 9, 11, 13, 15; region 2: slots 1, 3, 5, 7
        if constexpr (I == 1 && K >= 8 && (K & 1)) pw_dma_piece<(K - 8) / 4, ((K - 8) / 2) & 1>(dm, smem, wave);
        if constexpr (I == 2 && K < 8 && (K & 1)) pw_dma_piece<2 + K / 4, (K / 2) & 1>(dm, smem, wave);
    }
    if constexpr (K > 0) pw_sm_fin<BUF, (K > 0 ? K - 1 : 0)>(c);
    if constexpr (PAR == 1) PW_STAMP_SLOT();
}
// T0: the item's first key tile: the scores were taken against m = 0 (C = 0), so the row sum is checked on both sides, and the
// rescale path SETS the maximum (O[I] and l[I] are still untouched: nothing to bring along).
template <int I, int PAR, bool T0, bool MASK, bool PV, bool PV0, bool SN, bool SN0, bool KN, int... K>
__device__ __forceinline__ void pw_region_slots(PwCtx& c, int klim, const PwDma& dm, char* smem, int wave, std::integer_sequence<int, K...>) {
    constexpr int BUF = (PAR + I) & 1;
    pw_sum_begin(c);
    (pw_slot<I, PAR, MASK, PV, PV0, SN, SN0, KN, K>(c, klim, dm, smem, wave), ...);
    pw_sm_fin<BUF, 15>(c);
    const float ls = pw_sum_end(c);
    const bool off = T0 ? !(ls <= PW_HOT && ls >= PW_COLD) : !(ls <= PW_HOT);
    if (__builtin_expect(wave_any(off), 0)) pw_softmax_slow<I, BUF, T0, false>(c, klim);
    else c.l[I] += ls;
}
// (KN: region 1 also reads the next tile's K fragments -- there is a next tile in this item)
template <int I, int PAR, bool T0, bool MASK, bool PV, bool PV0, bool SN, bool SN0, bool KN = false>
__device__ __forceinline__ void pw_region(PwCtx& c, int klim, const PwDma& dm, char* smem, int wave) {
    pw_region_slots<I, PAR, T0, MASK, PV, PV0, SN, SN0, KN>(c, klim, dm, smem, wave, std::make_integer_sequence<int, 16>{});
}

__global__ __launch_bounds__(PW_NW * 64, 1) void attn_fwd_pw_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                    float* __restrict__ lse, int B, int N, float c2, int q_prescaled,
                                                                    int nrb, int total) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = (N + 63) >> 6;
#if PW_DEV
    // the accumulator registers this file owns (the clobber makes the kernel descriptor allocate them)
    asm volatile("" : : : "a0", "a95", "a143", "a175", "a207", "a239", "v96", "v245");
#endif

    // items of this workgroup: XCD x (= blockIdx % 8, observed) owns items [x * per, (x + 1) * per), dealt round-robin to its slots
    const int nslots = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, slot0 = (int)blockIdx.x >> 3;
    const int per = (total + 7) >> 3;
    const int lim = (xcd + 1) * per < total ? (xcd + 1) * per : total;
    const int first = xcd * per + slot0;
    if (first >= lim) return;

    PwCtx c;
#ifdef PW_PROF
    c.pp = reinterpret_cast<unsigned long long*>(smem + PW_LDS) + wave * 512;
    c.pidx = 0;
    c.plane = lane;
    c.pon = blockIdx.x == 5 && g_pw_prof != nullptr;
#endif
    {
#if PW_DEV
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#else
        const uint32_t lds0 = 0;
        c.lds = smem;
#endif
        const int row = lane & 31, f = swz128(row);
#pragma unroll
        for (int j = 0; j < 4; ++j) c.kaddr[j] = lds0 + (uint32_t)(row * 128 + ((((2 * j) | h) ^ f) << 4));
        const int q16 = lane & 15, g16 = (lane >> 4) & 1, r0 = 4 * h + (q16 >> 2);
        const int ch = 2 * g16 + ((q16 >> 1) & 1), sub = 8 * (q16 & 1);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            c.vaddr[db][0] = lds0 + (uint32_t)(PW_V0 + r0 * 128 + ((((4 * db) | ch) ^ swz128(r0)) << 4) + sub);
            c.vaddr[db][1] = lds0 + (uint32_t)(PW_V0 + (r0 + 8) * 128 + ((((4 * db) | ch) ^ swz128(r0 + 8)) << 4) + sub);
        }
    }
    const int rlast = N - 1 - (T - 1) * 64;                // last valid row of an item's last tile
    static_assert(QKV_LD * 2 == 4096 + 512, "row pitch of the qkv tensor");
    const int lr = lane >> 3;
    // swz128(8 p + lr): row bits 1, 2 come from lr, row bit 3 is the piece's parity
    const uint32_t col[2] = {(uint32_t)(((lane & 7) ^ swz128(lr)) << 4), (uint32_t)(((lane & 7) ^ swz128(lr + 8)) << 4)};
    PwDma dm;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = 8 * (4 * wave + e) + lr, rc = row < rlast ? row : rlast;
        dm.voff[e] = (uint32_t)((row << 12) + (row << 9)) + col[e & 1];          // row * 4608 bytes
        dm.voff_last[e] = (uint32_t)((rc << 12) + (rc << 9)) + col[e & 1];
    }
    auto item_k = [&](int blk) -> const bf16_t* {        // K rows of the item's (batch, head), row 0
        const int bh = (PW_ABLATE & 128) ? 0 : blk / nrb;      // (bit 7: every item reads the first (batch, head)'s K / V: all L2 hits)
        return qkv + (int64_t)(bh / NHEADS) * N * QKV_LD + NHEADS * HD + (bh % NHEADS) * HD;
    };
    dm.blk = first;
    dm.t = 0;
    dm.slot = 0;
    dm.base = item_k(first);
    dm.last = T == 1;
    c.kslot = 0;
    c.vslot = 0;
    // advance the request cursor by one tile; behind the last tile of the last item it stays (the slot it then refills is dead)
    auto dma_advance = [&]() {
        dm.slot = dm.slot == PW_RING - 1 ? 0 : dm.slot + 1;
        if (dm.t + 1 < T) {
            ++dm.t;
            dm.base += 64 * QKV_LD;
        } else if (dm.blk + nslots < lim) {
            dm.blk += nslots;
            dm.t = 0;
            dm.base = item_k(dm.blk);
        }
        dm.last = dm.t == T - 1;
    };
    auto dma_tile = [&]() {
        pw_dma_piece<0, 0>(dm, smem, wave); pw_dma_piece<0, 1>(dm, smem, wave);
        pw_dma_piece<1, 0>(dm, smem, wave); pw_dma_piece<1, 1>(dm, smem, wave);
        pw_dma_piece<2, 0>(dm, smem, wave); pw_dma_piece<2, 1>(dm, smem, wave);
        pw_dma_piece<3, 0>(dm, smem, wave); pw_dma_piece<3, 1>(dm, smem, wave);
        dma_advance();
    };
    dma_tile();
    dma_tile();
    dma_tile();

    // an item's Q rows go through LDS one item ahead: every wave requests its own 96 rows (12 pieces) and reads them back itself
    // (row-per-lane 16-byte chunks, the K tile's swizzle), so no barrier is involved -- only its own vmcnt
    const uint32_t q_dst = (uint32_t)(PW_Q0 + wave * (32 * PW_QB * 128));
    char* const q_lds = smem + q_dst;
    auto q_fetch = [&](int blk) {
        const int bh = blk / nrb, rb = blk - bh * nrb;
        const bf16_t* qb = qkv + (int64_t)(bh / NHEADS) * N * QKV_LD + (bh % NHEADS) * HD;
        const int q0 = rb * PW_ROWS + wave * (32 * PW_QB);
#pragma unroll
        for (int p = 0; p < 4 * PW_QB; ++p) {
            const int q = q0 + 8 * p + lr < N ? q0 + 8 * p + lr : N - 1;
            pw_dma<0>(qb, (uint32_t)((q << 12) + (q << 9)) + col[p & 1], smem, q_dst + p * 1024);
        }
    };
    auto q_take = [&]() {           // LDS -> pre-scaled fragments (this lane's row: the K fragment addressing, without the ring slot)
        const int q_off = (lane & 31) * 128, q_swz = swz128(lane & 31);
#define PW_Q_TAKE(I_, J_) pw_q_write<I_, J_>(c, pw_scale_chunk(*reinterpret_cast<const chunk16*>(q_lds + (I_) * 4096 + q_off + ((((2 * (J_)) | h) ^ q_swz) << 4)), c2))
        PW_Q_TAKE(0, 0); PW_Q_TAKE(0, 1); PW_Q_TAKE(0, 2); PW_Q_TAKE(0, 3);
        PW_Q_TAKE(1, 0); PW_Q_TAKE(1, 1); PW_Q_TAKE(1, 2); PW_Q_TAKE(1, 3);
        PW_Q_TAKE(2, 0); PW_Q_TAKE(2, 1); PW_Q_TAKE(2, 2); PW_Q_TAKE(2, 3);
#undef PW_Q_TAKE
    };
    // MAEST_BF16_QS: the rows are q' already -- 12 reads straight into the fragment registers (no scaling pass, no second rounding)
    auto q_take_qs = [&]() {
#if PW_DEV
        const uint32_t qa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)q_lds + (uint32_t)((lane & 31) * 128);
#else
        const uint32_t qa = q_dst + (uint32_t)((lane & 31) * 128);
#endif
        const int q_swz = swz128(lane & 31);
        const uint32_t a0 = qa + (uint32_t)(((0 | h) ^ q_swz) << 4), a1 = qa + (uint32_t)(((2 | h) ^ q_swz) << 4);
        const uint32_t a2 = qa + (uint32_t)(((4 | h) ^ q_swz) << 4), a3 = qa + (uint32_t)(((6 | h) ^ q_swz) << 4);
        pw_q_read<0, 0>(c, a0); pw_q_read<0, 1>(c, a1); pw_q_read<0, 2>(c, a2); pw_q_read<0, 3>(c, a3);
        pw_q_read<1, 0>(c, a0); pw_q_read<1, 1>(c, a1); pw_q_read<1, 2>(c, a2); pw_q_read<1, 3>(c, a3);
        pw_q_read<2, 0>(c, a0); pw_q_read<2, 1>(c, a1); pw_q_read<2, 2>(c, a2); pw_q_read<2, 3>(c, a3);
        pw_wait_lds();
    };
    q_fetch(first);
    pw_wait_all();                               // the first three tiles of the stream and the first item's Q rows have landed
    pw_barrier();
    if (q_prescaled) q_take_qs(); else q_take();

    for (int blk = first; blk < lim; blk += nslots) {
        const int bh = blk / nrb, rb = blk - bh * nrb, b = bh / NHEADS, head = bh - b * NHEADS;
        const int next = blk + nslots < lim ? blk + nslots : blk;
        const int klim_last = rlast + 1 - 4 * h;

        // per tile: region 0 | barrier: this wave's pieces of tile t + 1 have landed and its fragment reads of tile t are done, so
        // behind it tile t + 1 is complete and tile t's slots take tile t + 3 | region 1 | region 2
        // (TILE1: the tile behind an item's first: the products that region 0 finishes belong to tile 0)
#define PW_TILE_BODY(PAR, LASTT)                                                                               \
        do {                                                                                                   \
            pw_region<0, PAR, false, LASTT, true, false, true, false>(c, klim_last, dm, smem, wave);           \
            PW_STAMP();                                                                                        \
            pw_fence();                                                                                        \
            pw_wait_tile();                                                                                     \
            pw_barrier();                                                                                      \
            PW_STAMP();                                                                                        \
            if (LASTT) q_fetch(next);                                                                          \
            pw_region<1, PAR, false, LASTT, true, false, true, false, !(LASTT)>(c, klim_last, dm, smem, wave); \
            PW_STAMP();                                                                                        \
            pw_region<2, PAR, false, LASTT, true, false, !(LASTT), false>(c, klim_last, dm, smem, wave);       \
            dma_advance();                                                                                     \
            PW_STAMP();                                                                                        \
        } while (0)

        // item start: m = 0, -m = 0 (tile 0's S' products take the inline constant, later tiles the registers), l = 0
#if PW_DEV
        pw_vset_n<PW_V_NM>(0.0f, std::make_integer_sequence<int, 16 * PW_QB>{});
#endif
#pragma unroll
        for (int i = 0; i < PW_QB; ++i) {
            c.m[i] = 0.0f;
            c.l[i] = 0.0f;
#if !PW_DEV
#pragma unroll
            for (int r = 0; r < 16; ++r) c.negm[i][r] = 0.0f;
#endif
        }
        PW_STAMP();
        pw_read_k(c);
        pw_wait_lds();
        pw_s_products<0, 0, true>(c);
        PW_STAMP();
        if (T == 1) {
            // one key tile: every product of the item is a tile-0 product
            pw_region<0, 0, true, true, false, false, true, true>(c, klim_last, dm, smem, wave);
            pw_fence();
            pw_wait_tile();
            pw_barrier();
            q_fetch(next);
            pw_region<1, 0, true, true, true, true, true, true>(c, klim_last, dm, smem, wave);
            pw_region<2, 0, true, true, true, true, false, false>(c, klim_last, dm, smem, wave);
            dma_advance();
            pw_fence();
            pw_pv_products<2, 0, true>(c);
        } else {
            // tile 0: its S' products take C = 0, its P V products start O[0], O[1]; O[2]'s start a tile later, on zeros
            pw_o_zero<2>(c);
            pw_region<0, 0, true, false, false, false, true, true>(c, klim_last, dm, smem, wave);
            PW_STAMP();
            pw_fence();
            pw_wait_tile();
            pw_barrier();
            PW_STAMP();
            pw_region<1, 0, true, false, true, true, true, true, true>(c, klim_last, dm, smem, wave);
            PW_STAMP();
            pw_region<2, 0, true, false, true, true, true, false>(c, klim_last, dm, smem, wave);
            dma_advance();
            PW_STAMP();
            int t = 1;
            for (; t + 1 < T - 1; t += 2) {
                PW_TILE_BODY(1, false);
                PW_TILE_BODY(0, false);
            }
            if (t < T - 1) {
                PW_TILE_BODY(1, false);
                PW_TILE_BODY(0, true);
                pw_fence();
                pw_pv_products<2, 0, false>(c);
            } else {
                PW_TILE_BODY(1, true);
                pw_fence();
                pw_pv_products<2, 1, false>(c);
            }
        }
#undef PW_TILE_BODY

        // the next item's Q rows (requested a tile ago) before this item's stores join the vector-memory queue
        PW_STAMP();
        pw_fence();
        pw_wait_q();
        PW_STAMP();
        if (!(PW_ABLATE & 64)) { if (q_prescaled) q_take_qs(); else q_take(); }
        PW_STAMP();
        // normalise and store this wave's 96 rows (16-byte row pieces), log-sum-exp for the backward
#if PW_DEV
        asm volatile("s_nop 7\n\ts_nop 7");      // the last MFMAs' results -> v_accvgpr_read
#endif
#define PW_STORE(I_)                                                                                           \
        do {                                                                                                   \
            const int q = rb * PW_ROWS + wave * (32 * PW_QB) + 32 * (I_) + (lane & 31);                        \
            const float l_tot = c.l[I_] + __shfl_xor(c.l[I_], 32, 64);                                         \
            const float inv = 1.0f / l_tot;                                                                    \
            const bool ok = q < N && !((PW_ABLATE & 32) && l_tot != 12345.0f);                                 \
            bf16_t* op = out + ((int64_t)b * N + (ok ? q : 0)) * OUT_LD + head * HD;                           \
            store_32d_rows16(pw_o_read<I_, 0>(c), op, lane, inv, ok);                                          \
            store_32d_rows16(pw_o_read<I_, 1>(c), op + 32, lane, inv, ok);                                     \
            if (ok && lse != nullptr && h == 0) lse[((int64_t)b * NHEADS + head) * N + q] = (c.m[I_] + log2f(l_tot)) * LN2; \
            pw_fence();                                                                                        \
        } while (0)
        PW_STORE(0);
        PW_STORE(1);
        PW_STORE(2);
#undef PW_STORE
        PW_STAMP();
    }
#ifdef PW_PROF
    if (g_pw_prof != nullptr && blockIdx.x < 64 && tid == 0) {       // where the dispatcher put this workgroup
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_pw_prof[1024 + blockIdx.x] = xcc;
    }
    if (c.pon) {
        __syncthreads();
        for (int i = lane; i < 512; i += 64) g_pw_prof[wave * 512 + i] = i < c.pidx ? c.pp[i] : 0ull;
    }
#endif
}

#ifdef PW_PROF
extern "C" int maest_debug_pw_prof(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pw_prof), &p, sizeof(p)); }
#endif

int attn_fwd_pw_launch(const void* qkv, void* out, float* lse, int B, int N, AttnScale sc, bool q_prescaled, hipStream_t st) {
    const int nrb = (N + PW_ROWS - 1) / PW_ROWS;
    const int total = nrb * NHEADS * B;
    const int per = (total + 7) / 8;
    const int nslots = per < 64 ? per : 64;           // 2 workgroups on each of an XCD's 32 CUs
    static DeviceOnce once;
#ifdef PW_PROF
    constexpr int lds_bytes = PW_LDS + 2 * 512 * 8;
#else
    constexpr int lds_bytes = PW_LDS;
#endif
    ensure_dynamic_lds(once, &attn_fwd_pw_kernel, lds_bytes);
    hipLaunchKernelGGL(attn_fwd_pw_kernel, dim3(8 * nslots), dim3(PW_NW * 64), lds_bytes, st, (const bf16_t*)qkv, (bf16_t*)out, lse,
                       B, N, sc.c2, q_prescaled ? 1 : 0, nrb, total);
    return check_launch("maest_attn_fwd(persistent)");
}

}  // namespace maest
#endif  // MAEST_OWNED_DISABLED
