// TN GEMM (wgrad + bias grad), 256 x 256 tile, ONE WAVE PER SIMD (bf16 operands):
//   C[i][j] += sum_k A[k][i] B[k][j],  colsum[i] += sum_k A[k][i]      (A = dY [tokens][out], B = X [tokens][in], K = tokens)
// Same contract, operand layout, split-K plan and XCD numbering as gemm256.hip:gemm_tn256_kernel, which it replaces for bf16 inputs
// with the atomic combine (reference call sites: the weight / bias gradients of nn.Linear, models/maest.py:353-376, 197-208 through
// autograd).  The structure is gemm_nt_ow.hip's: four waves, each owning 128 x 128 outputs in the accumulator half of the register
// file (a0 .. a255), fragments of two 16-deep k-steps in v192 .. v255, every MFMA / fragment read / LDS-DMA request an inline-asm
// statement in program order, registers audited in the code object (maest_amd/build.py).  What is specific here:
//   * a K slice is 32 token rows of the tile's 256 columns of A and of B (2 x 16 KiB, rows of 512 B: whole cache lines, one LDS-DMA
//     request = 2 rows), FIVE slices in the ring (160 KiB): a slice is requested four slices ahead, its A half behind the barrier
//     of slice s - 5, its B half in the first k-step of slice s - 4, and retired by a counted vmcnt(24);
//   * fragments come transposed out of the token-major tiles by ds_read_b64_tr_b16 (two per fragment), with gemm_tn256_kernel's
//     source-side swizzle (16-byte chunk ^= (row & 3) << 2); 16 reads per k-step of 16 MFMAs, B's first, so that counted lgkmcnt
//     waits (6 / 8 / 10 / 12 outstanding) let the next k-step start on the reads that have landed;
//   * ONE barrier per slice, four MFMAs into the slice's second k-step: by then every wave's last reads of the slice are four MFMAs
//     old (lgkmcnt(0) costs nothing), behind it the next slice is read and this slice's buffer is refilled;
//   * the bias gradient is v_dot2c_f32_bf16 of the A fragments against a pair of ones (4 accumulators per 32-column block so that
//     no two consecutive ones depend on each other), on the slices that are this tile's turn;
//   * the split-K partials leave by fp32 atomics straight from the accumulator registers.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"
#include "gemm256_epi.h"

#ifdef MAEST_OWNED_DISABLED
// see gemm_nt_ow.hip: left out of a build whose register audit failed; gemm256.hip keeps the 8-wave wgrad kernel
namespace maest {
bool gemm_tn256o_available() { return false; }
int gemm_tn256o_launch(GemmTn256Params&, int, hipStream_t) {
    set_error("maest_gemm_tn(256o): the one-wave-per-SIMD kernel was left out of this build (register audit failed)");
    return MAEST_ERR_INVALID;
}
}  // namespace maest
#else

namespace maest {

bool gemm_tn256o_available() { return true; }

constexpr int TW_HALF = 32 * 512;             // one operand's slice: 32 token rows x 512 B
constexpr int TW_SLICE = 2 * TW_HALF;         // A then B
constexpr int TW_NBUF = 5;
constexpr int TW_SMEM = TW_NBUF * TW_SLICE;   // 163840
// register map (device build): accumulator tile (a, b) = a[16 (4 a + b) ..+15]; fragment set s (k-step parity):
// A[a] = v[192 + 32 s + 4 a ..+3], B[b] = v[208 + 32 s + 4 b ..+3]; bias-gradient accumulators v[176 + 4 blk + e], ones in v184
constexpr int TW_V_F = 192, TW_V_CS = 176, TW_V_ONE = 184;
constexpr int TW_V_LO = 176, TW_V_HI = 255;   // (the audited range)

#if defined(__AMDGCN__)
#define TW_DEV 1
#else
#define TW_DEV 0
#endif
#ifndef TW_ABLATE
#define TW_ABLATE 0       // timing experiments only (results wrong on purpose): bit 0 no LDS-DMA requests, 1 no barrier / vmcnt wait,
#endif                    // 2 no MFMAs, 3 no fragment reads, 4 no epilogue

// The owned arch VGPRs (TW_FRAGS, on every main-loop statement) and the whole accumulator half (TW_ACCS, on the waits and barriers) as
// clobber lists: hipcc cannot keep a value in them across the main loop (gemm_nt_ow.hip has the story), and may use v176 .. v255 behind it.
#define TW_FRAGS "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
#define TW_ACCS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"

struct TwCtx {
    uint32_t pa[4], pb[4];       // LDS byte offsets of this lane's transpose-read piece of A block a / B block b (k-step 0, first read)
    uint32_t lds0;
#if !TW_DEV
    f32x16_t acc[4][4];          // (host emulator: the state the device keeps in owned registers)
    chunk16 fa[2][4], fb[2][4];
    float cs[2][4];
    char* lds;
#endif
};

// half HF (token rows +0 / +4 of the lane's piece) of fragment T of the A (ISB = false) or B operand, k-step KS of the slice at addr
template <int SET, int T, bool ISB, int HF, int KS>
__device__ __forceinline__ void tw_read(TwCtx& c, uint32_t addr) {
#if TW_DEV
    constexpr int V = TW_V_F + 32 * SET + (ISB ? 16 : 0) + 4 * T + 2 * HF;
    if (!(TW_ABLATE & 8))
        asm volatile("ds_read_b64_tr_b16 v[%c1:%c2], %0 offset:%c3" : : "v"(addr), "i"(V), "i"(V + 1), "i"(KS * 8192 + HF * 2048) : TW_FRAGS);
#else
    typedef short v4i16_t __attribute__((ext_vector_type(4)));
    const v4i16_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16_t*)(c.lds + addr + KS * 8192 + HF * 2048));
    const chunk8 r2 = __builtin_bit_cast(chunk8, r);
    chunk16& f = ISB ? c.fb[SET][T] : c.fa[SET][T];
    f[2 * HF] = r2[0];
    f[2 * HF + 1] = r2[1];
#endif
}
// acc(a, b) (+)= A[a]^T B[b]: rows of the result tile = i (register-indexed), lane = column j
template <int SET, int A, int B, bool ZERO>
__device__ __forceinline__ void tw_mfma(TwCtx& c) {
#if TW_DEV
    constexpr int D = 16 * (4 * A + B), FA = TW_V_F + 32 * SET + 4 * A, FB = TW_V_F + 32 * SET + 16 + 4 * B;
    if (TW_ABLATE & 4) return;
    if constexpr (ZERO)
        asm volatile("v_mfma_f32_32x32x16_" MAEST_T16 " a[%c0:%c1], v[%c2:%c3], v[%c4:%c5], 0"
                     : : "i"(D), "i"(D + 15), "i"(FA), "i"(FA + 3), "i"(FB), "i"(FB + 3) : TW_FRAGS);
    else
        asm volatile("v_mfma_f32_32x32x16_" MAEST_T16 " a[%c0:%c1], v[%c2:%c3], v[%c4:%c5], a[%c0:%c1]"
                     : : "i"(D), "i"(D + 15), "i"(FA), "i"(FA + 3), "i"(FB), "i"(FB + 3) : TW_FRAGS);
#else
    if (ZERO) {
#pragma unroll
        for (int r = 0; r < 16; ++r) c.acc[A][B][r] = 0.0f;
    }
    mma_chunk<bf16_t>(c.acc[A][B], c.fa[SET][A], c.fb[SET][B]);
#endif
}
// One LDS-DMA request (1 KiB = 2 token rows x 512 B); voff then moves on by one slice (`step` bytes, wave-uniform)
__device__ __forceinline__ void tw_dma(const char* base, uint32_t& voff, uint32_t step, uint32_t dst, TwCtx& c) {
#if TW_DEV
    if (TW_ABLATE & 1) return;
    const uint32_t lds = __builtin_amdgcn_readfirstlane(dst);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tv_add_u32 %0, %3, %0"
                 : "+v"(voff) : "s"(base), "s"(lds), "s"(step) : "memory", TW_FRAGS);
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + voff),
                                     (__attribute__((address_space(3))) void*)(c.lds + dst), 16, 0, 0);
    voff += step;
#endif
}
template <int VM, int LGKM>
__device__ __forceinline__ void tw_wait() {          // VM / LGKM < 0: that counter is not waited for
#if TW_DEV
    if constexpr (VM >= 0 && LGKM >= 0) {
        if (TW_ABLATE & 2) asm volatile("s_waitcnt lgkmcnt(%c0)" : : "i"(LGKM) : "memory", TW_FRAGS, TW_ACCS);
        else asm volatile("s_waitcnt vmcnt(%c0) lgkmcnt(%c1)" : : "i"(VM), "i"(LGKM) : "memory", TW_FRAGS, TW_ACCS);
    } else if constexpr (VM >= 0) {
        if (!(TW_ABLATE & 2)) asm volatile("s_waitcnt vmcnt(%c0)" : : "i"(VM) : "memory", TW_FRAGS, TW_ACCS);
    } else {
        asm volatile("s_waitcnt lgkmcnt(%c0)" : : "i"(LGKM) : "memory", TW_FRAGS, TW_ACCS);
    }
#endif
}
__device__ __forceinline__ void tw_barrier() {
#if TW_DEV
    if (TW_ABLATE & 2) return;
    asm volatile("s_barrier" : : : "memory", TW_FRAGS, TW_ACCS);
#else
    __syncthreads();
#endif
}
#if TW_DEV
template <int A>
__device__ __forceinline__ float tw_acc_read1() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "i"(A));
    return x;
}
template <int A, int... R>
__device__ __forceinline__ void tw_acc_read16(f32x16_t& v, std::integer_sequence<int, R...>) {
    ((v[R] = tw_acc_read1<A + R>()), ...);
}
#endif
template <int A, int B>
__device__ __forceinline__ f32x16_t tw_acc_read(TwCtx& c) {
#if TW_DEV
    f32x16_t v;
    tw_acc_read16<16 * (4 * A + B)>(v, std::make_integer_sequence<int, 16>{});
    return v;
#else
    return c.acc[A][B];
#endif
}
// bias gradient: the A fragments of set SET, blocks A0 and A0 + 1, summed over their 8 k values per lane (pairs against 1.0, 1.0)
template <int SET, int A0>
__device__ __forceinline__ void tw_colsum(TwCtx& c) {
#if TW_DEV
#define TW_DOT(BLK, E) asm volatile("v_dot2c_f32_" MAEST_T16 " v%c0, v%c1, v%c2" : : "i"(TW_V_CS + 4 * (BLK) + (E)), \
                                    "i"(TW_V_F + 32 * SET + 4 * (A0 + (BLK)) + (E)), "i"(TW_V_ONE))
    TW_DOT(0, 0); TW_DOT(0, 1); TW_DOT(0, 2); TW_DOT(0, 3);
    TW_DOT(1, 0); TW_DOT(1, 1); TW_DOT(1, 2); TW_DOT(1, 3);
#undef TW_DOT
#else
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t w = c.fa[SET][A0 + blk][e];
            c.cs[blk][e] += lo16f(w) + hi16f(w);
        }
#endif
}
__device__ __forceinline__ void tw_colsum_init(TwCtx& c) {
#if TW_DEV
    asm volatile("v_mov_b32 v%c0, 0\n\tv_mov_b32 v%c1, 0\n\tv_mov_b32 v%c2, 0\n\tv_mov_b32 v%c3, 0" : : "i"(TW_V_CS), "i"(TW_V_CS + 1),
                 "i"(TW_V_CS + 2), "i"(TW_V_CS + 3));
    asm volatile("v_mov_b32 v%c0, 0\n\tv_mov_b32 v%c1, 0\n\tv_mov_b32 v%c2, 0\n\tv_mov_b32 v%c3, 0" : : "i"(TW_V_CS + 4), "i"(TW_V_CS + 5),
                 "i"(TW_V_CS + 6), "i"(TW_V_CS + 7));
    asm volatile("v_mov_b32 v%c0, " MAEST_ONE16X2_STR : : "i"(TW_V_ONE));
#else
    for (int blk = 0; blk < 2; ++blk)
        for (int e = 0; e < 4; ++e) c.cs[blk][e] = 0.0f;
#endif
}
template <int BLK>
__device__ __forceinline__ float tw_colsum_get(TwCtx& c) {      // (behind s_nop: the dot products' results)
#if TW_DEV
    float s0, s1, s2, s3;
    asm volatile("s_nop 3\n\tv_mov_b32 %0, v%c4\n\tv_mov_b32 %1, v%c5\n\tv_mov_b32 %2, v%c6\n\tv_mov_b32 %3, v%c7"
                 : "=v"(s0), "=v"(s1), "=v"(s2), "=v"(s3)
                 : "i"(TW_V_CS + 4 * BLK), "i"(TW_V_CS + 4 * BLK + 1), "i"(TW_V_CS + 4 * BLK + 2), "i"(TW_V_CS + 4 * BLK + 3));
    return (s0 + s1) + (s2 + s3);
#else
    return (c.cs[BLK][0] + c.cs[BLK][1]) + (c.cs[BLK][2] + c.cs[BLK][3]);
#endif
}

// One slot = an MFMA and what rides in its shadow.
//   k-step 0 of slice s (KS = 0): multiplies set 0; reads the slice's second k-step into set 1, one transpose read per slot (B's
//   eight first, then A's); requests the B half of slice s + 4 in slots 2, 6, 10, 14 (NDMA = 4);
//   k-step 1 (KS = 1): multiplies set 1; behind slot 3 stands the slice's barrier; slots 4 .. 15 read the NEXT slice's first k-step
//   into set 0 (two reads per slot in slots 4 .. 7) and request the A half of slice s + 5 in slots 6, 9, 12, 15.
// Counted waits in front of slots 0 / 4 / 8 / 12 of k-step 0 (fragments A[0] / A[1] / A[2] / A[3] of set 0 are 6 / 4 / 2 / 0 reads
// from the end of the previous k-step's sequence, plus what this k-step has issued since).
struct TwDma {
    const char* base;
    uint32_t step, dst;
};
template <int KS, int Q, bool ZERO, int NDMA>
__device__ __forceinline__ void tw_slot(TwCtx& c, uint32_t (&ra)[4], uint32_t (&rb)[4], const TwDma& d, uint32_t (&vo)[4]) {
    constexpr int SET = KS;
    if constexpr (KS == 0 && Q == 0) tw_wait<-1, 6>();
    if constexpr (KS == 0 && Q == 4) tw_wait<-1, 8>();
    if constexpr (KS == 0 && Q == 8) tw_wait<-1, 10>();
    if constexpr (KS == 0 && Q == 12) tw_wait<-1, 12>();
    tw_mfma<SET, (Q >> 2), (Q & 3), ZERO>(c);
    if constexpr (KS == 0) {
        // slice s, k-step 1 -> set 1: slot q reads half (q & 1) of B[q >> 1] (q < 8) / A[(q - 8) >> 1]
        if constexpr (Q < 8) tw_read<1, (Q >> 1), true, (Q & 1), 1>(c, rb[Q >> 1]);
        else tw_read<1, ((Q - 8) >> 1), false, (Q & 1), 1>(c, ra[(Q - 8) >> 1]);
        if constexpr (NDMA == 4 && (Q & 3) == 2) tw_dma(d.base, vo[Q >> 2], d.step, d.dst + (Q >> 2) * 1024, c);
    } else {
        // slice s + 1, k-step 0 -> set 0 (ra / rb point into the next slice's buffer): 16 reads in slots 4 .. 15
        if constexpr (Q >= 4 && Q < 8) {
            tw_read<0, (Q - 4), true, 0, 0>(c, rb[Q - 4]);
            tw_read<0, (Q - 4), true, 1, 0>(c, rb[Q - 4]);
        } else if constexpr (Q >= 8) {
            tw_read<0, ((Q - 8) >> 1), false, (Q & 1), 0>(c, ra[(Q - 8) >> 1]);
        }
        if constexpr (NDMA == 4 && Q >= 6 && (Q % 3) == 0) tw_dma(d.base, vo[(Q - 6) / 3], d.step, d.dst + ((Q - 6) / 3) * 1024, c);
    }
}

__global__ __launch_bounds__(256, 1) void gemm_tn256o_kernel(GemmTn256Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
#if TW_DEV
    // the registers this file owns (the clobber makes the kernel descriptor allocate them)
    asm volatile("" : : : "a0", "a255", "v176", "v255");
#endif
    // (split, tile) pairs are numbered split-major and each XCD takes a contiguous range of them (gemm_tn256_kernel)
    const int ntiles = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, ntiles * p.split_k);
    const int split = wg / ntiles;
    const int tile = wg - split * ntiles;
    const int tile_i = tile / p.tiles_n;
    const int tile_j = tile - tile_i * p.tiles_n;
    const int i0 = tile_i * 256, j0 = tile_j * 256;
    const int total_slices = p.K >> 5;
    const int s_begin = split * p.k_slices_per_split;
    int s_end = s_begin + p.k_slices_per_split;
    if (s_end > total_slices) s_end = total_slices;
    const int n = s_end - s_begin;                 // slices of this workgroup
    if (n <= 0) return;

    TwCtx c;
#if TW_DEV
    c.lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#else
    c.lds0 = 0;
    c.lds = smem;
#endif
    {
        // gemm256.hip: frag_tn256<bf16_t> -- row = 16 ks + 8 h + (q >> 2), logical byte column (blk + 16 g16 + 4 (q & 3)) * 2,
        // physical 16-byte chunk = logical ^ ((row & 3) << 2); the second read of a fragment sits 4 rows (2048 B) below
        const int h = lane >> 5, g16 = (lane >> 4) & 1, q = lane & 15;
        const int row = 8 * h + (q >> 2);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int cba = (wm * 128 + t * 32 + 16 * g16 + 4 * (q & 3)) * 2, cbb = (wn * 128 + t * 32 + 16 * g16 + 4 * (q & 3)) * 2;
            c.pa[t] = (uint32_t)(row * 512 + ((((cba >> 4) ^ ((row & 3) << 2)) << 4) | (cba & 15)));
            c.pb[t] = (uint32_t)(TW_HALF + row * 512 + ((((cbb >> 4) ^ ((row & 3) << 2)) << 4) | (cbb & 15)));
        }
    }
    // LDS-DMA sources: piece i of this wave = token rows 8 wave + 2 i, + 1 of a slice; lane l: row + (l >> 5), physical chunk l & 31
    const char* abase = p.A + ((int64_t)s_begin * 32 * p.lda + i0) * 2;
    const char* bbase = p.B + ((int64_t)s_begin * 32 * p.ldb + j0) * 2;
    const uint32_t a_step = (uint32_t)(32 * p.lda * 2), b_step = (uint32_t)(32 * p.ldb * 2);
    uint32_t voa[4], vob[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * wave + 2 * i + (lane >> 5);
        const uint32_t lc = (uint32_t)(((lane & 31) ^ ((row & 3) << 2)) << 4);
        voa[i] = (uint32_t)(row * (int)p.lda * 2) + lc;
        vob[i] = (uint32_t)(row * (int)p.ldb * 2) + lc;
    }
    const uint32_t piece0 = c.lds0 + (uint32_t)(wave * 4 * 1024);
    auto request_a = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) tw_dma(abase, voa[i], a_step, piece0 + (uint32_t)(buf * TW_SLICE + i * 1024), c);
    };
    auto request_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) tw_dma(bbase, vob[i], b_step, piece0 + (uint32_t)(buf * TW_SLICE + TW_HALF + i * 1024), c);
    };
    const bool do_colsum = p.colsum != nullptr;    // (uniform)
    int cs_turn = tile_j;                          // the j-tiles of an i-row take the slices round-robin (gemm_tn256_kernel)
    tw_colsum_init(c);

    // prologue: slices 0 .. 3 and the A half of slice 4, as far as they exist
    request_a(0); request_b(0);
    if (n > 1) { request_a(1); request_b(1); }
    if (n > 2) { request_a(2); request_b(2); }
    if (n > 3) { request_a(3); request_b(3); }
    if (n > 4) request_a(4);
    if (n > 4) tw_wait<28, -1>();
    else if (n == 4) tw_wait<24, -1>();
    else if (n == 3) tw_wait<16, -1>();
    else if (n == 2) tw_wait<8, -1>();
    else tw_wait<0, -1>();
    tw_barrier();
    {
        uint32_t ra[4], rb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { ra[t] = c.lds0 + c.pa[t]; rb[t] = c.lds0 + c.pb[t]; }
        // (the order of the reads of a k-step: B's eight, then A's -- the counted waits of tw_slot rely on it)
        tw_read<0, 0, true, 0, 0>(c, rb[0]); tw_read<0, 0, true, 1, 0>(c, rb[0]); tw_read<0, 1, true, 0, 0>(c, rb[1]); tw_read<0, 1, true, 1, 0>(c, rb[1]);
        tw_read<0, 2, true, 0, 0>(c, rb[2]); tw_read<0, 2, true, 1, 0>(c, rb[2]); tw_read<0, 3, true, 0, 0>(c, rb[3]); tw_read<0, 3, true, 1, 0>(c, rb[3]);
        tw_read<0, 0, false, 0, 0>(c, ra[0]); tw_read<0, 0, false, 1, 0>(c, ra[0]); tw_read<0, 1, false, 0, 0>(c, ra[1]); tw_read<0, 1, false, 1, 0>(c, ra[1]);
        tw_read<0, 2, false, 0, 0>(c, ra[2]); tw_read<0, 2, false, 1, 0>(c, ra[2]); tw_read<0, 3, false, 0, 0>(c, ra[3]); tw_read<0, 3, false, 1, 0>(c, ra[3]);
    }
    // Slice s at ring phase PH = s % 5.  KIND 2: slices s + 4 and s + 5 exist; 1: s + 4 only; 0: neither (the last four slices, which
    // wait for everything in front of their barrier).
    auto slice_body = [&](auto ph_tag, auto first_tag, auto kind_tag) {
        constexpr int PH = decltype(ph_tag)::value, KIND = decltype(kind_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int BUF = PH, BUF_N = (PH + 1) % 5, BUF_4 = (PH + 4) % 5;
        uint32_t ra[4], rb[4], rna[4], rnb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ra[t] = c.lds0 + (uint32_t)(BUF * TW_SLICE) + c.pa[t];
            rb[t] = c.lds0 + (uint32_t)(BUF * TW_SLICE) + c.pb[t];
            rna[t] = c.lds0 + (uint32_t)(BUF_N * TW_SLICE) + c.pa[t];
            rnb[t] = c.lds0 + (uint32_t)(BUF_N * TW_SLICE) + c.pb[t];
        }
        const TwDma db{bbase, b_step, piece0 + (uint32_t)(BUF_4 * TW_SLICE + TW_HALF)};    // B half of slice s + 4
        const TwDma da{abase, a_step, piece0 + (uint32_t)(BUF * TW_SLICE)};               // A half of slice s + 5 -> this slice's buffer
        const bool turn = do_colsum && cs_turn == 0;
        constexpr int N0 = KIND >= 1 ? 4 : 0, N1 = KIND == 2 ? 4 : 0;
        tw_slot<0, 0, FIRST, N0>(c, ra, rb, db, vob); tw_slot<0, 1, FIRST, N0>(c, ra, rb, db, vob);
        tw_slot<0, 2, FIRST, N0>(c, ra, rb, db, vob); tw_slot<0, 3, FIRST, N0>(c, ra, rb, db, vob);
        tw_slot<0, 4, FIRST, N0>(c, ra, rb, db, vob); tw_slot<0, 5, FIRST, N0>(c, ra, rb, db, vob);
        tw_slot<0, 6, FIRST, N0>(c, ra, rb, db, vob); tw_slot<0, 7, FIRST, N0>(c, ra, rb, db, vob);
        tw_slot<0, 8, FIRST, N0>(c, ra, rb, db, vob); tw_slot<0, 9, FIRST, N0>(c, ra, rb, db, vob);
        tw_slot<0, 10, FIRST, N0>(c, ra, rb, db, vob); tw_slot<0, 11, FIRST, N0>(c, ra, rb, db, vob);
        tw_slot<0, 12, FIRST, N0>(c, ra, rb, db, vob); tw_slot<0, 13, FIRST, N0>(c, ra, rb, db, vob);
        tw_slot<0, 14, FIRST, N0>(c, ra, rb, db, vob); tw_slot<0, 15, FIRST, N0>(c, ra, rb, db, vob);
        if (turn) {
            if (wn == 0) tw_colsum<0, 0>(c);
            else tw_colsum<0, 2>(c);
        }
        tw_wait<-1, 6>();             // set 1: B's fragments and A[0]
        tw_slot<1, 0, false, N1>(c, rna, rnb, da, voa); tw_slot<1, 1, false, N1>(c, rna, rnb, da, voa);
        tw_slot<1, 2, false, N1>(c, rna, rnb, da, voa); tw_slot<1, 3, false, N1>(c, rna, rnb, da, voa);
        tw_wait<(KIND >= 1 ? 24 : 0), 0>();       // every read of this slice has landed; so have this wave's pieces of slice s + 1
        tw_barrier();
        tw_slot<1, 4, false, N1>(c, rna, rnb, da, voa); tw_slot<1, 5, false, N1>(c, rna, rnb, da, voa);
        tw_slot<1, 6, false, N1>(c, rna, rnb, da, voa); tw_slot<1, 7, false, N1>(c, rna, rnb, da, voa);
        tw_slot<1, 8, false, N1>(c, rna, rnb, da, voa); tw_slot<1, 9, false, N1>(c, rna, rnb, da, voa);
        tw_slot<1, 10, false, N1>(c, rna, rnb, da, voa); tw_slot<1, 11, false, N1>(c, rna, rnb, da, voa);
        tw_slot<1, 12, false, N1>(c, rna, rnb, da, voa); tw_slot<1, 13, false, N1>(c, rna, rnb, da, voa);
        tw_slot<1, 14, false, N1>(c, rna, rnb, da, voa); tw_slot<1, 15, false, N1>(c, rna, rnb, da, voa);
        if (turn) {
            if (wn == 0) tw_colsum<1, 0>(c);
            else tw_colsum<1, 2>(c);
        }
        if (do_colsum) {
            if (cs_turn == 0) cs_turn = p.tiles_n;
            --cs_turn;
        }
    };
    auto run = [&](auto kind_tag, int& s, int send, int& ph) {
        using std::integral_constant;
        while (s < send) {
            switch (ph) {
            case 1: slice_body(integral_constant<int, 1>{}, std::false_type{}, kind_tag); ph = 2; if (++s == send) break; [[fallthrough]];
            case 2: slice_body(integral_constant<int, 2>{}, std::false_type{}, kind_tag); ph = 3; if (++s == send) break; [[fallthrough]];
            case 3: slice_body(integral_constant<int, 3>{}, std::false_type{}, kind_tag); ph = 4; if (++s == send) break; [[fallthrough]];
            case 4: slice_body(integral_constant<int, 4>{}, std::false_type{}, kind_tag); ph = 0; if (++s == send) break; [[fallthrough]];
            default: slice_body(integral_constant<int, 0>{}, std::false_type{}, kind_tag); ph = 1; ++s;
            }
        }
    };
    {
        using std::integral_constant;
        if (n > 5) slice_body(integral_constant<int, 0>{}, std::true_type{}, integral_constant<int, 2>{});
        else if (n == 5) slice_body(integral_constant<int, 0>{}, std::true_type{}, integral_constant<int, 1>{});
        else slice_body(integral_constant<int, 0>{}, std::true_type{}, integral_constant<int, 0>{});
        int s = 1, ph = 1;
        run(integral_constant<int, 2>{}, s, n - 5, ph);
        run(integral_constant<int, 1>{}, s, n - 4, ph);
        run(integral_constant<int, 0>{}, s, n, ph);
    }
    tw_wait<0, 0>();
#if TW_DEV
    if (TW_ABLATE & 16) return;
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");      // the last MFMAs' results are in the accumulator registers
#endif
    // split-K partial -> C by fp32 atomics, as the accumulator layout has them (a half-wave = 32 consecutive columns of one row)
    auto flush = [&](auto a_tag, auto b_tag, float* cbase) {
        constexpr int A = decltype(a_tag)::value, B = decltype(b_tag)::value;
        const f32x16_t t = tw_acc_read<A, B>(c);
        const int col = j0 + wn * 128 + B * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + wm * 128 + A * 32 + frag_row(r, lane);
            unsafeAtomicAdd(cbase + (int64_t)row * p.ldc + col, t[r]);
        }
    };
    // (the 256 addresses are formed row by row from a pointer hipcc cannot see through: hoisted as a whole they spill)
    auto flush_row = [&](auto a_tag) {
        using std::integral_constant;
        float* cb = p.C;
#if TW_DEV
        asm volatile("" : "+s"(cb));
#endif
        flush(a_tag, integral_constant<int, 0>{}, cb); flush(a_tag, integral_constant<int, 1>{}, cb);
        flush(a_tag, integral_constant<int, 2>{}, cb); flush(a_tag, integral_constant<int, 3>{}, cb);
    };
    flush_row(std::integral_constant<int, 0>{});
    flush_row(std::integral_constant<int, 1>{});
    flush_row(std::integral_constant<int, 2>{});
    flush_row(std::integral_constant<int, 3>{});
    if (do_colsum) {
        // this wave summed blocks 2 wn, 2 wn + 1 of its i half: column lane & 31, this lane's k half
        const float t0 = tw_colsum_get<0>(c), t1 = tw_colsum_get<1>(c);
        const float s0 = t0 + __shfl_xor(t0, 32, 64), s1 = t1 + __shfl_xor(t1, 32, 64);
        if (lane < 32) {
            unsafeAtomicAdd(p.colsum + i0 + wm * 128 + (2 * wn) * 32 + lane, s0);
            unsafeAtomicAdd(p.colsum + i0 + wm * 128 + (2 * wn + 1) * 32 + lane, s1);
        }
    }
}

int gemm_tn256o_launch(GemmTn256Params& p, int split_k, hipStream_t stream) {
    static DeviceOnce once;
    ensure_dynamic_lds(once, &gemm_tn256o_kernel, TW_SMEM);
    p.split_k = split_k;
    hipLaunchKernelGGL(gemm_tn256o_kernel, dim3(p.tiles_m * p.tiles_n * split_k), dim3(256), TW_SMEM, stream, p);
    return check_launch("maest_gemm_tn(256o)");
}

}  // namespace maest
#endif  // MAEST_OWNED_DISABLED
