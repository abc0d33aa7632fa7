// NT GEMM, 256 x 256 tile, one wave per SIMD, with the C tile's store DEFERRED into the next tile's main loop (bf16 operands, plain
// bf16 output with an optional bias: the qkv / proj / fc2 linears of a forward and the dgrads without a second operand -- 57 of the 91
// NT GEMM calls of a training step, every GEMM of an inference step but fc1; reference call sites: nn.Linear, models/maest.py:353-376,
// 197-208).  Same operand ring, register map and main-loop statements as gemm_nt256o_kernel (gemm_nt_ow.h); what differs is how a tile
// leaves.
//
// Why.  gemm_nt256o_kernel's epilogue is 8 k ticks per tile (stage through LDS 4.5 k, two barriers per pass, drain 3.1 k), the next tile's
// prologue another 3 - 4 k, against 25 k of main loop at K = 768 (profiles/r04_gemm_ow_timeline.txt): a third of a K = 768 tile is spent with
// the matrix pipe idle, and all 256 CUs write their 128 KiB at the same time -- 32 MiB at the HBM write rate is 5 us = the epilogue.  Here
//   * at the end of a tile's K loop every wave converts its 256 accumulators (+ bias) to 128 registers of packed bf16 pairs (v64 .. v191,
//     owned like the fragment registers): 16 v_accvgpr_read + 16 v_add (bias by DPP row broadcast out of four registers) + 8
//     v_cvt_pk_bf16_f32 per 32 x 32 block -- the only part that is not overlapped; the next tile's first operand units are requested in
//     front of it and land underneath;
//   * the packed tile is stored STRAIGHT FROM THE REGISTERS during stages 0 .. 3 of the next tile: two v_permlane32_swap give every lane
//     16 contiguous bytes of a row (a lane pair = 32 bytes, 32 rows per instruction), one global_store_dwordx4 per MFMA gap in the
//     k-steps that carry the A requests; no LDS staging (the ring keeps all 160 KiB), no barrier, and the chip's C writes are spread
//     over the main loops instead of bunched into bursts.  Row-piece stores stream at the same bytes / ns as whole lines once the footprint
//     leaves the L2s (profiles/r03_store_issue_probe.txt).
//   * vmcnt counts the stores with the LDS-DMA requests, in issue order: a store stage waits with vmcnt(16) (its 8 requests + 8 stores
//     may fly) where the other stages say vmcnt(8).
// Results: the same products in the same order, acc + bias rounded once to bf16 -- bit-equal to gemm_nt256o_kernel<2, 0, 0>.
// Shapes: M a multiple of 256 (complete tile rows: the stores carry no row mask), K >= 6 stages; everything else stays with
// gemm_nt256o_kernel (gemm_nt256o_launch decides).
#include "gemm256_epi.h"

#ifdef MAEST_OWNED_DISABLED
namespace maest {
bool gemm_nt256d_available() { return false; }
int gemm_nt256d_launch(Gemm256Params&, hipStream_t) {
    set_error("maest_gemm_nt(256d): the deferred-store kernel was left out of this build (register audit failed)");
    return MAEST_ERR_INVALID;
}
}  // namespace maest
#else

// the packed C tile: owned on every wait / barrier of the main loop (gemm_nt_ow.h appends this to their clobber lists)
#define OW_MORE_OWNED , "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187"
#include "gemm_nt_ow.h"

namespace maest {

bool gemm_nt256d_available() { return true; }

constexpr int OD_P = 64;              // packed C: block (nt, mt) = v[OD_P + 8 (4 nt + mt) ..+7], register p = columns 32 nt + 8 (p / 2) + 4 h + 2 (p % 2) ..+1 of row 32 mt + (lane & 31)
constexpr int OD_T = OW_V_F;          // the pack's temporaries: the fragment registers (nothing is in flight into them at a tile boundary)
constexpr int OD_BV = OW_V_BIAS;      // v188 .. v191: BV[nt], lane 16 rho + k = bias[n0 + 128 wn + 32 nt + 8 (k / 4) + 4 (rho / 2) + k % 4].  These are ALSO the
                                      // last four registers of the packed tile (block 15, second half): the bias is loaded behind the stage that stores them
                                      // (stage 3), and the pack writes them last -- block 15's conversions follow its bias adds
static_assert(OD_BV == OD_P + 124, "the bias registers are the packed tile's last four");
constexpr int OD_MIN_STAGES = 6;

struct OdCtx : OwCtx {
#if !OW_DEV
    uint32_t pk[128];                 // (host emulator: the packed tile and the bias values by accumulator register)
    float bv[4][16];
#endif
};

// ---- the tile's bias values, laid out for DPP row broadcasts: one dword load per 32-column block; bias == nullptr: zeros are never added
__device__ __forceinline__ void od_bias_load(OdCtx& c, const float* bias, int n0, int wn, int lane) {
    if (bias == nullptr) return;
    const int rho = lane >> 4, k = lane & 15;
    const float* src = bias + n0 + wn * 128;                                              // (wave-uniform: an SGPR pair)
    const uint32_t off = (uint32_t)((8 * (k >> 2) + 4 * (rho >> 1) + (k & 3)) * 4);       // this lane's place in a 32-column block
#if OW_DEV
    asm volatile("global_load_dword v%c2, %0, %1\n\tglobal_load_dword v%c3, %0, %1 offset:128\n\t"
                 "global_load_dword v%c4, %0, %1 offset:256\n\tglobal_load_dword v%c5, %0, %1 offset:384"
                 : : "v"(off), "s"(src), "i"(OD_BV), "i"(OD_BV + 1), "i"(OD_BV + 2), "i"(OD_BV + 3) : "memory", OW_FRAGS);
#else
    (void)src; (void)off;
    const int h = lane >> 5;
    for (int nt = 0; nt < 4; ++nt)
        for (int r = 0; r < 16; ++r) c.bv[nt][r] = bias[n0 + wn * 128 + 32 * nt + 8 * (r >> 2) + 4 * h + (r & 3)];
#endif
}

// ---- pack: accumulator block BLK (= 4 nt + mt) -> 8 registers of bf16 pairs.  ONE asm statement per block (hipcc puts a wait state
// behind every inline-asm statement: 640 one-instruction statements would have doubled the only exposed part of this kernel); register
// numbers are formed by the assembler from four bases: %c0 accumulators, %c1 temporaries, %c2 packed registers, %c3 the bias register
#if OW_DEV
#define OD_RD(r) "v_accvgpr_read_b32 v[%c1+" #r "], a[%c0+" #r "]\n\t"
#define OD_AD(r) "v_add_f32_dpp v[%c1+" #r "], v[%c3], v[%c1+" #r "] row_newbcast:" #r " row_mask:0xf bank_mask:0xf\n\t"
#define OD_CV(q, r0, r1) "v_cvt_pk_bf16_f32 v[%c2+" #q "], v[%c1+" #r0 "], v[%c1+" #r1 "]\n\t"
#define OD_RD16 OD_RD(0) OD_RD(1) OD_RD(2) OD_RD(3) OD_RD(4) OD_RD(5) OD_RD(6) OD_RD(7) OD_RD(8) OD_RD(9) OD_RD(10) OD_RD(11) OD_RD(12) OD_RD(13) OD_RD(14) OD_RD(15)
#define OD_AD16 OD_AD(0) OD_AD(1) OD_AD(2) OD_AD(3) OD_AD(4) OD_AD(5) OD_AD(6) OD_AD(7) OD_AD(8) OD_AD(9) OD_AD(10) OD_AD(11) OD_AD(12) OD_AD(13) OD_AD(14) OD_AD(15)
#define OD_CV8 OD_CV(0, 0, 1) OD_CV(1, 2, 3) OD_CV(2, 4, 5) OD_CV(3, 6, 7) OD_CV(4, 8, 9) OD_CV(5, 10, 11) OD_CV(6, 12, 13) OD_CV(7, 14, 15)
template <int BLK, bool BIAS>
__device__ __forceinline__ void od_pack_block_dev() {
    constexpr int T = OD_T + 16 * (BLK & 3);          // four temporary sets: a block's chain never waits for the previous block's
    if constexpr (BIAS)
        asm volatile(OD_RD16 OD_AD16 OD_CV8 : : "i"(16 * BLK), "i"(T), "i"(OD_P + 8 * BLK), "i"(OD_BV + (BLK >> 2)) : OW_FRAGS);
    else
        asm volatile(OD_RD16 OD_CV8 : : "i"(16 * BLK), "i"(T), "i"(OD_P + 8 * BLK), "i"(OD_BV + (BLK >> 2)) : OW_FRAGS);
}
#endif
template <int BLK, bool BIAS>
__device__ __forceinline__ void od_pack_block(OdCtx& c) {
#if OW_DEV
    od_pack_block_dev<BLK, BIAS>();
#else
    const f32x16_t& a = c.acc[BLK >> 2][BLK & 3];
    for (int p = 0; p < 8; ++p) {
        float x0 = a[2 * p], x1 = a[2 * p + 1];
        if (BIAS) { x0 += c.bv[BLK >> 2][2 * p]; x1 += c.bv[BLK >> 2][2 * p + 1]; }
        c.pk[8 * BLK + p] = pack_bf2(x0, x1);
    }
#endif
}
template <bool BIAS, int... BLK>
__device__ __forceinline__ void od_pack_all(OdCtx& c, std::integer_sequence<int, BLK...>) {
    (od_pack_block<BLK, BIAS>(c), ...);
}

// ---- one deferred store: chunk CI (0 .. 31) = the 16-column half J = CI & 1 of block CI >> 1.  Two lane-half swaps make the four registers of
// the half hold 8 consecutive columns: lanes 0-31 columns 16 J ..+7, lanes 32-63 columns 16 J + 8 ..+7 of row 32 mt + (lane & 31).
// voff: this lane's byte offset ((128 wm + (lane & 31)) ldc + 128 wn) 2 + 16 (lane >> 5); cb: the tile's C pointer moved down 32 mt rows.
template <int CI>
__device__ __forceinline__ void od_store(OdCtx& c, uint32_t voff, const char* cb) {
    constexpr int BLK = CI >> 1, J = CI & 1, NT = BLK >> 2;
    constexpr int P = OD_P + 8 * BLK + 4 * J, OFF = (32 * NT + 16 * J) * 2;
#if OW_DEV
    asm volatile("v_permlane32_swap_b32 v%c0, v%c1\n\tv_permlane32_swap_b32 v%c2, v%c3\n\tglobal_store_dwordx4 %4, v[%c0:%c3], %5 offset:%c6 nt"
                 : : "i"(P), "i"(P + 2), "i"(P + 1), "i"(P + 3), "v"(voff), "s"(cb), "i"(OFF) : "memory", OW_FRAGS);
#else
    uint32_t* q = c.pk + 8 * BLK + 4 * J;
    const auto r0 = __builtin_amdgcn_permlane32_swap(q[0], q[2], false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(q[1], q[3], false, false);
    q[0] = r0[0]; q[2] = r0[1]; q[1] = r1[0]; q[3] = r1[1];
    *reinterpret_cast<chunk16*>(const_cast<char*>(cb) + voff + OFF) = chunk16{q[0], q[1], q[2], q[3]};
#endif
}
// C pointers of a tile for the four 32-row blocks of a wave (the store picks by its block's mt)
struct OdBases {
    const char* b[4];
};
template <int CI>
__device__ __forceinline__ void od_store_ci(OdCtx& c, uint32_t voff, const OdBases& cb) {
    od_store<CI>(c, voff, cb.b[(CI >> 1) & 3]);
}
template <int... CI>
__device__ __forceinline__ void od_store_all(OdCtx& c, uint32_t voff, const OdBases& cb, std::integer_sequence<int, CI...>) {
    (od_store_ci<CI>(c, voff, cb), ...);
}

// One slot of a k-step (gemm_nt_ow.h: ow_slot) with the deferred stores: SST = 0 .. 3 (store stage; -1 none): the odd slots 9 .. 15 of
// k-steps 1 and 2 store chunks 8 SST + 4 (S - 1) + (Q - 9) / 2
template <int S, int Q, bool ZERO, int NDMA, int I0, int RA, int RB, int KS, int DBUF, int SST>
__device__ __forceinline__ void od_slot(OdCtx& c, const char* base, uint32_t (&vo)[8], uint32_t piece0, uint32_t voff, const OdBases& cb) {
    ow_slot<S, Q, ZERO, NDMA, I0, RA, RB, KS, DBUF>(c, base, vo, piece0);
    if constexpr (SST >= 0 && (S == 1 || S == 2) && Q >= 9 && (Q & 1) == 1)
        od_store_ci<8 * SST + 4 * (S - 1) + ((Q - 9) >> 1)>(c, voff, cb);
}
template <int S, bool ZERO, int NDMA, int I0, int RA, int RB, int KS, int DBUF, int SST, int... Q>
__device__ __forceinline__ void od_step_slots(OdCtx& c, const char* base, uint32_t (&vo)[8], uint32_t piece0, uint32_t voff, const OdBases& cb,
                                              std::integer_sequence<int, Q...>) {
    (od_slot<S, Q, ZERO, NDMA, I0, RA, RB, KS, DBUF, SST>(c, base, vo, piece0, voff, cb), ...);
}
template <int S, bool ZERO, int NDMA, int I0, int RA, int RB, int KS, int DBUF, int SST>
__device__ __forceinline__ void od_step(OdCtx& c, const char* base, uint32_t (&vo)[8], uint32_t piece0, uint32_t voff, const OdBases& cb) {
    od_step_slots<S, ZERO, NDMA, I0, RA, RB, KS, DBUF, SST>(c, base, vo, piece0, voff, cb, std::make_integer_sequence<int, 16>{});
}

template <bool BIAS>
__global__ __launch_bounds__(256, 1) void gemm_nt256d_kernel(Gemm256Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
#if OW_DEV
    asm volatile("" : : : "a0", "a255", "v64", "v255");       // the registers this file owns (the clobber makes the kernel descriptor allocate them)
#endif
    const int nwg = p.tiles_m * p.tiles_n;
    const int nstages = p.K >> 6;
    OdCtx c;
    c.wave = wave;
#if OW_DEV
    c.lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#else
    c.lds0 = 0;
    c.lds = smem;
#endif
    {
        const int ra = wm * 128 + (lane & 31), rb = wn * 128 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                c.pa[g][ks] = c.lds0 + (uint32_t)(g * 2 * OW_UNIT + ra * 128 + ((((2 * ks) | h) ^ ((ra >> 1) & 7)) << 4));
                c.pb[g][ks] = c.lds0 + (uint32_t)(g * 2 * OW_UNIT + rb * 128 + ((((2 * ks) | h) ^ ((rb >> 1) & 7)) << 4));
            }
        }
    }
    // tile order: as gemm_nt256o_kernel (XCD-contiguous ranges, optional column panels)
    auto tile_of = [&](int v, int& tm0, int& tn0) {
        const int wg = xcd_remap(v, nwg);
        int tile_m, tile_n;
        if (p.panel_w > 0) {
            const int per = p.tiles_m * p.panel_w;
            const int pn = wg / per, rem = wg - pn * per;
            const int left = p.tiles_n - pn * p.panel_w;
            const int w = left < p.panel_w ? left : p.panel_w;
            tile_m = rem / w;
            tile_n = pn * p.panel_w + rem - tile_m * w;
        } else {
            tile_m = wg / p.tiles_n;
            tile_n = wg - tile_m * p.tiles_n;
        }
        tm0 = tile_m * 256;
        tn0 = tile_n * 256;
    };
    const char* abase = nullptr;
    const char* bbase = nullptr;
    uint32_t voa[8], vob[8];
    // (per-lane values of the per-tile code are formed from a lane id hipcc cannot see through: hoisted out of the tile loop -- sixteen row
    // offsets, the bias pointer -- they were kept across the main loop in registers the kernel does not have: 21 spills)
    auto opaque_lane = [&]() {
        int l = lane;
#if OW_DEV
        asm volatile("" : "+v"(l));
#endif
        return l;
    };
    auto set_sources = [&](int tm0, int tn0) {       // (M is a multiple of 256 here; N's last rows are clamped as in gemm_nt256o_kernel)
        const int lane = opaque_lane();
        abase = p.A + (int64_t)tm0 * p.lda * 2;
        bbase = p.B + (int64_t)tn0 * p.ldb * 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = (wave * 8 + i) * 8 + (lane >> 3);
            const uint32_t csrc = (uint32_t)((((lane & 7) ^ ((r >> 1) & 7))) << 4);
            const int rb = tn0 + r < p.N ? r : p.N - 1 - tn0;
            voa[i] = (uint32_t)(r * (int)p.lda * 2) + csrc;
            vob[i] = (uint32_t)(rb * (int)p.ldb * 2) + csrc;
        }
    };
    const uint32_t piece0 = c.lds0 + (uint32_t)(wave * 8 * 1024);
    auto request = [&](const char* base, uint32_t (&vo)[8], auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        ow_dma<BUF * OW_UNIT + 0 * 1024>(base, vo[0], piece0, c); ow_dma<BUF * OW_UNIT + 1 * 1024>(base, vo[1], piece0, c);
        ow_dma<BUF * OW_UNIT + 2 * 1024>(base, vo[2], piece0, c); ow_dma<BUF * OW_UNIT + 3 * 1024>(base, vo[3], piece0, c);
        ow_dma<BUF * OW_UNIT + 4 * 1024>(base, vo[4], piece0, c); ow_dma<BUF * OW_UNIT + 5 * 1024>(base, vo[5], piece0, c);
        ow_dma<BUF * OW_UNIT + 6 * 1024>(base, vo[6], piece0, c); ow_dma<BUF * OW_UNIT + 7 * 1024>(base, vo[7], piece0, c);
    };
    using std::integral_constant;
    // the stores' addressing: one per-lane byte offset for the whole kernel, four wave-uniform pointers per tile
    const uint32_t voff = (uint32_t)(((wm * 128 + (lane & 31)) * (int)p.ldc + wn * 128) * 2 + 16 * h);
    auto bases_of = [&](int tm0, int tn0) {
        OdBases cb;
        const char* c0 = reinterpret_cast<const char*>(p.C) + ((int64_t)tm0 * p.ldc + tn0) * 2;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) cb.b[mt] = c0 + (int64_t)(32 * mt) * p.ldc * 2;
        return cb;
    };
    int v = blockIdx.x, m0, n0;
    tile_of(v, m0, n0);
    set_sources(m0, n0);
    bool fresh = true;                // this tile's A_0 / B_0 are still to be requested (the first tile of the workgroup)
    bool have_prev = false;           // the previous tile's packed C is still to be stored (every tile but the workgroup's first)
    OdBases prev = bases_of(m0, n0);
    for (;;) {
    // prologue: as gemm_nt256o_kernel (a later tile finds A_0 / B_0 requested in front of the previous tile's pack)
    if (fresh) {
        request(abase, voa, integral_constant<int, 0>{});
        request(bbase, vob, integral_constant<int, 1>{});
    }
    request(abase, voa, integral_constant<int, 2>{});
    ow_dma<3 * OW_UNIT + 0 * 1024>(bbase, vob[0], piece0, c); ow_dma<3 * OW_UNIT + 1 * 1024>(bbase, vob[1], piece0, c);
    ow_dma<3 * OW_UNIT + 2 * 1024>(bbase, vob[2], piece0, c); ow_dma<3 * OW_UNIT + 3 * 1024>(bbase, vob[3], piece0, c);
    ow_wait_vm<12>();                 // stage 0 has landed (this wave's share): A_1 and half of B_1 may fly
    ow_barrier();
    {
        const uint32_t la = c.pa[0][0], lb = c.pb[0][0];
        ow_read<0, 0, false>(c, la); ow_read<0, 1, false>(c, la); ow_read<0, 2, false>(c, la); ow_read<0, 3, false>(c, la);
        ow_read<0, 0, true, OW_UNIT>(c, lb); ow_read<0, 1, true, OW_UNIT>(c, lb); ow_read<0, 2, true, OW_UNIT>(c, lb); ow_read<0, 3, true, OW_UNIT>(c, lb);
    }
    // Stages as in gemm_nt256o_kernel (see there): PH = ring phase, KIND 2 steady / 1 last but one / 0 last; SST = 0 .. 3: the stage also
    // stores a quarter of the previous tile (4 chunks in k-step 1, 4 in k-step 2, between the A requests), and its counted wait lets
    // those 16 operations fly
    auto stage_body = [&](auto ph_tag, auto first_tag, auto kind_tag, auto sst_tag) {
        constexpr int PH = decltype(ph_tag)::value, KIND = decltype(kind_tag)::value, SST = decltype(sst_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        static_assert(SST < 0 || KIND == 2, "store stages are steady stages");
        constexpr int ABUF = (2 * PH) % 5, BBUF = (2 * PH + 1) % 5, ABUF_N = (2 * PH + 2) % 5, BBUF_N = (2 * PH + 3) % 5;
        constexpr int BBUF_P = (2 * PH + 4) % 5;
        ow_wait_lds();
        od_step<0, FIRST, (KIND >= 1 ? 4 : 0), 4, ABUF, BBUF, 1, BBUF_N, SST>(c, bbase, vob, piece0, voff, prev);
        ow_wait_lds();
        od_step<1, false, (KIND == 2 ? 4 : 0), 0, ABUF, BBUF, 2, BBUF_P, SST>(c, abase, voa, piece0, voff, prev);
        ow_wait_lds();
        od_step<2, false, (KIND == 2 ? 4 : 0), 4, ABUF, BBUF, 3, BBUF_P, SST>(c, abase, voa, piece0, voff, prev);
        ow_wait_lds();
        ow_wait_vm<(KIND == 2 ? (SST >= 0 ? 16 : 8) : 0)>();
        ow_barrier();                 // b_j
        od_step<3, false, (KIND == 2 ? 4 : 0), 0, ABUF_N, BBUF_N, 0, ABUF, SST>(c, bbase, vob, piece0, voff, prev);
    };
    constexpr integral_constant<int, -1> NOST{};
    auto run = [&](auto kind_tag, int& j, int jend, int& ph) {
        while (j < jend) {
            switch (ph) {
            case 1: stage_body(integral_constant<int, 1>{}, std::false_type{}, kind_tag, NOST); ph = 2; if (++j == jend) break; [[fallthrough]];
            case 2: stage_body(integral_constant<int, 2>{}, std::false_type{}, kind_tag, NOST); ph = 3; if (++j == jend) break; [[fallthrough]];
            case 3: stage_body(integral_constant<int, 3>{}, std::false_type{}, kind_tag, NOST); ph = 4; if (++j == jend) break; [[fallthrough]];
            case 4: stage_body(integral_constant<int, 4>{}, std::false_type{}, kind_tag, NOST); ph = 0; if (++j == jend) break; [[fallthrough]];
            default: stage_body(integral_constant<int, 0>{}, std::false_type{}, kind_tag, NOST); ph = 1; ++j;
            }
        }
    };
    {
        constexpr integral_constant<int, 2> K2{};
        if (have_prev) {              // (nstages >= 6: stages 0 .. 3 are steady stages)
            stage_body(integral_constant<int, 0>{}, std::true_type{}, K2, integral_constant<int, 0>{});
            stage_body(integral_constant<int, 1>{}, std::false_type{}, K2, integral_constant<int, 1>{});
            stage_body(integral_constant<int, 2>{}, std::false_type{}, K2, integral_constant<int, 2>{});
            stage_body(integral_constant<int, 3>{}, std::false_type{}, K2, integral_constant<int, 3>{});
        } else {
            stage_body(integral_constant<int, 0>{}, std::true_type{}, K2, NOST);
            stage_body(integral_constant<int, 1>{}, std::false_type{}, K2, NOST);
            stage_body(integral_constant<int, 2>{}, std::false_type{}, K2, NOST);
            stage_body(integral_constant<int, 3>{}, std::false_type{}, K2, NOST);
        }
        // the tile's bias values, behind the last store of the previous tile (their registers are the packed tile's last four); stage 4's
        // counted wait -- or the last stages' vmcnt(0) -- covers the loads
        if (BIAS) od_bias_load(c, p.bias, n0, wn, opaque_lane());
        int j = 4, ph = 4;
        run(integral_constant<int, 2>{}, j, nstages - 2, ph);
        run(integral_constant<int, 1>{}, j, nstages - 1, ph);
        run(integral_constant<int, 0>{}, j, nstages, ph);
    }
    // ---- tile boundary.  Behind the last stage's barrier no wave reads operands out of the ring any more (its last k-step read stale
    // bytes nobody multiplies): the next tile's A_0 / B_0 are requested at once and land under the pack.
    ow_wait_lds();                    // (those last reads: the fragment registers become the pack's temporaries)
    const int vn = v + (int)gridDim.x;
    const bool more = vn < nwg;       // (wave-uniform)
    int m0n = 0, n0n = 0;
    if (more) {
        tile_of(vn, m0n, n0n);
        set_sources(m0n, n0n);
        request(abase, voa, integral_constant<int, 0>{});
        request(bbase, vob, integral_constant<int, 1>{});
    }
    prev = bases_of(m0, n0);
#if OW_DEV
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : : : OW_FRAGS);      // the last MFMAs' results are in the accumulator registers
#endif
    od_pack_all<BIAS>(c, std::make_integer_sequence<int, 16>{});
    if (!more) {
        od_store_all(c, voff, prev, std::make_integer_sequence<int, 32>{});
        break;
    }
    have_prev = true;
    v = vn;
    m0 = m0n;
    n0 = n0n;
    fresh = false;
    }
}

template <bool BIAS>
static int launch256d(Gemm256Params& p, hipStream_t stream) {
    static DeviceOnce once;
    ensure_dynamic_lds(once, &gemm_nt256d_kernel<BIAS>, OW_SMEM);
    const int tiles = p.tiles_m * p.tiles_n;
    int cap = option(MAEST_OPT_GEMM_WGS);
    cap = cap < 1 ? tiles : (cap > 8 ? cap & ~7 : cap);
    hipLaunchKernelGGL((gemm_nt256d_kernel<BIAS>), dim3(tiles < cap ? tiles : cap), dim3(256), OW_SMEM, stream, p);
    return check_launch("maest_gemm_nt(256d)");
}

// whether gemm_nt256d_kernel takes the call (gemm_nt256o_launch asks): plain bf16 output, complete tile rows, enough K stages
bool gemm_nt256d_takes(const Gemm256Params& p) {
    return p.epi == MAEST_EPI_NONE && p.out_dtype == MAEST_BF16 && (p.M % 256) == 0 && (p.K >> 6) >= OD_MIN_STAGES &&
           (int64_t)256 * p.ldc * 2 + 512 < ((int64_t)1 << 31);      // (the stores' per-lane offset is 32 bits)
}
int gemm_nt256d_launch(Gemm256Params& p, hipStream_t stream) {
    return p.bias != nullptr ? launch256d<true>(p, stream) : launch256d<false>(p, stream);
}

}  // namespace maest
#endif  // MAEST_OWNED_DISABLED
