// NT GEMM, 256 x 256 tile, ONE WAVE PER SIMD (bf16 operands): C[m][n] = epilogue(sum_k A[m][k] B[n][k]) for the large ViT
// linears (same contract and epilogues as gemm256.hip:gemm_nt256w_kernel, which it replaces for bf16 inputs; reference call
// sites: nn.Linear forward / dgrad, models/maest.py:353-376, 197-208).
//
// Why another kernel.  gemm_nt256w_kernel runs 8 waves of 128 x 64 outputs: per 64-deep K stage its waves read 192 KiB of fragments
// out of LDS against 64 KiB of LDS-DMA coming in -- 768 + 512 LDS cycles against 2048 matrix-pipe cycles -- and they hide their
// fragment reads behind the other wave group's MFMAs with four barriers per stage.  Its removal ablation
// (profiles/r02c_nt_mainloop_ablation.txt) puts the MFMA stream alone and the load stream alone at 60 % of the kernel each: what is
// lost is their overlap.  Here a workgroup is FOUR waves, one per SIMD, each owning 128 x 128 outputs:
//   * the 256 fp32 accumulators of a wave live in the accumulator half of the register file (a0 .. a255), the fragments of the two
//     k32 halves of a stage in v128 .. v255; both ranges are owned by this file -- touched only by the inline asm of gemm_nt_ow.h,
//     audited in the code object by maest_amd/build.py (maest_amd/pw_audit.py), exactly as attn_fwd_pw.hip does it;
//   * the matrix instruction is v_mfma_f32_16x16x32_bf16 (round 6; rounds 4 - 5: 32x32x16).  The kernel is bound by the clock the power
//     limit leaves, and the 16 x 16 shape costs less energy per flop: back-to-back MFMAs on random operands sustain 2090 against 1825
//     TFLOP/s, this kernel gained 3 - 6 % (profiles/r06_mfma_shape_power.txt; the vendor library's kernels at these shapes are 16 x 16
//     kernels and clock 9 % higher than the 32 x 32 form did: profiles/r06_gemm_vs_library.txt).  64 blocks of 16 x 16 per wave;
//   * per K stage a wave reads 32 KiB of fragments for 128 MFMAs (LDS: 512 + 512 cycles against 2048): one ds_read_b128 per three
//     MFMAs during the first 46 slots of a half, a whole half ahead of their use; the statements execute in program order, so the
//     source is the schedule -- one instruction stream per SIMD, no wave to take turns with, ONE barrier per stage (between its halves);
//   * an MFMA and what rides in its shadow are ONE asm statement (hipcc puts a wait state behind every asm statement, and a 16-cycle
//     MFMA has four issue slots);
//   * the operand ring is gemm_nt256w_kernel's: 128-byte rows (whole cache lines), units A_j / B_j of 256 rows = 32 KiB through five
//     buffers, source-side swizzle  chunk ^= (row >> 1) & 7  (conflict-free for the 16-row x 4-chunk fragment reads too), LDS-DMA
//     requests dealt out between the MFMAs (four per quarter stage), a counted vmcnt(8) in front of the barrier (only the unit
//     requested last may fly);
//   * the stage loop is unrolled over the ring's period with one body per stage kind, so that a stage boundary costs a counter, a
//     compare and a branch not taken;
//   * the first half's MFMAs of a tile take the constant 0 as their C operand (no clearing pass over 256 registers).
// The C tile leaves through LDS in four 64-row passes with the bias / GELU / GELU' / multiply / row-dot epilogues of gemm256_epi.h; 16-bit
// outputs (bf16, split rows) in every form, fp32 outputs in the plain and RESIDUAL forms -- the rest stays with gemm_nt256w_kernel.  Results: the products of a k32 half are
// rounded together where the 32 x 32 kernels round per 16: last-bit differences in fp32, at most one bf16 ulp in < 2 % of the outputs
// (tests/kernel_cases.py: _same_products); bit-equal to the 8-wave kernel in the host emulator, whose MFMA twins add term by term.
// Measured and dropped in round 6: the C tile packed to bf16 registers and stored from inside the next tile's main loop
// (profiles/r06_gemm_deferred_store.txt: register-direct row-piece stores are slower than the LDS-staged whole lines; full overlap would be
// worth 16 - 19 % at K = 768), column-panel tile order (MAEST_GEMM_PANEL: fewer fabric reads, no time).
#include "gemm256_epi.h"

#ifdef MAEST_OWNED_DISABLED
// maest_amd/build.py compiles this file with MAEST_OWNED_DISABLED when the audit of the code object fails (a hipcc that allocates
// registers differently from the validated one): the kernel is left out, the dispatch in gemm256.hip keeps the 8-wave kernel.
namespace maest {
bool gemm_nt256o_available() { return false; }
int gemm_nt256o_launch(Gemm256Params&, hipStream_t) {
    set_error("maest_gemm_nt(256o): the one-wave-per-SIMD kernel was left out of this build (register audit failed)");
    return MAEST_ERR_INVALID;
}
}  // namespace maest
#else

#define OW_PROF_VAR 1       // (this file defines the profiling variable of OW_PROF builds)
#include "gemm_nt_ow.h"     // the ring, the register map and the main-loop statements 

namespace maest {

bool gemm_nt256o_available() { return true; }

// ---- C tile -> LDS -> HBM: four passes, pass ps = m-tile ps of both wave rows (tile rows 128 wm + 32 ps ..+31: every wave stages
// 32 rows x 128 columns per pass), two staging buffers, one barrier per pass: the 16-byte stores of pass ps go in flight, then
// pass ps + 1 is read out of the accumulators, converted / activated and staged underneath them.  With one wave per SIMD nothing
// hides a latency for free, so: the tile's 256 bias values sit in LDS (fetched into a register per lane at kernel start: a
// global load per use was two thirds of this epilogue), a pass's LDS reads are issued as one batch, and the second operand of
// the RESIDUAL / MUL forms is fetched into registers a pass ahead (AuxRegs), behind the previous pass's drain.
// GMODE: 0 none, 1 GELU, 3 GELU + GELU' side output, 4 GELU written as MAEST_SPLIT3_A rows (hi and lo parts staged as the two regions of the
// pair form; C is bf16 [M, 3 N]: hi -> columns n and N + n, lo -> 2 N + n); MODE: 0 plain, 1 RESIDUAL (+ aux), 2 MUL (* aux), 3 ROWDOT (plain + row-dot side output)
template <int OSZ, int GMODE, int MODE, typename NEXT>
__device__ __forceinline__ void ow_epilogue_run(char* smem0, OwCtx& c, const Gemm256Params& p, int m0, int n0, int wm, int wn,
                                                int lane, int tid, NEXT&& request_next) {
    using E = EpiT<OSZ, 256>;
    constexpr bool SPLIT = GMODE == 4;
    constexpr bool PAIR = GMODE == 3 || SPLIT;
    static_assert(!SPLIT || (OSZ == 2 && MODE == 0), "split rows: bf16 thirds, no second operand");
    constexpr int REGION = 64 * E::PITCH;                       // one 64-row staging region: 33792 / 66560
    constexpr int BUF = (PAIR ? 2 : 1) * REGION;
    constexpr int BIAS0 = OW_BIAS0;                             // 256 floats behind the (largest) staging buffer, stored by the caller
    static_assert(BUF <= OW_BIAS0 && OW_EPI0 + BIAS0 + 1024 <= OW_SMEM, "the staging buffer and the bias row");
    char* smem = smem0 + OW_EPI0;                               // (the ring's first two units stay free for the next tile's A_0 B_0)
    constexpr int NCH = 32 * E::CPR / 256;                      // 16-byte chunks per thread and 32-row group: 4 / 8
    constexpr int RS = 256 / E::CPR;                            // rows between a thread's consecutive chunks: 8 / 4
    const int q16 = lane >> 4;
    // Accumulator block (n16, m16) of the 16 x 16 MFMA: this lane holds row 16 m16 + (lane & 15), columns 16 n16 + 4 q16 ..+3 of the wave's
    // 128 x 128.  A pass stages the wave's rows 32 ps ..+31 = the blocks m16 = 2 ps, 2 ps + 1 of all eight n16; this lane's bias quadruple of
    // a 16-column block (columns 128 wn + 16 n16 + 4 q16 ..+3) comes out of LDS one block ahead of its use.
    auto bias_blk = [&](int n16) { return *reinterpret_cast<const f32x4_t*>(smem + BIAS0 + (wn * 128 + n16 * 16 + 4 * q16) * 4); };
    auto stage_blk = [&](auto n_tag, auto m_tag, char* row, const f32x4_t& b4) {
        constexpr int N16 = decltype(n_tag)::value, M16 = decltype(m_tag)::value;
        const f32x4_t t = ow_acc_read<N16, M16>(c);
        float v[4], d[4];
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            const f32x2_t xv = {t[e] + b4[e], t[e + 1] + b4[e + 1]};
            f32x2_t gv = xv, dv = {0.0f, 0.0f};
            if (GMODE != 0) gelu_pair2<false>(xv, gv, dv);
            v[e] = gv[0]; v[e + 1] = gv[1];
            d[e] = dv[0]; d[e + 1] = dv[1];
        }
        char* dst = row + (N16 * 16) * OSZ;
        if (OSZ == 4) {
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            if (PAIR) *reinterpret_cast<float4*>(dst + REGION) = make_float4(d[0], d[1], d[2], d[3]);
        } else {
            chunk8 o;
            if (SPLIT) {
                uint32_t h0, l0, h1, l1;
                split_bf2(v[0], v[1], h0, l0);
                split_bf2(v[2], v[3], h1, l1);
                o[0] = h0; o[1] = h1;
                *reinterpret_cast<chunk8*>(dst) = o;
                const chunk8 q = {l0, l1};
                *reinterpret_cast<chunk8*>(dst + REGION) = q;
                return;
            }
            o[0] = pack_bf2(v[0], v[1]); o[1] = pack_bf2(v[2], v[3]);
            *reinterpret_cast<chunk8*>(dst) = o;
            if (PAIR) {
                chunk8 q;
                q[0] = pack_bf2(d[0], d[1]); q[1] = pack_bf2(d[2], d[3]);
                *reinterpret_cast<chunk8*>(dst + REGION) = q;
            }
        }
    };
    auto stage = [&](auto ps_tag, char* buf) {
        using std::integral_constant;
        constexpr int PS = decltype(ps_tag)::value;
        char* row0 = buf + (wm * 32 + (lane & 15)) * E::PITCH + (wn * 128 + 4 * q16) * OSZ;
        char* row1 = row0 + 16 * E::PITCH;
        constexpr integral_constant<int, 2 * PS> M0{};
        constexpr integral_constant<int, 2 * PS + 1> M1{};
        f32x4_t b = bias_blk(0), bn = bias_blk(1);
        stage_blk(integral_constant<int, 0>{}, M0, row0, b); stage_blk(integral_constant<int, 0>{}, M1, row1, b); b = bias_blk(2);
        stage_blk(integral_constant<int, 1>{}, M0, row0, bn); stage_blk(integral_constant<int, 1>{}, M1, row1, bn); bn = bias_blk(3);
        stage_blk(integral_constant<int, 2>{}, M0, row0, b); stage_blk(integral_constant<int, 2>{}, M1, row1, b); b = bias_blk(4);
        stage_blk(integral_constant<int, 3>{}, M0, row0, bn); stage_blk(integral_constant<int, 3>{}, M1, row1, bn); bn = bias_blk(5);
        stage_blk(integral_constant<int, 4>{}, M0, row0, b); stage_blk(integral_constant<int, 4>{}, M1, row1, b); b = bias_blk(6);
        stage_blk(integral_constant<int, 5>{}, M0, row0, bn); stage_blk(integral_constant<int, 5>{}, M1, row1, bn); bn = bias_blk(7);
        stage_blk(integral_constant<int, 6>{}, M0, row0, b); stage_blk(integral_constant<int, 6>{}, M1, row1, b);
        stage_blk(integral_constant<int, 7>{}, M0, row0, bn); stage_blk(integral_constant<int, 7>{}, M1, row1, bn);
    };
    // drain: thread t moves chunk cc = t % CPR of rows r0 + RS i (r0 = t / CPR) of a 32-row group; its pointers into C / aux are
    // formed once, a group's rows are wave-uniform multiples of the row pitch away.  Rows beyond M exist in the last tile row only.
    const int r0 = tid / E::CPR, cc = tid - r0 * E::CPR;
    const bool full = m0 + 256 <= p.M;                          // (block-uniform)
    const int64_t col = (int64_t)(n0 + cc * E::EPC) * OSZ;
    char* c_thr = reinterpret_cast<char*>(p.C) + (int64_t)(m0 + r0) * p.ldc * OSZ + col;
    int64_t c_row = p.ldc * OSZ, x_row = p.ld_aux * OSZ;        // (aux_in and aux_out have the output's element size)
#if OW_DEV
    asm volatile("" : "+s"(c_row), "+s"(x_row));                // (their multiples are formed where they are used, not in front of the tile loop)
#endif
    const char* a_col = reinterpret_cast<const char*>(p.aux_in) + col;
    const char* a_thr = a_col + (int64_t)(m0 + r0) * x_row;
    char* o_thr = reinterpret_cast<char*>(p.aux_out) + (int64_t)(m0 + r0) * x_row + col;
    const int l_thr = r0 * E::PITCH + cc * 16;
    chunk16 ax[2][NCH];                                         // [32-row group]: the RESIDUAL / MUL operand of the pass to drain
    auto prefetch = [&](int ps) {
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ro = half * 128 + ps * 32 + i * RS;   // (wave-uniform)
                const char* src = a_thr + ro * x_row;
                if (!full) src = m0 + r0 + ro < p.M ? src : a_col;      // (row 0: loaded, never stored)
                ax[half][i] = *reinterpret_cast<const chunk16*>(src);
            }
    };
    auto drain_one = [&](const char* src, char* dthr, int64_t drow, const chunk16* aux, int ro0) {
        chunk16 v[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) v[i] = *reinterpret_cast<const chunk16*>(src + l_thr + i * RS * E::PITCH);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            chunk16 o = v[i];
            if (aux != nullptr) o = apply_aux<OSZ, MODE>(o, aux[i]);
            if constexpr (MODE == 3) {
                // row-dot side output (gemm256.hip: drain256): the stored values times `other` over each 64-column group -- the
                // 8 / 16 lanes that hold a group's chunks are neighbours
                const chunk16 r = aux[i];
                float d = 0.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (OSZ == 4) d += u2f(o[e]) * u2f(r[e]);
                    else d += lo16f(o[e]) * lo16f(r[e]) + hi16f(o[e]) * hi16f(r[e]);
                }
                constexpr int GL = 64 * OSZ / 16;
#pragma unroll
                for (int m = 1; m < GL; m <<= 1) d += __shfl_xor(d, m, 64);
                const int gm = m0 + r0 + ro0 + i * RS, gn = n0 + cc * E::EPC;
                if ((cc & (GL - 1)) == 0 && gm < p.M) {
                    const int item = (gm + p.row0) / p.ntok, q = gm + p.row0 - item * p.ntok;
                    p.rowdot[((int64_t)item * (p.N >> 6) + (gn >> 6)) * p.ntok + q] = d;
                }
            }
            // streaming output: written once, re-read by a later kernel after > L2-size of other traffic
            if (full || m0 + r0 + ro0 + i * RS < p.M)
            {
                if (OW_ABLATE & 128) *reinterpret_cast<chunk16*>(dthr + (ro0 + i * RS) * drow) = o;       // (A/B: plain instead of streaming stores)
                else __builtin_nontemporal_store(o, reinterpret_cast<chunk16*>(dthr + (ro0 + i * RS) * drow));
            }
        }
    };
    auto drain = [&](int ps, const char* buf) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {                  // the two 32-row groups of a pass are 128 rows apart
            const char* src = buf + half * 32 * E::PITCH;
            drain_one(src, c_thr, c_row, MODE != 0 ? ax[half] : nullptr, half * 128 + ps * 32);
            if (SPLIT) {
                drain_one(src, c_thr + (int64_t)p.N * 2, c_row, nullptr, half * 128 + ps * 32);
                drain_one(src + REGION, c_thr + (int64_t)p.N * 4, c_row, nullptr, half * 128 + ps * 32);
            } else if (PAIR) drain_one(src + REGION, o_thr, x_row, nullptr, half * 128 + ps * 32);
        }
    };
    // One staging buffer: a pass is  stage -> barrier -> drain (its LDS reads, then the stores) -> barrier.  The barriers are raw
    // (LDS traffic only: __syncthreads would also wait for every store and for the next tile's requests).  The next tile's A_0 / B_0
    // are requested as early as no load of this epilogue can queue up behind them (loads return in order): at once without a second
    // operand, behind the last pass's operand fetch otherwise.
    using std::integral_constant;
    auto sync = [&]() { ow_sync_epilogue(); };
    auto consume_aux = [&]() {          // every ax register has arrived (hipcc places the wait) before what follows is issued
#if OW_DEV
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int i = 0; i < NCH; ++i) asm volatile("" : "+v"(ax[half][i]));
#endif
    };
    if (MODE == 0) request_next();
    if (MODE != 0) prefetch(0);
    stage(integral_constant<int, 0>{}, smem);
    OW_TICK(13);
    sync();
    OW_TICK(14);
    drain(0, smem);
    OW_TICK(15);
    if (MODE != 0) prefetch(1);
    sync();
    stage(integral_constant<int, 1>{}, smem);
    OW_TICK(13);
    sync();
    OW_TICK(14);
    drain(1, smem);
    OW_TICK(15);
    if (MODE != 0) prefetch(2);
    sync();
    stage(integral_constant<int, 2>{}, smem);
    OW_TICK(13);
    sync();
    OW_TICK(14);
    drain(2, smem);
    OW_TICK(15);
    if (MODE != 0) prefetch(3);
    sync();
    stage(integral_constant<int, 3>{}, smem);
    OW_TICK(13);
    sync();
    OW_TICK(14);
    if (MODE != 0) {
        consume_aux();
        request_next();
    }
    drain(3, smem);
    OW_TICK(15);
}
// One kernel per epilogue form (OSZ: bytes per output element; GMODE / MODE as above): the launcher picks.  (All forms inlined into one
// persistent kernel made hipcc hoist every form's loop invariants in front of the tile loop and spill them -- into the accumulator half.)
template <int OSZ, int GMODE, int MODE>
__global__ __launch_bounds__(256, 1) void gemm_nt256o_kernel(Gemm256Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
#if OW_DEV
    // the registers this file owns (the clobber makes the kernel descriptor allocate them)
    asm volatile("" : : : "a0", "a255", "v124", "v255");
#endif
    const int nwg = p.tiles_m * p.tiles_n;
    const int nstages = p.K >> 6;
    OwCtx c;
    c.wave = wave;
#if OW_DEV
    c.lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#else
    c.lds0 = 0;
    c.lds = smem;
#endif
    {
        // this lane's fragment chunk: row (lane & 15) of a 16-row block, 16-byte k chunk 4 kk + (lane >> 4) of the stage's 128-byte row, at the
        // place the source-side swizzle put it (chunk ^ (row >> 1) & 7; the block's first row is a multiple of 16, so only the lane enters)
        const int ra = wm * 128 + (lane & 15), rb = wn * 128 + (lane & 15);
        const int sw = (lane & 15) >> 1, q = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                c.pa[g][kk] = c.lds0 + (uint32_t)(g * 2 * OW_UNIT + ra * 128 + ((((4 * kk) | q) ^ sw) << 4));
                c.pb[g][kk] = c.lds0 + (uint32_t)(g * 2 * OW_UNIT + rb * 128 + ((((4 * kk) | q) ^ sw) << 4));
            }
        }
    }
    // The kernel can run PERSISTENT (gemm_nt256o_launch): workgroup b (XCD b % 8) walks the virtual block ids b, b + gridDim.x, ...
    // (gridDim.x a multiple of 8 whenever there is more than one round), so that an XCD still works through its contiguous tile range
    // round by round.
    // Tile order inside that range: row-major by default.  With p.panel_w > 0 the grid is walked in COLUMN PANELS of panel_w tiles,
    // row-major inside a panel (the last panel may be narrower): the 32 tiles an XCD has in flight then meet panel_w column tiles of B
    // instead of all tiles_n of them.  At N = 3072, K = 768 the whole B (4.7 MB) does not fit the XCD's 4 MiB L2 beside the A stream and
    // was re-fetched from the Infinity Cache every round (5.8 x the algorithmic reads, profiles/r05f_pmc_traffic.json); a panel of 6
    // (2.4 MB) stays resident and A is read once per panel (gemm_nt256o_launch picks the width).
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
    // LDS-DMA sources: piece i of this wave = rows 64 wave + 8 i ..+7 of a unit, lane l = row l >> 3, 16-byte chunk (l & 7) ^ swizzle;
    // offsets are relative to the tile's first row (rows beyond M / N repeat the last one: loaded, never stored)
    const char* abase = nullptr;
    const char* bbase = nullptr;
    uint32_t voa[8], vob[8];
    auto set_sources = [&](int tm0, int tn0) {
        abase = p.A + ((OW_ABLATE & 32) ? 0 : (int64_t)tm0 * p.lda * 2);      // (bit 5: every workgroup loads tile 0)
        bbase = p.B + ((OW_ABLATE & 32) ? 0 : (int64_t)tn0 * p.ldb * 2);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = (wave * 8 + i) * 8 + (lane >> 3);
            const uint32_t csrc = (uint32_t)((((lane & 7) ^ ((r >> 1) & 7))) << 4);
            const int ra = tm0 + r < p.M ? r : p.M - 1 - tm0;
            const int rb = tn0 + r < p.N ? r : p.N - 1 - tn0;
            voa[i] = (uint32_t)(ra * (int)p.lda * 2) + csrc;
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
#ifdef OW_PROF
    for (int i = 0; i < 24; ++i) c.prof[i] = 0;
    c.tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = c.tprev;
#endif
    int v = blockIdx.x, m0, n0;
    tile_of(v, m0, n0);
    set_sources(m0, n0);
    bool fresh = true;                // this tile's A_0 / B_0 are still to be requested (the first tile of the workgroup)
    for (;;) {
    // prologue: A_0 B_0 (a later tile finds them requested by the previous tile's epilogue), then A_1 and B_1 (unit 2 j + (B ? 1 : 0) lives in
    // buffer unit % 5) -- what stage 0 would find requested by the stages "-2" and "-1"; A_2 rides in stage 0 like in any other stage.
    if (fresh) {
        request(abase, voa, integral_constant<int, 0>{});
        request(bbase, vob, integral_constant<int, 1>{});
    }
    if (nstages > 1) {
        request(abase, voa, integral_constant<int, 2>{});
        request(bbase, vob, integral_constant<int, 3>{});
        ow_wait_vm<16>();             // stage 0 has landed (this wave's share): A_1 and B_1 may fly
    } else {
        ow_wait_vm<0>();
    }
    ow_barrier();
    {                                 // fragment set 0 <- k32 half 0 of stage 0
        const uint32_t la = c.pa[0][0], lb = c.pb[0][0];
        ow_read<0, 0, false>(c, la); ow_read<0, 1, false>(c, la); ow_read<0, 2, false>(c, la); ow_read<0, 3, false>(c, la);
        ow_read<0, 4, false>(c, la); ow_read<0, 5, false>(c, la); ow_read<0, 6, false>(c, la); ow_read<0, 7, false>(c, la);
        ow_read<0, 0, true, OW_UNIT>(c, lb); ow_read<0, 1, true, OW_UNIT>(c, lb); ow_read<0, 2, true, OW_UNIT>(c, lb); ow_read<0, 3, true, OW_UNIT>(c, lb);
        ow_read<0, 4, true, OW_UNIT>(c, lb); ow_read<0, 5, true, OW_UNIT>(c, lb); ow_read<0, 6, true, OW_UNIT>(c, lb); ow_read<0, 7, true, OW_UNIT>(c, lb);
    }
    const float* bias_src = p.bias != nullptr ? p.bias + n0 + 4 * lane : nullptr;
    const uint32_t bias_dst = c.lds0 + (uint32_t)(OW_EPI0 + OW_BIAS0 + lane * 16);
    ow_bias_load(c, bias_src);
    OW_TICK(10);                      // (prologue)
    // Stage j = two k32 halves of 64 MFMAs.  Half 0 multiplies fragment set 0 (read during the previous stage's half 1) and reads set 1 =
    // the stage's second k32 half; half 1 multiplies set 1 and reads set 0 = the NEXT stage's first half (the last stage reads the ring's next
    // buffers there: stale bytes nobody multiplies).  The stage's one barrier b_j stands BETWEEN the halves: in front of it every wave has read
    // its last fragments of stage j (the reads ride in the first 46 slots of half 0; lgkmcnt(0)) and has seen its pieces of stage j + 1 land
    // (vmcnt(8): only A_{j+2}, requested last, may fly), so behind it stage j + 1 may be read and stage j's two buffers refilled.  Requests, four
    // per 32-slot quarter (the memory front end takes a request per ~30 cycles and CU):
    //   half 0 of stage j:   A_{j+2} -> B_{j-1}'s buffer (free since b_{j-1})
    //   half 1 of stage j:   B_{j+2} -> A_j's buffer (free since b_j)
    // i.e. a request has at least a half stage (A: a whole one) to land.  Stage 0 is a stage like any other (its first half's MFMAs take C = 0).
    // The stage loop is unrolled over the ring's period (PH = j % 5): buffer addresses are constants, the per-stage scalar work is a counter and
    // two compares.  KIND 2: the units A_{j+2}, B_{j+2} exist (j + 2 < nstages); 0: the last two stages (no requests; they wait for everything).
    auto stage_body = [&](auto ph_tag, auto first_tag, auto kind_tag) {
        constexpr int PH = decltype(ph_tag)::value, KIND = decltype(kind_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int ABUF = (2 * PH) % 5, BBUF = (2 * PH + 1) % 5, ABUF_N = (2 * PH + 2) % 5, BBUF_N = (2 * PH + 3) % 5;
        constexpr int BBUF_P = (2 * PH + 4) % 5;             // B_{j-1}'s buffer
        constexpr int ND = KIND == 2 ? 4 : 0;
        OW_TICK(0);
        ow_wait_lds();                // set 0 is in the registers
        OW_TICK(1);
        ow_half<0, FIRST, ABUF, BBUF, 1, ND, 0, BBUF_P, 4, BBUF_P>(c, abase, voa, abase, voa, piece0);
        OW_TICK(2);
        ow_wait_lds();                // set 1 is in the registers (its reads are >= 16 slots old)
        OW_TICK(3);
        ow_wait_vm<(KIND == 2 ? 8 : 0)>();
        OW_TICK(8);
        ow_barrier();                 // b_j
        OW_TICK(9);
        ow_half<1, false, ABUF_N, BBUF_N, 0, ND, 0, ABUF, 4, ABUF>(c, bbase, vob, bbase, vob, piece0);
    };
    // stages [j, jend) starting at ring phase ph (= j % 5), all of one kind; steady state: a counter, a compare, a branch not taken
    auto run = [&](auto kind_tag, int& j, int jend, int& ph) {
        while (j < jend) {
            switch (ph) {
            case 1: stage_body(integral_constant<int, 1>{}, std::false_type{}, kind_tag); ph = 2; if (++j == jend) break; [[fallthrough]];
            case 2: stage_body(integral_constant<int, 2>{}, std::false_type{}, kind_tag); ph = 3; if (++j == jend) break; [[fallthrough]];
            case 3: stage_body(integral_constant<int, 3>{}, std::false_type{}, kind_tag); ph = 4; if (++j == jend) break; [[fallthrough]];
            case 4: stage_body(integral_constant<int, 4>{}, std::false_type{}, kind_tag); ph = 0; if (++j == jend) break; [[fallthrough]];
            default: stage_body(integral_constant<int, 0>{}, std::false_type{}, kind_tag); ph = 1; ++j;
            }
        }
    };
    {
        if (nstages > 2) stage_body(integral_constant<int, 0>{}, std::true_type{}, integral_constant<int, 2>{});
        else stage_body(integral_constant<int, 0>{}, std::true_type{}, integral_constant<int, 0>{});
        int j = 1, ph = 1;
        run(integral_constant<int, 2>{}, j, nstages - 2, ph);
        run(integral_constant<int, 0>{}, j, nstages, ph);
    }

    OW_TICK(0);                       // (the last half 1 counts as slot 0 of the next stage)
    ow_wait_vm<0>();
    ow_wait_lds();
    ow_barrier();                     // the ring is drained and read: LDS becomes the C staging area
    if (wave == 0) ow_bias_store(c, bias_dst);
    ow_wait_lds();
    ow_barrier();
#if OW_DEV
    if (OW_ABLATE & 16) return;
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");      // the last MFMAs' results are in the accumulator registers
#endif
    OW_TICK(11);                      // (drain)
    const int vn = v + (int)gridDim.x;
    const bool more = vn < nwg;       // (wave-uniform)
    int m0n = 0, n0n = 0;
    if (more) tile_of(vn, m0n, n0n);
    auto request_next = [&]() {
        if (more) {
            set_sources(m0n, n0n);
            request(abase, voa, integral_constant<int, 0>{});
            request(bbase, vob, integral_constant<int, 1>{});
        }
    };
    // (the epilogue's per-thread pointers are formed per tile, from values hipcc cannot see through: hoisted out of the tile loop --
    // ten inlined epilogue forms' worth of them -- they spilled over the main loop)
    int tid_e = tid, wave_e = wave;
#if OW_DEV
    asm volatile("" : "+v"(tid_e));
    asm volatile("" : "+s"(wave_e));
#endif
    ow_epilogue_run<OSZ, GMODE, MODE>(smem, c, p, m0, n0, wave_e >> 1, wave_e & 1, tid_e & 63, tid_e, request_next);
#ifdef OW_PROF
    if (v == (int)blockIdx.x) {       // (the workgroup's first tile)
        OW_TICK(12);
        if ((blockIdx.x == 5 || blockIdx.x == gridDim.x - 3) && g_ow_prof != nullptr && lane == 0) {
            unsigned long long* out = g_ow_prof + (blockIdx.x == 5 ? 0 : 96);
            for (int i = 0; i < 22; ++i) out[wave * 24 + i] = c.prof[i];
            out[wave * 24 + 22] = t_begin;
            out[wave * 24 + 23] = c.tprev;
        }
    }
#endif
    if (!more) break;
    v = vn;
    m0 = m0n;
    n0 = n0n;
    fresh = false;
    ow_sync_epilogue();               // every wave has read the last pass out of the staging area: A_1 / B_1 may be requested into it
    }
}
#ifdef OW_PROF
extern "C" int maest_debug_ow_prof(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ow_prof), &p, sizeof(p)); }
#endif

template <int OSZ, int GMODE, int MODE>
static int launch256o(Gemm256Params& p, hipStream_t stream) {
    static DeviceOnce once;
    ensure_dynamic_lds(once, &gemm_nt256o_kernel<OSZ, GMODE, MODE>, OW_SMEM);
    const int tiles = p.tiles_m * p.tiles_n;
    // Workgroups: one per tile by default -- the tile loop then runs once and the hardware deals the tiles out as CUs come free, which
    // matters when something else holds a few CUs (the gradient all-reduce's kernels during the backward: with a fixed tile list per
    // workgroup the ones that start late would double the kernel's time).  MAEST_OPT_GEMM_WGS = n > 0: at most n workgroups, each
    // walking tiles b, b + n, ... (256 = one per CU: 2-4 % faster stand-alone on the K = 768 shapes, equal in the single-GPU step).
    int cap = option(MAEST_OPT_GEMM_WGS);
    cap = cap < 1 ? tiles : (cap > 8 ? cap & ~7 : cap);   // more than one XCD's worth: a multiple of 8 (the tile -> XCD map)
    hipLaunchKernelGGL((gemm_nt256o_kernel<OSZ, GMODE, MODE>), dim3(tiles < cap ? tiles : cap), dim3(256), OW_SMEM, stream, p);
    return check_launch("maest_gemm_nt(256o)");
}

int gemm_nt256o_launch(Gemm256Params& p, hipStream_t stream) {
    const bool bf = p.out_dtype == MAEST_BF16, gelu = p.epi == MAEST_EPI_GELU;
    const int tiles = p.tiles_m * p.tiles_n;
    // Column panels (tile_of): MAEST_OPT_GEMM_PANEL = -1 chooses by a traffic estimate -- B beyond ~3 MB is re-fetched by every XCD in
    // every round (8 x rounds x B bytes); panels of ~2.5 MB of B stay resident and cost one more pass over A per extra panel --,
    // 0 = never, n > 0 = panels of n tiles wherever there are more than n column tiles.
    int pw = option(MAEST_OPT_GEMM_PANEL);
    if (pw < 0) {
        pw = 0;
        const int64_t bbytes = (int64_t)p.N * p.K * 2, abytes = (int64_t)p.M * p.K * 2;
        if (bbytes > ((int64_t)3 << 20)) {
            const int np0 = (int)((bbytes + ((int64_t)5 << 19) - 1) / ((int64_t)5 << 19));
            const int w = (p.tiles_n + np0 - 1) / np0;
            const int np = (p.tiles_n + w - 1) / w;
            const int64_t rounds = (tiles + 255) / 256;
            if (np > 1 && abytes * (np - 1) < bbytes * 8 * (rounds - 1)) pw = w;
        }
    }
    p.panel_w = pw > 0 && pw < p.tiles_n ? pw : 0;
    if (gelu && p.aux_out == nullptr && p.out_dtype == MAEST_SPLIT3_A) return launch256o<2, 4, 0>(p, stream);
    // (fp32 outputs: the plain and the RESIDUAL form only -- what the default evaluation mode's 3 K GEMMs write (bf16x3: qkv in fp32, proj / fc2 with
    // the fp32 residual add) and the patch embedding; their epilogues spill a few dozen registers around the tile loop with 128 fragment registers owned
    // here, which costs nothing measurable.  The other fp32-output forms stay with the eight-wave kernel.)
    if (p.epi == MAEST_EPI_RESIDUAL && !bf) return launch256o<4, 0, 1>(p, stream);
    if (p.epi == MAEST_EPI_NONE && p.out_dtype == MAEST_F32) return launch256o<4, 0, 0>(p, stream);
    if (gelu && p.aux_out != nullptr && bf) return launch256o<2, 3, 0>(p, stream);
    if (p.epi == MAEST_EPI_MUL && bf) return launch256o<2, 0, 2>(p, stream);
    if (gelu && p.aux_out == nullptr && bf) return launch256o<2, 1, 0>(p, stream);
    if (p.epi == MAEST_EPI_ROWDOT && bf) return launch256o<2, 0, 3>(p, stream);
    if (p.epi == MAEST_EPI_NONE && bf) return launch256o<2, 0, 0>(p, stream);
    set_error("maest_gemm_nt(256o): epilogue %d with output dtype %d is not served by this kernel", p.epi, p.out_dtype);
    return MAEST_ERR_INVALID;
}

}  // namespace maest
#endif  // MAEST_OWNED_DISABLED
