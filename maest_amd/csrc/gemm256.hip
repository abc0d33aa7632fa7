// 256x256-tile NT GEMM for the large ViT linears (same contract and epilogues as gemm.hip:gemm_nt_kernel;
// reference call sites: nn.Linear forward / dgrad, models/maest.py:353-376, 197-208).
//
// Why a second kernel: at 128x128 the global->LDS operand traffic per flop (0.0152 B/flop, ~11 TB/s at
// 700 TFLOP/s) is what caps the MFMA pipe; a 256x256 tile halves it.  Structure (gfx950):
//   * 512 threads = 8 waves as 2 (m) x 4 (n); each wave owns 128 x 64 outputs = 4 x 2 MFMA 32x32 tiles,
//     128 fp32 accumulators per lane (one workgroup per CU, 2 waves per SIMD).
//   * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction), no
//     VGPR round trip.  K is walked in 64-BYTE slices per row (32 bf16 / 16 fp32) through a 4-deep
//     ring of 32 KiB LDS buffers: the loads of slice s+3 are issued right after the barrier that opens
//     slice s, and a COUNTED s_waitcnt vmcnt(8) (never 0 in the loop) retires only slice s's loads, so
//     three slices are always in flight across the (raw) barriers and HBM/L2 latency never reaches the
//     MFMA pipe.  One barrier per slice.
//   * LDS-DMA writes are lane-linear (16 rows of 64 B per instruction), so the bank-conflict swizzle is
//     applied on the SOURCE address: the 16-byte chunk that lands at position p of row r is global chunk
//     p ^ ((r >> 2) & 3); fragment reads XOR the same value.  A 16-lane ds_read_b128 group then covers
//     all 16 bank slots exactly once.
//   * epilogue as in gemm.hip: weight tile is the MFMA A operand, so lanes hold 4 consecutive output
//     columns; the C tile is staged through (the now idle) LDS in the output dtype in row groups and
//     written with 16-byte coalesced stores with bias / GELU / residual / GELU' fused.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm256_epi.h"

namespace maest {

constexpr int G2_ROWB = 64;                 // bytes per row per K slice
constexpr int G2_TILE = 256 * G2_ROWB;      // 16384: one operand tile of one slice
constexpr int G2_STAGES = 4;
constexpr int G2_SMEM = G2_STAGES * 2 * G2_TILE;   // 131072
// s_waitcnt immediate (gfx9 encoding): vmcnt = N, expcnt / lgkmcnt = "no wait"
#define MAEST_WAIT_VMCNT(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (((N) >> 4) << 14) | 0x0F70)


template <int OSZ>
struct Epi256 {
    static constexpr int PITCH = 256 * OSZ + 16;              // 528 / 1040
    static constexpr int ROWS = OSZ == 2 ? 128 : 64;          // m rows staged per pass
    static constexpr int MT = ROWS / 32;                      // wave m-tiles per pass: 4 / 2
    static constexpr int PASSES = 256 / ROWS;                 // 2 / 4
    static constexpr int CPR = 256 * OSZ / 16;                // chunks per row: 32 / 64
    static constexpr int EPC = 16 / OSZ;
};

template <int OSZ, int GMODE, bool EXACT>
__device__ __forceinline__ void stage256(char* smem, const f32x16_t (&acc)[2][4], const float* bias, int n0, int N,
                                         int mt0, int wn, int lane) {
    using E = Epi256<OSZ>;
    const int h = lane >> 5;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = wn * 64 + nt * 32 + 8 * g + 4 * h;
            float b4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (bias != nullptr && n0 + nl < N) {
                const float4 t = *reinterpret_cast<const float4*>(bias + n0 + nl);
                b4[0] = t.x; b4[1] = t.y; b4[2] = t.z; b4[3] = t.w;
            }
#pragma unroll
            for (int mi = 0; mi < E::MT; ++mi) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    f32x2_t xv = {acc[nt][mt0 + mi][4 * g + e] + b4[e], acc[nt][mt0 + mi][4 * g + e + 1] + b4[e + 1]};
                    if (GMODE != 0) {
                        f32x2_t gv, dv;
                        gelu_pair2<EXACT>(xv, gv, dv);
                        xv = GMODE == 1 ? gv : dv;
                    }
                    v[e] = xv[0];
                    v[e + 1] = xv[1];
                }
                char* dst = smem + (mi * 32 + (lane & 31)) * E::PITCH + nl * OSZ;
                if (OSZ == 4) {
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    chunk8 o;
                    o[0] = pack_bf2(v[0], v[1]);
                    o[1] = pack_bf2(v[2], v[3]);
                    *reinterpret_cast<chunk8*>(dst) = o;
                }
            }
        }
}

// GELU with the derivative side output: value and derivative come out of ONE gelu_pair per element and are
// staged together, 64 rows (2 wave m-tiles) at a time, into two LDS regions (`smem` and `smem + region`).
template <int OSZ, bool EXACT, int NMT>
__device__ __forceinline__ void stage256_pair(char* smem, int region, const f32x16_t (&acc)[2][4], const float* bias,
                                              int n0, int N, int mt0, int wn, int lane) {
    using E = Epi256<OSZ>;
    const int h = lane >> 5;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = wn * 64 + nt * 32 + 8 * g + 4 * h;
            float b4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (bias != nullptr && n0 + nl < N) {
                const float4 t = *reinterpret_cast<const float4*>(bias + n0 + nl);
                b4[0] = t.x; b4[1] = t.y; b4[2] = t.z; b4[3] = t.w;
            }
#pragma unroll
            for (int mi = 0; mi < NMT; ++mi) {
                float v[4], d[4];
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const f32x2_t xv = {acc[nt][mt0 + mi][4 * g + e] + b4[e], acc[nt][mt0 + mi][4 * g + e + 1] + b4[e + 1]};
                    f32x2_t gv, dv;
                    gelu_pair2<EXACT>(xv, gv, dv);
                    v[e] = gv[0]; v[e + 1] = gv[1];
                    d[e] = dv[0]; d[e + 1] = dv[1];
                }
                char* dst = smem + (mi * 32 + (lane & 31)) * E::PITCH + nl * OSZ;
                if (OSZ == 4) {
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(dst + region) = make_float4(d[0], d[1], d[2], d[3]);
                } else {
                    chunk8 o, q;
                    o[0] = pack_bf2(v[0], v[1]); o[1] = pack_bf2(v[2], v[3]);
                    q[0] = pack_bf2(d[0], d[1]); q[1] = pack_bf2(d[2], d[3]);
                    *reinterpret_cast<chunk8*>(dst) = o;
                    *reinterpret_cast<chunk8*>(dst + region) = q;
                }
            }
        }
}


// MODE 0 plain, 1 RESIDUAL (+ aux), 2 MUL (* aux), 3 ROWDOT: plain store, and rowdot[item, column group of 64, row in item]
// = sum of the stored values times aux over the group (the 8 / 16 lanes that hold a group's chunks are neighbours)
template <int OSZ, int MODE, int ROWS = Epi256<OSZ>::ROWS>
__device__ __forceinline__ void drain256(const char* smem, void* dst, int64_t ld, const AuxRegs<OSZ, ROWS>* aux,
                                         int mbase, int n0, int M, int N, int tid, float* rowdot = nullptr, int ntok = 1,
                                         int row0 = 0) {
    using E = Epi256<OSZ>;
#pragma unroll
    for (int i = 0; i < ROWS * E::CPR / 512; ++i) {
        const int c = tid + i * 512;
        const int row = c / E::CPR, cc = c - row * E::CPR;
        const int gm = mbase + row, gn = n0 + cc * E::EPC;
        chunk16 v = *reinterpret_cast<const chunk16*>(smem + row * E::PITCH + cc * 16);
        if (MODE == 1 || MODE == 2) v = apply_aux<OSZ, MODE>(v, aux->v[i]);
        if (MODE == 3) {
            const chunk16 r = aux->v[i];
            float d = 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (OSZ == 4) d += u2f(v[e]) * u2f(r[e]);
                else d += lo16f(v[e]) * lo16f(r[e]) + hi16f(v[e]) * hi16f(r[e]);
            }
            constexpr int GL = 64 * OSZ / 16;          // lanes per 64-column group: 8 / 16
#pragma unroll
            for (int m = 1; m < GL; m <<= 1) d += __shfl_xor(d, m, 64);
            if ((cc & (GL - 1)) == 0 && gm < M && gn < N) {
                const int item = (gm + row0) / ntok, q = gm + row0 - item * ntok;
                rowdot[((int64_t)item * (N >> 6) + (gn >> 6)) * ntok + q] = d;
            }
        }
        if (gm >= M || gn >= N) continue;
        // streaming output: written once, re-read by a later kernel after > L2-size of other traffic
        __builtin_nontemporal_store(v, reinterpret_cast<chunk16*>(reinterpret_cast<char*>(dst) + ((int64_t)gm * ld + gn) * OSZ));
    }
}

template <int OSZ, bool EXACT>
__device__ __forceinline__ void epilogue256(char* smem, const f32x16_t (&acc)[2][4], const Gemm256Params& p, int m0,
                                            int n0, int wm, int wn, int lane, int tid) {
    using E = Epi256<OSZ>;
    if (p.epi == MAEST_EPI_GELU && p.aux_out != nullptr) {
        constexpr int PR = OSZ == 2 ? 64 : 32;       // rows per pass (fp32 rows are twice as wide)
        constexpr int PMT = PR / 32;
        constexpr int REGION = PR * E::PITCH;        // 33792 (bf16) / 33280 (fp32); two regions per pass
#pragma unroll
        for (int ps = 0; ps < 256 / PR; ++ps) {
            const int pwm = ps / (4 / PMT);
            const int mt0 = (ps % (4 / PMT)) * PMT;
            const int mbase = m0 + pwm * 128 + mt0 * 32;
            if (wm == pwm) stage256_pair<OSZ, EXACT, PMT>(smem, REGION, acc, p.bias, n0, p.N, mt0, wn, lane);
            __syncthreads();
            drain256<OSZ, 0, PR>(smem, p.C, p.ldc, nullptr, mbase, n0, p.M, p.N, tid);
            drain256<OSZ, 0, PR>(smem + REGION, p.aux_out, p.ld_aux, nullptr, mbase, n0, p.M, p.N, tid);
            __syncthreads();
        }
        return;
    }
    const bool with_aux = p.epi == MAEST_EPI_RESIDUAL || p.epi == MAEST_EPI_MUL || p.epi == MAEST_EPI_ROWDOT;   // block-uniform
#pragma unroll
    for (int ps = 0; ps < E::PASSES; ++ps) {
        const int pwm = ps / (4 / E::MT);
        const int mt0 = (ps % (4 / E::MT)) * E::MT;
        const int mbase = m0 + pwm * 128 + mt0 * 32;
        const bool mine = (wm == pwm);   // wave-uniform
        if (p.epi == MAEST_EPI_GELU) {
            if (mine) stage256<OSZ, 1, EXACT>(smem, acc, p.bias, n0, p.N, mt0, wn, lane);
            __syncthreads();
            drain256<OSZ, 0, E::ROWS>(smem, p.C, p.ldc, nullptr, mbase, n0, p.M, p.N, tid);
        } else {
            AuxRegs<OSZ, E::ROWS> ax;
            if (with_aux) ax.load(p.aux_in, p.ld_aux, mbase, n0, p.M, tid);      // in flight across the staging
            if (mine) stage256<OSZ, 0, EXACT>(smem, acc, p.bias, n0, p.N, mt0, wn, lane);
            __syncthreads();
            if (p.epi == MAEST_EPI_RESIDUAL)
                drain256<OSZ, 1, E::ROWS>(smem, p.C, p.ldc, &ax, mbase, n0, p.M, p.N, tid);
            else if (p.epi == MAEST_EPI_MUL)
                drain256<OSZ, 2, E::ROWS>(smem, p.C, p.ldc, &ax, mbase, n0, p.M, p.N, tid);
            else if (p.epi == MAEST_EPI_ROWDOT)
                drain256<OSZ, 3, E::ROWS>(smem, p.C, p.ldc, &ax, mbase, n0, p.M, p.N, tid, p.rowdot, p.ntok, p.row0);
            else
                drain256<OSZ, 0, E::ROWS>(smem, p.C, p.ldc, nullptr, mbase, n0, p.M, p.N, tid);
        }
        __syncthreads();
    }
}

// (gemm_nt256_kernel -- the first 256 x 256 kernel, K in 64-byte slices through a four-deep ring: MAEST_GEMM_VARIANT = 1 -- was removed in round 6;
// gemm_nt256w_kernel below, whole-line stages, has served every shape since round 2.  Its epilogue forms above are gemm_nt256w_kernel's too.)

template <int OSZ, bool EXACT, int TNC, int NTH>
__device__ __forceinline__ void epilogueT(char* smem, const f32x16_t (&acc)[2][4], const Gemm256Params& p, int m0,
                                          int n0, int wm, int wn, int lane, int tid) {
    using E = EpiT<OSZ, TNC>;
    const bool gelu = p.epi == MAEST_EPI_GELU;
    const bool pair = gelu && p.aux_out != nullptr;
    // rows per pass so that the staging area (two regions for the pair form) fits the LDS ring it reuses
    auto run = [&](auto rp_tag, auto pair_tag) {
        constexpr int RP = decltype(rp_tag)::value;
        constexpr bool PAIR = decltype(pair_tag)::value;
        constexpr int REGION = RP * E::PITCH;
#pragma unroll
        for (int ps = 0; ps < 256 / RP; ++ps) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int r0 = wm * 128 + mt * 32;             // wave-uniform
                if (r0 >= ps * RP && r0 < (ps + 1) * RP) {
                    if (PAIR) stageT<OSZ, 3, EXACT, TNC>(smem, REGION, acc[0][mt], acc[1][mt], p.bias, n0, p.N, r0 - ps * RP, wn, lane);
                    else if (gelu) stageT<OSZ, 1, EXACT, TNC>(smem, 0, acc[0][mt], acc[1][mt], p.bias, n0, p.N, r0 - ps * RP, wn, lane);
                    else stageT<OSZ, 0, EXACT, TNC>(smem, 0, acc[0][mt], acc[1][mt], p.bias, n0, p.N, r0 - ps * RP, wn, lane);
                }
            }
            __syncthreads();
            const int mbase = m0 + ps * RP;
            if (PAIR) {
                drainT<OSZ, 0, TNC, NTH>(smem, RP, p.C, p.ldc, nullptr, 0, mbase, n0, p.M, p.N, tid);
                drainT<OSZ, 0, TNC, NTH>(smem + REGION, RP, p.aux_out, p.ld_aux, nullptr, 0, mbase, n0, p.M, p.N, tid);
            } else if (p.epi == MAEST_EPI_RESIDUAL) {
                drainT<OSZ, 1, TNC, NTH>(smem, RP, p.C, p.ldc, p.aux_in, p.ld_aux, mbase, n0, p.M, p.N, tid);
            } else if (p.epi == MAEST_EPI_MUL) {
                drainT<OSZ, 2, TNC, NTH>(smem, RP, p.C, p.ldc, p.aux_in, p.ld_aux, mbase, n0, p.M, p.N, tid);
            } else {
                drainT<OSZ, 0, TNC, NTH>(smem, RP, p.C, p.ldc, nullptr, 0, mbase, n0, p.M, p.N, tid);
            }
            if (ps + 1 < 256 / RP) __syncthreads();
        }
    };
    if (pair) run(std::integral_constant<int, (OSZ == 2 ? 128 : 64)>{}, std::true_type{});
    else run(std::integral_constant<int, (OSZ == 2 ? 256 : 128)>{}, std::false_type{});
}

constexpr int W2_ROWB = 128;
constexpr int W2_UNIT = 256 * W2_ROWB;      // 32768
constexpr int W2_NBUF = 5;
constexpr int W2_SMEM = W2_NBUF * W2_UNIT;  // 163840

// ---- pipelined C-tile epilogue of the full-line kernel (512 threads, 160 KiB of LDS).
// Measured: staging the whole tile in one pass is SLOWER than two 128-row passes (0.36 vs 0.33 ms for the qkv
// shape) -- with more passes the LDS staging of one overlaps the HBM stores of the previous one.  So: four passes,
// pass ps = m-tile ps of BOTH wave groups (rows wm*128 + ps*32 ..+32: every wave stages one m-tile per pass, none
// idles), two staging buffers, ONE barrier per pass: each wave first drains pass ps (its 16-byte stores go in
// flight), then converts / activates and stages pass ps+1 into the other buffer underneath them.
template <int OSZ, bool EXACT, bool PAIR, int MODE>
__device__ __forceinline__ void epilogueW_run(char* smem, const f32x16_t (&acc)[2][4], const Gemm256Params& p, int m0,
                                              int n0, int wm, int wn, int lane, int tid, bool gelu) {
    using E = EpiT<OSZ, 256>;
    constexpr int REGION = 64 * E::PITCH;                       // one 64-row staging region: 33792 / 66560
    constexpr int BUF = (PAIR ? 2 : 1) * REGION;
    constexpr bool DOUBLE = 2 * BUF <= W2_SMEM;                 // everything but the fp32 value + GELU' pair
    auto stage = [&](int ps, char* buf) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            if (mt == ps) {
                if (PAIR) stageT<OSZ, 3, EXACT, 256>(buf, REGION, acc[0][mt], acc[1][mt], p.bias, n0, p.N, wm * 32, wn, lane);
                else if (gelu) stageT<OSZ, 1, EXACT, 256>(buf, 0, acc[0][mt], acc[1][mt], p.bias, n0, p.N, wm * 32, wn, lane);
                else stageT<OSZ, 0, EXACT, 256>(buf, 0, acc[0][mt], acc[1][mt], p.bias, n0, p.N, wm * 32, wn, lane);
            }
    };
    // aux_in of the RESIDUAL / MUL forms: fetched one pass ahead into registers (see AuxRegs)
    AuxRegs<OSZ, 32> ax[2][2];                                  // [pass parity][32-row group]
    auto prefetch = [&](int ps) {
#pragma unroll
        for (int half = 0; half < 2; ++half)
            ax[ps & 1][half].load(p.aux_in, p.ld_aux, m0 + half * 128 + ps * 32, n0, p.M, tid);
    };
    auto drain = [&](int ps, const char* buf) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {                  // the two 32-row groups of the pass are 128 rows apart
            const int mbase = m0 + half * 128 + ps * 32;
            const char* src = buf + half * 32 * E::PITCH;
            drain256<OSZ, MODE, 32>(src, p.C, p.ldc, &ax[ps & 1][half], mbase, n0, p.M, p.N, tid);
            if (PAIR) drain256<OSZ, 0, 32>(src + REGION, p.aux_out, p.ld_aux, nullptr, mbase, n0, p.M, p.N, tid);
        }
    };
    if (MODE != 0) prefetch(0);
    stage(0, smem);
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        char* cur = smem + (DOUBLE ? (ps & 1) * BUF : 0);
        if (MODE != 0 && ps < 3) prefetch(ps + 1);
        drain(ps, cur);
        if (!DOUBLE && ps < 3) __syncthreads();                 // single buffer: everybody has drained before the refill
        if (ps < 3) stage(ps + 1, smem + (DOUBLE ? ((ps + 1) & 1) * BUF : 0));
        if (ps < 3) __syncthreads();
    }
}
template <int OSZ, bool EXACT>
__device__ __forceinline__ void epilogueW(char* smem, const f32x16_t (&acc)[2][4], const Gemm256Params& p, int m0,
                                          int n0, int wm, int wn, int lane, int tid) {
    const bool gelu = p.epi == MAEST_EPI_GELU;
    if (gelu && p.aux_out != nullptr) epilogueW_run<OSZ, EXACT, true, 0>(smem, acc, p, m0, n0, wm, wn, lane, tid, true);
    else if (p.epi == MAEST_EPI_RESIDUAL) epilogueW_run<OSZ, EXACT, false, 1>(smem, acc, p, m0, n0, wm, wn, lane, tid, false);
    else if (p.epi == MAEST_EPI_MUL) epilogueW_run<OSZ, EXACT, false, 2>(smem, acc, p, m0, n0, wm, wn, lane, tid, false);
    else epilogueW_run<OSZ, EXACT, false, 0>(smem, acc, p, m0, n0, wm, wn, lane, tid, gelu);
}

// ================================================================================================
// 256x256 tile, FULL-CACHE-LINE operand stages ("wide" kernel).
// Measured on MI355X (scratch/probe/dma_bw.hip, ablate.sh): the vector-memory front end retires roughly one
// cache-LINE request per ~3.7 clk per CU whatever part of the line is used, so the 64-byte row slices of the
// kernel above move 16-18 B/clk/CU and their DMA -- not the MFMA pipe (that loop runs at 2200 TFLOP/s with the
// DMA removed) -- caps it near 1250 TFLOP/s.  Whole 128-byte lines move 35 B/clk/CU.  Hence:
//   * a K STAGE is 128 bytes per row (64 bf16 / 32 fp32); one LDS-DMA instruction fetches 8 rows x 128 B.
//   * an operand UNIT is 256 rows x 128 B = 32 KiB; units alternate A_0 B_0 A_1 B_1 ... through a ring of
//     NBUF = 5 buffers (all 160 KiB of LDS): when stage j has been read, its two buffers take B_{j+2} and
//     A_{j+3}, so the DMA queue always holds 1.5 stages and only its last unit may still fly at a stage boundary
//     (counted vmcnt(4)).
//   * a stage is consumed in two k halves (LOAD 12 ds_read_b128 / COMPUTE 16 MFMAs each) with the same
//     one-barrier stagger between the two wave groups as above.
//   * bank swizzle for 128-byte rows: physical 16-byte chunk p of row r holds logical chunk p ^ ((r >> 1) & 7)
//     (applied on the DMA source address); a 16-lane ds_read_b128 group then covers all 16 slots of 256 B.
// ================================================================================================

// MAEST_ABLATE_* : timing experiments only (scratch/probe/ablate_w.sh builds the kernel with parts of the main loop
// removed; results are wrong on purpose).  Never defined in the product build.
// MAEST_NT_PRIO (timing experiment; the product build leaves it undefined = 0): wave priority in the main loop of
// gemm_nt256w_kernel.  0: raised for every COMPUTE phase (the shipped form); 1: never; 2: static -- the second-dispatched
// half (waves 4-7) at priority 1 for the whole loop, no per-phase flips; 3: COMPUTE phases at priority 3.
#ifndef MAEST_NT_PRIO
#define MAEST_NT_PRIO 0
#endif
#if MAEST_NT_PRIO == 0
#define MAEST_NT_PRIO_RAISE() __builtin_amdgcn_s_setprio(1)
#define MAEST_NT_PRIO_DROP() __builtin_amdgcn_s_setprio(0)
#elif MAEST_NT_PRIO == 3
#define MAEST_NT_PRIO_RAISE() __builtin_amdgcn_s_setprio(3)
#define MAEST_NT_PRIO_DROP() __builtin_amdgcn_s_setprio(0)
#else
#define MAEST_NT_PRIO_RAISE() ((void)0)
#define MAEST_NT_PRIO_DROP() ((void)0)
#endif
#ifdef MAEST_ABLATE_NO_BARRIER
#define MAEST_LOOP_BARRIER() ((void)0)
#else
#define MAEST_LOOP_BARRIER() __builtin_amdgcn_s_barrier()
#endif
// ---- epilogue of the 128-row variant (MTW = 2: each wave owns 64 x 64 outputs): both wave groups stage their m-tiles at
// once -- the whole 128 x 256 tile (two passes of 64 rows for the fp32 value + GELU' pair, which would not fit) -- one
// barrier, then the drain with aux_in prefetched into registers before the staging.
template <int OSZ, bool EXACT, int MODE, bool PAIR>
__device__ __forceinline__ void epilogueH_run(char* smem, const f32x16_t (&acc)[2][2], const Gemm256Params& p, int m0,
                                              int n0, int wm, int wn, int lane, int tid, bool gelu) {
    using E = EpiT<OSZ, 256>;
    constexpr int PMT = (PAIR && OSZ == 4) ? 1 : 2;             // m-tiles per wave and pass
    constexpr int GR = 32 * PMT;                                // rows per wave group and pass
    constexpr int REGION = 2 * GR * E::PITCH;                   // value region; the GELU' region follows
#pragma unroll
    for (int ps = 0; ps < 2 / PMT; ++ps) {
        AuxRegs<OSZ, GR> ax[2];
        if (MODE != 0) {
#pragma unroll
            for (int g = 0; g < 2; ++g) ax[g].load(p.aux_in, p.ld_aux, m0 + g * 64 + ps * GR, n0, p.M, tid);
        }
#pragma unroll
        for (int mi = 0; mi < PMT; ++mi) {
            const int mt = ps * PMT + mi;
            const int lrow = wm * GR + mi * 32;
            if (PAIR) stageT<OSZ, 3, EXACT, 256>(smem, REGION, acc[0][mt], acc[1][mt], p.bias, n0, p.N, lrow, wn, lane);
            else if (gelu) stageT<OSZ, 1, EXACT, 256>(smem, 0, acc[0][mt], acc[1][mt], p.bias, n0, p.N, lrow, wn, lane);
            else stageT<OSZ, 0, EXACT, 256>(smem, 0, acc[0][mt], acc[1][mt], p.bias, n0, p.N, lrow, wn, lane);
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int mbase = m0 + g * 64 + ps * GR;
            const char* src = smem + g * GR * E::PITCH;
            drain256<OSZ, MODE, GR>(src, p.C, p.ldc, &ax[g], mbase, n0, p.M, p.N, tid, p.rowdot, p.ntok, p.row0);
            if (PAIR) drain256<OSZ, 0, GR>(src + REGION, p.aux_out, p.ld_aux, nullptr, mbase, n0, p.M, p.N, tid);
        }
        if (ps + 1 < 2 / PMT) __syncthreads();
    }
}
template <int OSZ, bool EXACT>
__device__ __forceinline__ void epilogueH(char* smem, const f32x16_t (&acc)[2][2], const Gemm256Params& p, int m0,
                                          int n0, int wm, int wn, int lane, int tid) {
    const bool gelu = p.epi == MAEST_EPI_GELU;
    if (gelu && p.aux_out != nullptr) epilogueH_run<OSZ, EXACT, 0, true>(smem, acc, p, m0, n0, wm, wn, lane, tid, true);
    else if (p.epi == MAEST_EPI_RESIDUAL) epilogueH_run<OSZ, EXACT, 1, false>(smem, acc, p, m0, n0, wm, wn, lane, tid, false);
    else if (p.epi == MAEST_EPI_MUL) epilogueH_run<OSZ, EXACT, 2, false>(smem, acc, p, m0, n0, wm, wn, lane, tid, false);
    else if (p.epi == MAEST_EPI_ROWDOT) epilogueH_run<OSZ, EXACT, 3, false>(smem, acc, p, m0, n0, wm, wn, lane, tid, false);
    else epilogueH_run<OSZ, EXACT, 0, false>(smem, acc, p, m0, n0, wm, wn, lane, tid, gelu);
}

// MTW = 32-row m-tiles per wave: 4 -> the 256 x 256 tile described above; 2 -> a 128 x 256 tile (64 x 64 outputs per wave,
// A units of 128 rows in the same 32 KiB ring slots, EPIV ignored), used for the LAST PARTIAL ROUND of a launch: with
// 870 tiles on 256 CUs (the N = 768 shapes at M = 74240) the fourth round runs 102 workgroups on an otherwise idle chip;
// as 204 half tiles it fills the chip once and takes about half the time (gemm_nt256_try).
template <typename T, int EPIV, bool X3 = false, int MTW = 4>
__global__ __launch_bounds__(512) void gemm_nt256w_kernel(Gemm256Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;          // 0..7
    const int wm = wave >> 2, wn = wave & 3;
    const int h = lane >> 5;
    constexpr int NA = MTW;             // LDS-DMA instructions per wave and A unit (8 rows each): 4 / 2
    constexpr int BM = 64 * MTW;        // tile rows: 256 / 128

    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int tile_m = wg / p.tiles_n;
    const int tile_n = wg - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * 256;

    constexpr int ELT = (int)sizeof(T);
    constexpr int KS = W2_ROWB / ELT;        // 64 / 32 elements per stage
    const int nstages = p.K / KS;

    // LDS-DMA map: instruction i of this wave fills rows [(wave*NA+i)*8, +8) of an A unit, [(wave*4+i)*8, +8) of a B unit
    const char* a_src[NA];
    const char* b_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int csrc = (lane & 7) ^ ((r >> 1) & 7);        // source-side swizzle
        int rb = n0 + r;
        if (rb > p.N - 1) rb = p.N - 1;
        b_src[i] = p.B + (int64_t)rb * p.ldb * ELT + csrc * 16;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int r = (wave * NA + i) * 8 + (lane >> 3);
        const int csrc = (lane & 7) ^ ((r >> 1) & 7);
        int ra = m0 + r;
        if (ra > p.M - 1) ra = p.M - 1;
        a_src[i] = p.A + (int64_t)ra * p.lda * ELT + csrc * 16;
    }
    const int dma_off_a = wave * NA * 1024, dma_off_b = wave * 4 * 1024;

    f32x16_t acc[2][MTW];   // [nt][mt]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < MTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // one LDS-DMA instruction of unit (stage, A | B) into ring buffer `buf`; past-the-end stages re-load the last stage
    // into a dead buffer (keeps the vmcnt arithmetic uniform)
    auto issue_one = [&](int stage, bool is_b, int buf, int i) {
        const int sc = stage < nstages ? stage : nstages - 1;
        const char* src = (is_b ? b_src[i] : a_src[i < NA ? i : 0]) + (int64_t)sc * W2_ROWB;
        char* dst = smem + buf * W2_UNIT + (is_b ? dma_off_b : dma_off_a) + i * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    // unit u = 2*stage + (0: A, 1: B) lives in buffer u % NBUF
    auto issue_unit = [&](int stage, bool is_b, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (is_b || i < NA) issue_one(stage, is_b, buf, i);
    };

    int a_off[MTW], b_off[2], a_swz[MTW], b_swz[2];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int row = wm * (32 * MTW) + mt * 32 + (lane & 31);
        a_off[mt] = row * W2_ROWB;
        a_swz[mt] = (row >> 1) & 7;
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int row = wn * 64 + nt * 32 + (lane & 31);
        b_off[nt] = row * W2_ROWB;
        b_swz[nt] = (row >> 1) & 7;
    }

    chunk16 fa[2][MTW], fb[2][2];
#ifdef MAEST_ABLATE_NO_DSREAD
    for (int i = 0; i < 2; ++i) {
        for (int j = 0; j < MTW; ++j) fa[i][j] = chunk16{MAEST_ONE16X2 + (uint32_t)lane, MAEST_ONE16X2, MAEST_ONE16X2, MAEST_ONE16X2};
        for (int j = 0; j < 2; ++j) fb[i][j] = chunk16{MAEST_ONE16X2, MAEST_ONE16X2 + (uint32_t)lane, MAEST_ONE16X2, MAEST_ONE16X2};
    }
#endif
    auto load_frags = [&](int abuf, int bbuf, int kh) {
#ifdef MAEST_ABLATE_NO_DSREAD
        return;
#endif
        const char* la = smem + abuf * W2_UNIT;
        const char* lb = smem + bbuf * W2_UNIT;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int kc = 4 * kh + 2 * ks + h;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
                fa[ks][mt] = *reinterpret_cast<const chunk16*>(la + a_off[mt] + ((kc ^ a_swz[mt]) << 4));
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                fb[ks][nt] = *reinterpret_cast<const chunk16*>(lb + b_off[nt] + ((kc ^ b_swz[nt]) << 4));
        }
    };
    // COMPUTE phase with one operand unit's DMA (this wave's 4 -- or NA -- instructions) spread between the MFMAs: the
    // memory front end accepts about one 8-line instruction per 30 clk per CU, so the four waves of a group
    // feed it at exactly its rate, never in a burst, and a wave is never parked in the queue while it owes MFMAs.
#ifdef MAEST_ABLATE_ROLLING
    int roll_q = 0;
#endif
    auto compute = [&](bool dma, int stage, bool is_b, int buf) {
        const int ndma = is_b ? 4 : NA;
        MAEST_NT_PRIO_RAISE();
        if constexpr (X3) {
            // split-bf16: the two k chunks of a fragment pair feed ONE K = 16 MFMA triple (common.h: mma_chunk2)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    mma_chunk2<T, true>(acc[nt][mt], fb[0][nt], fb[1][nt], fa[0][mt], fa[1][mt]);
                    if (MTW == 4) {
                        if (dma && (mt & 1)) issue_one(stage, is_b, buf, nt * 2 + (mt >> 1));
                    } else {
                        const int i = nt * 2 + mt;
                        if (dma && i < ndma) issue_one(stage, is_b, buf, i);
                    }
                }
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
#ifndef MAEST_ABLATE_NO_MFMA
                    mma_chunk<T>(acc[nt][mt], fb[ks][nt], fa[ks][mt]);
#endif
                }
#ifndef MAEST_ABLATE_NO_DMA
                const int i = ks * 2 + nt;
                if (dma && i < ndma) issue_one(stage, is_b, buf, i);
#endif
            }
        }
#ifdef MAEST_ABLATE_ROLLING
        // timing experiment: what would the stores of the PREVIOUS tile's C cost if they rode in this tile's main loop?
        // One full-line 16-byte store per COMPUTE phase (16 per wave and tile = its 128 x 64 bf16 sub-tile), data from
        // a fragment register (wrong on purpose), optionally through a wave-private LDS bounce (ROLLING=2).
        if (roll_q < 16) {
            const int row = wm * 128 + roll_q * 8 + (lane >> 3);
            chunk16 v = fa[0][0];
#if MAEST_ABLATE_ROLLING >= 2
            char* bounce = smem + 4 * W2_UNIT + wave * 4096 + lane * 16;     // (buffer 4: wrong on purpose)
            *reinterpret_cast<chunk8*>(bounce) = chunk8{v[0], v[1]};
            *reinterpret_cast<chunk8*>(bounce + 8) = chunk8{v[2], v[3]};
            v = *reinterpret_cast<const chunk16*>(bounce + ((lane & 7) ^ 5) * 16 - (lane & 7) * 16);
#endif
            __builtin_nontemporal_store(v, reinterpret_cast<chunk16*>(reinterpret_cast<char*>(p.C) +
                                        ((int64_t)(m0 + row) * p.ldc + n0 + wn * 64) * 2 + (lane & 7) * 16));
        }
        ++roll_q;
#endif
        MAEST_NT_PRIO_DROP();
    };
    auto next = [](int b, int by) { b += by; return b >= W2_NBUF ? b - W2_NBUF : b; };

    // prologue: units 0..4 = A_0 B_0 A_1 B_1 A_2
    issue_unit(0, false, 0);
    issue_unit(0, true, 1);
    issue_unit(1, false, 2);
    issue_unit(1, true, 3);
    issue_unit(2, false, 4);
    MAEST_WAIT_VMCNT(2 * NA + 4);    // stage 0 landed (this wave's share): A_1 B_1 A_2 may still fly
    __builtin_amdgcn_s_barrier();
#if MAEST_NT_PRIO == 2
    if (wm == 1) __builtin_amdgcn_s_setprio(1);
#endif
    if (wm == 1) __builtin_amdgcn_s_barrier();          // stagger (wave-uniform)
    // Per stage j, group A (wm == 0) passes barriers  b1 b2 b3 b4  as
    //   LOAD(j,0) b1 COMPUTE(j,0) b2 LOAD(j,1) b3 COMPUTE(j,1) [vmcnt] b4
    // and group B, one barrier behind, as
    //   b1 LOAD(j,0) b2 COMPUTE(j,0) b3 LOAD(j,1) [vmcnt] b4 COMPUTE(j,1).
    // At b4 everybody has finished reading stage j's two buffers and has retired its own share of stage j+1.
    // Refills ride inside COMPUTE phases (see compute()):
    //   group A: COMPUTE(j,0) B_{j+1} -> A_{j-1}'s buffer,  COMPUTE(j,1) A_{j+2} -> B_{j-1}'s buffer
    //   group B: COMPUTE(j,0) A_{j+2} -> B_{j-1}'s buffer,  COMPUTE(j,1) B_{j+2} -> A_j's buffer (free since b4(j))
    // so that every load has at least three phases to land; only the A unit issued last (NA instructions) may still
    // fly at b4.
    int abuf = 0, bbuf = 1;          // buffers of A_j, B_j
    for (int j = 0; j < nstages; ++j) {
        const int abuf_prev = next(abuf, W2_NBUF - 2), bbuf_prev = next(bbuf, W2_NBUF - 2);
        load_frags(abuf, bbuf, 0);
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) only
        MAEST_LOOP_BARRIER();
        if (wm == 0) compute(j > 0, j + 1, true, abuf_prev);
        else compute(j > 0, j + 2, false, bbuf_prev);
        MAEST_LOOP_BARRIER();
        load_frags(abuf, bbuf, 1);
        if (wm == 0) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            MAEST_LOOP_BARRIER();           // b3
            compute(j > 0, j + 2, false, bbuf_prev);
#ifndef MAEST_ABLATE_NO_VMWAIT
            MAEST_WAIT_VMCNT(NA);
#endif
            MAEST_LOOP_BARRIER();           // b4
        } else {
#ifndef MAEST_ABLATE_NO_VMWAIT
            __builtin_amdgcn_s_waitcnt(0x0070 | NA);     // vmcnt(NA) lgkmcnt(0)
#else
            __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
            MAEST_LOOP_BARRIER();           // b4
            compute(true, j + 2, true, abuf);
            MAEST_LOOP_BARRIER();
        }
        abuf = next(abuf, 2);
        bbuf = next(bbuf, 2);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();          // un-stagger
    MAEST_WAIT_VMCNT(0);   // drain the past-the-end loads before LDS is reused
    __syncthreads();       // LDS becomes the C staging area
#if defined(MAEST_ABLATE_NO_EPILOGUE) || defined(MAEST_ABLATE_ROLLING)
    // timing experiments: no C-tile epilogue at all (results are not stored); the accumulators are kept alive
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < MTW; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" :: "v"(acc[i][j]));
#endif
        }
    return;
#endif
    if constexpr (MTW == 2) {
        if (p.out_dtype == MAEST_BF16) epilogueH<2, sizeof(T) == 4>(smem, acc, p, m0, n0, wm, wn, lane, tid);
        else epilogueH<4, sizeof(T) == 4>(smem, acc, p, m0, n0, wm, wn, lane, tid);
    } else if (EPIV == 0) { // two-pass epilogue of the 64-byte-slice kernel
        if (p.out_dtype == MAEST_BF16) epilogue256<2, sizeof(T) == 4>(smem, acc, p, m0, n0, wm, wn, lane, tid);
        else epilogue256<4, sizeof(T) == 4>(smem, acc, p, m0, n0, wm, wn, lane, tid);
    } else if (EPIV == 1) { // four passes, double buffered
        if (p.out_dtype == MAEST_BF16) epilogueW<2, sizeof(T) == 4>(smem, acc, p, m0, n0, wm, wn, lane, tid);
        else epilogueW<4, sizeof(T) == 4>(smem, acc, p, m0, n0, wm, wn, lane, tid);
    } else {                // whole tile in one pass
        if (p.out_dtype == MAEST_BF16) epilogueT<2, sizeof(T) == 4, 256, 512>(smem, acc, p, m0, n0, wm, wn, lane, tid);
        else epilogueT<4, sizeof(T) == 4, 256, 512>(smem, acc, p, m0, n0, wm, wn, lane, tid);
    }
}

template <typename T, int EPIV, bool X3 = false, int MTW = 4>
static int launch256w(Gemm256Params& p, hipStream_t stream) {
    static DeviceOnce once;
    ensure_dynamic_lds(once, &gemm_nt256w_kernel<T, EPIV, X3, MTW>, W2_SMEM);
    hipLaunchKernelGGL((gemm_nt256w_kernel<T, EPIV, X3, MTW>), dim3(p.tiles_m * p.tiles_n), dim3(512), W2_SMEM, stream, p);
    return check_launch("maest_gemm_nt(256w)");
}

// (gemm_nt256x128_kernel -- 256 x 128 tiles, two workgroups per CU: MAEST_GEMM_VARIANT = 2 and the N % 256 != 0 shapes -- was removed in round 6: no
// shape of the model has N % 256 != 0 at large M, and such a call is served by gemm.hip's 128 x 128 kernel.)

// Called by maest_gemm_nt for large, 16-byte-friendly problems.  Returns -1 when the shape does not qualify.
int gemm_nt256_try(const void* A, int64_t lda, const void* B, int64_t ldb, int in_dtype, void* C, int64_t ldc,
                   int out_dtype, int M, int N, int K, const float* bias, int epi, const void* aux_in, void* aux_out,
                   int64_t ld_aux, hipStream_t stream, float* rowdot, int ntok) {
    if (epi == MAEST_EPI_ATOMIC) return -1;
    // the row-dot side output lives in the epilogues of the full-line kernel only (two-pass form / 128-row tiles)
    if (epi == MAEST_EPI_ROWDOT && ((N % 256) != 0 || (K * (in_dtype == MAEST_BF16 ? 2 : 4)) % W2_ROWB != 0))
        return -1;
    const bool x3 = in_dtype == MAEST_F32X3;     // fp32 tensors, split-bf16 products (full-line kernel only)
    if (x3) in_dtype = MAEST_F32;
    // below ~8k rows the 256-row tiles leave most CUs idle (M = 560: 9-36 workgroups); the 128x128 kernel's finer
    // grid wins there (measured: one 10 s clip 2.13 -> 1.50 ms, batch 8 2.44 -> 2.23 ms, batch 16 equal).
    // MAEST_OPT_GEMM_MIN_M overrides the threshold (the emulator tests run the big kernels at M = 512).
    const int min_m = option(MAEST_OPT_GEMM_MIN_M);
    if (M < (min_m > 512 ? min_m : 512) || N < 256 || (N % 256) != 0) return -1;
    if ((K * (in_dtype == MAEST_BF16 ? 2 : 4)) % W2_ROWB != 0) return -1;
    const int variant = option(MAEST_OPT_GEMM_VARIANT);   // experiment switch (A/B timing, tests)
    if (out_dtype == MAEST_SPLIT3_A &&                    // (the split-row output exists in gemm_nt256o_kernel's epilogue only)
        (variant == 3 || !gemm_nt256o_available())) return -1;
    Gemm256Params p;
    p.A = (const char*)A; p.B = (const char*)B; p.C = C;
    p.bias = bias; p.aux_in = aux_in; p.aux_out = aux_out;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ld_aux = ld_aux;
    p.M = M; p.N = N; p.K = K; p.out_dtype = out_dtype; p.epi = epi;
    p.rowdot = rowdot; p.ntok = ntok > 0 ? ntok : 1; p.row0 = 0; p.panel_w = 0;
    p.tiles_m = (M + 255) / 256;
    {
        p.tiles_n = N / 256;
        // epilogue form, measured side by side in one run (scratch/gemm_ab.py): the two-pass form wins for bf16
        // outputs by 1-5 %, the four-pass double-buffered form for the fp32 residual outputs by 3-5 %, the
        // one-pass form never.  MAEST_OPT_GEMM_EPILOGUE = 0 / 1 / 2 forces one of them (-1 = this default).
        const int eopt = option(MAEST_OPT_GEMM_EPILOGUE);
        const int ev = epi == MAEST_EPI_ROWDOT ? 0 : (eopt >= 0 ? eopt : (epi == MAEST_EPI_RESIDUAL ? 1 : 0));
        auto full = [&](Gemm256Params& q) {
            // bf16 operands: the one-wave-per-SIMD kernel (gemm_nt_ow.hip); MAEST_OPT_GEMM_VARIANT = 3 keeps the 8-wave kernel (A/B, tests)
            // (every 16-bit-output form but RESIDUAL; fp32 outputs in the plain and RESIDUAL forms; the other fp32-output forms take the kernel below)
            if (!x3 && in_dtype == MAEST_BF16 && variant != 3 && gemm_nt256o_available() &&
                ((out_dtype == MAEST_BF16 && epi != MAEST_EPI_RESIDUAL) || (out_dtype == MAEST_SPLIT3_A && epi == MAEST_EPI_GELU && aux_out == nullptr) ||
                 (out_dtype == MAEST_F32 && (epi == MAEST_EPI_NONE || epi == MAEST_EPI_RESIDUAL))))
                return gemm_nt256o_launch(q, stream);
            if (x3) return ev == 1 ? launch256w<float, 1, true>(q, stream) : launch256w<float, 0, true>(q, stream);
            if (ev == 1) return in_dtype == MAEST_BF16 ? launch256w<bf16_t, 1>(q, stream) : launch256w<float, 1>(q, stream);
            if (ev == 2) return in_dtype == MAEST_BF16 ? launch256w<bf16_t, 2>(q, stream) : launch256w<float, 2>(q, stream);
            return in_dtype == MAEST_BF16 ? launch256w<bf16_t, 0>(q, stream) : launch256w<float, 0>(q, stream);
        };
        auto half = [&](Gemm256Params& q) {        // 128 x 256 tiles (gemm_nt256w_kernel, MTW = 2)
            q.tiles_m = (q.M + 127) / 128;
            if (x3) return launch256w<float, 0, true, 2>(q, stream);
            return in_dtype == MAEST_BF16 ? launch256w<bf16_t, 0, false, 2>(q, stream) : launch256w<float, 0, false, 2>(q, stream);
        };
        // The last partial round.  One workgroup per CU and equal tiles: a launch runs in ceil(tiles / 256) rounds and the
        // last one may be nearly empty (N = 768 at M = 74240: 870 tiles = 3.4 rounds -- measured 1093 TFLOP/s against
        // 1226 at M = 65280, exactly 3 rounds).  When at most half a round is left over, the rows of the full rounds
        // go to one launch and the remaining rows to a second one in 128-row tiles (twice the workgroups, about half the
        // time each): 3 + ~0.6 instead of 4 rounds.  MAEST_OPT_GEMM_TAIL: 0 off, 1 (default) automatic, 2 every tile a
        // 128-row tile (tests).
        const int tail = out_dtype == MAEST_SPLIT3_A ? 0 : option(MAEST_OPT_GEMM_TAIL);      // (the 128-row tiles have no split-row epilogue)
        if (tail == 2) return half(p);
        const int ncu = 256;
        const int tiles = p.tiles_m * p.tiles_n;
        const int rounds = tiles / ncu;
        if (tail == 1 && rounds >= 1 && tiles % ncu != 0) {
            const int mf = rounds * ncu / p.tiles_n;            // m-tiles of the full rounds
            const int rows_left = M - mf * 256;
            const int half_wgs = ((rows_left + 127) / 128) * p.tiles_n;
            if (mf > 0 && rows_left > 0 && half_wgs <= ncu) {
                const int64_t esz = in_dtype == MAEST_BF16 ? 2 : 4, osz = out_dtype == MAEST_BF16 ? 2 : 4;
                Gemm256Params a = p, b = p;
                a.M = mf * 256;
                a.tiles_m = mf;
                const int rc = full(a);
                if (rc != MAEST_OK) return rc;
                const int64_t r0 = (int64_t)mf * 256;
                b.M = rows_left;
                b.A = p.A + r0 * lda * esz;
                b.C = (char*)C + r0 * ldc * osz;
                // aux_in: fp32 for RESIDUAL, the output dtype for MUL; aux_out (GELU'): the output dtype
                if (aux_in) b.aux_in = (const char*)aux_in + r0 * ld_aux * (epi == MAEST_EPI_RESIDUAL ? 4 : osz);
                // (row-dot: the second launch sees rows r0.. as its rows 0..; r0 is a multiple of 256, not of ntok in
                // general, so it gets the item / row arithmetic through an offset of its own)
                b.row0 = (int)r0;
                if (aux_out) b.aux_out = (char*)aux_out + r0 * ld_aux * osz;
                return half(b);
            }
        }
        return full(p);
    }
}


// ================================================================================================
// 256x256-tile TN GEMM (wgrad + bias grad):  C[i][j] += sum_k A[k][i] * B[k][j],  colsum[i] += sum_k A[k][i]
// Same ring / stagger structure as gemm_nt256_kernel; operands stay token-major.  A slice is 32 (bf16) /
// 16 (fp32) token rows of 256 columns = 16 KiB per operand; one LDS-DMA instruction moves 1 KiB = 2 (bf16)
// or 1 (fp32) whole rows.  Fragments come from ds_read_b64_tr_b16 (bf16) / ds_read_b32 (fp32).  The 4 token
// rows of a transpose read sit 512 B apart (same banks), so the SOURCE-side swizzle XORs (row & 3) into
// bits 2-3 of the 16-byte chunk index: the four 64-byte row segments then land on the four bank quarters.
// Requires M % 256 == 0, N % 256 == 0 and K % slice == 0 (else the caller uses gemm_tn_kernel).
// ================================================================================================
template <typename T>
struct Tn256 {
    static constexpr int ELT = (int)sizeof(T);
    static constexpr int RB = 256 * ELT;             // bytes per tile row: 512 / 1024
    static constexpr int KS = G2_TILE / RB;          // token rows per slice: 32 / 16
    static constexpr int RPI = 1024 / RB;            // rows per DMA instruction: 2 / 1
    static constexpr int KSTEP = ELT == 2 ? 16 : 8;  // k per chunk step
};

typedef short v4i16b_t __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ chunk16 frag_tn256(const char* tile, int ks, int iblk, int lane);
template <>
__device__ __forceinline__ chunk16 frag_tn256<bf16_t>(const char* tile, int ks, int iblk, int lane) {
    const int h = lane >> 5, g16 = (lane >> 4) & 1, q = lane & 15;
    const int row = ks * 16 + 8 * h + (q >> 2);                 // row & 3 == (q >> 2) & 3 for both reads
    const int cb = (iblk + 16 * g16 + 4 * (q & 3)) * 2;         // logical byte column
    const int pb = ((((cb >> 4) ^ ((row & 3) << 2))) << 4) | (cb & 15);
    const char* p = tile + row * 512 + pb;
#ifdef __AMDGCN__
    // Issued as inline asm ON PURPOSE.  Through the builtin, the compiler's waitcnt pass treats a transpose read
    // as possibly aliasing the in-flight LDS-DMA writes and puts `s_waitcnt vmcnt(0)` in front of the first one
    // of every slice -- draining the whole prefetch ring each iteration (the pure-MFMA loop then ran at half
    // rate).  The ring's own counted vmcnt + barrier already order DMA writes before these reads; the caller waits
    // lgkmcnt(0) and pins the results (tn_pin) before the MFMAs consume them.
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
    chunk8 l2, h2;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(l2) : "v"(addr));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(h2) : "v"(addr));
#else
    const v4i16b_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16b_t*)(p));
    const v4i16b_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16b_t*)(p + 4 * 512));
    const chunk8 l2 = __builtin_bit_cast(chunk8, lo), h2 = __builtin_bit_cast(chunk8, hi);   // no repacking
#endif
    chunk16 c;
    c[0] = l2[0]; c[1] = l2[1]; c[2] = h2[0]; c[3] = h2[1];
    return c;
}
// keeps the consumers of asm-issued fragment reads behind the s_waitcnt / s_barrier that precede this call
__device__ __forceinline__ void tn_pin(chunk16 (&fa)[2][4], chunk16 (&fb)[2][2]) {
#ifdef __AMDGCN__
    asm volatile("" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fb[0][0]), "+v"(fb[0][1]));
    asm volatile("" : "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]), "+v"(fb[1][0]), "+v"(fb[1][1]));
#endif
}
template <>
__device__ __forceinline__ chunk16 frag_tn256<float>(const char* tile, int ks, int iblk, int lane) {
    const int h = lane >> 5;
    const int cb = (iblk + (lane & 31)) * 4;
    chunk16 c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = ks * 8 + 4 * h + q;
        const int pb = ((((cb >> 4) ^ ((row & 3) << 2))) << 4) | (cb & 15);
        c[q] = *reinterpret_cast<const uint32_t*>(tile + row * 1024 + pb);
    }
    return c;
}


template <typename T, bool X3 = false>
__global__ __launch_bounds__(512) void gemm_tn256_kernel(GemmTn256Params p) {
    using C = Tn256<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;

    // (split, tile) pairs are numbered split-major and each XCD takes a CONTIGUOUS range of them: the ~32
    // workgroups resident on an XCD then walk at most two K ranges, so every token row is pulled into at most
    // two private L2s instead of all eight (measured before: 4x the algorithmic HBM/MALL fetch).
    const int ntiles = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, ntiles * p.split_k);
    const int split = wg / ntiles;
    const int tile = wg - split * ntiles;
    const int tile_i = tile / p.tiles_n;
    const int tile_j = tile - tile_i * p.tiles_n;
    const int i0 = tile_i * 256, j0 = tile_j * 256;

    const int total_slices = p.K / C::KS;
    const int s_begin = split * p.k_slices_per_split;
    int s_end = s_begin + p.k_slices_per_split;
    if (s_end > total_slices) s_end = total_slices;
    const int nslices = s_end - s_begin;

    // LDS-DMA map: wave-instruction (wave, i) fills rows [(wave*2+i)*RPI, +RPI) of a slice tile
    const char* a_src[2];
    const char* b_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * C::RPI + (C::RPI == 2 ? (lane >> 5) : 0);
        const int pc = C::RPI == 2 ? (lane & 31) : lane;       // physical 16-byte chunk within the row
        const int lc = pc ^ ((row & 3) << 2);                   // logical chunk fetched from global
        a_src[i] = p.A + ((int64_t)(s_begin * C::KS + row) * p.lda + i0) * C::ELT + lc * 16;
        b_src[i] = p.B + ((int64_t)(s_begin * C::KS + row) * p.ldb + j0) * C::ELT + lc * 16;
    }
    const int dma_off = wave * 2 * 1024;
    const int64_t a_step = (int64_t)C::KS * p.lda * C::ELT, b_step = (int64_t)C::KS * p.ldb * C::ELT;

    f32x16_t acc[4][2];   // [a: i-block][b: j-block]
    float cs = 0.0f;   // bias-gradient partial: column lane&31 of i-block wn, this lane's k half
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    // bias gradient, spread evenly: the j-tiles of one i-row take the slices round-robin (slice s belongs to tile_j
    // == s % tiles_n) and, inside a workgroup, wave wn sums i-block a == wn -- every wave of every workgroup does
    // 1 / (4 * tiles_n) of the work (one wave per j == 0 tile doing all of it cost 16 % of the kernel)
    const bool do_colsum = p.colsum != nullptr;   // uniform
    int cs_turn = tile_j;                         // counts down to this tile's slice

    // piece q (0..3) of the 4 LDS-DMA instructions this wave owes to slice s: A0 B0 A1 B1
    auto issue_piece = [&](int s, int q) {
        const int sc = s < nslices ? s : nslices - 1;
        char* la = smem + (s & (G2_STAGES - 1)) * 2 * G2_TILE + dma_off;
        const int i = q >> 1;
        if ((q & 1) == 0)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + sc * a_step),
                                             (__attribute__((address_space(3))) void*)(la + i * 1024), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + sc * b_step),
                                             (__attribute__((address_space(3))) void*)(la + G2_TILE + i * 1024), 16, 0, 0);
    };
    auto issue = [&](int s) {
#pragma unroll
        for (int q = 0; q < 4; ++q) issue_piece(s, q);
    };

    if (nslices > 0) {
        issue(0);
        issue(1);
        issue(2);
    }
    MAEST_WAIT_VMCNT(8);
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();          // stagger the second wave group by one phase
    chunk16 fa[2][4], fb[2][2];
    for (int s = 0; s < nslices; ++s) {
        const char* la = smem + (s & (G2_STAGES - 1)) * 2 * G2_TILE;
        const char* lb = la + G2_TILE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#ifdef MAEST_ABLATE_NO_DSREAD
            if (s > 0) continue;
#endif
#pragma unroll
            for (int a = 0; a < 4; ++a) fa[ks][a] = frag_tn256<T>(la, ks, wm * 128 + a * 32, lane);
#pragma unroll
            for (int b = 0; b < 2; ++b) fb[ks][b] = frag_tn256<T>(lb, ks, wn * 64 + b * 32, lane);
        }
#ifndef MAEST_ABLATE_NO_DMA
        issue(s + 3);   // (in the LOAD phase: spread between the MFMAs of COMPUTE it measured 9 % slower here)
#endif
        __builtin_amdgcn_s_waitcnt(0x0078);   // vmcnt(8) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        tn_pin(fa, fb);
        __builtin_amdgcn_s_setprio(1);
#ifndef MAEST_ABLATE_NO_MFMA
        if constexpr (X3) {     // split-bf16: the two k chunks of a slice feed one K = 16 MFMA triple
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) mma_chunk2<T, true>(acc[a][b], fa[0][a], fa[1][a], fb[0][b], fb[1][b]);
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) mma_chunk<T>(acc[a][b], fa[ks][a], fb[ks][b]);   // D rows = i, cols = j
        }
        }
#else
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a][0][ks] += u2f(fa[ks][a][0] ^ fb[ks][a & 1][1]);
#endif
        __builtin_amdgcn_s_setprio(0);
        if (do_colsum) {   // the fragments already hold A[k][i] for (i = lane&31, 8 or 4 k's): sum them on the VALU
            if (cs_turn == 0) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    if (a == wn) {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const uint32_t w = fa[ks][a][e];
                                if (C::ELT == 2) cs += lo16f(w) + hi16f(w);
                                else cs += u2f(w);
                            }
                    }
                cs_turn = p.tiles_n;
            }
            --cs_turn;
        }
        __builtin_amdgcn_s_barrier();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
    MAEST_WAIT_VMCNT(0);

#ifdef MAEST_ABLATE_NO_EPI
    if (acc[0][0][0] != 123.456f) return;
#endif
    if (p.ws != nullptr) {
        // Workspace form of the split-K combine (end of round 3).  The atomic form below issues 1024 global_atomic_add_f32 per
        // workgroup, 256 B each, and an fp32 atomic instruction takes ~50 ns of a CU's store path (scratch/probe/store_issue.hip):
        // the 40-50 us of a 0.3 ms launch.  Here the accumulators leave as they sit in the registers, four at a time: 256 dwordx4
        // stores of 1 KiB (~38 ns each when streaming), lane-linear -- the layout is private to this kernel and
        // tn256_reduce_kernel, which sums the partials of a tile in split order (deterministic) and adds them to C.
        // Measured (profiles/r03_ab_tn_workspace_combine.txt): GEMM + reduce against the atomic form alone -21 % (proj, 28 splits) ...
        // +-0 (fc1 / fc2, 7 splits); the training step equal within the pairs' spread: opt-in, for reproducible dW.
        float* base = p.ws + ((int64_t)(split * ntiles + tile) * 8 + wave) * 8192 + lane * 4;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4_t v = {acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t*>(base + ((a * 2 + b) * 4 + q) * 256));
                }
    } else {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = j0 + wn * 64 + b * 32 + (lane & 31);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 128 + a * 32 + frag_row(r, lane);
                unsafeAtomicAdd(p.C + (int64_t)row * p.ldc + col, acc[a][b][r]);
            }
    }
    }
    if (do_colsum) {
        const float tot = cs + __shfl_xor(cs, 32, 64);     // merge the two k halves
        if (lane < 32) unsafeAtomicAdd(p.colsum + i0 + wm * 128 + wn * 32 + lane, tot);
    }
}

// C[tile] += sum over the splits, in split order, of the partial tiles gemm_tn256_kernel left in the workspace.  One wave per
// (tile, GEMM wave, 32 x 32 block): it re-reads what that wave's lanes stored for the block (16 bytes per lane and chunk: 1 KiB per
// instruction, four splits in flight) and adds the sums to C as the accumulator layout has them -- for a fixed register the lanes of
// a half-wave hold 32 consecutive columns of one row, i.e. lane-linear 128-byte runs, which cost their bytes
// (scratch/probe/store_issue.hip).  No LDS, 40 registers: it fits beside any other kernel's workgroups.
__global__ __launch_bounds__(64) void tn256_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int64_t ldc, int ntiles,
                                                          int tiles_n, int split_k) {
    const int lane = threadIdx.x;
    const int ab = blockIdx.x & 7, wave = (blockIdx.x >> 3) & 7, tile = blockIdx.x >> 6;
    const int a = ab >> 1, b = ab & 1;
    const int wm = wave >> 2, wn = wave & 3;
    const int tile_i = tile / tiles_n, tile_j = tile - tile_i * tiles_n;
    f32x4_t s[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    const float* src = ws + ((int64_t)tile * 8 + wave) * 8192 + (ab * 4) * 256 + lane * 4;
    const int64_t step = (int64_t)ntiles * 8 * 8192;
    int sp = 0;
    for (; sp + 4 <= split_k; sp += 4) {          // four splits' loads in flight, added in split order
        f32x4_t v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                v[u][q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(src + u * step + q * 256));
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] += v[u][q];
        src += 4 * step;
    }
    for (; sp < split_k; ++sp) {
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] += __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(src + q * 256));
        src += step;
    }
    float* cp = C + ((int64_t)tile_i * 256 + wm * 128 + a * 32) * ldc + tile_j * 256 + wn * 64 + b * 32 + (lane & 31);
    float old[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) old[r] = cp[(int64_t)frag_row(r, lane) * ldc];
#pragma unroll
    for (int r = 0; r < 16; ++r) cp[(int64_t)frag_row(r, lane) * ldc] = old[r] + s[r >> 2][r & 3];
}

// split_k, K slices per split and workgroup count of the 256-tile TN kernel for a shape; false when it does not take the shape
static bool tn256_plan(int dtype, int M, int N, int K, int split_k, int* splits, int* per_split) {
    if (dtype == MAEST_F32X3) dtype = MAEST_F32;
    const int ks = dtype == MAEST_BF16 ? Tn256<bf16_t>::KS : Tn256<float>::KS;
    if ((M % 256) != 0 || (N % 256) != 0 || (K % ks) != 0 || K < 8 * ks) return false;
    // variant 4: take any qualifying shape (emulator tests)
    if ((int64_t)M * N < (int64_t)8 * 65536 && option(MAEST_OPT_GEMM_VARIANT) != 4)
        return false;                                      // few output tiles: the 128x128 kernel splits K finer
    const int total = K / ks;
    const int tiles = (M / 256) * (N / 256);
    // one workgroup per CU and ONE round: never more workgroups than the 256 CUs (rounding up gave 288 for the
    // 36-tile fc1/fc2 wgrads, i.e. a second round for 32 stragglers: 0.49 ms instead of 0.33 ms)
    if (split_k <= 0) split_k = tiles >= 256 ? 1 : 256 / tiles;
    if (split_k > total / 4) split_k = total / 4 > 0 ? total / 4 : 1;
    *per_split = (total + split_k - 1) / split_k;
    *splits = (total + *per_split - 1) / *per_split;
    return true;
}

// Bytes of workspace with which maest_gemm_tn_ws combines the split-K partials without atomics; 0: the shape takes a path
// that has no workspace form (or a single split)
int64_t gemm_tn256_workspace_bytes(int dtype, int M, int N, int K, int split_k) {
    int splits = 1, per = 0;
    if (option(MAEST_OPT_TN_REDUCE) == 0 || !tn256_plan(dtype, M, N, K, split_k, &splits, &per) || splits < 2) return 0;
    return (int64_t)splits * (M / 256) * (N / 256) * 65536 * (int64_t)sizeof(float);
}

template <typename T, bool X3 = false>
static int launch_tn256(GemmTn256Params& p, int split_k, hipStream_t stream) {
    static DeviceOnce once;
    ensure_dynamic_lds(once, &gemm_tn256_kernel<T, X3>, G2_SMEM);
    p.split_k = split_k;
    const int ntiles = p.tiles_m * p.tiles_n;
    hipLaunchKernelGGL((gemm_tn256_kernel<T, X3>), dim3(ntiles * split_k), dim3(512), G2_SMEM, stream, p);
    const int rc = check_launch("maest_gemm_tn(256)");
    if (rc != MAEST_OK || p.ws == nullptr) return rc;
    hipLaunchKernelGGL(tn256_reduce_kernel, dim3(ntiles * 64), dim3(64), 0, stream, (const float*)p.ws, p.C, p.ldc, ntiles, p.tiles_n,
                       split_k);
    return check_launch("maest_gemm_tn(256, reduce)");
}

// Called by maest_gemm_tn; returns -1 when the shape does not qualify.  split_k <= 0 = automatic.  ws / ws_bytes: optional
// workspace (gemm_tn256_workspace_bytes); too small or NULL = atomics.
int gemm_tn256_try(const void* A, int64_t lda, const void* B, int64_t ldb, int dtype, float* C, int64_t ldc, int M,
                   int N, int K, float* colsum, int split_k, hipStream_t stream, void* ws, int64_t ws_bytes) {
    const bool x3 = dtype == MAEST_F32X3;        // fp32 tensors, split-bf16 products
    int per = 0;
    if (!tn256_plan(dtype, M, N, K, split_k, &split_k, &per)) return -1;
    if (x3) dtype = MAEST_F32;
    GemmTn256Params p;
    p.A = (const char*)A; p.B = (const char*)B; p.C = C; p.colsum = colsum;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.tiles_m = M / 256;
    p.tiles_n = N / 256;
    p.k_slices_per_split = per;
    const int64_t need = (int64_t)split_k * p.tiles_m * p.tiles_n * 65536 * (int64_t)sizeof(float);
    const bool use_ws = ws != nullptr && split_k >= 2 && ws_bytes >= need && option(MAEST_OPT_TN_REDUCE) != 0 &&
                        ((uintptr_t)ws % 16) == 0 && ((uintptr_t)C % 16) == 0 && (ldc % 4) == 0;
    p.ws = use_ws ? (float*)ws : nullptr;
    // bf16 operands, atomic combine: the one-wave-per-SIMD kernel (gemm_tn_ow.hip; 32-bit slice offsets); MAEST_OPT_GEMM_VARIANT = 3
    // keeps the 8-wave kernel (A/B, tests)
    if (!x3 && dtype == MAEST_BF16 && !use_ws && option(MAEST_OPT_GEMM_VARIANT) != 3 && gemm_tn256o_available() &&
        (int64_t)K * lda * 2 < ((int64_t)1 << 31) && (int64_t)K * ldb * 2 < ((int64_t)1 << 31))
        return gemm_tn256o_launch(p, split_k, stream);
    if (x3) return launch_tn256<float, true>(p, split_k, stream);
    return dtype == MAEST_BF16 ? launch_tn256<bf16_t>(p, split_k, stream) : launch_tn256<float>(p, split_k, stream);
}

}  // namespace maest
