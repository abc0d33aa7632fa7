// Pieces of the 256-column C-tile epilogues shared by the NT GEMM kernels of gemm256.hip and gemm_nt_ow.hip: the launch
// parameter block, the LDS staging of one 32-row m-tile (bias / GELU / GELU' fused), the 16-byte coalesced drain and the
// register prefetch of the second epilogue operand.
#pragma once
#include "common.h"

namespace maest {

struct Gemm256Params {
    const char* A;
    const char* B;
    void* C;
    const float* bias;
    const void* aux_in;
    void* aux_out;
    int64_t lda, ldb, ldc, ld_aux;
    int M, N, K;
    int out_dtype, epi;
    int tiles_m, tiles_n;
    float* rowdot;      // MAEST_EPI_ROWDOT: fp32 [rows / ntok, N / 64, ntok]
    int ntok;
    int row0;           // row of the whole problem this launch's row 0 is (second launch of a split problem)
    int panel_w;        // gemm_nt256o_kernel: > 0 = tiles are walked in column panels of this many tiles (gemm_nt_ow.hip: tile_of)
};

// The second operand of the RESIDUAL / MUL epilogues (aux_in, 16 bytes per output chunk) for one staging pass, fetched
// into registers BEFORE the pass is staged: the loads fly under the LDS staging and the barrier instead of sitting, four
// at a time, between the LDS read and the store of the drain loop (each batch a full L2-miss latency: the dgrad through
// GELU' -- 128 KiB of aux per tile, nowhere in cache -- spent a third of its tile time there).  Rows beyond M are clamped
// (loaded, never stored): an unconditional load keeps the compiler from waiting for it at a join.
template <int OSZ, int ROWS, int NTH = 512>
struct AuxRegs {
    static constexpr int CPR = 256 * OSZ / 16;
    static constexpr int NCH = ROWS * CPR / NTH;
    chunk16 v[NCH];
    __device__ __forceinline__ void load(const void* aux, int64_t ld_aux, int mbase, int n0, int M, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * NTH;
            const int row = c / CPR, cc = c - row * CPR;
            int gm = mbase + row;
            gm = gm < M ? gm : M - 1;
            v[i] = *reinterpret_cast<const chunk16*>(reinterpret_cast<const char*>(aux) +
                                                     ((int64_t)gm * ld_aux + n0 + cc * (16 / OSZ)) * OSZ);
        }
    }
};

template <int OSZ, int MODE>
__device__ __forceinline__ chunk16 apply_aux(chunk16 v, const chunk16& r) {
    if (MODE == 1 || (MODE == 2 && OSZ == 4)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = MODE == 1 ? f2u(u2f(v[e]) + u2f(r[e])) : f2u(u2f(v[e]) * u2f(r[e]));
    } else if (MODE == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t vw = v[e], rw = r[e];
            const float lo = lo16f(vw) * lo16f(rw);
            const float hi = hi16f(vw) * hi16f(rw);
            v[e] = pack_bf2(lo, hi);
        }
    }
    return v;
}

// ---- C-tile epilogue shared by the full-line 256x256 kernel (TNC = 256 columns, 512 threads, 160 KiB of LDS) and the
// 256x128 kernel (TNC = 128, 256 threads, 72 KiB): the whole tile -- or the largest row group that fits, with two
// regions for the GELU + GELU' pair -- is staged in ONE pass by ALL waves at once (the older epilogue256 staged one
// wave group at a time in 128 KiB), then drained with 16-byte non-temporal stores.
template <int OSZ, int TNC>
struct EpiT {
    static constexpr int PITCH = TNC * OSZ + 16;      // 272 / 528 (TNC 128), 528 / 1040 (TNC 256)
    static constexpr int CPR = TNC * OSZ / 16;        // 16-byte chunks per row
    static constexpr int EPC = 16 / OSZ;
};

// one 32-row m-tile of this wave's block -> LDS rows [lrow0, lrow0 + 32); GMODE 0 none, 1 GELU value, 3 value + GELU'
template <int OSZ, int GMODE, bool EXACT, int TNC>
__device__ __forceinline__ void stageT(char* smem, int region, const f32x16_t& a0, const f32x16_t& a1,
                                       const float* bias, int n0, int N, int lrow0, int wn, int lane) {
    using E = EpiT<OSZ, TNC>;
    const int h = lane >> 5;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = wn * 64 + nt * 32 + 8 * g + 4 * h;
            f32x4_t b4 = {0.0f, 0.0f, 0.0f, 0.0f};
            if (bias != nullptr && n0 + nl < N) b4 = *reinterpret_cast<const f32x4_t*>(bias + n0 + nl);
            float v[4], d[4];
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const f32x2_t xv = {(nt == 0 ? a0[4 * g + e] : a1[4 * g + e]) + b4[e],
                                    (nt == 0 ? a0[4 * g + e + 1] : a1[4 * g + e + 1]) + b4[e + 1]};
                f32x2_t gv = xv, dv = {0.0f, 0.0f};
                if (GMODE != 0) gelu_pair2<EXACT>(xv, gv, dv);
                v[e] = gv[0]; v[e + 1] = gv[1];
                d[e] = dv[0]; d[e + 1] = dv[1];
            }
            char* dst = smem + (lrow0 + (lane & 31)) * E::PITCH + nl * OSZ;
            if (OSZ == 4) {
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                if (GMODE == 3) *reinterpret_cast<float4*>(dst + region) = make_float4(d[0], d[1], d[2], d[3]);
            } else {
                chunk8 o;
                o[0] = pack_bf2(v[0], v[1]); o[1] = pack_bf2(v[2], v[3]);
                *reinterpret_cast<chunk8*>(dst) = o;
                if (GMODE == 3) {
                    chunk8 q;
                    q[0] = pack_bf2(d[0], d[1]); q[1] = pack_bf2(d[2], d[3]);
                    *reinterpret_cast<chunk8*>(dst + region) = q;
                }
            }
        }
}

template <int OSZ, int MODE, int TNC, int NTH>
__device__ __forceinline__ void drainT(const char* smem, int rows, void* dst, int64_t ld, const void* aux,
                                       int64_t ld_aux, int mbase, int n0, int M, int N, int tid) {
    using E = EpiT<OSZ, TNC>;
#pragma unroll 4
    for (int c = tid; c < rows * E::CPR; c += NTH) {
        const int row = c / E::CPR, cc = c - row * E::CPR;
        const int gm = mbase + row, gn = n0 + cc * E::EPC;
        if (gm >= M || gn >= N) continue;
        chunk16 v = *reinterpret_cast<const chunk16*>(smem + row * E::PITCH + cc * 16);
        if (MODE == 1) {
            const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(aux) + (int64_t)gm * ld_aux + gn);
            v[0] = f2u(u2f(v[0]) + r.x); v[1] = f2u(u2f(v[1]) + r.y);
            v[2] = f2u(u2f(v[2]) + r.z); v[3] = f2u(u2f(v[3]) + r.w);
        } else if (MODE == 2) {
            if (OSZ == 4) {
                const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(aux) + (int64_t)gm * ld_aux + gn);
                v[0] = f2u(u2f(v[0]) * r.x); v[1] = f2u(u2f(v[1]) * r.y);
                v[2] = f2u(u2f(v[2]) * r.z); v[3] = f2u(u2f(v[3]) * r.w);
            } else {
                const chunk16 r = *reinterpret_cast<const chunk16*>(reinterpret_cast<const bf16_t*>(aux) + (int64_t)gm * ld_aux + gn);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t vw = v[e], rw = r[e];
                    const float lo = lo16f(vw) * lo16f(rw);
                    const float hi = hi16f(vw) * hi16f(rw);
                    v[e] = pack_bf2(lo, hi);
                }
            }
        }
        // streaming output: written once, re-read by a later kernel after > L2-size of other traffic
        __builtin_nontemporal_store(v, reinterpret_cast<chunk16*>(reinterpret_cast<char*>(dst) + ((int64_t)gm * ld + gn) * OSZ));
    }
}

struct GemmTn256Params {
    const char* A;
    const char* B;
    float* C;
    float* colsum;
    int64_t lda, ldb, ldc;
    int M, N, K;
    int tiles_m, tiles_n;
    int k_slices_per_split, split_k;
    float* ws;      // NULL: the split-K partials are added to C with fp32 atomics; else [split][tile][wave][32 chunks][64 lanes][4] fp32
};
// gemm_tn_ow.hip: the one-wave-per-SIMD 256 x 256 wgrad kernel (bf16 operands, atomic split-K combine: p.ws == NULL)
int gemm_tn256o_launch(GemmTn256Params& p, int split_k, hipStream_t stream);
bool gemm_tn256o_available();   // false in a build whose register audit failed (maest_amd/build.py): the 8-wave kernel serves

// gemm_nt_ow.hip: the one-wave-per-SIMD 256 x 256 kernel (bf16 operands; N % 256 == 0, K % 64 == 0, fp32 output for RESIDUAL, bf16
// output for the GELU + GELU' pair)
int gemm_nt256o_launch(Gemm256Params& p, hipStream_t stream);
bool gemm_nt256o_available();

}  // namespace maest
