// MFMA GEMMs with fused epilogues for the MAEST ViT linears (reference: nn.Linear call sites
// models/maest.py:353,355,361,376 (qkv / proj), :197-199,203-206 (fc1 / GELU / fc2), :572 (head),
// the im2col form of nn.Conv2d :238-240, and their autograd dgrad / wgrad / bias-grad).
//
//   maest_gemm_nt:  C[M,N]  = epilogue( sum_k A[m,k] * B[n,k] )      forward + dgrad ("NT": k-contiguous)
//   maest_gemm_tn:  C[I,J] += sum_k A[k,i] * B[k,j]  (+ colsum_i A)  wgrad + bias grad ("TN": k = token)
//
// Shared design (gfx950): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave
// = 2x2 MFMA 32x32 tiles, 64 fp32 accumulators per lane).  K is walked in slices of 64 (bf16) / 32
// (fp32) so the bf16 perf path and the exact-fp32 parity path share all address arithmetic.
// Global -> registers -> LDS staging with the next slice's loads issued before the MFMAs of the
// current one (double-buffered LDS, one barrier per slice).  Workgroup ids are remapped so that each
// XCD sweeps a contiguous range of tiles and re-reads operand panels from its own L2.
//
// NT: LDS rows are 128-byte k-runs padded to 144 B (ds_read_b128 of 16 rows covers all 64 banks once).
//     The MFMA is issued with the WEIGHT tile as the A operand, so a lane ends up holding 4 consecutive
//     output columns per accumulator quad; the tile is then staged through LDS in the output dtype and
//     written with 16-byte coalesced stores (bias / GELU / residual / GELU' fused on the way).
// TN: both operands are token-major ([k][i], i contiguous) exactly as the activations / gradients sit in
//     HBM, so NO transposed copies are ever made: tiles are staged row-for-row and the MFMA fragments
//     are produced by the gfx950 LDS transpose read ds_read_b64_tr_b16 (bf16) or plain ds_read_b32
//     (fp32).  LDS pitch 320 B puts the 4 k-rows of a transpose read on disjoint bank quarters.
//     The bias gradient (column sum of A over the tokens) rides along as one extra MFMA against a
//     fragment of ones in the tile_j == 0 workgroups.  Split-K partials are combined with fp32 atomics.
#include "common.h"

namespace maest {

constexpr int GEMM_BM = 128;
constexpr int GEMM_BN = 128;
constexpr int GEMM_ROWB = 128;   // NT: payload bytes per tile row per K slice
constexpr int GEMM_PITCH = 144;  // NT: padded LDS row pitch (bytes)
constexpr int GEMM_TILE_BYTES = GEMM_BM * GEMM_PITCH;      // 18432
constexpr int GEMM_SMEM_BYTES = 4 * GEMM_TILE_BYTES;       // A,B x 2 buffers = 73728

struct GemmParams {
    const char* A;
    const char* B;
    void* C;
    const float* bias;
    const void* aux_in;
    void* aux_out;
    float* colsum;                  // TN only
    int64_t lda, ldb, ldc, ld_aux;  // in elements
    int M, N, K;
    int out_dtype;   // MAEST_F32 / MAEST_BF16
    int epi;         // MAEST_EPI_*
    int vec_ok;      // NT: 16-byte staged epilogue allowed (alignment / divisibility checked on the host)
    int tiles_m, tiles_n;
    int k_slices_per_split;  // K slices handled by one blockIdx.y
};

// ------------------------------------------------------------------------------------------------
// NT epilogue helpers
// ------------------------------------------------------------------------------------------------
template <int OSZ>
struct EpiCfg {
    static constexpr int PITCH = 128 * OSZ + 16;   // staged C tile pitch: 528 (fp32) / 272 (bf16)
    static constexpr int CPR = 128 * OSZ / 16;     // 16-byte chunks per row: 32 / 16
    static constexpr int EPC = 16 / OSZ;           // elements per chunk: 4 / 8
};

// registers -> LDS, in the output dtype.  acc[jn][im]: rows n = wn*64 + jn*32 + frag_row, col m = lane.
// GMODE: 0 = acc + bias, 1 = gelu(acc + bias), 2 = gelu'(acc + bias)
template <int OSZ, int GMODE, bool EXACT>
__device__ __forceinline__ void epi_stage(char* smem, const f32x16_t (&acc)[2][2], const float* bias, int n0, int N,
                                          int wm, int wn, int lane) {
    using E = EpiCfg<OSZ>;
    const int h = lane >> 5;
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = wn * 64 + jn * 32 + 8 * g + 4 * h;
            float b4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (bias != nullptr && n0 + nl < N) {   // N % 4 == 0 in the vector path
                const float4 t = *reinterpret_cast<const float4*>(bias + n0 + nl);
                b4[0] = t.x; b4[1] = t.y; b4[2] = t.z; b4[3] = t.w;
            }
#pragma unroll
            for (int im = 0; im < 2; ++im) {
                const int ml = wm * 64 + im * 32 + (lane & 31);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[jn][im][4 * g + e] + b4[e];
                    if (GMODE != 0) {
                        float gv, dv;
                        gelu_pair<EXACT>(v[e], gv, dv);
                        v[e] = GMODE == 1 ? gv : dv;
                    }
                }
                char* dst = smem + ml * E::PITCH + nl * OSZ;
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

// LDS -> global with 16-byte stores; MODE 0 plain, 1 += fp32 residual, 2 *= aux (aux in out dtype)
template <int OSZ, int MODE>
__device__ __forceinline__ void epi_drain(const char* smem, void* dst, int64_t ld, const void* aux, int64_t ld_aux,
                                          int m0, int n0, int M, int N, int tid) {
    using E = EpiCfg<OSZ>;
#pragma unroll 4
    for (int c = tid; c < 128 * E::CPR; c += 256) {
        const int row = c / E::CPR, cc = c - row * E::CPR;
        const int gm = m0 + row, gn = n0 + cc * E::EPC;
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
                    const float lo = bf2f((bf16_t)(vw & 0xffffu)) * bf2f((bf16_t)(rw & 0xffffu));
                    const float hi = bf2f((bf16_t)(vw >> 16)) * bf2f((bf16_t)(rw >> 16));
                    v[e] = pack_bf2(lo, hi);
                }
            }
        }
        __builtin_nontemporal_store(v, reinterpret_cast<chunk16*>(reinterpret_cast<char*>(dst) + ((int64_t)gm * ld + gn) * OSZ));
    }
}

template <int OSZ, bool EXACT>
__device__ __forceinline__ void epi_vector(char* smem, const f32x16_t (&acc)[2][2], const GemmParams& p, int m0, int n0,
                                           int wm, int wn, int lane, int tid) {
    switch (p.epi) {
        case MAEST_EPI_GELU:
            if (p.aux_out != nullptr) {
                epi_stage<OSZ, 2, EXACT>(smem, acc, p.bias, n0, p.N, wm, wn, lane);
                __syncthreads();
                epi_drain<OSZ, 0>(smem, p.aux_out, p.ld_aux, nullptr, 0, m0, n0, p.M, p.N, tid);
                __syncthreads();
            }
            epi_stage<OSZ, 1, EXACT>(smem, acc, p.bias, n0, p.N, wm, wn, lane);
            __syncthreads();
            epi_drain<OSZ, 0>(smem, p.C, p.ldc, nullptr, 0, m0, n0, p.M, p.N, tid);
            break;
        case MAEST_EPI_RESIDUAL:
            epi_stage<OSZ, 0, EXACT>(smem, acc, p.bias, n0, p.N, wm, wn, lane);
            __syncthreads();
            epi_drain<OSZ, 1>(smem, p.C, p.ldc, p.aux_in, p.ld_aux, m0, n0, p.M, p.N, tid);
            break;
        case MAEST_EPI_MUL:
            epi_stage<OSZ, 0, EXACT>(smem, acc, p.bias, n0, p.N, wm, wn, lane);
            __syncthreads();
            epi_drain<OSZ, 2>(smem, p.C, p.ldc, p.aux_in, p.ld_aux, m0, n0, p.M, p.N, tid);
            break;
        default:
            epi_stage<OSZ, 0, EXACT>(smem, acc, p.bias, n0, p.N, wm, wn, lane);
            __syncthreads();
            epi_drain<OSZ, 0>(smem, p.C, p.ldc, nullptr, 0, m0, n0, p.M, p.N, tid);
            break;
    }
}

// element-wise fallback (ragged N / unaligned leading dims, and the split-K atomic accumulate)
template <bool EXACT>
__device__ __forceinline__ void epi_scalar(const f32x16_t (&acc)[2][2], const GemmParams& p, int m0, int n0, int wm,
                                           int wn, int lane, bool add_bias) {
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        const int row = m0 + wm * 64 + im * 32 + (lane & 31);
        if (row >= p.M) continue;
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = n0 + wn * 64 + jn * 32 + frag_row(r, lane);
                if (col >= p.N) continue;
                float v = acc[jn][im][r] + ((p.bias != nullptr && add_bias) ? p.bias[col] : 0.0f);
                const int64_t ci = (int64_t)row * p.ldc + col;
                const int64_t xi = (int64_t)row * p.ld_aux + col;
                switch (p.epi) {
                    case MAEST_EPI_GELU: {
                        float gv, dv;
                        gelu_pair<EXACT>(v, gv, dv);
                        if (p.aux_out != nullptr) {
                            if (p.out_dtype == MAEST_BF16) reinterpret_cast<bf16_t*>(p.aux_out)[xi] = f2bf(dv);
                            else reinterpret_cast<float*>(p.aux_out)[xi] = dv;
                        }
                        v = gv;
                        break;
                    }
                    case MAEST_EPI_RESIDUAL:
                        v += reinterpret_cast<const float*>(p.aux_in)[xi];
                        break;
                    case MAEST_EPI_MUL: {
                        const float d = (p.out_dtype == MAEST_BF16)
                                            ? bf2f(reinterpret_cast<const bf16_t*>(p.aux_in)[xi])
                                            : reinterpret_cast<const float*>(p.aux_in)[xi];
                        v *= d;
                        break;
                    }
                    case MAEST_EPI_ATOMIC:
                        unsafeAtomicAdd(reinterpret_cast<float*>(p.C) + ci, v);
                        continue;
                    default:
                        break;
                }
                if (p.out_dtype == MAEST_SPLIT3_A) {          // [ hi | hi | lo ] thirds of a bf16 [M, 3 N] row (small M: the big-tile kernel has the fast form)
                    const bf16_t hi = f2bf(v), lo = f2bf(v - bf2f(hi));
                    bf16_t* c3 = reinterpret_cast<bf16_t*>(p.C) + ci;
                    c3[0] = hi;
                    c3[p.N] = hi;
                    c3[2 * (int64_t)p.N] = lo;
                } else if (p.out_dtype == MAEST_BF16) reinterpret_cast<bf16_t*>(p.C)[ci] = f2bf(v);
                else reinterpret_cast<float*>(p.C)[ci] = v;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// NT kernel
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5;

    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int tile_m = wg / p.tiles_n;
    const int tile_n = wg - tile_m * p.tiles_n;
    const int m0 = tile_m * GEMM_BM, n0 = tile_n * GEMM_BN;

    constexpr int ELT = (int)sizeof(T);
    constexpr int KS = GEMM_ROWB / ELT;  // elements per K slice
    const int total_slices = p.K / KS;
    const int s_begin = blockIdx.y * p.k_slices_per_split;
    int s_end = s_begin + p.k_slices_per_split;
    if (s_end > total_slices) s_end = total_slices;
    const int nslices = s_end - s_begin;

    // staging map: thread -> (row = tid>>3 (+32 i), 16-byte chunk = tid&7)
    const int ld_row = tid >> 3, ld_chunk = tid & 7;
    const char* a_src[4];
    const char* b_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int ra = m0 + ld_row + 32 * i;
        if (ra > p.M - 1) ra = p.M - 1;  // clamp: rows >= M are computed on garbage and never stored
        int rb = n0 + ld_row + 32 * i;
        if (rb > p.N - 1) rb = p.N - 1;
        a_src[i] = p.A + ((int64_t)ra * p.lda + (int64_t)s_begin * KS) * ELT + ld_chunk * 16;
        b_src[i] = p.B + ((int64_t)rb * p.ldb + (int64_t)s_begin * KS) * ELT + ld_chunk * 16;
    }
    const int st_off = ld_row * GEMM_PITCH + ld_chunk * 16;

    f32x16_t acc[2][2];   // [jn][im]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    chunk16 ra[4], rb[4];
    if (nslices > 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const chunk16*>(a_src[i]);
            rb[i] = *reinterpret_cast<const chunk16*>(b_src[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<chunk16*>(smem + st_off + i * 32 * GEMM_PITCH) = ra[i];
            *reinterpret_cast<chunk16*>(smem + GEMM_TILE_BYTES + st_off + i * 32 * GEMM_PITCH) = rb[i];
        }
    }
    __syncthreads();

    const int a_rd = (wm * 64 + (lane & 31)) * GEMM_PITCH;
    const int b_rd = (wn * 64 + (lane & 31)) * GEMM_PITCH;

    for (int s = 0; s < nslices; ++s) {
        const int cur = s & 1;
        const bool more = (s + 1) < nslices;
        if (more) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = *reinterpret_cast<const chunk16*>(a_src[i] + (int64_t)(s + 1) * GEMM_ROWB);
                rb[i] = *reinterpret_cast<const chunk16*>(b_src[i] + (int64_t)(s + 1) * GEMM_ROWB);
            }
        }
        const char* la = smem + cur * 2 * GEMM_TILE_BYTES;
        const char* lb = la + GEMM_TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int coff = (2 * ks + h) * 16;
            chunk16 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = *reinterpret_cast<const chunk16*>(la + a_rd + i * 32 * GEMM_PITCH + coff);
                fb[i] = *reinterpret_cast<const chunk16*>(lb + b_rd + i * 32 * GEMM_PITCH + coff);
            }
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int im = 0; im < 2; ++im) mma_chunk<T>(acc[jn][im], fb[jn], fa[im]);   // D rows = n, cols = m
        }
        if (more) {
            char* da = smem + (cur ^ 1) * 2 * GEMM_TILE_BYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<chunk16*>(da + st_off + i * 32 * GEMM_PITCH) = ra[i];
                *reinterpret_cast<chunk16*>(da + GEMM_TILE_BYTES + st_off + i * 32 * GEMM_PITCH) = rb[i];
            }
        }
        __syncthreads();
    }

    if (p.vec_ok && p.epi != MAEST_EPI_ATOMIC) {
        constexpr bool EXACT = sizeof(T) == 4;   // fp32 parity mode keeps libm erf
        if (p.out_dtype == MAEST_BF16) epi_vector<2, EXACT>(smem, acc, p, m0, n0, wm, wn, lane, tid);
        else epi_vector<4, EXACT>(smem, acc, p, m0, n0, wm, wn, lane, tid);
    } else {
        epi_scalar<sizeof(T) == 4>(acc, p, m0, n0, wm, wn, lane, blockIdx.y == 0);
    }
}

// ------------------------------------------------------------------------------------------------
// TN kernel:  C[i][j] += sum_k A[k][i] * B[k][j]   (+ colsum[i] += sum_k A[k][i])
// ------------------------------------------------------------------------------------------------
template <typename T>
struct TnCfg {
    static constexpr int ELT = (int)sizeof(T);
    static constexpr int KS = 128 / ELT;             // k rows per slice: 64 / 32
    static constexpr int ROWB = 128 * ELT;           // bytes per tile row (128 i): 256 / 512
    static constexpr int PITCH = ELT == 2 ? 320 : 528;
    static constexpr int CPR = ROWB / 16;            // 16 / 32 chunks per row
    static constexpr int RPP = 256 / CPR;            // rows per staging pass: 16 / 8
    static constexpr int EPC = 16 / ELT;
    static constexpr int TILE = KS * PITCH;          // 20480 / 16896
    static constexpr int SMEM = 4 * TILE;            // 81920 / 67584
};

typedef short v4i16_t __attribute__((ext_vector_type(4)));

// MFMA operand chunk of k-step `ks` for the 32-wide i-block starting at `iblk` of a [k][i] LDS tile
template <typename T>
__device__ __forceinline__ chunk16 load_frag_tn(const char* tile, int ks, int iblk, int lane);
template <>
__device__ __forceinline__ chunk16 load_frag_tn<bf16_t>(const char* tile, int ks, int iblk, int lane) {
    using C = TnCfg<bf16_t>;
    const int h = lane >> 5, g16 = (lane >> 4) & 1, q = lane & 15;
    const char* p = tile + (ks * 16 + 8 * h + (q >> 2)) * C::PITCH + (iblk + 16 * g16 + 4 * (q & 3)) * 2;
    // lane receives T[kb + j][iblk + (lane & 31)], j = 0..3  (ds_read_b64_tr_b16)
    const v4i16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16_t*)(p));
    const v4i16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16_t*)(p + 4 * C::PITCH));
    const chunk8 l2 = __builtin_bit_cast(chunk8, lo), h2 = __builtin_bit_cast(chunk8, hi);   // no repacking
    chunk16 c;
    c[0] = l2[0]; c[1] = l2[1]; c[2] = h2[0]; c[3] = h2[1];
    return c;
}
template <>
__device__ __forceinline__ chunk16 load_frag_tn<float>(const char* tile, int ks, int iblk, int lane) {
    using C = TnCfg<float>;
    const int h = lane >> 5;
    const char* p = tile + (ks * 8 + 4 * h) * C::PITCH + (iblk + (lane & 31)) * 4;
    chunk16 c;
#pragma unroll
    for (int q = 0; q < 4; ++q) c[q] = *reinterpret_cast<const uint32_t*>(p + q * C::PITCH);
    return c;
}
template <typename T>
__device__ __forceinline__ chunk16 ones_chunk();
template <>
__device__ __forceinline__ chunk16 ones_chunk<bf16_t>() {
    chunk16 c;
    c[0] = MAEST_ONE16X2; c[1] = MAEST_ONE16X2; c[2] = MAEST_ONE16X2; c[3] = MAEST_ONE16X2;
    return c;
}
template <>
__device__ __forceinline__ chunk16 ones_chunk<float>() {
    chunk16 c;
    c[0] = 0x3f800000u; c[1] = 0x3f800000u; c[2] = 0x3f800000u; c[3] = 0x3f800000u;
    return c;
}

template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmParams p) {
    using C = TnCfg<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int tile_i = wg / p.tiles_n;
    const int tile_j = wg - tile_i * p.tiles_n;
    const int i0 = tile_i * 128, j0 = tile_j * 128;

    const int total_slices = (p.K + C::KS - 1) / C::KS;
    const int s_begin = blockIdx.y * p.k_slices_per_split;
    int s_end = s_begin + p.k_slices_per_split;
    if (s_end > total_slices) s_end = total_slices;
    const int nslices = s_end - s_begin;

    const int ld_c = tid % C::CPR, ld_r = tid / C::CPR;
    const int st_off = ld_r * C::PITCH + ld_c * 16;
    // a chunk is loaded when it lies inside the row allocation (lda / ldb); columns >= M / N only feed
    // output rows / columns that are never stored
    const bool a_col_ok = (i0 + (ld_c + 1) * C::EPC) <= p.lda;
    const bool b_col_ok = (j0 + (ld_c + 1) * C::EPC) <= p.ldb;
    const char* a_col = p.A + (int64_t)(i0 + ld_c * C::EPC) * C::ELT;
    const char* b_col = p.B + (int64_t)(j0 + ld_c * C::EPC) * C::ELT;

    f32x16_t acc[2][2];   // [a: i-block][b: j-block]
    f32x16_t acc_cs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_cs[i][r] = 0.0f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    }
    const bool do_colsum = (p.colsum != nullptr) && (tile_j == 0) && (wn == 0);   // wave-uniform
    const chunk16 ones = ones_chunk<T>();

    chunk16 ra[4], rb[4];
    auto load_slice = [&](int s) {
        const int k0 = (s_begin + s) * C::KS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + ld_r + i * C::RPP;
            chunk16 z;
            z[0] = 0; z[1] = 0; z[2] = 0; z[3] = 0;
            ra[i] = (k < p.K && a_col_ok) ? *reinterpret_cast<const chunk16*>(a_col + (int64_t)k * p.lda * C::ELT) : z;
            rb[i] = (k < p.K && b_col_ok) ? *reinterpret_cast<const chunk16*>(b_col + (int64_t)k * p.ldb * C::ELT) : z;
        }
    };
    auto store_slice = [&](int buf) {
        char* da = smem + buf * 2 * C::TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<chunk16*>(da + st_off + i * C::RPP * C::PITCH) = ra[i];
            *reinterpret_cast<chunk16*>(da + C::TILE + st_off + i * C::RPP * C::PITCH) = rb[i];
        }
    };

    if (nslices > 0) {
        load_slice(0);
        store_slice(0);
    }
    __syncthreads();

    for (int s = 0; s < nslices; ++s) {
        const int cur = s & 1;
        const bool more = (s + 1) < nslices;
        if (more) load_slice(s + 1);
        const char* la = smem + cur * 2 * C::TILE;
        const char* lb = la + C::TILE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            chunk16 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = load_frag_tn<T>(la, ks, wm * 64 + i * 32, lane);
                fb[i] = load_frag_tn<T>(lb, ks, wn * 64 + i * 32, lane);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) mma_chunk<T>(acc[a][b], fa[a], fb[b]);   // D rows = i, cols = j
            if (do_colsum) {
                mma_chunk<T>(acc_cs[0], fa[0], ones);
                mma_chunk<T>(acc_cs[1], fa[1], ones);
            }
        }
        if (more) store_slice(cur ^ 1);
        __syncthreads();
    }

    float* Cp = reinterpret_cast<float*>(p.C);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = j0 + wn * 64 + b * 32 + (lane & 31);
        if (col >= p.N) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + a * 32 + frag_row(r, lane);
                if (row < p.M) unsafeAtomicAdd(Cp + (int64_t)row * p.ldc + col, acc[a][b][r]);
            }
    }
    if (do_colsum && (lane & 31) == 0) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + a * 32 + frag_row(r, lane);
                if (row < p.M) unsafeAtomicAdd(p.colsum + row, acc_cs[a][r]);
            }
    }
}

int gemm_nt256_try(const void* A, int64_t lda, const void* B, int64_t ldb, int in_dtype, void* C, int64_t ldc,
                   int out_dtype, int M, int N, int K, const float* bias, int epi, const void* aux_in, void* aux_out,
                   int64_t ld_aux, hipStream_t stream, float* rowdot = nullptr, int ntok = 0);   // gemm256.hip

int gemm_tn256_try(const void* A, int64_t lda, const void* B, int64_t ldb, int dtype, float* C, int64_t ldc, int M,
                   int N, int K, float* colsum, int split_k, hipStream_t stream, void* ws, int64_t ws_bytes);   // gemm256.hip
int64_t gemm_tn256_workspace_bytes(int dtype, int M, int N, int K, int split_k);                              // gemm256.hip

// rowdot[item, g, q] = sum_{c < 64} c_mat[m, 64 g + c] * other[m, 64 g + c], m = item * ntok + q: four lanes per (row, group).
// The stand-alone form of the MAEST_EPI_ROWDOT epilogue (shapes the 256-row-tile kernels do not take).
template <typename T>
__global__ __launch_bounds__(256) void rowdot_kernel(const T* __restrict__ c_mat, int64_t ldc, const T* __restrict__ other,
                                                     int64_t ld_other, float* __restrict__ rowdot, int M, int groups, int ntok) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)M * groups;
    int64_t item = gid >> 2;
    const int quarter = (int)(gid & 3);
    const bool valid = item < total;
    if (!valid) item = total - 1;
    const int64_t row = item / groups;
    const int g = (int)(item - row * groups);
    const T* pc = c_mat + row * ldc + g * 64 + quarter * 16;
    const T* po = other + row * ld_other + g * 64 + quarter * 16;
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += elem_traits<T>::to_f32(pc[i]) * elem_traits<T>::to_f32(po[i]);
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (valid && quarter == 0) {
        const int64_t it = row / ntok, q = row - it * ntok;
        rowdot[(it * groups + g) * ntok + q] = acc;
    }
}

template <typename T>
static int launch_gemm_nt(GemmParams& p, int split_k, hipStream_t stream) {
    static DeviceOnce once;
    ensure_dynamic_lds(once, &gemm_nt_kernel<T>, GEMM_SMEM_BYTES);
    dim3 grid(p.tiles_m * p.tiles_n, split_k, 1);
    hipLaunchKernelGGL(gemm_nt_kernel<T>, grid, dim3(256), GEMM_SMEM_BYTES, stream, p);
    return check_launch("maest_gemm_nt");
}

template <typename T>
static int launch_gemm_tn(GemmParams& p, int split_k, hipStream_t stream) {
    static DeviceOnce once;
    ensure_dynamic_lds(once, &gemm_tn_kernel<T>, TnCfg<T>::SMEM);
    dim3 grid(p.tiles_m * p.tiles_n, split_k, 1);
    hipLaunchKernelGGL(gemm_tn_kernel<T>, grid, dim3(256), TnCfg<T>::SMEM, stream, p);
    return check_launch("maest_gemm_tn");
}

}  // namespace maest

using namespace maest;

extern "C" int maest_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, int in_dtype,
                             void* C, int64_t ldc, int out_dtype, int M, int N, int K,
                             const float* bias, int epi, const void* aux_in, void* aux_out,
                             int64_t ld_aux, int split_k, void* stream) {
    MAEST_REQUIRE(A && B && C, "maest_gemm_nt: null operand");
    MAEST_REQUIRE(M > 0 && N > 0 && K > 0, "maest_gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
    MAEST_REQUIRE(in_dtype == MAEST_F32 || in_dtype == MAEST_BF16 || in_dtype == MAEST_F32X3,
                  "maest_gemm_nt: bad in_dtype %d", in_dtype);
    // split-bf16 products exist in the big-tile kernel; shapes it does not take run the exact fp32 kernel (a superset)
    const bool x3 = in_dtype == MAEST_F32X3;
    if (x3) in_dtype = MAEST_F32;
    // MAEST_SPLIT3_A output: gelu(acc + bias) written as [ hi | hi | lo ] bf16 thirds of a [M, 3 N] tensor (ldc >= 3 N) -- the A operand of
    // the next split product run as one bf16 GEMM over 3 K: a staged epilogue form of the one-wave-per-SIMD kernel, element-wise elsewhere
    const bool split_out = out_dtype == MAEST_SPLIT3_A;
    MAEST_REQUIRE(out_dtype == MAEST_F32 || out_dtype == MAEST_BF16 || split_out, "maest_gemm_nt: bad out_dtype %d", out_dtype);
    MAEST_REQUIRE(!split_out || (in_dtype == MAEST_BF16 && !x3 && epi == MAEST_EPI_GELU && !aux_in && !aux_out && ldc >= 3 * (int64_t)N && split_k == 1),
                  "maest_gemm_nt: MAEST_SPLIT3_A output needs bf16 operands, the GELU epilogue without side output and ldc >= 3 N");
    const int elt = in_dtype == MAEST_BF16 ? 2 : 4;
    const int ks = GEMM_ROWB / elt;
    MAEST_REQUIRE(K % ks == 0, "maest_gemm_nt: K=%d must be a multiple of %d (pad the operands)", K, ks);
    MAEST_REQUIRE((lda * elt) % 16 == 0 && (ldb * elt) % 16 == 0, "maest_gemm_nt: lda/ldb rows must be 16-byte multiples");
    MAEST_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "maest_gemm_nt: A/B must be 16-byte aligned");
    MAEST_REQUIRE(epi >= MAEST_EPI_NONE && epi <= MAEST_EPI_ATOMIC, "maest_gemm_nt: bad epilogue %d", epi);
    MAEST_REQUIRE(split_k >= 1, "maest_gemm_nt: split_k must be >= 1");
    MAEST_REQUIRE(split_k == 1 || epi == MAEST_EPI_ATOMIC, "maest_gemm_nt: split_k > 1 needs MAEST_EPI_ATOMIC");
    MAEST_REQUIRE(epi != MAEST_EPI_ATOMIC || out_dtype == MAEST_F32, "maest_gemm_nt: atomic epilogue accumulates fp32");
    MAEST_REQUIRE(epi != MAEST_EPI_RESIDUAL || (aux_in && out_dtype == MAEST_F32), "maest_gemm_nt: residual epilogue needs fp32 aux_in and fp32 out");
    MAEST_REQUIRE(epi != MAEST_EPI_MUL || aux_in, "maest_gemm_nt: mul epilogue needs aux_in");
    GemmParams p;
    p.A = (const char*)A; p.B = (const char*)B; p.C = C;
    p.bias = bias; p.aux_in = aux_in; p.aux_out = aux_out; p.colsum = nullptr;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ld_aux = ld_aux;
    p.M = M; p.N = N; p.K = K;
    p.out_dtype = out_dtype; p.epi = epi;
    const int osz = out_dtype == MAEST_F32 ? 4 : 2;
    const int epc = 16 / osz;
    bool vec = (N % epc == 0) && (N % 4 == 0) && (ldc % epc == 0) && ((uintptr_t)C % 16 == 0);
    if (bias) vec = vec && ((uintptr_t)bias % 16 == 0);
    if (aux_out) vec = vec && (ld_aux % epc == 0) && ((uintptr_t)aux_out % 16 == 0);
    if (aux_in) {
        const int aepc = (epi == MAEST_EPI_RESIDUAL) ? 4 : epc;
        vec = vec && (ld_aux % aepc == 0) && ((uintptr_t)aux_in % 16 == 0);
    }
    p.vec_ok = vec ? 1 : 0;
    if (vec && split_k == 1) {   // large, aligned problems go to the 256x256 LDS-DMA kernel
        const int rc = gemm_nt256_try(A, lda, B, ldb, x3 ? MAEST_F32X3 : in_dtype, C, ldc, out_dtype, M, N, K, bias, epi,
                                      aux_in, aux_out, ld_aux, (hipStream_t)stream);
        if (rc >= 0) return rc;
    }
    if (split_out) p.vec_ok = 0;       // (shapes the 256-row-tile kernel does not take: the element-wise epilogue writes the three thirds)
    p.tiles_m = (M + GEMM_BM - 1) / GEMM_BM;
    p.tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
    const int total = K / ks;
    if (split_k > total) split_k = total;
    p.k_slices_per_split = (total + split_k - 1) / split_k;
    split_k = (total + p.k_slices_per_split - 1) / p.k_slices_per_split;
    return in_dtype == MAEST_BF16 ? launch_gemm_nt<bf16_t>(p, split_k, (hipStream_t)stream)
                                  : launch_gemm_nt<float>(p, split_k, (hipStream_t)stream);
}

extern "C" int maest_gemm_nt_rowdot(const void* A, int64_t lda, const void* B, int64_t ldb, int in_dtype, void* C,
                                    int64_t ldc, int out_dtype, int M, int N, int K, const float* bias, const void* other,
                                    int64_t ld_other, float* rowdot, int rows_per_item, void* stream) {
    MAEST_REQUIRE(A && B && C && other && rowdot, "maest_gemm_nt_rowdot: null pointer");
    MAEST_REQUIRE(M > 0 && N > 0 && K > 0 && (N % 64) == 0, "maest_gemm_nt_rowdot: bad shape M=%d N=%d K=%d (N %% 64 == 0)", M, N, K);
    MAEST_REQUIRE(rows_per_item > 0 && (M % rows_per_item) == 0, "maest_gemm_nt_rowdot: M=%d is not a multiple of rows_per_item=%d",
                  M, rows_per_item);
    MAEST_REQUIRE(out_dtype == MAEST_F32 || out_dtype == MAEST_BF16, "maest_gemm_nt_rowdot: bad out_dtype %d", out_dtype);
    const int osz = out_dtype == MAEST_BF16 ? 2 : 4, epc = 16 / osz;
    MAEST_REQUIRE((ld_other % epc) == 0 && ((uintptr_t)other % 16) == 0 && (ldc % epc) == 0 && ((uintptr_t)C % 16) == 0,
                  "maest_gemm_nt_rowdot: C / other rows must be 16-byte aligned");
    const bool x3 = in_dtype == MAEST_F32X3;
    const int in_plain = x3 ? MAEST_F32 : in_dtype;
    MAEST_REQUIRE(in_plain == MAEST_F32 || in_plain == MAEST_BF16, "maest_gemm_nt_rowdot: bad in_dtype %d", in_dtype);
    const int elt = in_plain == MAEST_BF16 ? 2 : 4;
    if (K % (GEMM_ROWB / elt) == 0 && (lda * elt) % 16 == 0 && (ldb * elt) % 16 == 0 && ((uintptr_t)A % 16) == 0 &&
        ((uintptr_t)B % 16) == 0 && (bias == nullptr || ((uintptr_t)bias % 16) == 0)) {
        const int rc = gemm_nt256_try(A, lda, B, ldb, in_dtype, C, ldc, out_dtype, M, N, K, bias, MAEST_EPI_ROWDOT, other,
                                      nullptr, ld_other, (hipStream_t)stream, rowdot, rows_per_item);
        if (rc >= 0) return rc;
    }
    const int rc = maest_gemm_nt(A, lda, B, ldb, in_dtype, C, ldc, out_dtype, M, N, K, bias, MAEST_EPI_NONE, nullptr, nullptr,
                                 0, 1, stream);
    if (rc != MAEST_OK) return rc;
    const int groups = N / 64;
    const int64_t threads = (int64_t)M * groups * 4;
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (out_dtype == MAEST_BF16)
        hipLaunchKernelGGL(rowdot_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)C, ldc,
                           (const bf16_t*)other, ld_other, rowdot, M, groups, rows_per_item);
    else
        hipLaunchKernelGGL(rowdot_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)C, ldc,
                           (const float*)other, ld_other, rowdot, M, groups, rows_per_item);
    return check_launch("maest_gemm_nt_rowdot");
}

extern "C" int maest_gemm_tn_workspace_bytes(int dtype, int M, int N, int K, int split_k, int64_t* bytes) {
    MAEST_REQUIRE(bytes != nullptr, "maest_gemm_tn_workspace_bytes: null result pointer");
    MAEST_REQUIRE(M > 0 && N > 0 && K > 0 && split_k >= 0, "maest_gemm_tn_workspace_bytes: bad shape M=%d N=%d K=%d", M, N, K);
    MAEST_REQUIRE(dtype == MAEST_F32 || dtype == MAEST_BF16 || dtype == MAEST_F32X3, "maest_gemm_tn_workspace_bytes: bad dtype %d", dtype);
    *bytes = gemm_tn256_workspace_bytes(dtype, M, N, K, split_k);
    return MAEST_OK;
}

extern "C" int maest_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int dtype, float* C,
                             int64_t ldc, int M, int N, int K, float* colsum, int split_k, void* stream) {
    return maest_gemm_tn_ws(A, lda, B, ldb, dtype, C, ldc, M, N, K, colsum, split_k, nullptr, 0, stream);
}

extern "C" int maest_gemm_tn_ws(const void* A, int64_t lda, const void* B, int64_t ldb, int dtype, float* C,
                                int64_t ldc, int M, int N, int K, float* colsum, int split_k, void* workspace,
                                int64_t workspace_bytes, void* stream) {
    MAEST_REQUIRE(A && B && C, "maest_gemm_tn: null operand");
    MAEST_REQUIRE(workspace_bytes >= 0, "maest_gemm_tn_ws: negative workspace size");
    MAEST_REQUIRE(M > 0 && N > 0 && K > 0, "maest_gemm_tn: bad shape M=%d N=%d K=%d", M, N, K);
    MAEST_REQUIRE(dtype == MAEST_F32 || dtype == MAEST_BF16 || dtype == MAEST_F32X3, "maest_gemm_tn: bad dtype %d", dtype);
    const bool x3 = dtype == MAEST_F32X3;        // split-bf16 products exist in the 256-tile kernel; other shapes: exact fp32
    if (x3) dtype = MAEST_F32;
    const int elt = dtype == MAEST_BF16 ? 2 : 4;
    MAEST_REQUIRE(lda >= M && ldb >= N, "maest_gemm_tn: leading dims smaller than the matrix width");
    MAEST_REQUIRE((lda * elt) % 16 == 0 && (ldb * elt) % 16 == 0, "maest_gemm_tn: lda/ldb rows must be 16-byte multiples");
    MAEST_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "maest_gemm_tn: A/B must be 16-byte aligned");
    MAEST_REQUIRE(split_k >= 0, "maest_gemm_tn: split_k must be >= 0 (0 = automatic)");
    {   // large aligned problems go to the 256x256 LDS-DMA kernel
        const int rc = gemm_tn256_try(A, lda, B, ldb, x3 ? MAEST_F32X3 : dtype, C, ldc, M, N, K, colsum, split_k,
                                      (hipStream_t)stream, workspace, workspace_bytes);
        if (rc >= 0) return rc;
    }
    if (split_k == 0) {
        const int t = ((M + 127) / 128) * ((N + 127) / 128);
        split_k = 1024 / t > 0 ? 1024 / t : 1;
    }
    GemmParams p;
    p.A = (const char*)A; p.B = (const char*)B; p.C = C;
    p.bias = nullptr; p.aux_in = nullptr; p.aux_out = nullptr; p.colsum = colsum;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ld_aux = 0;
    p.M = M; p.N = N; p.K = K;
    p.out_dtype = MAEST_F32; p.epi = MAEST_EPI_ATOMIC; p.vec_ok = 0;
    p.tiles_m = (M + 127) / 128;
    p.tiles_n = (N + 127) / 128;
    const int ks = 128 / elt;
    const int total = (K + ks - 1) / ks;
    if (split_k > total) split_k = total;
    p.k_slices_per_split = (total + split_k - 1) / split_k;
    split_k = (total + p.k_slices_per_split - 1) / p.k_slices_per_split;
    return dtype == MAEST_BF16 ? launch_gemm_tn<bf16_t>(p, split_k, (hipStream_t)stream)
                               : launch_gemm_tn<float>(p, split_k, (hipStream_t)stream);
}
