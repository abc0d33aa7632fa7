// Fused softmax attention, forward and backward, for the MAEST ViT blocks
// (reference: Attention.forward, models/maest.py:358-378 -- qkv reshape/permute :362-364,
//  (q @ k^T) * scale :371, softmax :372, attn @ v :375; backward = autograd of the same).
//
// Layout contract (no head-major copies anywhere): q/k/v are read in place from the qkv Linear's
// output [B*N, 2304] (column = s*768 + head*64 + d) and the result is written as [B*N, 768]
// (column = head*64 + d), which IS the reference's (attn @ v).transpose(1, 2).reshape(B, N, C).
//
// MFMA scheme (wave64, 32x32 tiles, head_dim = 64).  A 32x32 MFMA leaves D with lane <-> column and
// registers <-> rows; such a D can be fed back, register-for-register, as the B operand of a second
// MFMA whose reduction runs over D's ROW index, provided the A operand is gathered with the same
// row permutation.  That gather is done straight from the ROW-MAJOR LDS tile by the gfx950 transpose
// read ds_read_b64_tr_b16 (bf16; 4 consecutive rows of one column per lane) or ds_read_b32 (fp32):
// no transposed copies of K / V / Q / dO are ever built (common.h: acc_to_chunk, frag_from_rows).
// Hence, with "^T" meaning "keys/queries on the D-row axis":
//   forward   S^T = K Q^T            -> softmax stats are per lane (lane = query)
//             O^T = V^T P^T          (A = V^T tile in LDS, B = P^T registers)
//   dK/dV     S   = Q K^T, dP = dO V^T   (lane = key), dV^T = dO^T P, dK^T = Q^T dS
//   dQ        S^T = K Q^T, dP^T = V dO^T (lane = query), dQ^T = K^T dS^T
// so the probability tile never leaves registers and no cross-lane shuffles are needed beyond one
// half-wave exchange for the running max.  Scores are never written to HBM; the only saved
// statistic is the per-row log-sum-exp.  N is ragged (560, 290, 281, 875, 1685 ...): tail keys are
// masked, tail rows are zero-filled in LDS and never stored.
//
// One template serves both numeric modes: T = bf16 (v_mfma_f32_32x32x16_bf16, fp32 accumulate,
// fp32 softmax) and T = float (v_mfma_f32_32x32x2_f32, exact fp32 -- parity mode).
#include <type_traits>

#include "attn_common.h"

namespace maest {

template <typename T>
struct AttnCfg {
    static constexpr int ELT = (int)sizeof(T);
    static constexpr int ROWB = HD * ELT;            // bytes of one head row: 128 / 256
    static constexpr int NCH = ROWB / 16;            // 16-byte chunks per row: 8 / 16
    static constexpr int EPC = 16 / ELT;             // elements per chunk: 8 / 4
    static constexpr int STEPS = NCH / 2;            // chunk steps over d: 4 / 8
    static constexpr int PITCH = ROWB + 16;          // row-major LDS pitch: 144 / 272
    static constexpr int PITCH_T = 64 * ELT + (ELT == 2 ? 8 : 16);  // transposed tile pitch: 136 / 272
    static constexpr int ROWS_PER_PASS = 256 / NCH;  // rows staged by 256 threads per pass: 32 / 16
    static constexpr int PASSES64 = 64 / ROWS_PER_PASS;  // 2 / 4
    static constexpr int TILE = 64 * PITCH;          // bytes of a 64-row row-major tile
    static constexpr int TILE_T = 64 * PITCH_T;      // bytes of a [64 d][64 rows] transposed tile
    static constexpr int ASTEPS = acc_steps<T>::value;
};

// Workgroup -> (row block, head, batch).  The row blocks of one (batch, head) re-read the same K/V (or Q/dO)
// tiles; dispatched as a plain 3-D grid they land on different XCDs (round-robin) and each pulls its own copy
// from HBM (measured 2.3x the algorithmic fetch at N = 290).  A 1-D grid remapped per XCD keeps them on one L2.
struct AttnBlock { int rb, head, b; };
__device__ __forceinline__ AttnBlock attn_block(int nrb, int B) {
    const int total = nrb * NHEADS * B;
    const int q = xcd_remap(blockIdx.x, total);
    AttnBlock a;
    a.rb = q % nrb;
    const int bh = q / nrb;
    a.head = bh % NHEADS;
    a.b = bh / NHEADS;
    return a;
}

// ---- 64-row tile staging: global -> registers -> LDS (row-major and/or transposed) -------------
template <typename T>
struct TileRegs {
    chunk16 c[AttnCfg<T>::PASSES64];
};

// rows [r0, r0+64) of a 64-wide head slice; `base` points at (row 0, d 0); rows >= nrows read as 0
template <typename T>
__device__ __forceinline__ void tile_load(TileRegs<T>& t, const T* base, int64_t row_stride, int r0,
                                          int nrows, int tid) {
    using C = AttnCfg<T>;
    const int c = tid % C::NCH;
    const int rr = tid / C::NCH;
#pragma unroll
    for (int p = 0; p < C::PASSES64; ++p) {
        const int r = r0 + rr + p * C::ROWS_PER_PASS;
        if (r < nrows) {
            t.c[p] = *reinterpret_cast<const chunk16*>(base + (int64_t)r * row_stride + c * C::EPC);
        } else {
            t.c[p][0] = 0; t.c[p][1] = 0; t.c[p][2] = 0; t.c[p][3] = 0;
        }
    }
}
template <typename T>
__device__ __forceinline__ void tile_store_rows(const TileRegs<T>& t, char* lds, int tid) {
    using C = AttnCfg<T>;
    const int c = tid % C::NCH;
    const int rr = tid / C::NCH;
#pragma unroll
    for (int p = 0; p < C::PASSES64; ++p)
        *reinterpret_cast<chunk16*>(lds + (rr + p * C::ROWS_PER_PASS) * C::PITCH + c * 16) = t.c[p];
}
// lds_t[d][row]  (pitch PITCH_T)
template <typename T>
__device__ __forceinline__ void tile_store_transposed(const TileRegs<T>& t, char* lds_t, int tid) {
    using C = AttnCfg<T>;
    const int c = tid % C::NCH;
    const int rr = tid / C::NCH;
#pragma unroll
    for (int p = 0; p < C::PASSES64; ++p) {
        const int r = rr + p * C::ROWS_PER_PASS;
        if constexpr (C::ELT == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint16_t v = (uint16_t)(t.c[p][e >> 1] >> ((e & 1) * 16));
                *reinterpret_cast<uint16_t*>(lds_t + (c * 8 + e) * C::PITCH_T + r * 2) = v;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                *reinterpret_cast<uint32_t*>(lds_t + (c * 4 + e) * C::PITCH_T + r * 4) = t.c[p][e];
        }
    }
}

// per-lane operand fragments of one row (row = lane&31 of a 32-row group), chunks 2s+h
template <typename T>
__device__ __forceinline__ void row_frags_load(chunk16 (&f)[AttnCfg<T>::STEPS], const T* base,
                                               int64_t row_stride, int row, int nrows, int h) {
    using C = AttnCfg<T>;
    const int r = row < nrows ? row : nrows - 1;  // clamped rows are never stored
#pragma unroll
    for (int s = 0; s < C::STEPS; ++s)
        f[s] = *reinterpret_cast<const chunk16*>(base + (int64_t)r * row_stride + (2 * s + h) * C::EPC);
}

// acc[32 x 32] += sum_d  A_lds[row0 + (lane&31)][d] * frag[d]   (A from a row-major LDS tile)
template <typename T, bool X3 = false>
__device__ __forceinline__ void mma_rows(f32x16_t& acc, const char* lds, int row0, int lane,
                                         const chunk16 (&frag)[AttnCfg<T>::STEPS]) {
    using C = AttnCfg<T>;
    const char* rp = lds + (row0 + (lane & 31)) * C::PITCH + (lane >> 5) * 16;
#pragma unroll
    for (int s = 0; s < C::STEPS; s += 2) {      // chunk steps in pairs (STEPS is 4 / 8): see common.h mma_chunk2
        const chunk16 a0 = *reinterpret_cast<const chunk16*>(rp + s * 32);
        const chunk16 a1 = *reinterpret_cast<const chunk16*>(rp + (s + 1) * 32);
        mma_chunk2<T, X3>(acc, a0, a1, frag[s], frag[s + 1]);
    }
}
// A-operand chunk At[d][rho] = Tile[rho][d] for step s of the 32-row group rho0, d = dblk*32 + (lane&31),
// gathered from the ROW-MAJOR tile with the row permutation of acc_to_chunk:
//   bf16: rows 16s + 4h + (0..3) and 16s + 8 + 4h + (0..3)  -> two ds_read_b64_tr_b16
//   fp32: rows  8s + 4h + (0..3)                            -> four ds_read_b32
template <typename T>
__device__ __forceinline__ chunk16 frag_from_rows(const char* tile, int rho0, int s, int dblk, int lane);
// the same gather from a row-major bf16 tile of ANY row pitch (bytes, multiple of 8)
template <int PITCH>
__device__ __forceinline__ chunk16 frag_from_rows_bf16(const char* tile, int rho0, int s, int dblk, int lane) {
    const int h = lane >> 5, g16 = (lane >> 4) & 1, q = lane & 15;
    const char* p = tile + (rho0 + 16 * s + 4 * h + (q >> 2)) * PITCH + (dblk * 32 + 16 * g16 + 4 * (q & 3)) * 2;
    const v4i16a_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(lds_cast<v4i16a_t>(p));
    const v4i16a_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(lds_cast<v4i16a_t>(p + 8 * PITCH));
    const chunk8 l2 = __builtin_bit_cast(chunk8, lo), h2 = __builtin_bit_cast(chunk8, hi);   // no repacking
    chunk16 c;
    c[0] = l2[0]; c[1] = l2[1]; c[2] = h2[0]; c[3] = h2[1];
    return c;
}
template <>
__device__ __forceinline__ chunk16 frag_from_rows<bf16_t>(const char* tile, int rho0, int s, int dblk, int lane) {
    return frag_from_rows_bf16<AttnCfg<bf16_t>::PITCH>(tile, rho0, s, dblk, lane);
}
template <>
__device__ __forceinline__ chunk16 frag_from_rows<float>(const char* tile, int rho0, int s, int dblk, int lane) {
    using C = AttnCfg<float>;
    const int h = lane >> 5;
    const char* p = tile + (rho0 + 8 * s + 4 * h) * C::PITCH + (dblk * 32 + (lane & 31)) * 4;
    chunk16 c;
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = *reinterpret_cast<const uint32_t*>(p + j * C::PITCH);
    return c;
}
// acc2[d-block db][32 d x 32 cols] += sum_{rho in 32-row group rho0} Tile[rho][d] * P[rho][col]
template <typename T, bool X3 = false>
__device__ __forceinline__ void mma_transposed(f32x16_t (&acc)[2], const char* tile, int rho0, int lane,
                                               const f32x16_t& p) {
    using C = AttnCfg<T>;
#pragma unroll
    for (int s = 0; s < C::ASTEPS; s += 2) {     // ASTEPS is 2 / 4
        const chunk16 b0 = acc_to_chunk<T>(p, s), b1 = acc_to_chunk<T>(p, s + 1);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const chunk16 a0 = frag_from_rows<T>(tile, rho0, s, db, lane);
            const chunk16 a1 = frag_from_rows<T>(tile, rho0, s + 1, db, lane);
            mma_chunk2<T, X3>(acc[db], a0, a1, b0, b1);
        }
    }
}

template <typename T>
__device__ __forceinline__ void store4(T* dst, float a, float b, float c, float d);
template <>
__device__ __forceinline__ void store4<bf16_t>(bf16_t* dst, float a, float b, float c, float d) {
    chunk8 v;
    v[0] = pack_bf2(a, b);
    v[1] = pack_bf2(c, d);
    *reinterpret_cast<chunk8*>(dst) = v;
}
template <>
__device__ __forceinline__ void store4<float>(float* dst, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(dst) = make_float4(a, b, c, d);
}
// write a [64 d][32 cols] accumulator pair (lane = col = row of the output matrix) to global
template <typename T>
__device__ __forceinline__ void store_dT(const f32x16_t (&acc)[2], T* row_ptr, int lane, float mul) {
    const int h = lane >> 5;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            store4<T>(row_ptr + db * 32 + 8 * g + 4 * h, acc[db][4 * g] * mul, acc[db][4 * g + 1] * mul,
                      acc[db][4 * g + 2] * mul, acc[db][4 * g + 3] * mul);
}

// the same accumulator pair as MAEST_SPLIT3_A thirds of a bf16 [.., 3 * 768] row (`row_ptr`: the head's first column in the first third);
// call it from converged code (store_32d_words16)
__device__ __forceinline__ void store_dT_split3(const f32x16_t (&acc)[2], bf16_t* row_ptr, int lane, float mul, bool ok) {
    store_32d_split3(acc[0], row_ptr, lane, mul, ok, OUT_LD);
    store_32d_split3(acc[1], row_ptr + 32, lane, mul, ok, OUT_LD);
}

__device__ __forceinline__ void store_dT_rows16(const f32x16_t (&acc)[2], bf16_t* row_ptr, int lane, float mul, bool ok) {
    store_32d_rows16(acc[0], row_ptr, lane, mul, ok);
    store_32d_rows16(acc[1], row_ptr + 32, lane, mul, ok);
}

// store_dT for any T; for bf16 the 16-byte-piece form (the caller's row predicate goes in as `ok`)
template <typename T>
__device__ __forceinline__ void store_dT_ok(const f32x16_t (&acc)[2], T* row_ptr, int lane, float mul, bool ok) {
    if constexpr (sizeof(T) == 2) store_dT_rows16(acc, row_ptr, lane, mul, ok);
    else if (ok) store_dT<T>(acc, row_ptr, lane, mul);
}

// =================================================================================== forward
template <typename T, bool X3 = false>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 4 : 1) void attn_fwd_kernel(const T* __restrict__ qkv,
                                                                                T* __restrict__ out,
                                                                                float* __restrict__ lse, int B, int N,
                                                                                float sc_c2, int q_rows, int out_a3) {
    // out_a3 (fp32 operands only): `out` is a bf16 [B * N, 3 * 768] tensor of MAEST_SPLIT3_A rows (hi | hi | lo of the fp32 result)
    using C = AttnCfg<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x { K[key][d], V[key][d] }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const AttnBlock blk = attn_block((N + 127) / 128, B);
    const int head = blk.head, b = blk.b;
    if (blk.rb * 128 >= q_rows) return;   // block-uniform: only the first q_rows queries are wanted (the head's tokens)
    const int q0 = blk.rb * 128 + wave * 32;
    const int q = q0 + (lane & 31);
    const bool wave_active = q0 < N && q0 < q_rows;   // wave-uniform
    const T* qbase = qkv + (int64_t)b * N * QKV_LD + head * HD;
    const T* kbase = qbase + NHEADS * HD;
    const T* vbase = qbase + 2 * NHEADS * HD;

    chunk16 qf[C::STEPS];
    row_frags_load<T>(qf, qbase, QKV_LD, q, N, h);

    f32x16_t o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.0f;
    float m_run = NEG_BIG, l_run = 0.0f;
    const float c2 = sc_c2;

#ifndef MAEST_ABLATE_FWD
#define MAEST_ABLATE_FWD 0     // timing experiments only (scratch/attn_ablate.sh fwd; results wrong on purpose): bit 0 no softmax math,
#endif                         // 1 no P V products, 2 no S products, 3 no K / V tile refills, 4 no O store
    const int ntiles = (N + 63) / 64;
    TileRegs<T> kr, vr;
    tile_load<T>(kr, kbase, QKV_LD, 0, N, tid);
    tile_load<T>(vr, vbase, QKV_LD, 0, N, tid);
    tile_store_rows<T>(kr, smem, tid);
    tile_store_rows<T>(vr, smem + C::TILE, tid);
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        const char* k_lds = smem + (kt & 1) * 2 * C::TILE;
        const char* v_lds = k_lds + C::TILE;
        const bool more = kt + 1 < ntiles && !(MAEST_ABLATE_FWD & 8);
        if (more) {
            tile_load<T>(kr, kbase, QKV_LD, (kt + 1) * 64, N, tid);
            tile_load<T>(vr, vbase, QKV_LD, (kt + 1) * 64, N, tid);
        }
        if (wave_active) {   // waves whose 32 queries are all padding only help staging the tiles
        // S^T[key][q]
        f32x16_t s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = (MAEST_ABLATE_FWD & 4) ? (float)(lane + r) : 0.0f;
            if (!(MAEST_ABLATE_FWD & 4)) mma_rows<T, X3>(s[kb], k_lds, kb * 32, lane, qf);
        }
        // online softmax in the scaled log2 domain: p = 2^(s*c2 - m).  Only the ragged last tile pays for
        // key masking; the elementwise work is written on float pairs (v_pk_fma_f32 / v_pk_add_f32).
        if (kt == ntiles - 1 && (N & 63) != 0) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * 64 + kb * 32 + frag_row(r, lane) >= N) s[kb][r] = NEG_BIG;
        }
        float mx = NEG_BIG;
        if (!(MAEST_ABLATE_FWD & 1)) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * c2);
        const float alpha = fast_exp2<T>(m_run - m_new);
        m_run = m_new;
        const f32x2_t c2v = {c2, c2}, nm = {-m_new, -m_new};
        f32x2_t ps = {0.0f, 0.0f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2_t sv = {s[kb][r], s[kb][r + 1]};
                const f32x2_t e = __builtin_elementwise_fma(sv, c2v, nm);
                const f32x2_t pv = (MAEST_ABLATE_FWD & 1) ? sv : f32x2_t{fast_exp2<T>(e[0]), fast_exp2<T>(e[1])};
                s[kb][r] = pv[0];
                s[kb][r + 1] = pv[1];
                ps += pv;
            }
        l_run = l_run * alpha + (ps[0] + ps[1]);  // per half-wave partial; halves are merged at the end
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        // O^T[d][q] += V^T[d][key] P^T[key][q]   (V^T gathered from the row-major V tile)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (!(MAEST_ABLATE_FWD & 2)) mma_transposed<T, X3>(o, v_lds, kb * 32, lane, s[kb]);
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[kb][r] += s[kb][r];
            }
        }
        }
        if (more) {
            char* nk = smem + ((kt + 1) & 1) * 2 * C::TILE;
            tile_store_rows<T>(kr, nk, tid);
            tile_store_rows<T>(vr, nk + C::TILE, tid);
        }
        __syncthreads();
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (wave_active) {      // (wave-uniform)
        // (staging O through LDS for whole-row stores was measured: no gain at N = 290, -16 % at N = 560;
        // profiles/r03_attn_fwd_ablation.txt.  The 16-byte pieces of store_dT_ok need no LDS and no barrier.)
        const bool ok = q < N && (!(MAEST_ABLATE_FWD & 16) || l_tot == 12345.0f);
        if (sizeof(T) == 4 && out_a3) {
            store_dT_split3(o, reinterpret_cast<bf16_t*>(out) + ((int64_t)b * N + (ok ? q : 0)) * (3 * OUT_LD) + head * HD, lane, inv, ok);
        } else {
            store_dT_ok<T>(o, out + ((int64_t)b * N + q) * OUT_LD + head * HD, lane, inv, ok);
        }
        if (ok && lse != nullptr && h == 0) lse[((int64_t)b * NHEADS + head) * N + q] = m_run * LN2 + logf(l_tot);
    }
}

// =================================================================================== delta
// delta[b,head,q] = sum_d dO[b,q,head,d] * O[b,q,head,d]      (4 lanes per (row, head))
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ o, const T* __restrict__ dout,
                                                         float* __restrict__ delta, int B, int N, int q_rows) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)B * N * NHEADS;
    int64_t item = gid >> 2;
    const int quarter = (int)(gid & 3);
    const bool valid = item < total;
    if (!valid) item = total - 1;
    const int64_t row = item / NHEADS;
    const int head = (int)(item - row * NHEADS);
    // rows beyond the query tiles the backward visits are never read: they re-read row 0 (one hot line) and store nothing
    const bool skip = q_rows < N && (int)(row % N) >= ((q_rows + 31) & ~31);
    const int64_t srow = skip ? 0 : row;
    const T* po = o + srow * OUT_LD + head * HD + quarter * 16;
    const T* pd = dout + srow * OUT_LD + head * HD + quarter * 16;
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += elem_traits<T>::to_f32(po[i]) * elem_traits<T>::to_f32(pd[i]);
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (valid && !skip && quarter == 0) {
        const int64_t bb = row / N;
        const int64_t qq = row - bb * N;
        delta[(bb * NHEADS + head) * N + qq] = acc;
    }
}

// =================================================================================== dK, dV
template <typename T, bool X3 = false>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void attn_bwd_dkdv_kernel(
    const T* __restrict__ qkv, const T* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ delta, T* __restrict__ dqkv, int B, int N, float sc_c2, float sc_dk) {
    using C = AttnCfg<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // 2 x { Q[q][d], dO[q][d], lse[64] (pre-multiplied by log2e), delta[64] }
    constexpr int BUF = 2 * C::TILE + 512;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const AttnBlock blk = attn_block((N + 127) / 128, B);
    const int head = blk.head, b = blk.b;
    const int key = blk.rb * 128 + wave * 32 + (lane & 31);
    const T* qbase = qkv + (int64_t)b * N * QKV_LD + head * HD;
    const T* kbase = qbase + NHEADS * HD;
    const T* vbase = qbase + 2 * NHEADS * HD;
    const T* dobase = dout + (int64_t)b * N * OUT_LD + head * HD;
    const float* lse_b = lse + ((int64_t)b * NHEADS + head) * N;
    const float* dl_b = delta + ((int64_t)b * NHEADS + head) * N;

    chunk16 kf[C::STEPS], vf[C::STEPS];
    row_frags_load<T>(kf, kbase, QKV_LD, key, N, h);
    row_frags_load<T>(vf, vbase, QKV_LD, key, N, h);
    const bool key_ok = key < N;
    const bool wave_active = blk.rb * 128 + wave * 32 < N;   // wave-uniform: all 32 keys of this wave are padding otherwise
    const f32x2_t c2v = {sc_c2, sc_c2};

    f32x16_t dk[2], dv[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[db][r] = 0.0f; dv[db][r] = 0.0f; }
    const float c2 = sc_c2;

    const int ntiles = (N + 63) / 64;
    TileRegs<T> qr, dr;
    float lse_r = 0.0f, dl_r = 0.0f;
    auto load_tile = [&](int qt) {
        tile_load<T>(qr, qbase, QKV_LD, qt * 64, N, tid);
        tile_load<T>(dr, dobase, OUT_LD, qt * 64, N, tid);
        if (tid < 64) {
            const int qq = qt * 64 + tid;
            lse_r = qq < N ? lse_b[qq] * LOG2E : -NEG_BIG;   // padded rows: P = 2^(-BIG) = 0
            dl_r = qq < N ? dl_b[qq] : 0.0f;
        }
    };
    auto store_tile = [&](int buf) {
        char* base = smem + buf * BUF;
        tile_store_rows<T>(qr, base, tid);
        tile_store_rows<T>(dr, base + C::TILE, tid);
        if (tid < 64) {
            reinterpret_cast<float*>(base + 2 * C::TILE)[tid] = lse_r;
            reinterpret_cast<float*>(base + 2 * C::TILE)[64 + tid] = dl_r;
        }
    };
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int qt = 0; qt < ntiles; ++qt) {
        const char* q_lds = smem + (qt & 1) * BUF;
        const char* do_lds = q_lds + C::TILE;
        const float* lse_lds = reinterpret_cast<const float*>(q_lds + 2 * C::TILE);
        const float* dl_lds = lse_lds + 64;
        const bool more = qt + 1 < ntiles;
        if (more) load_tile(qt + 1);
        if (wave_active) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
            mma_rows<T, X3>(s, q_lds, qb * 32, lane, kf);     // S[q][key]
            mma_rows<T, X3>(dp, do_lds, qb * 32, lane, vf);   // dP[q][key]
            // P = 2^(S*c2 - lse), dS = P * (dP - delta) on float pairs.  No masks: padded query rows carry
            // lse = +BIG (P = 0 exactly), and a padded key column only feeds its own never-stored lane.
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ql = qb * 32 + 8 * g + 4 * h;  // local q of register 4g (4 consecutive rows)
                const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse_lds + ql);
                const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(dl_lds + ql);
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const int r = 4 * g + e;
                    const f32x2_t sv = {s[r], s[r + 1]}, nl = {-l4[e], -l4[e + 1]};
                    const f32x2_t dpv = {dp[r], dp[r + 1]}, dl = {d4[e], d4[e + 1]};
                    const f32x2_t ev = __builtin_elementwise_fma(sv, c2v, nl);
                    const f32x2_t pv = {fast_exp2<T>(ev[0]), fast_exp2<T>(ev[1])};
                    const f32x2_t ds = pv * (dpv - dl);
                    s[r] = pv[0]; s[r + 1] = pv[1];       // P
                    dp[r] = ds[0]; dp[r + 1] = ds[1];     // dS (unscaled)
                }
            }
            mma_transposed<T, X3>(dv, do_lds, qb * 32, lane, s);   // dV^T[d][key] += dO^T[d][q] P[q][key]
            mma_transposed<T, X3>(dk, q_lds, qb * 32, lane, dp);   // dK^T[d][key] += Q^T[d][q] dS[q][key]
        }
        }
        if (more) store_tile((qt + 1) & 1);
        __syncthreads();
    }
    {
        T* row = dqkv + ((int64_t)b * N + key) * QKV_LD + head * HD;
        store_dT_ok<T>(dk, row + NHEADS * HD, lane, sc_dk, key_ok);
        store_dT_ok<T>(dv, row + 2 * NHEADS * HD, lane, 1.0f, key_ok);
    }
}

// =================================================================================== dQ
template <typename T, bool X3 = false>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void attn_bwd_dq_kernel(
    const T* __restrict__ qkv, const T* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ delta, T* __restrict__ dqkv, int B, int N, float sc_c2, float sc_dq) {
    using C = AttnCfg<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x { K[key][d], V[key][d] }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const AttnBlock blk = attn_block((N + 127) / 128, B);
    const int head = blk.head, b = blk.b;
    const int q = blk.rb * 128 + wave * 32 + (lane & 31);
    const T* qbase = qkv + (int64_t)b * N * QKV_LD + head * HD;
    const T* kbase = qbase + NHEADS * HD;
    const T* vbase = qbase + 2 * NHEADS * HD;
    const T* dobase = dout + (int64_t)b * N * OUT_LD + head * HD;

    chunk16 qf[C::STEPS], dof[C::STEPS];
    row_frags_load<T>(qf, qbase, QKV_LD, q, N, h);
    row_frags_load<T>(dof, dobase, OUT_LD, q, N, h);
    const bool q_ok = q < N;
    const bool wave_active = blk.rb * 128 + wave * 32 < N;   // wave-uniform
    const int qc = q_ok ? q : N - 1;
    const float lse_q = lse[((int64_t)b * NHEADS + head) * N + qc] * LOG2E;
    const float dl_q = delta[((int64_t)b * NHEADS + head) * N + qc];
    const f32x2_t nlse = {-lse_q, -lse_q}, dlv = {dl_q, dl_q}, c2v = {sc_c2, sc_c2};

    f32x16_t dq[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[db][r] = 0.0f;
    const float c2 = sc_c2;

    const int ntiles = (N + 63) / 64;
    TileRegs<T> kr, vr;
    tile_load<T>(kr, kbase, QKV_LD, 0, N, tid);
    tile_load<T>(vr, vbase, QKV_LD, 0, N, tid);
    tile_store_rows<T>(kr, smem, tid);
    tile_store_rows<T>(vr, smem + C::TILE, tid);
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        const char* k_lds = smem + (kt & 1) * 2 * C::TILE;
        const char* v_lds = k_lds + C::TILE;
        const bool more = kt + 1 < ntiles;
        if (more) {
            tile_load<T>(kr, kbase, QKV_LD, (kt + 1) * 64, N, tid);
            tile_load<T>(vr, vbase, QKV_LD, (kt + 1) * 64, N, tid);
        }
        if (wave_active) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
            mma_rows<T, X3>(s, k_lds, kb * 32, lane, qf);     // S^T[key][q]
            mma_rows<T, X3>(dp, v_lds, kb * 32, lane, dof);   // dP^T[key][q]
            // a padded key has a zero K row in LDS, so its dS column multiplies zeros below; the mask on the
            // ragged last tile only keeps 2^(-lse) from overflowing there
            if (kt == ntiles - 1 && (N & 63) != 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * 64 + kb * 32 + frag_row(r, lane) >= N) s[r] = NEG_BIG;
            }
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2_t sv = {s[r], s[r + 1]}, dpv = {dp[r], dp[r + 1]};
                const f32x2_t ev = __builtin_elementwise_fma(sv, c2v, nlse);
                const f32x2_t pv = {fast_exp2<T>(ev[0]), fast_exp2<T>(ev[1])};
                const f32x2_t ds = pv * (dpv - dlv);
                dp[r] = ds[0]; dp[r + 1] = ds[1];   // dS^T (unscaled)
            }
            mma_transposed<T, X3>(dq, k_lds, kb * 32, lane, dp);   // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
        }
        }
        if (more) {
            char* nk = smem + ((kt + 1) & 1) * 2 * C::TILE;
            tile_store_rows<T>(kr, nk, tid);
            tile_store_rows<T>(vr, nk + C::TILE, tid);
        }
        __syncthreads();
    }
    store_dT_ok<T>(dq, dqkv + ((int64_t)b * N + q) * QKV_LD + head * HD, lane, sc_dq, q_ok);
}

// =================================================================================== fused backward (bf16, N <= 320)
// ONE workgroup per (batch, head) and ONE pass over the query tiles for dQ, dK and dV: 5 MFMA products per (query
// block, key block) instead of the 7 of the two-kernel form above (S and dP are computed once), no delta kernel.
// The waves are SPECIALISED, each role with its own loop (so that their register live ranges do not add up):
//   * NKW = ceil(N / 32) KEY waves: wave w owns keys [32w, 32w + 32) for the whole launch -- its K and V rows are the
//     B operands (registers) of  S = Q K^T  and  dP = dO V^T, and its dK^T / dV^T accumulators never leave registers.
//     Per 32-row query tile: 16 MFMAs, no global memory access at all.
//   * 2 AUX waves.  (a) They feed the query tiles: wave A the Q tile, wave B the dO tile, computing
//     delta = rowsum(dO * O) on the way, global -> registers -> LDS with the loads issued TWO tiles ahead (the
//     exposed load latency of a one-tile-ahead version cost 27 % of the kernel).  (b) They compute dQ, one 32-wide
//     half of the head dim each: dQ needs every key of a query row, i.e. every key wave's dS, which the key waves
//     drop (bf16, [key][q]) into an LDS exchange tile; one barrier later
//     dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]  with both operands gathered by the transpose read from
//     row-major LDS (K resident for the whole launch), while the key waves already work on the next query tile.
//     With 10 + 2 waves every SIMD carries 3 waves and 48..52 MFMAs per tile.
// One barrier per tile.  Padded query rows carry lse = +BIG (P = 0); padded keys have zero K rows in LDS (their dS
// multiplies zeros) and their dK / dV rows are never stored.  dK / dV leave through LDS as whole 128-byte rows.
constexpr int FB_MAXW = 12;                       // 10 key waves + 2 aux waves -> 3 waves per SIMD, <= 168 VGPRs
constexpr int FB_DS_PITCH = 72;                   // dS exchange tile [key][32 q] bf16: 64 B + 8 (conflict-free 8-byte stores)
// (The first, register-fed form of this kernel -- query tiles staged through registers by the aux waves, delta computed in flight: MAEST_ATTN_BWD = 2 --
// was removed in round 6: superseded since round 2 by the DMA-fed forms below, which are 9 % faster with the separate delta pass.)

// =================================================================================== fused backward, DMA-fed (bf16, N <= 320)
// The decomposition above, with the aux waves relieved of the tile staging (their loop was the
// critical path: 260 us of staging + dQ against 240 us of key-wave work): K (resident) and the Q / dO query tiles are
// UNPADDED 128-byte-row tiles filled by LDS-DMA (no register round trip, no ds_write pass; two tiles ahead, three
// buffers), with the bank swizzle  chunk ^= row[1] row[2] row[3]  applied on the DMA source address so that the
// row-per-lane 16-byte reads AND the transpose reads are conflict-free (scratch/lds_banks.py; the padded pitch costs
// the transpose reads 2x).  delta = rowsum(dO * O) comes from attn_delta_kernel again (the DMA cannot compute it).
// DMA'd rows beyond N repeat row N - 1: padded queries carry lse = +BIG (P = 0 exactly), padded keys write dS = 0.
constexpr int F2_QBUF = 2 * 32 * 128 + 256;       // Q tile | dO tile | lse[32] | delta[32]
// LDS-DMA of 8-row groups [j0, j1) of an unpadded 128-byte-row tile (instruction j = rows 8j .. 8j + 7 = 1 KiB)
__device__ __forceinline__ void dma_rows128(char* tile, const bf16_t* base, int ld, int row0, int j0, int j1, int jstep,
                                            int nvalid, int lane) {
    for (int j = j0; j < j1; j += jstep) {
        const int lrow = 8 * j + (lane >> 3);                  // row inside the tile (the swizzle is a function of it)
        int grow = row0 + lrow;
        grow = grow < nvalid ? grow : nvalid - 1;
        const bf16_t* src = base + (uint32_t)(grow * ld + (((lane & 7) ^ swz128(lrow)) << 3));
        dma16(src, tile + j * 1024);
    }
}
__device__ __forceinline__ void mma_rows_swz(f32x16_t& acc, const char* tile, int row0, int lane, const chunk16 (&frag)[4]) {
    const int row = row0 + (lane & 31), f = swz128(row), h = lane >> 5;
    const char* rp = tile + row * 128;
#pragma unroll
    for (int s = 0; s < 4; s += 2) {
        const chunk16 a0 = *reinterpret_cast<const chunk16*>(rp + (((2 * s + h) ^ f) << 4));
        const chunk16 a1 = *reinterpret_cast<const chunk16*>(rp + (((2 * s + 2 + h) ^ f) << 4));
        mma_chunk2<bf16_t, false>(acc, a0, a1, frag[s], frag[s + 1]);
    }
}
__device__ __forceinline__ chunk16 frag_from_rows_swz(const char* tile, int rho0, int s, int dblk, int lane) {
    const int h = lane >> 5, g16 = (lane >> 4) & 1, q = lane & 15;
    const int row = rho0 + 16 * s + 4 * h + (q >> 2);
    const int cb = (dblk * 32 + 16 * g16 + 4 * (q & 3)) * 2;          // byte column
    const char* p0 = tile + row * 128 + ((((cb >> 4) ^ swz128(row)) << 4) | (cb & 15));
    const char* p1 = tile + (row + 8) * 128 + ((((cb >> 4) ^ swz128(row + 8)) << 4) | (cb & 15));
    const v4i16a_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(lds_cast<v4i16a_t>(p0));
    const v4i16a_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(lds_cast<v4i16a_t>(p1));
    const chunk8 l2 = __builtin_bit_cast(chunk8, lo), h2 = __builtin_bit_cast(chunk8, hi);
    chunk16 c;
    c[0] = l2[0]; c[1] = l2[1]; c[2] = h2[0]; c[3] = h2[1];
    return c;
}
__device__ __forceinline__ void mma_transposed_swz(f32x16_t (&acc)[2], const char* tile, int rho0, int lane, const f32x16_t& p) {
    const chunk16 b0 = acc_to_chunk<bf16_t>(p, 0), b1 = acc_to_chunk<bf16_t>(p, 1);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        const chunk16 a0 = frag_from_rows_swz(tile, rho0, 0, db, lane);
        const chunk16 a1 = frag_from_rows_swz(tile, rho0, 1, db, lane);
        mma_chunk2<bf16_t, false>(acc[db], a0, a1, b0, b1);
    }
}

// =================================================================================== forward, DMA-fed tiles (bf16)
// attn_fwd_kernel with its K / V tiles on the staging path of the fused backward: UNPADDED 128-byte-row tiles filled by LDS-DMA
// (two instructions of K and two of V per wave and tile, issued right behind the barrier that frees the buffer, for the tile
// after the one being consumed), bank-swizzled on the DMA source address so that the 16-byte row reads AND the transpose reads
// are conflict-free -- no staging registers, no ds_write pass, 32 KiB instead of 36.  The removal ablation of the
// register-staged form (profiles/r03_attn_fwd_ablation.txt) prices its K / V refills at a quarter of the kernel and its
// LDS bank-conflict cycles at 23 % (padded pitch: the transpose reads cost twice their ideal cycles).  Rows beyond N repeat row
// N - 1 (the DMA clamps): their scores are masked on the last tile as before, so P = 0 meets a finite V row.
#ifndef MAEST_FWD_RING
#define MAEST_FWD_RING 2      // ring depth of the K / V tiles: 2 = one tile ahead (32 KiB, 4 workgroups per CU); 3 = two ahead (48 KiB, 3 per CU)
#endif
// NW = waves per workgroup = 32-query blocks per workgroup (round 3, last part).  Every workgroup of a (batch, head) streams
// ALL of its K / V tiles from L2 into its own LDS, so fewer, larger workgroups move fewer bytes (N = 560: 3 x 192 or 3 x 256
// rows instead of 5 x 128).  Measured (scratch/attn_fwd_waves.py, profiles/r03_attn_fwd_waves.txt; bit-equal at every NW):
// it does not pay at the production shapes -- NW = 5 / 6 leave one SIMD with two waves of a workgroup whose per-tile barrier
// then waits for that SIMD (N = 290: 193 / 187 us against 147), NW = 8 pays its padding (N = 560: 429 against 407 us) and
// breaks even where the padding is equal (N = 1685: 756 against 767 us).  The refills are latency the four co-resident
// workgroups already hide, not bandwidth.  Only N <= 256 with more than 128 rows gains (N = 129: 56 against 65 us).
// attn_fwd_waves() therefore keeps NW = 4; MAEST_OPT_ATTN_FWD_WAVES forces another for tests and A/B.
template <int NW>
struct FwdWgs { static constexpr int value = MAEST_FWD_RING == 3 ? (NW <= 5 ? 3 : 2) : (16 / NW); };
template <int NW>
__global__ __launch_bounds__(NW * 64, FwdWgs<NW>::value) void attn_fwd_dma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                              float* __restrict__ lse, int B, int N, float sc_c2, int q_rows) {
    using T = bf16_t;
    using C = AttnCfg<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x { K[64 keys][128 B], V[64 keys][128 B] }
    constexpr int TILE128 = 64 * 128;

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int QB = NW * 32;           // query rows per workgroup
    const AttnBlock blk = attn_block((N + QB - 1) / QB, B);
    const int head = blk.head, b = blk.b;
    if (blk.rb * QB >= q_rows) return;    // block-uniform: only the first q_rows queries are wanted (the head's tokens)
    const int q0 = blk.rb * QB + wave * 32;
    const int q = q0 + (lane & 31);
    const bool wave_active = q0 < N && q0 < q_rows;   // wave-uniform
    const T* qbase = qkv + (int64_t)b * N * QKV_LD + head * HD;
    const T* kbase = qbase + NHEADS * HD;
    const T* vbase = qbase + 2 * NHEADS * HD;

    const int ntiles = (N + 63) / 64;
    constexpr int RING = MAEST_FWD_RING;      // 2: one tile ahead (32 KiB, 4 workgroups per CU); 3: two ahead (48 KiB, 3 per CU)
    static_assert(RING == 2 || NW == 4, "the counted waits of the three-deep ring assume four pieces per wave and tile");
    // this wave's share of the 16 one-KiB pieces (8 of K, 8 of V) of key tile kt -> ring buffer kt % RING.  Pieces 2w, 2w + 1 of
    // K and of V at NW = 4; dealt round-robin otherwise (the waits below are vmcnt(0): the count per wave does not matter)
    auto tile_dma = [&](int kt) {
        char* kb = smem + (kt % RING) * 2 * TILE128;
        if constexpr (NW == 4) {
            dma_rows128(kb, kbase, QKV_LD, kt * 64, 2 * wave, 2 * wave + 2, 1, N, lane);
            dma_rows128(kb + TILE128, vbase, QKV_LD, kt * 64, 2 * wave, 2 * wave + 2, 1, N, lane);
        } else {
#pragma unroll
            for (int p0 = 0; p0 < 16; p0 += NW) {
                const int p = p0 + wave;
                if (p < 8) dma_rows128(kb, kbase, QKV_LD, kt * 64, p, p + 1, 1, N, lane);
                else if (p < 16) dma_rows128(kb + TILE128, vbase, QKV_LD, kt * 64, p - 8, p - 7, 1, N, lane);
            }
        }
    };
    tile_dma(0);
    if (RING == 3 && ntiles > 1) tile_dma(1);
    chunk16 qf[C::STEPS];
    row_frags_load<T>(qf, qbase, QKV_LD, q, N, h);

    f32x16_t o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.0f;
    float m_run = NEG_BIG, l_run = 0.0f;
    const float c2 = sc_c2;
    if (RING == 3 && ntiles > 1) __builtin_amdgcn_s_waitcnt(0x0F74);      // vmcnt(4): tile 0 and the Q fragments landed, tile 1 may fly
    else MAEST_ATTN_WAIT_VM0();
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < ntiles; ++kt) {
        const char* k_lds = smem + (kt % RING) * 2 * TILE128;
        const char* v_lds = k_lds + TILE128;
        const bool issued = kt + RING - 1 < ntiles;
        if (issued) tile_dma(kt + RING - 1);     // its buffer was read a tile ago: everybody has passed the barrier since
        if (wave_active) {
            f32x16_t s[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
                mma_rows_swz(s[kb], k_lds, kb * 32, lane, qf);           // S^T[key][q]
            }
            if (kt == ntiles - 1 && (N & 63) != 0) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kt * 64 + kb * 32 + frag_row(r, lane) >= N) s[kb][r] = NEG_BIG;
            }
            float mx = NEG_BIG;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx * c2);
            const float alpha = fast_exp2<T>(m_run - m_new);
            m_run = m_new;
            const f32x2_t c2v = {c2, c2}, nm = {-m_new, -m_new};
            f32x2_t ps = {0.0f, 0.0f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2_t sv = {s[kb][r], s[kb][r + 1]};
                    const f32x2_t e = __builtin_elementwise_fma(sv, c2v, nm);
                    const f32x2_t pv = {fast_exp2<T>(e[0]), fast_exp2<T>(e[1])};
                    s[kb][r] = pv[0];
                    s[kb][r + 1] = pv[1];
                    ps += pv;
                }
            l_run = l_run * alpha + (ps[0] + ps[1]);  // per half-wave partial; halves are merged at the end
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) mma_transposed_swz(o, v_lds, kb * 32, lane, s[kb]);   // O^T[d][q] += V^T[d][key] P^T[key][q]
        }
        // this wave's share of the next tile has landed; behind the barrier everybody's has, and nobody reads this tile any more
        if (RING == 3 && issued) __builtin_amdgcn_s_waitcnt(0x0074);      // vmcnt(4) lgkmcnt(0): only the newest tile may fly
        else __builtin_amdgcn_s_waitcnt(0x0070);        // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (wave_active) {      // (wave-uniform)
        store_dT_ok<T>(o, out + ((int64_t)b * N + q) * OUT_LD + head * HD, lane, inv, q < N);
        if (q < N && lse != nullptr && h == 0) lse[((int64_t)b * NHEADS + head) * N + q] = m_run * LN2 + logf(l_tot);
    }
}

// =================================================================================== forward, split-bf16 on pre-split tiles (fp32 tensors)
// attn_fwd_kernel<float, X3> splits every fp32 operand chunk into its bf16 hi / lo parts at the point of use: every wave of a workgroup
// re-splits the same K and V tile data, the K rows / V^T fragments come out of fp32 tiles (V^T: four ds_read_b32 per chunk), and the
// VALU work of the splits sits between the MFMAs.  Here (round 5; complete passes, q_rows = N) a K / V tile is split ONCE while it is
// staged -- 16 fp32 values per thread and tensor -> K_hi, K_lo, V_hi, V_lo as unpadded bank-swizzled bf16 tiles, the layout of the bf16
// kernels above -- so the products read 16-byte bf16 row chunks and transpose-read fragments exactly like attn_fwd_dma_kernel, three MFMAs
// per chunk step (lo*hi, hi*lo, hi*hi: small terms first, as mma_chunk2<float, X3>), Q is split once per workgroup, and only the
// probabilities are split per tile.  Same contract as the X3 kernel (1e-4 of the output scale against fp64; not bit-equal: the k order
// inside a chunk step differs).  out_a3: the result leaves as MAEST_SPLIT3_A rows.
__device__ __forceinline__ void x3_split4(const chunk16& f, chunk8& hi, chunk8& lo) {     // four fp32 -> their bf16 hi and lo parts
    uint32_t h0, l0, h1, l1;
    split_bf2(u2f(f[0]), u2f(f[1]), h0, l0);
    split_bf2(u2f(f[2]), u2f(f[3]), h1, l1);
    hi[0] = h0; hi[1] = h1;
    lo[0] = l0; lo[1] = l1;
}
struct X3TileRegs { chunk16 c[4]; };
// rows [r0, r0 + 64) of a head's fp32 slice: thread t holds the four fp32 of 16-byte column t & 15 of rows (t >> 4) + 16 p
__device__ __forceinline__ void x3_tile_load(X3TileRegs& t, const float* base, int r0, int nrows, int tid) {
    const int c = tid & 15, rr = tid >> 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int r = r0 + rr + 16 * p;
        r = r < nrows ? r : nrows - 1;            // (rows beyond N repeat the last one: masked scores, P = 0 against a finite V row)
        t.c[p] = *reinterpret_cast<const chunk16*>(base + (int64_t)r * QKV_LD + c * 4);
    }
}
// ... split and stored as two bf16 tiles (64 rows x 128 B, 16-byte chunk ^= swz128(row))
__device__ __forceinline__ void x3_tile_store(const X3TileRegs& t, char* hi_tile, char* lo_tile, int tid) {
    const int c = tid & 15, rr = tid >> 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = rr + 16 * p;
        chunk8 hi, lo;
        x3_split4(t.c[p], hi, lo);
        const int off = row * 128 + ((((c >> 1) ^ swz128(row)) << 4) | ((c & 1) << 3));
        *reinterpret_cast<chunk8*>(hi_tile + off) = hi;
        *reinterpret_cast<chunk8*>(lo_tile + off) = lo;
    }
}
__global__ __launch_bounds__(256, 2) void attn_fwd_x3s_kernel(const float* __restrict__ qkv, void* __restrict__ out, float* __restrict__ lse,
                                                              int B, int N, float sc_c2, int out_a3) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x { K_hi, K_lo, V_hi, V_lo }: 64 KiB
    constexpr int TILE128 = 64 * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const AttnBlock blk = attn_block((N + 127) / 128, B);
    const int head = blk.head, b = blk.b;
    const int q0 = blk.rb * 128 + wave * 32;
    const int q = q0 + (lane & 31);
    const bool wave_active = q0 < N;              // wave-uniform
    const float* qbase = qkv + (int64_t)b * N * QKV_LD + head * HD;
    const float* kbase = qbase + NHEADS * HD;
    const float* vbase = qbase + 2 * NHEADS * HD;

    const int ntiles = (N + 63) / 64;
    X3TileRegs kr, vr;
    x3_tile_load(kr, kbase, 0, N, tid);
    x3_tile_load(vr, vbase, 0, N, tid);
    // Q: chunk step s of the bf16 layout = d 16 s + 8 h ..+7 of this lane's row, split once
    chunk16 qh[4], ql[4];
    {
        const float* qp = qbase + (int64_t)(q < N ? q : N - 1) * QKV_LD;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const chunk16 f0 = *reinterpret_cast<const chunk16*>(qp + 16 * st + 8 * h);
            const chunk16 f1 = *reinterpret_cast<const chunk16*>(qp + 16 * st + 8 * h + 4);
            chunk8 h0, l0, h1, l1;
            x3_split4(f0, h0, l0);
            x3_split4(f1, h1, l1);
            qh[st][0] = h0[0]; qh[st][1] = h0[1]; qh[st][2] = h1[0]; qh[st][3] = h1[1];
            ql[st][0] = l0[0]; ql[st][1] = l0[1]; ql[st][2] = l1[0]; ql[st][3] = l1[1];
        }
    }
    x3_tile_store(kr, smem, smem + TILE128, tid);
    x3_tile_store(vr, smem + 2 * TILE128, smem + 3 * TILE128, tid);
    f32x16_t o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.0f;
    float m_run = NEG_BIG, l_run = 0.0f;
    const float c2 = sc_c2;
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        const char* buf = smem + (kt & 1) * 4 * TILE128;
        const char *khi = buf, *klo = buf + TILE128, *vhi = buf + 2 * TILE128, *vlo = buf + 3 * TILE128;
        const bool more = kt + 1 < ntiles;
        if (more) {
            x3_tile_load(kr, kbase, (kt + 1) * 64, N, tid);
            x3_tile_load(vr, vbase, (kt + 1) * 64, N, tid);
        }
        if (wave_active) {   // waves whose 32 queries are all padding only help staging the tiles
            // S^T[key][q] = K Q^T: per 16-d chunk step lo*hi, hi*lo, hi*hi
            f32x16_t s[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
                const int row = kb * 32 + (lane & 31), f = swz128(row);
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int off = row * 128 + (((2 * st + h) ^ f) << 4);
                    const chunk16 ah = *reinterpret_cast<const chunk16*>(khi + off);
                    const chunk16 al = *reinterpret_cast<const chunk16*>(klo + off);
                    mma_chunk<bf16_t>(s[kb], al, qh[st]);
                    mma_chunk<bf16_t>(s[kb], ah, ql[st]);
                    mma_chunk<bf16_t>(s[kb], ah, qh[st]);
                }
            }
            if (kt == ntiles - 1 && (N & 63) != 0) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kt * 64 + kb * 32 + frag_row(r, lane) >= N) s[kb][r] = NEG_BIG;
            }
            float mx = NEG_BIG;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx * c2);
            const float alpha = fast_exp2<float>(m_run - m_new);
            m_run = m_new;
            const f32x2_t c2v = {c2, c2}, nm = {-m_new, -m_new};
            f32x2_t ps = {0.0f, 0.0f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2_t sv = {s[kb][r], s[kb][r + 1]};
                    const f32x2_t e = __builtin_elementwise_fma(sv, c2v, nm);
                    const f32x2_t pv = {fast_exp2<float>(e[0]), fast_exp2<float>(e[1])};
                    s[kb][r] = pv[0];
                    s[kb][r + 1] = pv[1];
                    ps += pv;
                }
            l_run = l_run * alpha + (ps[0] + ps[1]);  // per half-wave partial; halves are merged at the end
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
            // O^T[d][q] += V^T[d][key] P^T[key][q]: P split per 16-key step (the register -> key map of acc_to_chunk)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    chunk16 bh, bl;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        uint32_t hw, lw;
                        split_bf2(s[kb][8 * st + 2 * j], s[kb][8 * st + 2 * j + 1], hw, lw);
                        bh[j] = hw;
                        bl[j] = lw;
                    }
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const chunk16 ah = frag_from_rows_swz(vhi, kb * 32, st, db, lane);
                        const chunk16 al = frag_from_rows_swz(vlo, kb * 32, st, db, lane);
                        mma_chunk<bf16_t>(o[db], al, bh);
                        mma_chunk<bf16_t>(o[db], ah, bl);
                        mma_chunk<bf16_t>(o[db], ah, bh);
                    }
                }
        }
        if (more) {
            char* nb = smem + ((kt + 1) & 1) * 4 * TILE128;
            x3_tile_store(kr, nb, nb + TILE128, tid);
            x3_tile_store(vr, nb + 2 * TILE128, nb + 3 * TILE128, tid);
        }
        __syncthreads();
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (wave_active) {      // (wave-uniform)
        const bool ok = q < N;
        if (out_a3) {
            store_dT_split3(o, reinterpret_cast<bf16_t*>(out) + ((int64_t)b * N + (ok ? q : 0)) * (3 * OUT_LD) + head * HD, lane, inv, ok);
        } else if (ok) {
            store_dT<float>(o, reinterpret_cast<float*>(out) + ((int64_t)b * N + q) * OUT_LD + head * HD, lane, inv);
        }
        if (ok && lse != nullptr && h == 0) lse[((int64_t)b * NHEADS + head) * N + q] = m_run * LN2 + logf(l_tot);
    }
}

// =================================================================================== two-kernel backward, DMA-fed tiles (bf16)
// attn_bwd_dkdv_kernel / attn_bwd_dq_kernel (the forms that serve N > 320: the 30 s shapes) with their streamed tiles on the
// staging path of attn_fwd_dma_kernel: UNPADDED 128-byte-row tiles filled by LDS-DMA (two pieces of each of the two tiles per
// wave and step, requested right behind the barrier that frees the buffer), bank-swizzled on the DMA source address so that
// the row reads AND the transpose reads are conflict-free (the padded pitch costs the transpose reads -- half of these kernels'
// LDS traffic, and they are LDS-bandwidth bound -- twice their ideal cycles); no staging registers, no ds_write pass.  The same
// products in the same order: bit-equal to the register-staged forms (rows beyond N repeat row N - 1 instead of being zero:
// padded queries carry lse = +BIG and padded keys a masked score, so P = dS = 0 exactly against a finite operand).
// (launch bound: three waves per SIMD -- 168 VGPRs, no spill; at two the compiler spent 217 and the kernel ran 3.5 .. 4.4 % slower at N = 875 / 560)
__global__ __launch_bounds__(256, 3) void attn_bwd_dkdv_dma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                   const float* __restrict__ lse, const float* __restrict__ delta,
                                                                   bf16_t* __restrict__ dqkv, int B, int N, float sc_c2, float sc_dk) {
    using T = bf16_t;
    using C = AttnCfg<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE128 = 64 * 128;
    constexpr int BUF = 2 * TILE128 + 512;      // { Q[q][d], dO[q][d], lse[64] (pre-multiplied by log2e), delta[64] }

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const AttnBlock blk = attn_block((N + 127) / 128, B);
    const int head = blk.head, b = blk.b;
    const int key = blk.rb * 128 + wave * 32 + (lane & 31);
    const T* qbase = qkv + (int64_t)b * N * QKV_LD + head * HD;
    const T* kbase = qbase + NHEADS * HD;
    const T* vbase = qbase + 2 * NHEADS * HD;
    const T* dobase = dout + (int64_t)b * N * OUT_LD + head * HD;
    const float* lse_b = lse + ((int64_t)b * NHEADS + head) * N;
    const float* dl_b = delta + ((int64_t)b * NHEADS + head) * N;

    const int ntiles = (N + 63) / 64;
    auto tile_dma = [&](int qt) {
        char* base = smem + (qt & 1) * BUF;
        dma_rows128(base, qbase, QKV_LD, qt * 64, 2 * wave, 2 * wave + 2, 1, N, lane);
        dma_rows128(base + TILE128, dobase, OUT_LD, qt * 64, 2 * wave, 2 * wave + 2, 1, N, lane);
    };
    // The statistics of the next tile are REQUESTED here and first touched in stat_store, behind the tile's products: a use right behind
    // the load (the scaling, the padding select) makes hipcc put its counted vmcnt wait there -- in front of the products -- and that wait
    // also covers the four LDS-DMA pieces of the next tile requested just before (hidden from the compiler, not from the counter): the whole
    // prefetch would be waited for before the tile it should overlap with (round 5: profiles/r05_attn_bwd_two_kernel.txt).
    float lse_r = 0.0f, dl_r = 0.0f;
    auto stat_load = [&](int qt) {
        if (tid < 64) {
            const int qq = qt * 64 + tid, qc = qq < N ? qq : N - 1;
            lse_r = lse_b[qc];
            dl_r = dl_b[qc];
        }
    };
    auto stat_store = [&](int qt) {
        if (tid < 64) {
            pin_loaded(lse_r);
            pin_loaded(dl_r);
            const int qq = qt * 64 + tid;
            float* st = reinterpret_cast<float*>(smem + (qt & 1) * BUF + 2 * TILE128);
            st[tid] = qq < N ? lse_r * LOG2E : -NEG_BIG;     // padded rows: P = 2^(-BIG) = 0
            st[64 + tid] = qq < N ? dl_r : 0.0f;
        }
    };
    tile_dma(0);
    stat_load(0);
    chunk16 kf[C::STEPS], vf[C::STEPS];
    row_frags_load<T>(kf, kbase, QKV_LD, key, N, h);
    row_frags_load<T>(vf, vbase, QKV_LD, key, N, h);
#pragma unroll
    for (int s = 0; s < C::STEPS; ++s) { pin_loaded(kf[s]); pin_loaded(vf[s]); }       // (see pin_loaded)
    const bool key_ok = key < N;
    const bool wave_active = blk.rb * 128 + wave * 32 < N;   // wave-uniform
    const f32x2_t c2v = {sc_c2, sc_c2};
    f32x16_t dk[2], dv[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[db][r] = 0.0f; dv[db][r] = 0.0f; }
    stat_store(0);
    __builtin_amdgcn_s_waitcnt(0x0070);            // vmcnt(0) lgkmcnt(0): this wave's pieces, fragments and statistics
    __builtin_amdgcn_s_barrier();
    for (int qt = 0; qt < ntiles; ++qt) {
        const char* q_lds = smem + (qt & 1) * BUF;
        const char* do_lds = q_lds + TILE128;
        const float* lse_lds = reinterpret_cast<const float*>(q_lds + 2 * TILE128);
        const float* dl_lds = lse_lds + 64;
        const bool more = qt + 1 < ntiles;
        if (more) {                                 // the other buffer was read a tile ago: everybody has passed the barrier since
            tile_dma(qt + 1);
            stat_load(qt + 1);
        }
        if (wave_active) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                f32x16_t sc, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sc[r] = 0.0f; dp[r] = 0.0f; }
                mma_rows_swz(sc, q_lds, qb * 32, lane, kf);      // S[q][key]
                mma_rows_swz(dp, do_lds, qb * 32, lane, vf);     // dP[q][key]
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ql = qb * 32 + 8 * g + 4 * h;
                    const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse_lds + ql);
                    const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(dl_lds + ql);
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const int r = 4 * g + e;
                        const f32x2_t sv = {sc[r], sc[r + 1]}, nl = {-l4[e], -l4[e + 1]};
                        const f32x2_t dpv = {dp[r], dp[r + 1]}, dl = {d4[e], d4[e + 1]};
                        const f32x2_t ev = __builtin_elementwise_fma(sv, c2v, nl);
                        const f32x2_t pv = {fast_exp2<T>(ev[0]), fast_exp2<T>(ev[1])};
                        const f32x2_t ds = pv * (dpv - dl);
                        sc[r] = pv[0]; sc[r + 1] = pv[1];       // P
                        dp[r] = ds[0]; dp[r + 1] = ds[1];       // dS (unscaled)
                    }
                }
                mma_transposed_swz(dv, do_lds, qb * 32, lane, sc);   // dV^T[d][key] += dO^T[d][q] P[q][key]
                mma_transposed_swz(dk, q_lds, qb * 32, lane, dp);    // dK^T[d][key] += Q^T[d][q] dS[q][key]
            }
        }
        if (more) stat_store(qt + 1);
        __builtin_amdgcn_s_waitcnt(0x0070);        // this wave's share of the next tile has landed; behind the barrier everybody's
        __builtin_amdgcn_s_barrier();
    }
    T* row = dqkv + ((int64_t)b * N + (key_ok ? key : 0)) * QKV_LD + head * HD;
    store_dT_ok<T>(dk, row + NHEADS * HD, lane, sc_dk, key_ok);
    store_dT_ok<T>(dv, row + 2 * NHEADS * HD, lane, 1.0f, key_ok);
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dq_dma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                 const float* __restrict__ lse, const float* __restrict__ delta,
                                                                 bf16_t* __restrict__ dqkv, int B, int N, float sc_c2, float sc_dq) {
    using T = bf16_t;
    using C = AttnCfg<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x { K[key][128 B], V[key][128 B] }
    constexpr int TILE128 = 64 * 128;

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const AttnBlock blk = attn_block((N + 127) / 128, B);
    const int head = blk.head, b = blk.b;
    const int q = blk.rb * 128 + wave * 32 + (lane & 31);
    const T* qbase = qkv + (int64_t)b * N * QKV_LD + head * HD;
    const T* kbase = qbase + NHEADS * HD;
    const T* vbase = qbase + 2 * NHEADS * HD;
    const T* dobase = dout + (int64_t)b * N * OUT_LD + head * HD;

    const int ntiles = (N + 63) / 64;
    auto tile_dma = [&](int kt) {
        char* kb = smem + (kt & 1) * 2 * TILE128;
        dma_rows128(kb, kbase, QKV_LD, kt * 64, 2 * wave, 2 * wave + 2, 1, N, lane);
        dma_rows128(kb + TILE128, vbase, QKV_LD, kt * 64, 2 * wave, 2 * wave + 2, 1, N, lane);
    };
    tile_dma(0);
    chunk16 qf[C::STEPS], dof[C::STEPS];
    row_frags_load<T>(qf, qbase, QKV_LD, q, N, h);
    row_frags_load<T>(dof, dobase, OUT_LD, q, N, h);
    const bool q_ok = q < N;
    const bool wave_active = blk.rb * 128 + wave * 32 < N;   // wave-uniform
    const int qc = q_ok ? q : N - 1;
    float lse_q = lse[((int64_t)b * NHEADS + head) * N + qc] * LOG2E;
    float dl_q = delta[((int64_t)b * NHEADS + head) * N + qc];
#pragma unroll
    for (int s = 0; s < C::STEPS; ++s) { pin_loaded(qf[s]); pin_loaded(dof[s]); }     // (see pin_loaded: no counted vmcnt waits inside the loop)
    pin_loaded(lse_q);
    pin_loaded(dl_q);
    const f32x2_t nlse = {-lse_q, -lse_q}, dlv = {dl_q, dl_q}, c2v = {sc_c2, sc_c2};
    f32x16_t dq[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[db][r] = 0.0f;
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < ntiles; ++kt) {
        const char* k_lds = smem + (kt & 1) * 2 * TILE128;
        const char* v_lds = k_lds + TILE128;
        if (kt + 1 < ntiles) tile_dma(kt + 1);
        if (wave_active) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x16_t sc, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sc[r] = 0.0f; dp[r] = 0.0f; }
                mma_rows_swz(sc, k_lds, kb * 32, lane, qf);      // S^T[key][q]
                mma_rows_swz(dp, v_lds, kb * 32, lane, dof);     // dP^T[key][q]
                if (kt == ntiles - 1 && (N & 63) != 0) {         // padded keys: P = 0, hence dS = 0
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kt * 64 + kb * 32 + frag_row(r, lane) >= N) sc[r] = NEG_BIG;
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2_t sv = {sc[r], sc[r + 1]}, dpv = {dp[r], dp[r + 1]};
                    const f32x2_t ev = __builtin_elementwise_fma(sv, c2v, nlse);
                    const f32x2_t pv = {fast_exp2<T>(ev[0]), fast_exp2<T>(ev[1])};
                    const f32x2_t ds = pv * (dpv - dlv);
                    dp[r] = ds[0]; dp[r + 1] = ds[1];            // dS^T (unscaled)
                }
                mma_transposed_swz(dq, k_lds, kb * 32, lane, dp);    // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0070);
        __builtin_amdgcn_s_barrier();
    }
    store_dT_ok<T>(dq, dqkv + ((int64_t)b * N + (q_ok ? q : 0)) * QKV_LD + head * HD, lane, sc_dq, q_ok);
}

// MAEST_ATTN_PROF: timing instrumentation only (scratch/attn_prof.py builds a second library with it; never defined in
// the product build): shader-clock stamps of every wave of the workgroups with blockIdx % 256 == 5, per query tile.
#ifdef MAEST_ATTN_PROF
__device__ unsigned long long* g_attn_prof = nullptr;
#define PROF_DECL() unsigned long long pst[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PROF_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); pst[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PROF_FLUSH(t) do { if (g_attn_prof != nullptr && (blockIdx.x & 255) == 5 && lane == 0) { \
        unsigned long long* d_ = g_attn_prof + ((((blockIdx.x >> 8) * FB_MAXW + wave) * 16 + (t)) * 8); \
        for (int k_ = 0; k_ < 8; ++k_) d_[k_] = pst[k_]; } } while (0)
#define PROF_FLUSH3(g) do { if (g_attn_prof != nullptr && blockIdx.x == 5 && lane == 0 && (g) < 128) { \
        unsigned long long* d_ = g_attn_prof + ((wave * 128 + (g)) * 8); \
        for (int k_ = 0; k_ < 8; ++k_) d_[k_] = pst[k_]; } } while (0)
#else
#define PROF_DECL() ((void)0)
#define PROF_STAMP(k) ((void)0)
#define PROF_FLUSH(t) ((void)0)
#define PROF_FLUSH3(g) ((void)0)
#endif
__global__ __launch_bounds__(FB_MAXW * 64) void attn_bwd_fused2_kernel(const bf16_t* __restrict__ qkv,
                                                                        const bf16_t* __restrict__ dout,
                                                                        const float* __restrict__ lse,
                                                                        const float* __restrict__ delta,
                                                                        bf16_t* __restrict__ dqkv, int B, int N,
                                                                        float sc_c2, float sc_dq, float sc_dk, int q_rows) {
    using T = bf16_t;
    using C = AttnCfg<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nkw = (N + 31) >> 5;                 // key waves = key blocks
    // query tiles that carry gradient: all of them, or only the first ones (the last block of the network, where only
    // the head's tokens -- queries 0 .. q_rows-1 -- have a non-zero dO; rows of those tiles beyond q_rows hold dO = 0)
    const int nqt = q_rows < N ? (q_rows + 31) >> 5 : nkw;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    // LDS map: K [nkw*32][128 B] | 2 x dS [nkw*32][32 q] | 3 x { Q tile, dO tile, lse[32], delta[32] }
    const int DSBUF = nkw * 32 * FB_DS_PITCH;
    char* k_lds = smem;
    char* ds0 = smem + nkw * 32 * 128;
    char* qbuf0 = ds0 + 2 * DSBUF;

    const int bh = xcd_remap(blockIdx.x, B * NHEADS);
    const int head = bh % NHEADS, b = bh / NHEADS;
    const T* qbase = qkv + (int64_t)b * N * QKV_LD + head * HD;
    const T* kbase = qbase + NHEADS * HD;
    const T* vbase = qbase + 2 * NHEADS * HD;
    const T* dobase = dout + (int64_t)b * N * OUT_LD + head * HD;
    const float* lse_b = lse + ((int64_t)b * NHEADS + head) * N;
    const float* dl_b = delta + ((int64_t)b * NHEADS + head) * N;
    T* dq_out = dqkv + (int64_t)b * N * QKV_LD + head * HD;

    const bool key_wave = wave < nkw;
    const int aux = wave - nkw;                    // 0: Q feeder + dQ[:, 0:32], 1: dO feeder + dQ[:, 32:64]; >= 2: filler
#ifdef MAEST_ATTN_PROF
    const unsigned long long prof_t0 = __builtin_amdgcn_s_memtime();
#endif

    // ---- prologue, all waves: K of every key block -> LDS by DMA
    dma_rows128(k_lds, kbase, QKV_LD, 0, wave, nkw * 4, nwaves, N, lane);

    if (key_wave) {
        // =============================================================================== key waves
        const int key = wave * 32 + (lane & 31);
        const bool key_ok = key < N;
        const f32x2_t c2v = {sc_c2, sc_c2};
        chunk16 kf[C::STEPS], vf[C::STEPS];
        row_frags_load<T>(kf, kbase, QKV_LD, key, N, h);
        row_frags_load<T>(vf, vbase, QKV_LD, key, N, h);
        f32x16_t dk[2], dv[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[db][r] = 0.0f; dv[db][r] = 0.0f; }
        MAEST_ATTN_WAIT_VM0();                             // this wave's share of the K DMA (and its fragments) landed
        __builtin_amdgcn_s_barrier();                      // K in LDS, query tile 0 staged
        int buf = 0;
        PROF_DECL();
        for (int t = 0; t < nqt; ++t) {
            PROF_STAMP(0);
            const char* q_lds = qbuf0 + buf * F2_QBUF;
            const char* do_lds = q_lds + 32 * 128;
            const float* lse_lds = reinterpret_cast<const float*>(q_lds + 2 * 32 * 128);
            const float* dl_lds = lse_lds + 32;
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
            mma_rows_swz(s, q_lds, 0, lane, kf);         // S[q][key]
            mma_rows_swz(dp, do_lds, 0, lane, vf);       // dP[q][key]
            PROF_STAMP(1);
            char* ds_row = ds0 + (t & 1) * DSBUF + key * FB_DS_PITCH;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ql = 8 * g + 4 * h;            // local q of register 4g (4 consecutive rows)
                const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse_lds + ql);
                const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(dl_lds + ql);
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const int r = 4 * g + e;
                    const f32x2_t sv = {s[r], s[r + 1]}, nl = {-l4[e], -l4[e + 1]};
                    const f32x2_t dpv = {dp[r], dp[r + 1]}, dl = {d4[e], d4[e + 1]};
                    const f32x2_t ev = __builtin_elementwise_fma(sv, c2v, nl);
                    const f32x2_t pv = {fast_exp2<T>(ev[0]), fast_exp2<T>(ev[1])};
                    const f32x2_t dsv = pv * (dpv - dl);
                    s[r] = pv[0]; s[r + 1] = pv[1];       // P
                    dp[r] = dsv[0]; dp[r + 1] = dsv[1];   // dS (unscaled)
                }
                chunk8 w;                                 // a padded key contributes nothing to dQ (its K row is not zero here)
                w[0] = key_ok ? pack_bf2(dp[4 * g], dp[4 * g + 1]) : 0u;
                w[1] = key_ok ? pack_bf2(dp[4 * g + 2], dp[4 * g + 3]) : 0u;
                *reinterpret_cast<chunk8*>(ds_row + ql * 2) = w;
            }
            PROF_STAMP(2);
            mma_transposed_swz(dv, do_lds, 0, lane, s);   // dV^T[d][key] += dO^T[d][q] P[q][key]
            mma_transposed_swz(dk, q_lds, 0, lane, dp);   // dK^T[d][key] += Q^T[d][q] dS[q][key]
            PROF_STAMP(3);
            buf = buf == 2 ? 0 : buf + 1;
            __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0): the dS tile is written
            PROF_STAMP(4);
            __builtin_amdgcn_s_barrier();
            PROF_STAMP(5);
            PROF_FLUSH(t);
        }
        __builtin_amdgcn_s_barrier();                      // the aux waves are done with K and the last dS tile
        // dK / dV: registers -> this wave's private LDS patch (row = key, 128 B of d) -> whole rows, 16 B per lane
        char* patch = smem + wave * (2 * 32 * C::PITCH);
#pragma unroll
        for (int tsel = 0; tsel < 2; ++tsel)
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16_t& a = tsel == 0 ? dk[db] : dv[db];
                    const float m = tsel == 0 ? sc_dk : 1.0f;
                    chunk8 w;
                    w[0] = pack_bf2(a[4 * g] * m, a[4 * g + 1] * m);
                    w[1] = pack_bf2(a[4 * g + 2] * m, a[4 * g + 3] * m);
                    *reinterpret_cast<chunk8*>(patch + tsel * 32 * C::PITCH + (lane & 31) * C::PITCH +
                                               (db * 32 + 8 * g + 4 * h) * 2) = w;
                }
        __syncthreads();
#pragma unroll
        for (int tsel = 0; tsel < 2; ++tsel)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = lane + 64 * j, r = i >> 3, c = i & 7;
                const int krow = wave * 32 + r;
                if (krow < N) {
                    const chunk16 v = *reinterpret_cast<const chunk16*>(patch + tsel * 32 * C::PITCH + r * C::PITCH + c * 16);
                    *reinterpret_cast<chunk16*>(dq_out + (uint32_t)(krow * QKV_LD + (1 + tsel) * NHEADS * HD + c * 8)) = v;
                }
            }
    } else {
        // =============================================================================== aux (and filler) waves
        const T* fbase = aux == 0 ? qbase : dobase;        // wave-uniform
        const int fld = aux == 0 ? QKV_LD : OUT_LD;
        const float* sbase = aux == 0 ? lse_b : dl_b;      // per-row statistic this wave carries: lse (scaled) / delta
        auto tile_dma = [&](int t) {                       // this wave's 32 x 128 B tile of query tile t -> ring buffer t % 3
            if (t >= nqt || aux > 1) return;
            char* dst = qbuf0 + (t % 3) * F2_QBUF + (aux == 0 ? 0 : 32 * 128);
            dma_rows128(dst, fbase, fld, t * 32, 0, 4, 1, N, lane);
        };
        auto stat_load = [&](int t) -> float {             // (unconditional, clamped: a conditional load would make hipcc wait for it at the join)
            int row = t * 32 + (lane & 31);
            row = row < N ? row : N - 1;
            return (t < nqt && aux <= 1) ? sbase[row] : 0.0f;
        };
        auto stat_store = [&](float v, int t) {
            if (t >= nqt || aux > 1 || lane >= 32) return;
            const bool live = t * 32 + lane < N;
            float* dstp = reinterpret_cast<float*>(qbuf0 + (t % 3) * F2_QBUF + 2 * 32 * 128) + (aux == 0 ? 0 : 32) + lane;
            if (aux == 0) *dstp = live ? v * LOG2E : -NEG_BIG;   // padded rows: lse = +BIG -> P = 2^(-BIG) = 0
            else *dstp = live ? v : 0.0f;
        };
        // K^T[32 d of this wave][every key] stays in REGISTERS for the whole (batch, head) (80 registers at 10 key blocks: the
        // aux waves have them to spare under the 168 budget), gathered once from the K tile after the first barrier; a dQ
        // job then reads only its dS^T fragments -- two blocks ahead of the MFMAs that consume them.  (The per-tile
        // timeline, scratch/attn_prof.py, showed the aux waves arriving LAST at every barrier: 2200 of their 3650 cycles
        // per tile were this product, 16 transpose reads in front of every 4 MFMAs.)
        chunk16 ktf[FB_MAXW - 2][2];
        auto kt_load = [&]() {
            if (aux > 1) return;
#pragma unroll
            for (int kb = 0; kb < FB_MAXW - 2; ++kb)
                if (kb < nkw) {
                    ktf[kb][0] = frag_from_rows_swz(k_lds, kb * 32, 0, aux, lane);                 // K^T[d][key]
                    ktf[kb][1] = frag_from_rows_swz(k_lds, kb * 32, 1, aux, lane);
                }
        };
        auto dq_compute = [&](f32x16_t& acc, int t) {      // dQ^T[32 d of this wave][32 q of tile t]
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            if (aux > 1) return;
            const char* ds = ds0 + (t & 1) * DSBUF;
            chunk16 bq[3][2];                              // dS^T[key][q] of blocks kb, kb + 1, kb + 2
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                if (kb < nkw) {
                    bq[kb][0] = frag_from_rows_bf16<FB_DS_PITCH>(ds, kb * 32, 0, 0, lane);
                    bq[kb][1] = frag_from_rows_bf16<FB_DS_PITCH>(ds, kb * 32, 1, 0, lane);
                }
#pragma unroll
            for (int kb = 0; kb < FB_MAXW - 2; ++kb)
                if (kb < nkw) {
                    if (kb + 2 < nkw) {
                        bq[(kb + 2) % 3][0] = frag_from_rows_bf16<FB_DS_PITCH>(ds, (kb + 2) * 32, 0, 0, lane);
                        bq[(kb + 2) % 3][1] = frag_from_rows_bf16<FB_DS_PITCH>(ds, (kb + 2) * 32, 1, 0, lane);
                    }
                    mma_chunk<T>(acc, ktf[kb][0], bq[kb % 3][0]);
                    mma_chunk<T>(acc, ktf[kb][1], bq[kb % 3][1]);
                }
        };
        auto dq_store = [&](const f32x16_t& acc, int t) {
            if (t < 0 || aux > 1) return;
            const int q = t * 32 + (lane & 31);
            store_32d_rows16(acc, dq_out + (uint32_t)(q * QKV_LD + aux * 32), lane, sc_dq, q < N);     // (wave-uniform t, aux: converged)
        };
        // prologue: tiles 0 and 1 by DMA; statistics of tile 0 stored, of tile 1 in flight
        tile_dma(0);
        tile_dma(1);
        stat_store(stat_load(0), 0);
        float st_next = stat_load(1);
        f32x16_t dq_prev;                                  // dQ of the previous job, stored one step late (see below)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq_prev[r] = 0.0f;
        MAEST_ATTN_WAIT_VM0();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();                      // K in LDS, query tile 0 staged (tile 1 landed too)
        kt_load();
        PROF_DECL();
        for (int t = 0; t < nqt; ++t) {
            // Everything issued one step ago has landed by now: the DMA of tile t + 1, and the dQ stores of job
            // t - 2, which were issued BEFORE the dQ product of that step -- a store issued right in front of this
            // wait would put its latency on the critical path (vmcnt counts stores too on gfx950).
            PROF_STAMP(0);
            MAEST_ATTN_WAIT_VM0();
            PROF_STAMP(1);
            stat_store(st_next, t + 1);
            tile_dma(t + 2);                               // lands during this step and the next one
            st_next = stat_load(t + 2);
            dq_store(dq_prev, t - 2);
            PROF_STAMP(2);
            if (t > 0) dq_compute(dq_prev, t - 1);
            PROF_STAMP(3);
            __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): statistics of tile t + 1 are in LDS
            PROF_STAMP(4);
            __builtin_amdgcn_s_barrier();
            PROF_STAMP(5);
            PROF_FLUSH(t);
        }
        dq_store(dq_prev, nqt - 2);
        dq_compute(dq_prev, nqt - 1);
        dq_store(dq_prev, nqt - 1);
        if (aux <= 1) {                                    // queries without gradient: dQ = 0
            for (int q = nqt * 32 + (lane & 31); q < N; q += 32) {
                T* row = dq_out + (uint32_t)(q * QKV_LD + aux * 32);
#pragma unroll
                for (int g = 0; g < 4; ++g) store4<T>(row + 8 * g + 4 * h, 0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
        __builtin_amdgcn_s_barrier();                      // LDS may be reused by the key waves' epilogue
        __syncthreads();
    }
#ifdef MAEST_ATTN_PROF
    {
        unsigned long long pst[8] = {prof_t0, __builtin_amdgcn_s_memtime(), __builtin_amdgcn_s_memrealtime(), 0, 0, 0, 0, 0};
        PROF_FLUSH(15);
    }
#endif
}

static int attn_bwd_fused2_smem(int N) {
    const int nkw = (N + 31) / 32;
    return nkw * 32 * 128 + 2 * nkw * 32 * FB_DS_PITCH + 3 * F2_QBUF;
}

// =================================================================================== fused backward, persistent (bf16, 257 <= N <= 320)
// The per-tile timeline of attn_bwd_fused2_kernel (MAEST_ATTN_PROF, scratch/attn_prof.py; B = 256, N = 290) shows a workgroup
// living 58 k cycles of which 12.4 k pass before its first query tile (K, the K / V fragments and two query tiles fetched
// by all 256 workgroups of a round at once: 150 KB per CU at the ~11 B/clk/CU an all-CU burst gets) and 7 k after its last
// one (dK / dV staging and stores): a third of the time the matrix pipe has nothing to do, and HBM idles during the tiles.
// Here a workgroup is PERSISTENT -- one per CU, walking its (batch, head) items -- and everything item i + 1 needs is
// fetched while item i computes:
//   * K and V of the next item arrive by LDS-DMA (both live in LDS now, 2 x 40 KiB; the key waves take their K / V fragments
//     from there at an item's first tile and the aux waves their K^T registers, after which the tiles are free for the next
//     item's prefetch: one 1-KiB piece per key wave and step, steps 1 .. nqt - 1);
//   * the query-tile ring simply runs on across the item boundary (tiles of the next item follow two steps ahead), and so
//     do the statistics and the dQ jobs of the aux waves (job g - 1 computed at step g, stored at step g + 1);
//   * all LDS-DMA is issued by the KEY waves (it was 500 of the aux waves' 3650 cycles per tile, and they arrived last at
//     every barrier): waves 0..3 one Q piece, 4..7 one dO piece per step, retired by a counted vmcnt one step later;
//   * dK / dV leave the registers as 16-byte row pieces (one half-wave exchange per piece) at the first step of the NEXT
//     item -- no LDS staging (LDS is never free here), and issued behind the end-of-step wait so that the counted vmcnt
//     only ever keeps LOADS of the current step in flight (stores may complete out of order with loads).
// Shapes: q_rows == N and 9 <= ceil(N / 32) <= 10 (the prefetch needs nkw * (nkw - 1) >= 8 * nkw piece slots and eight
// key waves for the tile pieces): the 10 s training shapes N = 281, 290.  Everything else keeps attn_bwd_fused2_kernel.
__device__ __forceinline__ void row_frags_lds_swz(chunk16 (&f)[4], const char* tile, int row, int h) {
    const char* rp = tile + row * 128;
    const int fz = swz128(row);
#pragma unroll
    for (int s = 0; s < 4; ++s) f[s] = *reinterpret_cast<const chunk16*>(rp + (((2 * s + h) ^ fz) << 4));
}
__global__ __launch_bounds__(FB_MAXW * 64) void attn_bwd_fused3_kernel(const bf16_t* __restrict__ qkv,
                                                                        const bf16_t* __restrict__ dout,
                                                                        const float* __restrict__ lse,
                                                                        const float* __restrict__ delta,
                                                                        bf16_t* __restrict__ dqkv, int B, int N,
                                                                        float sc_c2, float sc_dq, float sc_dk) {
    using T = bf16_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nkw = (N + 31) >> 5, nqt = nkw;      // key waves = key blocks = query tiles (9 or 10)
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;            // nkw + 2
    // LDS map: K [nkw*32][128 B] | V [nkw*32][128 B] | 2 x dS [nkw*32][32 q] | 3 x { Q tile, dO tile, lse[32], delta[32] }
    const int DSBUF = nkw * 32 * FB_DS_PITCH;
    char* k_lds = smem;
    char* v_lds = smem + nkw * 32 * 128;
    char* ds0 = v_lds + nkw * 32 * 128;
    char* qbuf0 = ds0 + 2 * DSBUF;

    const int items = B * NHEADS, stride = gridDim.x, it0 = blockIdx.x;
    const int nitems = (items - it0 + stride - 1) / stride;
    const int total = nitems * nqt;                // steps (= query tiles) of this workgroup

    auto q_of = [&](int it) { return qkv + (int64_t)(it / NHEADS) * N * QKV_LD + (it % NHEADS) * HD; };
    auto do_of = [&](int it) { return dout + (int64_t)(it / NHEADS) * N * OUT_LD + (it % NHEADS) * HD; };
    auto dq_of = [&](int it) { return dqkv + (int64_t)(it / NHEADS) * N * QKV_LD + (it % NHEADS) * HD; };

    const bool key_wave = wave < nkw;
    const int aux = wave - nkw;                    // 0: lse + dQ[:, 0:32], 1: delta + dQ[:, 32:64]

    // ---- prologue, all waves: K and V of the first item -> LDS by DMA
    {
        const T* kb0 = q_of(it0) + NHEADS * HD;
        dma_rows128(k_lds, kb0, QKV_LD, 0, wave, nkw * 4, nwaves, N, lane);
        dma_rows128(v_lds, kb0 + NHEADS * HD, QKV_LD, 0, wave, nkw * 4, nwaves, N, lane);
    }

    if (key_wave) {
        // =============================================================================== key waves
        const int key = wave * 32 + (lane & 31);
        const bool key_ok = key < N;
        const f32x2_t c2v = {sc_c2, sc_c2};
        // this wave's piece of query tile t of item `it` -> ring slot (waves 0..3: 8 rows of Q, 4..7: 8 rows of dO)
        auto tile_piece = [&](int it, int t, int slot, int lv) {
            const bool isdo = wave >= 4;
            const int j = wave & 3;
            char* dst = qbuf0 + slot * F2_QBUF + (isdo ? 32 * 128 : 0);
            dma_rows128(dst, isdo ? do_of(it) : q_of(it), isdo ? OUT_LD : QKV_LD, t * 32, j, j + 1, 1, N, lv);
        };
        if (wave < 8) { tile_piece(it0, 0, 0, lane); tile_piece(it0, 1, 1, lane); }
        chunk16 kf[4], vf[4];
        f32x16_t dk[2], dv[2];
        MAEST_ATTN_WAIT_VM0();                             // this wave's share of K, V and of tiles 0, 1 landed
        __builtin_amdgcn_s_barrier();
        int it = it0, t = 0, slot = 0;                     // the tile of this step
        int it2 = it0, t2 = 2, slot2 = 2;                  // the tile two steps ahead (its DMA is issued now)
        PROF_DECL();
        for (int g = 0; g < total; ++g) {
            PROF_STAMP(0);
            const bool first = t == 0, last = t == nqt - 1;
            const int itn = it + stride;                   // next item of this workgroup
            // the lane index as the DMA address arithmetic sees it is redefined every step: hoisted out of the loop, the
            // per-lane source offsets are spilled, and the reload's compiler-placed vmcnt(0) drains the pieces in flight
            // (measured: 1900 cycles per step in front of the first MFMA)
            int lv = lane;
#if defined(__AMDGCN__)
            asm volatile("" : "+v"(lv));
#endif
#ifndef MAEST_ABLATE_F3
#define MAEST_ABLATE_F3 0      // timing experiments only (scratch/attn_ablate.sh; results wrong on purpose): bit 0 no dK / dV stores,
#endif                         // 1 no K / V prefetch, 2 no tile pieces, 3 no dQ product, 4 no dQ stores, 5 no softmax math, 6 no dV / dK products
            if (first && g > 0 && !(MAEST_ABLATE_F3 & 1)) {  // dK, dV of the item that ended a step ago
                T* row = dq_of(it - stride) + (uint32_t)(key * QKV_LD + NHEADS * HD);
                store_dT_rows16(dk, row, lane, sc_dk, key_ok);
                store_dT_rows16(dv, row + NHEADS * HD, lane, 1.0f, key_ok);
            }
            // LDS-DMA of this step: the next item's K / V piece first, the tile piece second
            bool pref = false;
            if (!first && itn < items && !(MAEST_ABLATE_F3 & 2)) {
                const int pidx = (t - 1) * nkw + wave;
                if (pidx < 8 * nkw) {
                    const bool isv = pidx >= 4 * nkw;
                    const int j = isv ? pidx - 4 * nkw : pidx;
                    const T* base = q_of(itn) + (isv ? 2 : 1) * NHEADS * HD;
                    dma_rows128(isv ? v_lds : k_lds, base, QKV_LD, 0, j, j + 1, 1, N, lv);
                    pref = true;
                }
            }
            const bool tp = wave < 8 && g + 2 < total && !(MAEST_ABLATE_F3 & 4);
            if (tp) tile_piece(it2, t2, slot2, lv);
            if (first) {
                row_frags_lds_swz(kf, k_lds, key, h);
                row_frags_lds_swz(vf, v_lds, key, h);
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { dk[db][r] = 0.0f; dv[db][r] = 0.0f; }
            }
            const char* q_lds = qbuf0 + slot * F2_QBUF;
            const char* do_lds = q_lds + 32 * 128;
            const float* lse_lds = reinterpret_cast<const float*>(q_lds + 2 * 32 * 128);
            const float* dl_lds = lse_lds + 32;
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
            PROF_STAMP(1);
            mma_rows_swz(s, q_lds, 0, lane, kf);         // S[q][key]
            mma_rows_swz(dp, do_lds, 0, lane, vf);       // dP[q][key]
            PROF_STAMP(2);
            char* ds_row = ds0 + (g & 1) * DSBUF + key * FB_DS_PITCH;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int ql = 8 * gq + 4 * h;           // local q of register 4 gq (4 consecutive rows)
                const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse_lds + ql);
                const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(dl_lds + ql);
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const int r = 4 * gq + e;
                    const f32x2_t sv = {s[r], s[r + 1]}, nl = {-l4[e], -l4[e + 1]};
                    const f32x2_t dpv = {dp[r], dp[r + 1]}, dl = {d4[e], d4[e + 1]};
                    const f32x2_t ev = __builtin_elementwise_fma(sv, c2v, nl);
                    const f32x2_t pv = (MAEST_ABLATE_F3 & 32) ? ev : f32x2_t{fast_exp2<T>(ev[0]), fast_exp2<T>(ev[1])};
                    const f32x2_t dsv = (MAEST_ABLATE_F3 & 32) ? dpv : pv * (dpv - dl);
                    s[r] = pv[0]; s[r + 1] = pv[1];       // P
                    dp[r] = dsv[0]; dp[r + 1] = dsv[1];   // dS (unscaled)
                }
                chunk8 w;                                 // a padded key contributes nothing to dQ
                w[0] = key_ok ? pack_bf2(dp[4 * gq], dp[4 * gq + 1]) : 0u;
                w[1] = key_ok ? pack_bf2(dp[4 * gq + 2], dp[4 * gq + 3]) : 0u;
                *reinterpret_cast<chunk8*>(ds_row + ql * 2) = w;
            }
            PROF_STAMP(3);
            if (!(MAEST_ABLATE_F3 & 64)) {
            mma_transposed_swz(dv, do_lds, 0, lane, s);   // dV^T[d][key] += dO^T[d][q] P[q][key]
            mma_transposed_swz(dk, q_lds, 0, lane, dp);   // dK^T[d][key] += Q^T[d][q] dS[q][key]
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { dv[0][r] += s[r]; dk[0][r] += dp[r]; }
            }
            PROF_STAMP(4);
            // end of step: the pieces issued a step ago have landed (only this step's may still fly); at an item's last
            // step its K / V prefetch piece too (the next step reads K and V)
            const int keep = (tp ? 1 : 0) + ((pref && !last) ? 1 : 0);
            if (keep == 0) __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0)
            else if (keep == 1) __builtin_amdgcn_s_waitcnt(0x0071);     // vmcnt(1) lgkmcnt(0)
            else __builtin_amdgcn_s_waitcnt(0x0072);                    // vmcnt(2) lgkmcnt(0)
            PROF_STAMP(5);
            __builtin_amdgcn_s_barrier();
            PROF_STAMP(6);
            PROF_FLUSH3(g);
            slot = slot == 2 ? 0 : slot + 1;
            slot2 = slot2 == 2 ? 0 : slot2 + 1;
            if (++t == nqt) { t = 0; it = itn; }
            if (++t2 == nqt) { t2 = 0; it2 += stride; }
        }
        {                                                  // dK, dV of the last item
            T* row = dq_of(it - stride) + (uint32_t)(key * QKV_LD + NHEADS * HD);
            store_dT_rows16(dk, row, lane, sc_dk, key_ok);
            store_dT_rows16(dv, row + NHEADS * HD, lane, 1.0f, key_ok);
        }
    } else {
        // =============================================================================== aux waves
#ifndef MAEST_F3_AUX_PRIO
#define MAEST_F3_AUX_PRIO 0
#endif
        if (MAEST_F3_AUX_PRIO > 0) __builtin_amdgcn_s_setprio(MAEST_F3_AUX_PRIO);   // (the youngest waves of the workgroup lose every arbitration)
        const float* sbase = aux == 0 ? lse : delta;       // per-row statistic this wave carries: lse (scaled) / delta
        auto stat_load = [&](int it, int t) -> float {     // (unconditional, clamped)
            int row = t * 32 + (lane & 31);
            row = row < N ? row : N - 1;
            return sbase[(int64_t)it * N + row];
        };
        auto stat_store = [&](float v, int t, int slot) {
            if (lane >= 32) return;
            const bool live = t * 32 + lane < N;
            float* dstp = reinterpret_cast<float*>(qbuf0 + slot * F2_QBUF + 2 * 32 * 128) + (aux == 0 ? 0 : 32) + lane;
            if (aux == 0) *dstp = live ? v * LOG2E : -NEG_BIG;   // padded rows: lse = +BIG -> P = 2^(-BIG) = 0
            else *dstp = live ? v : 0.0f;
        };
        chunk16 ktf[FB_MAXW - 2][2];                       // K^T[32 d of this wave][every key] of the current item
        auto kt_load = [&]() {
#pragma unroll
            for (int kb = 0; kb < FB_MAXW - 2; ++kb)
                if (kb < nkw) {
                    ktf[kb][0] = frag_from_rows_swz(k_lds, kb * 32, 0, aux, lane);
                    ktf[kb][1] = frag_from_rows_swz(k_lds, kb * 32, 1, aux, lane);
                }
        };
        auto dq_compute = [&](f32x16_t& acc, int gj) {     // dQ^T[32 d of this wave][32 q] of the tile of step gj
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const char* ds = ds0 + (gj & 1) * DSBUF;
            chunk16 bq[3][2];                              // dS^T[key][q] of blocks kb, kb + 1, kb + 2
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                bq[kb][0] = frag_from_rows_bf16<FB_DS_PITCH>(ds, kb * 32, 0, 0, lane);
                bq[kb][1] = frag_from_rows_bf16<FB_DS_PITCH>(ds, kb * 32, 1, 0, lane);
            }
#pragma unroll
            for (int kb = 0; kb < FB_MAXW - 2; ++kb)
                if (kb < nkw) {
                    if (kb + 2 < nkw) {
                        bq[(kb + 2) % 3][0] = frag_from_rows_bf16<FB_DS_PITCH>(ds, (kb + 2) * 32, 0, 0, lane);
                        bq[(kb + 2) % 3][1] = frag_from_rows_bf16<FB_DS_PITCH>(ds, (kb + 2) * 32, 1, 0, lane);
                    }
                    mma_chunk<T>(acc, ktf[kb][0], bq[kb % 3][0]);
                    mma_chunk<T>(acc, ktf[kb][1], bq[kb % 3][1]);
                }
        };
        auto dq_store = [&](const f32x16_t& acc, T* out_item, int t) {
            const int q = t * 32 + (lane & 31);
            store_32d_rows16(acc, out_item + (uint32_t)(q * QKV_LD + aux * 32), lane, sc_dq, q < N);
        };
        // prologue: statistics of tile 0 stored, of tile 1 in flight
        stat_store(stat_load(it0, 0), 0, 0);
        float st_next = stat_load(it0, 1);
        f32x16_t dq_prev;
#pragma unroll
        for (int r = 0; r < 16; ++r) dq_prev[r] = 0.0f;
        MAEST_ATTN_WAIT_VM0();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();                      // K, V in LDS, query tiles 0 and 1 staged
        kt_load();
        int it = it0, t = 0;                               // the tile of this step
        int t1 = 1, slot1 = 1;                             // the next tile (its statistics are stored now)
        int it2 = it0, t2 = 2;                             // two ahead (its statistic is loaded now)
        T* out1 = nullptr; int tj1 = 0;                    // job g - 1: computed now
        T* out2 = nullptr; int tj2 = 0;                    // job g - 2: stored now
        PROF_DECL();
        for (int g = 0; g < total; ++g) {
            // everything issued a step ago has landed: the statistic of tile g + 1 and the dQ stores of job g - 3
            PROF_STAMP(0);
            MAEST_ATTN_WAIT_VM0();
            PROF_STAMP(1);
            if (g + 1 < total) stat_store(st_next, t1, slot1);
            if (g + 2 < total) st_next = stat_load(it2, t2);
            if (out2 != nullptr && !(MAEST_ABLATE_F3 & 16)) dq_store(dq_prev, out2, tj2);
            PROF_STAMP(2);
            if (out1 != nullptr && !(MAEST_ABLATE_F3 & 8)) dq_compute(dq_prev, g - 1);
            PROF_STAMP(3);
            if (t == 0 && g > 0) kt_load();                // the item that starts now (its K arrived during the previous one)
            PROF_STAMP(4);
            __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): statistics of tile g + 1 are in LDS
            PROF_STAMP(5);
            __builtin_amdgcn_s_barrier();
            PROF_STAMP(6);
            PROF_FLUSH3(g);
            out2 = out1; tj2 = tj1;
            out1 = dq_of(it); tj1 = t;
            slot1 = slot1 == 2 ? 0 : slot1 + 1;
            if (++t == nqt) { t = 0; it += stride; }
            if (++t1 == nqt) t1 = 0;
            if (++t2 == nqt) { t2 = 0; it2 += stride; }
        }
        if (out2 != nullptr) dq_store(dq_prev, out2, tj2);
        dq_compute(dq_prev, total - 1);
        dq_store(dq_prev, out1, tj1);
    }
}

static int attn_bwd_fused3_smem(int N) {
    const int nkw = (N + 31) / 32;
    return 2 * nkw * 32 * 128 + 2 * nkw * 32 * FB_DS_PITCH + 3 * F2_QBUF;
}

// Waves (= 32-query blocks) per workgroup of attn_fwd_dma_kernel.  MAEST_OPT_ATTN_FWD_WAVES forces 4 / 5 / 6 / 8; 0 = by shape.
static int attn_fwd_waves(int N, int q_rows) {
    const int forced = option(MAEST_OPT_ATTN_FWD_WAVES);
    if (forced == 4 || forced == 5 || forced == 6 || forced == 8) return forced;
    (void)N; (void)q_rows;
    return 4;                                     // measured best at every production shape (see attn_fwd_dma_kernel)
}

int attn_fwd_pw_launch(const void* qkv, void* out, float* lse, int B, int N, AttnScale sc, bool q_prescaled, hipStream_t st);   // attn_fwd_pw.hip
bool attn_fwd_pw_available();   // false in a build whose register audit failed (maest_amd/build.py)

template <typename T, bool X3 = false>
static int attn_fwd_launch(const void* qkv, void* out, float* lse, int B, int N, AttnScale sc, bool q_prescaled, int q_rows, hipStream_t st,
                           bool out_a3 = false) {
    using C = AttnCfg<T>;
    if constexpr (sizeof(T) == 2 && !X3) {
        const int afw = option(MAEST_OPT_ATTN_FWD);
        // the long token rows (10 s inference, 30 s shapes): the persistent one-wave-per-SIMD form; 3 forces it at any N (tests)
        if (q_rows == N && ((afw == 0 && N > 320) || afw == 3) && attn_fwd_pw_available()) return attn_fwd_pw_launch(qkv, out, lse, B, N, sc, q_prescaled, st);
        if (afw == 0 || afw == 2 || afw == 3) {     // K / V tiles by LDS-DMA (unpadded, swizzled); 2 forces this form at any N
            const int nw = attn_fwd_waves(N, q_rows);
            const dim3 g(((N + nw * 32 - 1) / (nw * 32)) * NHEADS * B);
            const int lds = MAEST_FWD_RING * 2 * 64 * 128;
#define MAEST_FWD_LAUNCH(NW_) hipLaunchKernelGGL(attn_fwd_dma_kernel<NW_>, g, dim3(NW_ * 64), lds, st, (const bf16_t*)qkv, \
                                                 (bf16_t*)out, lse, B, N, sc.c2, q_rows)
            if (nw == 5) MAEST_FWD_LAUNCH(5);
            else if (nw == 6) MAEST_FWD_LAUNCH(6);
            else if (nw == 8) MAEST_FWD_LAUNCH(8);
            else MAEST_FWD_LAUNCH(4);
#undef MAEST_FWD_LAUNCH
            return check_launch("maest_attn_fwd(dma)");
        }
    }
    dim3 grid(((N + 127) / 128) * NHEADS * B);
    if constexpr (X3) {
        // split-bf16, complete passes: the kernel on pre-split tiles (MAEST_OPT_ATTN_FWD = 1 keeps the per-use split form: A/B, tests)
        if (q_rows == N && option(MAEST_OPT_ATTN_FWD) != 1) {
            static DeviceOnce once_x;
            ensure_dynamic_lds(once_x, &attn_fwd_x3s_kernel, 8 * 64 * 128);
            hipLaunchKernelGGL(attn_fwd_x3s_kernel, dim3(((N + 127) / 128) * NHEADS * B), dim3(256), 8 * 64 * 128, st, (const float*)qkv, out, lse,
                               B, N, sc.c2, out_a3 ? 1 : 0);
            return check_launch("maest_attn_fwd(x3, split tiles)");
        }
    }
    const int smem_bytes = 4 * C::TILE;
    static DeviceOnce once;
    ensure_dynamic_lds(once, &attn_fwd_kernel<T, X3>, smem_bytes);
    hipLaunchKernelGGL((attn_fwd_kernel<T, X3>), grid, dim3(256), smem_bytes, st, (const T*)qkv, (T*)out, lse, B, N, sc.c2,
                       q_rows, out_a3 ? 1 : 0);
    return check_launch("maest_attn_fwd");
}

template <typename T, bool X3 = false>
static int attn_bwd_launch(const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                           void* dqkv, int B, int N, AttnScale sc, int q_rows, hipStream_t st) {
    using C = AttnCfg<T>;
    if constexpr (sizeof(T) == 2) {
        // bf16 and at most 10 key blocks (the 10 s training shapes, N = 281 / 290): one fused pass per (batch, head)
        const int nkw = (N + 31) / 32;
        const int abw = option(MAEST_OPT_ATTN_BWD);
        if (nkw + 2 <= FB_MAXW && (abw == 0 || abw == 3)) {     // DMA-fed forms (+ the delta kernel)
            const int64_t items = (int64_t)B * N * NHEADS * 4;
            if (out != nullptr)      // (NULL: the caller filled `delta` already, maest_gemm_nt_rowdot)
                hipLaunchKernelGGL(attn_delta_kernel<T>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st,
                                   (const T*)out, (const T*)dout, delta, B, N, q_rows);
            if (abw == 0 && q_rows == N && nkw >= 9) {          // persistent form: one workgroup per CU walks the (batch, head) items
                static DeviceOnce once_p;
                ensure_dynamic_lds(once_p, &attn_bwd_fused3_kernel, attn_bwd_fused3_smem(32 * (FB_MAXW - 2)));
                const int ncu = 256, nitems = B * NHEADS;
                hipLaunchKernelGGL(attn_bwd_fused3_kernel, dim3(nitems < ncu ? nitems : ncu), dim3((nkw + 2) * 64),
                                   attn_bwd_fused3_smem(N), st, (const bf16_t*)qkv, (const bf16_t*)dout, lse,
                                   (const float*)delta, (bf16_t*)dqkv, B, N, sc.c2, sc.dq, sc.dk);
                return check_launch("maest_attn_bwd(fused, persistent)");
            }
            static DeviceOnce once_g;
            ensure_dynamic_lds(once_g, &attn_bwd_fused2_kernel, attn_bwd_fused2_smem(32 * (FB_MAXW - 2)));
            const int waves = nkw + 2 < 8 ? 8 : nkw + 2;
            hipLaunchKernelGGL(attn_bwd_fused2_kernel, dim3(B * NHEADS), dim3(waves * 64), attn_bwd_fused2_smem(N), st,
                               (const bf16_t*)qkv, (const bf16_t*)dout, lse, (const float*)delta, (bf16_t*)dqkv, B, N,
                               sc.c2, sc.dq, sc.dk, q_rows);
            return check_launch("maest_attn_bwd(fused, dma)");
        }
    }
    if (q_rows < N) {
        set_error("maest_attn_bwd_rows: q_rows = %d < N = %d is served by the fused bf16 kernel only (N <= %d, "
                  "MAEST_OPT_ATTN_BWD = 0 or 3)", q_rows, N, 32 * (FB_MAXW - 2));
        return MAEST_ERR_INVALID;
    }
    if constexpr (sizeof(T) == 2 && !X3) {
        if (option(MAEST_OPT_ATTN_BWD) != 4) {                            // DMA-fed tiles (4 = register-staged padded tiles: A/B, tests)
            constexpr int smem_da = 2 * (2 * 64 * 128 + 512), smem_db = 4 * 64 * 128;
            static DeviceOnce once_da, once_db;
            ensure_dynamic_lds(once_da, &attn_bwd_dkdv_dma_kernel, smem_da);
            ensure_dynamic_lds(once_db, &attn_bwd_dq_dma_kernel, smem_db);
            const int64_t items = (int64_t)B * N * NHEADS * 4;
            if (out != nullptr)
                hipLaunchKernelGGL(attn_delta_kernel<T>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st,
                                   (const T*)out, (const T*)dout, delta, B, N, N);
            dim3 grid(((N + 127) / 128) * NHEADS * B);
            hipLaunchKernelGGL(attn_bwd_dkdv_dma_kernel, grid, dim3(256), smem_da, st, (const bf16_t*)qkv, (const bf16_t*)dout,
                               lse, (const float*)delta, (bf16_t*)dqkv, B, N, sc.c2, sc.dk);
            hipLaunchKernelGGL(attn_bwd_dq_dma_kernel, grid, dim3(256), smem_db, st, (const bf16_t*)qkv, (const bf16_t*)dout,
                               lse, (const float*)delta, (bf16_t*)dqkv, B, N, sc.c2, sc.dq);
            return check_launch("maest_attn_bwd(dma tiles)");
        }
    }
    const int smem_a = 2 * (2 * C::TILE + 512);
    const int smem_b = 4 * C::TILE;
    static DeviceOnce once_a, once_b;
    ensure_dynamic_lds(once_a, &attn_bwd_dkdv_kernel<T, X3>, smem_a);
    ensure_dynamic_lds(once_b, &attn_bwd_dq_kernel<T, X3>, smem_b);
    const int64_t items = (int64_t)B * N * NHEADS * 4;
    if (out != nullptr)
        hipLaunchKernelGGL(attn_delta_kernel<T>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st,
                           (const T*)out, (const T*)dout, delta, B, N, N);
    dim3 grid(((N + 127) / 128) * NHEADS * B);
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<T, X3>), grid, dim3(256), smem_a, st, (const T*)qkv, (const T*)dout, lse,
                       (const float*)delta, (T*)dqkv, B, N, sc.c2, sc.dk);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T, X3>), grid, dim3(256), smem_b, st, (const T*)qkv, (const T*)dout, lse,
                       (const float*)delta, (T*)dqkv, B, N, sc.c2, sc.dq);
    return check_launch("maest_attn_bwd");
}

}  // namespace maest

using namespace maest;

#ifdef MAEST_ATTN_PROF
extern "C" int maest_debug_attn_prof(void* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_prof), &p, sizeof(p));
}
#endif

extern "C" int maest_attn_fwd_rows(const void* qkv, void* out, float* lse, int B, int N, int dtype, float scale,
                                   int q_rows, void* stream) {
    MAEST_REQUIRE(qkv && out, "maest_attn_fwd: null pointer");
    MAEST_REQUIRE(B > 0 && N > 0, "maest_attn_fwd: bad shape B=%d N=%d", B, N);
    MAEST_REQUIRE(q_rows > 0 && q_rows <= N, "maest_attn_fwd_rows: q_rows = %d outside 1..N", q_rows);
    MAEST_REQUIRE(dtype == MAEST_F32 || dtype == MAEST_BF16 || dtype == MAEST_F32X3 || dtype == MAEST_BF16_QS || dtype == MAEST_F32X3_A3,
                  "maest_attn_fwd: bad dtype %d", dtype);
    MAEST_REQUIRE(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 16) == 0, "maest_attn_fwd: 16-byte alignment");
    const bool qs = dtype == MAEST_BF16_QS;
    const AttnScale sc = attn_scale(scale, qs);
    if (dtype == MAEST_F32X3 || dtype == MAEST_F32X3_A3)
        return attn_fwd_launch<float, true>(qkv, out, lse, B, N, sc, false, q_rows, (hipStream_t)stream, dtype == MAEST_F32X3_A3);
    return dtype == MAEST_F32 ? attn_fwd_launch<float>(qkv, out, lse, B, N, sc, false, q_rows, (hipStream_t)stream)
                              : attn_fwd_launch<bf16_t>(qkv, out, lse, B, N, sc, qs, q_rows, (hipStream_t)stream);
}

extern "C" int maest_attn_fwd(const void* qkv, void* out, float* lse, int B, int N, int dtype, float scale,
                              void* stream) {
    return maest_attn_fwd_rows(qkv, out, lse, B, N, dtype, scale, N, stream);
}

extern "C" int maest_attn_bwd_rows(const void* qkv, const void* out, const void* dout, const float* lse,
                                   float* delta, void* dqkv, int B, int N, int dtype, float scale, int q_rows,
                                   void* stream) {
    MAEST_REQUIRE(qkv && dout && lse && delta && dqkv, "maest_attn_bwd: null pointer");
    MAEST_REQUIRE(B > 0 && N > 0, "maest_attn_bwd: bad shape B=%d N=%d", B, N);
    MAEST_REQUIRE(q_rows > 0 && q_rows <= N, "maest_attn_bwd_rows: q_rows = %d outside 1..N", q_rows);
    MAEST_REQUIRE(dtype == MAEST_F32 || dtype == MAEST_BF16 || dtype == MAEST_F32X3 || dtype == MAEST_BF16_QS,
                  "maest_attn_bwd: bad dtype %d", dtype);
    const AttnScale sc = attn_scale(scale, dtype == MAEST_BF16_QS);
    if (dtype == MAEST_F32X3)
        return attn_bwd_launch<float, true>(qkv, out, dout, lse, delta, dqkv, B, N, sc, q_rows, (hipStream_t)stream);
    return dtype == MAEST_F32
               ? attn_bwd_launch<float>(qkv, out, dout, lse, delta, dqkv, B, N, sc, q_rows, (hipStream_t)stream)
               : attn_bwd_launch<bf16_t>(qkv, out, dout, lse, delta, dqkv, B, N, sc, q_rows, (hipStream_t)stream);
}

extern "C" int maest_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                              float* delta, void* dqkv, int B, int N, int dtype, float scale, void* stream) {
    return maest_attn_bwd_rows(qkv, out, dout, lse, delta, dqkv, B, N, dtype, scale, N, stream);
}
