// LayerNorm forward / backward and the cls/dist head pooling for the MAEST ViT
// (reference: nn.LayerNorm call sites models/maest.py:395,405 (eps 1e-6, :499), :553 final norm,
//  :571 head norm (eps 1e-5); token pick :806-810; feature mean :905-906; early exit :825-829).
//
// HBM-bound row kernels: one wave64 per 768-wide row, 12 contiguous-by-4 elements per lane
// (3 x 16-byte loads per lane, fully coalesced), statistics by wave-wide butterfly reductions,
// fp32 math; the normalised row is emitted directly in the GEMM operand dtype (bf16 / fp32).
#include "common.h"

namespace maest {

constexpr int LN_COLS = 768;
constexpr int LN_VEC = 3;  // float4 per lane

__device__ __forceinline__ void ln_load_row(const float* row, int lane, float (&v)[12]) {
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(row + i * 256 + lane * 4);
        v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    }
}
__device__ __forceinline__ void ln_stats(const float (&v)[12], float eps, float& mu, float& rs) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += v[i];
    mu = wave_sum(s) * (1.0f / LN_COLS);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < 12; ++i) { const float d = v[i] - mu; q += d * d; }
    const float var = wave_sum(q) * (1.0f / LN_COLS);
    rs = 1.0f / sqrtf(var + eps);
}
__device__ __forceinline__ void store_row4(void* y, int dtype, int64_t off, float a, float b, float c, float d) {
    if (dtype == MAEST_SPLIT3_A) {        // [ hi | hi | lo ] thirds of a 3 x 768 bf16 row (`off` = row * ldy + column)
        uint32_t h0, l0, h1, l1;
        split_bf2(a, b, h0, l0);
        split_bf2(c, d, h1, l1);
        const chunk8 hi = {h0, h1}, lo = {l0, l1};
        bf16_t* yp = reinterpret_cast<bf16_t*>(y) + off;
        *reinterpret_cast<chunk8*>(yp) = hi;
        *reinterpret_cast<chunk8*>(yp + LN_COLS) = hi;
        *reinterpret_cast<chunk8*>(yp + 2 * LN_COLS) = lo;
    } else if (dtype == MAEST_BF16) {
        chunk8 o;
        o[0] = pack_bf2(a, b);
        o[1] = pack_bf2(c, d);
        *reinterpret_cast<chunk8*>(reinterpret_cast<bf16_t*>(y) + off) = o;
    } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + off) = make_float4(a, b, c, d);
    }
}
__device__ __forceinline__ void load_row4(const void* y, int dtype, int64_t off, float (&o)[4]) {
    if (dtype == MAEST_BF16) {
        const chunk8 t = *reinterpret_cast<const chunk8*>(reinterpret_cast<const bf16_t*>(y) + off);
        o[0] = bf2f((bf16_t)(t[0] & 0xffffu)); o[1] = bf2f((bf16_t)(t[0] >> 16));
        o[2] = bf2f((bf16_t)(t[1] & 0xffffu)); o[3] = bf2f((bf16_t)(t[1] >> 16));
    } else {
        const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(y) + off);
        o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
    }
}

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, void* __restrict__ y,
                                                            int64_t ldy, int y_dtype, float* __restrict__ mean,
                                                            float* __restrict__ rstd, int rows, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;  // wave-uniform
    float v[12];
    ln_load_row(x + (int64_t)row * ldx, lane, v);
    float mu, rs;
    ln_stats(v, eps, mu, rs);
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) {
        const int c = i * 256 + lane * 4;
        const float4 g = *reinterpret_cast<const float4*>(gamma + c);
        const float4 bt = *reinterpret_cast<const float4*>(beta + c);
        store_row4(y, y_dtype, (int64_t)row * ldy + c, (v[4 * i] - mu) * rs * g.x + bt.x,
                   (v[4 * i + 1] - mu) * rs * g.y + bt.y, (v[4 * i + 2] - mu) * rs * g.z + bt.z,
                   (v[4 * i + 3] - mu) * rs * g.w + bt.w);
    }
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
}

// Residual add fused into the LayerNorm that follows it (x += attn(...) ; LN2(x)  /  x += mlp(...) ; next block's LN1(x),
// models/maest.py:418-419): x_out = x + delta is written once (the new residual stream, also the saved input of this
// LayerNorm's backward) and normalised in the same pass.  The proj / fc2 GEMMs then emit `delta` (bias included) in
// the operand dtype instead of reading and rewriting the fp32 stream in their epilogue: the same bytes move, but in
// this streaming kernel (6 TB/s) rather than in a GEMM epilogue during which the matrix pipes idle.
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(const float* __restrict__ x, const void* __restrict__ delta,
                                                                int delta_dtype, float* __restrict__ x_out,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, void* __restrict__ y,
                                                                int y_dtype, float* __restrict__ mean,
                                                                float* __restrict__ rstd, int rows, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;  // wave-uniform
    float v[12];
    ln_load_row(x + (int64_t)row * LN_COLS, lane, v);
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) {
        const int64_t off = (int64_t)row * LN_COLS + i * 256 + lane * 4;
        float d[4];
        load_row4(delta, delta_dtype, off, d);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * i + e] = d[e] + v[4 * i + e];     // (acc + bias) + x, the epilogue's order
        *reinterpret_cast<float4*>(x_out + off) = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
    }
    float mu, rs;
    ln_stats(v, eps, mu, rs);
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) {
        const int c = i * 256 + lane * 4;
        const float4 g = *reinterpret_cast<const float4*>(gamma + c);
        const float4 bt = *reinterpret_cast<const float4*>(beta + c);
        store_row4(y, y_dtype, (int64_t)row * (y_dtype == MAEST_SPLIT3_A ? 3 * LN_COLS : LN_COLS) + c, (v[4 * i] - mu) * rs * g.x + bt.x,
                   (v[4 * i + 1] - mu) * rs * g.y + bt.y, (v[4 * i + 2] - mu) * rs * g.z + bt.z,
                   (v[4 * i + 3] - mu) * rs * g.w + bt.w);
    }
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma ;  dgamma += dy * xhat ; dbeta += dy
__global__ __launch_bounds__(256, 5) void layernorm_bwd_kernel(const void* __restrict__ dy, int64_t lddy, int dy_dtype,
                                                            const float* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ dres, float* __restrict__ dx_out,
                                                            void* __restrict__ dx_lp, int dx_lp_dtype,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int rows, int n_tok, int n_head) {
    // n_head > 0: `dres` is COMPACT -- [clips][n_head][768], the residual gradient of the first n_head tokens of every
    // clip of n_tok tokens, zero for the others (the last block of the network, whose patch tokens feed nothing)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][2][768]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (<= 96 VGPRs, launch bound 5 waves / SIMD: this HBM-bound kernel then fits beside a workgroup of the 256-tile
    // wgrad GEMM -- 2 x 208 VGPRs per SIMD, 128 KiB of LDS -- that runs on the side stream at the same time; gamma is
    // re-read per row from L1 instead of living in 12 registers)
    float ag[12], ab[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { ag[i] = 0.0f; ab[i] = 0.0f; }
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        float xv[12], dv[12];
        ln_load_row(x + (int64_t)row * ldx, lane, xv);
#pragma unroll
        for (int i = 0; i < LN_VEC; ++i) {
            float t[4];
            load_row4(dy, dy_dtype, (int64_t)row * lddy + i * 256 + lane * 4, t);
            dv[4 * i] = t[0]; dv[4 * i + 1] = t[1]; dv[4 * i + 2] = t[2]; dv[4 * i + 3] = t[3];
        }
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const float xh = (xv[i] - mu) * rs;
            const float g = dv[i] * gamma[(i >> 2) * 256 + lane * 4 + (i & 3)];
            s1 += g;
            s2 += g * xh;
            ag[i] += dv[i] * xh;
            ab[i] += dv[i];
            xv[i] = xh;
            dv[i] = g;
        }
        s1 = wave_sum(s1) * (1.0f / LN_COLS);
        s2 = wave_sum(s2) * (1.0f / LN_COLS);
#pragma unroll
        for (int i = 0; i < LN_VEC; ++i) {
            const int64_t off = (int64_t)row * LN_COLS + i * 256 + lane * 4;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rs * (dv[4 * i + e] - s1 - xv[4 * i + e] * s2);
            if (dres != nullptr) {
                int64_t roff = off;
                bool has = true;
                if (n_head > 0) {
                    const int clip = row / n_tok, tok = row - clip * n_tok;     // wave-uniform
                    has = tok < n_head;
                    roff = ((int64_t)clip * n_head + tok) * LN_COLS + i * 256 + lane * 4;
                }
                if (has) {
                    const float4 r = *reinterpret_cast<const float4*>(dres + roff);
                    o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
                }
            }
            if (dx_out != nullptr) *reinterpret_cast<float4*>(dx_out + off) = make_float4(o[0], o[1], o[2], o[3]);
            if (dx_lp != nullptr) store_row4(dx_lp, dx_lp_dtype, off, o[0], o[1], o[2], o[3]);
        }
    }
    // block reduce of the per-wave dgamma / dbeta partials, then one atomic per column per block
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = i * 256 + lane * 4 + e;
            red[(wave * 2 + 0) * LN_COLS + c] = ag[4 * i + e];
            red[(wave * 2 + 1) * LN_COLS + c] = ab[4 * i + e];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < LN_COLS; c += 256) {
        float sg = 0.0f, sb = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            sg += red[(w * 2 + 0) * LN_COLS + c];
            sb += red[(w * 2 + 1) * LN_COLS + c];
        }
        unsafeAtomicAdd(dgamma + c, sg);
        unsafeAtomicAdd(dbeta + c, sb);
    }
}

// cls/dist rows only: xn = LN(x[b, tok]); feat = (cls + dist) / 2.  One wave per (b, tok).
__global__ __launch_bounds__(256) void head_pool_fwd_kernel(const float* __restrict__ x, int B, int N,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            float* __restrict__ cls, float* __restrict__ dist,
                                                            float* __restrict__ mean, float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wave;  // = b * 2 + tok
    if (item >= B * 2) return;
    const int b = item >> 1, tok = item & 1;
    float v[12];
    ln_load_row(x + ((int64_t)b * N + tok) * LN_COLS, lane, v);
    float mu, rs;
    ln_stats(v, eps, mu, rs);
    float* dst = (tok == 0 ? cls : dist) + (int64_t)b * LN_COLS;
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) {
        const int c = i * 256 + lane * 4;
        const float4 g = *reinterpret_cast<const float4*>(gamma + c);
        const float4 bt = *reinterpret_cast<const float4*>(beta + c);
        *reinterpret_cast<float4*>(dst + c) =
            make_float4((v[4 * i] - mu) * rs * g.x + bt.x, (v[4 * i + 1] - mu) * rs * g.y + bt.y,
                        (v[4 * i + 2] - mu) * rs * g.z + bt.z, (v[4 * i + 3] - mu) * rs * g.w + bt.w);
    }
    if (lane == 0) {
        if (mean) mean[item] = mu;
        if (rstd) rstd[item] = rs;
    }
}
__global__ __launch_bounds__(256) void feat_avg_kernel(const float* __restrict__ cls, const float* __restrict__ dist,
                                                       float* __restrict__ feat, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) feat[i] = (cls[i] + dist[i]) / 2;
}

__global__ __launch_bounds__(256) void head_pool_bwd_kernel(const float* __restrict__ d_cls,
                                                            const float* __restrict__ d_dist,
                                                            const float* __restrict__ d_feat,
                                                            const float* __restrict__ x, int B, int N,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
    // grid-stride over the 2 B rows, a wave per row; dgamma / dbeta partials stay in registers across a wave's rows and
    // leave as ONE atomic per column and workgroup (an atomic per element and row -- 512-way contention on each of the
    // 1536 addresses at B = 256 -- took 99 us)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][2][768]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ag[12], ab[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { ag[i] = 0.0f; ab[i] = 0.0f; }
    for (int item = blockIdx.x * 4 + wave; item < B * 2; item += gridDim.x * 4) {
        const int b = item >> 1, tok = item & 1;
        const float* dsrc = tok == 0 ? d_cls : d_dist;
        float xv[12], dv[12];
        ln_load_row(x + ((int64_t)b * N + tok) * LN_COLS, lane, xv);
        const float mu = mean[item], rs = rstd[item];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < LN_VEC; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = i * 256 + lane * 4 + e;
                float d = 0.0f;
                if (dsrc != nullptr) d += dsrc[(int64_t)b * LN_COLS + c];
                if (d_feat != nullptr) d += 0.5f * d_feat[(int64_t)b * LN_COLS + c];
                const float xh = (xv[4 * i + e] - mu) * rs;
                const float g = d * gamma[c];
                s1 += g;
                s2 += g * xh;
                ag[4 * i + e] += d * xh;
                ab[4 * i + e] += d;
                xv[4 * i + e] = xh;
                dv[4 * i + e] = g;
            }
        s1 = wave_sum(s1) * (1.0f / LN_COLS);
        s2 = wave_sum(s2) * (1.0f / LN_COLS);
#pragma unroll
        for (int i = 0; i < LN_VEC; ++i) {
            const int64_t off = ((int64_t)b * N + tok) * LN_COLS + i * 256 + lane * 4;
            *reinterpret_cast<float4*>(dx + off) =
                make_float4(rs * (dv[4 * i] - s1 - xv[4 * i] * s2), rs * (dv[4 * i + 1] - s1 - xv[4 * i + 1] * s2),
                            rs * (dv[4 * i + 2] - s1 - xv[4 * i + 2] * s2), rs * (dv[4 * i + 3] - s1 - xv[4 * i + 3] * s2));
        }
    }
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = i * 256 + lane * 4 + e;
            red[(wave * 2 + 0) * LN_COLS + c] = ag[4 * i + e];
            red[(wave * 2 + 1) * LN_COLS + c] = ab[4 * i + e];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < LN_COLS; c += 256) {
        float sg = 0.0f, sb = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            sg += red[(w * 2 + 0) * LN_COLS + c];
            sb += red[(w * 2 + 1) * LN_COLS + c];
        }
        unsafeAtomicAdd(dgamma + c, sg);
        unsafeAtomicAdd(dbeta + c, sb);
    }
}

// emb[b] = cat(x[b,0], x[b,1], mean(x[b,2:], 0))
__global__ __launch_bounds__(256) void embed_pool_kernel(const float* __restrict__ x, int N, float* __restrict__ emb) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;  // < 768
    const float* xb = x + (int64_t)b * N * LN_COLS;
    float s = 0.0f;
    for (int n = 2; n < N; ++n) s += xb[(int64_t)n * LN_COLS + c];
    float* e = emb + (int64_t)b * 3 * LN_COLS;
    e[c] = xb[c];
    e[LN_COLS + c] = xb[LN_COLS + c];
    e[2 * LN_COLS + c] = s / (float)(N - 2);
}

}  // namespace maest

using namespace maest;

extern "C" int maest_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y,
                                   int64_t ldy, int y_dtype, float* mean, float* rstd, int rows, int cols,
                                   float eps, void* stream) {
    MAEST_REQUIRE(x && gamma && beta && y, "maest_layernorm_fwd: null pointer");
    MAEST_REQUIRE(cols == LN_COLS, "maest_layernorm_fwd: cols must be 768, got %d", cols);
    MAEST_REQUIRE(rows > 0, "maest_layernorm_fwd: rows=%d", rows);
    MAEST_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, "maest_layernorm_fwd: leading dims must be multiples of 4");
    MAEST_REQUIRE(y_dtype == MAEST_F32 || y_dtype == MAEST_BF16 || y_dtype == MAEST_SPLIT3_A, "maest_layernorm_fwd: bad dtype");
    MAEST_REQUIRE(y_dtype != MAEST_SPLIT3_A || ldy >= 3 * LN_COLS, "maest_layernorm_fwd: MAEST_SPLIT3_A rows are 3 x 768 wide (ldy = %lld)", (long long)ldy);
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma,
                       beta, y, ldy, y_dtype, mean, rstd, rows, eps);
    return check_launch("maest_layernorm_fwd");
}

extern "C" int maest_add_layernorm_fwd(const float* x, const void* delta, int delta_dtype, float* x_out,
                                       const float* gamma, const float* beta, void* y, int y_dtype, float* mean,
                                       float* rstd, int rows, int cols, float eps, void* stream) {
    MAEST_REQUIRE(x && delta && x_out && gamma && beta && y, "maest_add_layernorm_fwd: null pointer");
    MAEST_REQUIRE(cols == LN_COLS, "maest_add_layernorm_fwd: cols must be 768, got %d", cols);
    MAEST_REQUIRE(rows > 0, "maest_add_layernorm_fwd: rows=%d", rows);
    MAEST_REQUIRE((y_dtype == MAEST_F32 || y_dtype == MAEST_BF16 || y_dtype == MAEST_SPLIT3_A) && (delta_dtype == MAEST_F32 || delta_dtype == MAEST_BF16),
                  "maest_add_layernorm_fwd: bad dtype");
    hipLaunchKernelGGL(add_layernorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, delta,
                       delta_dtype, x_out, gamma, beta, y, y_dtype, mean, rstd, rows, eps);
    return check_launch("maest_add_layernorm_fwd");
}

extern "C" int maest_layernorm_bwd_headres(const void* dy, int64_t lddy, int dy_dtype, const float* x, int64_t ldx,
                                           const float* gamma, const float* mean, const float* rstd, const float* dres,
                                           float* dx_out, void* dx_lp, int dx_lp_dtype, float* dgamma, float* dbeta,
                                           int rows, int cols, int n_tok, int n_head, void* stream) {
    MAEST_REQUIRE(dy && x && gamma && mean && rstd && dgamma && dbeta, "maest_layernorm_bwd: null pointer");
    MAEST_REQUIRE(n_head >= 0 && (n_head == 0 || (dres && n_tok >= n_head && rows % n_tok == 0)),
                  "maest_layernorm_bwd_headres: bad token counts n_tok=%d n_head=%d rows=%d", n_tok, n_head, rows);
    MAEST_REQUIRE(cols == LN_COLS, "maest_layernorm_bwd: cols must be 768, got %d", cols);
    MAEST_REQUIRE(rows > 0, "maest_layernorm_bwd: rows=%d", rows);
    MAEST_REQUIRE(lddy % 4 == 0 && ldx % 4 == 0, "maest_layernorm_bwd: leading dims must be multiples of 4");
    int blocks = (rows + 3) / 4;
    const int cap = option(MAEST_OPT_LN_BWD_BLOCKS);     // grid-stride cap: per-block dgamma/dbeta partials vs waves in flight
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks), dim3(256), 4 * 2 * LN_COLS * 4, (hipStream_t)stream, dy,
                       lddy, dy_dtype, x, ldx, gamma, mean, rstd, dres, dx_out, dx_lp, dx_lp_dtype, dgamma, dbeta,
                       rows, n_tok > 0 ? n_tok : 1, n_head);
    return check_launch("maest_layernorm_bwd");
}

extern "C" int maest_layernorm_bwd(const void* dy, int64_t lddy, int dy_dtype, const float* x, int64_t ldx,
                                   const float* gamma, const float* mean, const float* rstd, const float* dres,
                                   float* dx_out, void* dx_lp, int dx_lp_dtype, float* dgamma, float* dbeta,
                                   int rows, int cols, void* stream) {
    return maest_layernorm_bwd_headres(dy, lddy, dy_dtype, x, ldx, gamma, mean, rstd, dres, dx_out, dx_lp, dx_lp_dtype,
                                       dgamma, dbeta, rows, cols, 1, 0, stream);
}

extern "C" int maest_head_pool_fwd(const float* x, int B, int N, const float* gamma, const float* beta, float eps,
                                   float* cls, float* dist, float* feat, float* mean, float* rstd, void* stream) {
    MAEST_REQUIRE(x && gamma && beta && cls && dist && feat, "maest_head_pool_fwd: null pointer");
    MAEST_REQUIRE(B > 0 && N >= 2, "maest_head_pool_fwd: bad shape B=%d N=%d", B, N);
    hipLaunchKernelGGL(head_pool_fwd_kernel, dim3((2 * B + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, B, N, gamma,
                       beta, eps, cls, dist, mean, rstd);
    const int64_t n = (int64_t)B * LN_COLS;
    hipLaunchKernelGGL(feat_avg_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)cls, (const float*)dist, feat, n);
    return check_launch("maest_head_pool_fwd");
}

extern "C" int maest_head_pool_bwd(const float* d_cls, const float* d_dist, const float* d_feat, const float* x,
                                   int B, int N, const float* gamma, const float* mean, const float* rstd, float* dx,
                                   float* dgamma, float* dbeta, void* stream) {
    MAEST_REQUIRE(x && gamma && mean && rstd && dx && dgamma && dbeta, "maest_head_pool_bwd: null pointer");
    MAEST_REQUIRE(B > 0 && N >= 2, "maest_head_pool_bwd: bad shape B=%d N=%d", B, N);
    if (hipMemsetAsync(dx, 0, (size_t)B * N * LN_COLS * sizeof(float), (hipStream_t)stream) != hipSuccess) {
        set_error("maest_head_pool_bwd: hipMemsetAsync failed");
        return MAEST_ERR_LAUNCH;
    }
    const int blocks = (2 * B + 3) / 4 < 32 ? (2 * B + 3) / 4 : 32;
    hipLaunchKernelGGL(head_pool_bwd_kernel, dim3(blocks), dim3(256), 4 * 2 * LN_COLS * 4, (hipStream_t)stream, d_cls, d_dist,
                       d_feat, x, B, N, gamma, mean, rstd, dx, dgamma, dbeta);
    return check_launch("maest_head_pool_bwd");
}

extern "C" int maest_embed_pool(const float* x, int B, int N, float* emb, void* stream) {
    MAEST_REQUIRE(x && emb, "maest_embed_pool: null pointer");
    MAEST_REQUIRE(B > 0 && N > 2, "maest_embed_pool: bad shape B=%d N=%d", B, N);
    hipLaunchKernelGGL(embed_pool_kernel, dim3(LN_COLS / 256, B), dim3(256), 0, (hipStream_t)stream, x, N, emb);
    return check_launch("maest_embed_pool");
}
