// Small HBM-bound helpers around the GEMMs: operand transposes for wgrad, parameter casts,
// bias-gradient column sums, BCE-with-logits (+ fused label mixup), predict_labels tail.
// (reference: F.binary_cross_entropy_with_logits models/module.py:90,299-301; label mixup :84-86;
//  predict_labels sigmoid/mean models/maest.py:937-938.)
#include "common.h"

namespace maest {

// ---- tiled transpose through LDS: dst[c][r] = src[r][c]; dst row pitch ld_dst >= rows, pad zeroed.
// 64x64 element tiles, 256 threads.  LDS tile rows are padded by one 4-byte word.
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, int64_t ld_src,
                                                        T* __restrict__ dst, int64_t ld_dst, int rows, int cols) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* tile = reinterpret_cast<T*>(smem);  // [64][64 + PAD]
    constexpr int PAD = 4 / (int)sizeof(T);
    constexpr int LD = 64 + PAD;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + i * 4 + ty, c = c0 + tx;
        T v = (T)0;
        if (r < rows && c < cols) v = src[(int64_t)r * ld_src + c];
        tile[(i * 4 + ty) * LD + tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + i * 4 + ty, r = r0 + tx;
        if (c < cols && r < ld_dst) dst[(int64_t)c * ld_dst + r] = tile[tx * LD + i * 4 + ty];
    }
}

// ---- fp32 parameters -> operand dtype, plus optional transposed copy (one pass over the source)
template <typename T>
__global__ __launch_bounds__(256) void cast_weights_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                           T* __restrict__ dst_t, int rows, int cols) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile = reinterpret_cast<float*>(smem);  // [64][65]
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + i * 4 + ty, c = c0 + tx;
        float v = 0.0f;
        if (r < rows && c < cols) {
            v = src[(int64_t)r * cols + c];
            if (dst != nullptr) dst[(int64_t)r * cols + c] = elem_traits<T>::from_f32(v);
        }
        tile[(i * 4 + ty) * 65 + tx] = v;
    }
    if (dst_t == nullptr) return;  // uniform
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + i * 4 + ty, r = r0 + tx;
        if (c < cols && r < rows) dst_t[(int64_t)c * rows + r] = elem_traits<T>::from_f32(tile[tx * 65 + i * 4 + ty]);
    }
}

// ---- the same for MANY parameters in one launch: after an optimizer step every weight needs fresh operand
// copies; ~100 tiny launches cost more in launch latency than in bytes.  The item table travels by value.
constexpr int CAST_MAX_ITEMS = 64;
struct CastTable {
    const float* src[CAST_MAX_ITEMS];
    void* dst[CAST_MAX_ITEMS];
    void* dst_t[CAST_MAX_ITEMS];
    int rows[CAST_MAX_ITEMS], cols[CAST_MAX_ITEMS];
    int scaled_rows[CAST_MAX_ITEMS];      // rows [0, scaled_rows) of dst (NOT of dst_t) are multiplied by row_scale before the rounding
    int tile_begin[CAST_MAX_ITEMS + 1];   // prefix sum of 64x64 tiles
    float row_scale;
    int split3b;                          // dst rows are [ hi | lo | hi ] bf16 thirds of 3 x cols (MAEST_SPLIT3_B); dst_t unused
    int n;
};
template <typename T>
__global__ __launch_bounds__(256) void cast_weights_multi_kernel(const CastTable tab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile = reinterpret_cast<float*>(smem);  // [64][65]
    int it = 0;
    while (it + 1 < tab.n && (int)blockIdx.x >= tab.tile_begin[it + 1]) ++it;   // block-uniform scalar search
    const int rows = tab.rows[it], cols = tab.cols[it], srows = tab.scaled_rows[it];
    const float* __restrict__ src = tab.src[it];
    T* __restrict__ dst = reinterpret_cast<T*>(tab.dst[it]);
    T* __restrict__ dst_t = reinterpret_cast<T*>(tab.dst_t[it]);
    const int local = blockIdx.x - tab.tile_begin[it];
    const int tcols = (cols + 63) / 64;
    const int r0 = (local / tcols) * 64, c0 = (local % tcols) * 64;
    if constexpr (sizeof(T) == 2) {
        // Quad path (bf16 copies of matrices whose sides are multiples of 4: every ViT weight): a thread moves four consecutive values --
        // 16-byte loads, 8-byte stores for the plain copy, and, through the LDS tile, 8-byte stores along the rows of the transposed
        // copy (the element-wise path below stores 2 bytes per lane: 255 us for the 48 block matrices against ~140 here).
        if (!tab.split3b && (rows & 3) == 0 && (cols & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 &&
            (dst == nullptr || (reinterpret_cast<uintptr_t>(dst) & 7) == 0) && (dst_t == nullptr || (reinterpret_cast<uintptr_t>(dst_t) & 7) == 0)) {
            const int tq = threadIdx.x & 15, tr = threadIdx.x >> 4;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int lr = tr + 16 * p, r = r0 + lr, c = c0 + 4 * tq;
                float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (r < rows && c < cols) {
                    v = *reinterpret_cast<const float4*>(src + (int64_t)r * cols + c);
                    if (dst != nullptr) {
                        const float m = r < srows ? tab.row_scale : 1.0f;
                        chunk8 o;
                        o[0] = pack_bf2(v.x * m, v.y * m);
                        o[1] = pack_bf2(v.z * m, v.w * m);
                        *reinterpret_cast<chunk8*>(dst + (int64_t)r * cols + c) = o;
                    }
                }
                float* tp = tile + lr * 65 + 4 * tq;       // (pitch 65: the transposed reads below hit 32 distinct banks per half-wave)
                tp[0] = v.x; tp[1] = v.y; tp[2] = v.z; tp[3] = v.w;
            }
            if (dst_t == nullptr) return;  // uniform
            __syncthreads();
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int lc = tr + 16 * p, c = c0 + lc, r = r0 + 4 * tq;
                if (c < cols && r < rows) {
                    chunk8 o;
                    o[0] = pack_bf2(tile[(4 * tq) * 65 + lc], tile[(4 * tq + 1) * 65 + lc]);
                    o[1] = pack_bf2(tile[(4 * tq + 2) * 65 + lc], tile[(4 * tq + 3) * 65 + lc]);
                    *reinterpret_cast<chunk8*>(dst_t + (int64_t)c * rows + r) = o;
                }
            }
            return;
        }
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + i * 4 + ty, c = c0 + tx;
        float v = 0.0f;
        if (r < rows && c < cols) {
            v = src[(int64_t)r * cols + c];
            if constexpr (sizeof(T) == 2) {
                if (tab.split3b) {
                    const bf16_t hi = f2bf(v), lo = f2bf(v - bf2f(hi));
                    bf16_t* d3 = reinterpret_cast<bf16_t*>(tab.dst[it]) + (int64_t)r * 3 * cols + c;
                    d3[0] = hi; d3[cols] = lo; d3[2 * cols] = hi;
                    continue;
                }
            }
            if (dst != nullptr) dst[(int64_t)r * cols + c] = elem_traits<T>::from_f32(r < srows ? v * tab.row_scale : v);
        }
        tile[(i * 4 + ty) * 65 + tx] = v;
    }
    if (dst_t == nullptr) return;  // uniform
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + i * 4 + ty, r = r0 + tx;
        if (c < cols && r < rows) dst_t[(int64_t)c * rows + r] = elem_traits<T>::from_f32(tile[tx * 65 + i * 4 + ty]);
    }
}

// ---- head-token rows of a [clips][n_tok][768] tensor <-> the compact [clips][n_head][768] form (the last block of the
// network runs its proj / MLP only on the tokens the head reads: maest.py, _Engine).  One wave-sized row segment per
// thread group: 768 columns = 96 x 16-byte (fp32: 192 x) chunks; grid-stride over (clip, token).
template <typename T>
__global__ __launch_bounds__(256) void gather_head_rows_kernel(const T* __restrict__ src, int n_tok, int n_head,
                                                               T* __restrict__ dst, int64_t n_chunks) {
    constexpr int CPR = 768 * (int)sizeof(T) / 16;      // 16-byte chunks per row
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n_chunks; c += (int64_t)gridDim.x * 256) {
        const int64_t r = c / CPR;
        const int k = (int)(c - r * CPR);
        const int64_t clip = r / n_head;
        const int tok = (int)(r - clip * n_head);
        const chunk16 v = *reinterpret_cast<const chunk16*>(reinterpret_cast<const char*>(src) +
                                                            ((clip * n_tok + tok) * CPR + k) * 16);
        *reinterpret_cast<chunk16*>(reinterpret_cast<char*>(dst) + c * 16) = v;
    }
}
// dst rows [0, n_pad) of every clip: the n_head compact rows, then zeros (dst rows >= n_pad are left alone)
template <typename T>
__global__ __launch_bounds__(256) void scatter_head_rows_kernel(const T* __restrict__ src, int n_tok, int n_head, int n_pad,
                                                                T* __restrict__ dst, int64_t n_chunks) {
    constexpr int CPR = 768 * (int)sizeof(T) / 16;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n_chunks; c += (int64_t)gridDim.x * 256) {
        const int64_t r = c / CPR;
        const int k = (int)(c - r * CPR);
        const int64_t clip = r / n_pad;
        const int tok = (int)(r - clip * n_pad);
        chunk16 v = {0u, 0u, 0u, 0u};
        if (tok < n_head)
            v = *reinterpret_cast<const chunk16*>(reinterpret_cast<const char*>(src) + ((clip * n_head + tok) * CPR + k) * 16);
        *reinterpret_cast<chunk16*>(reinterpret_cast<char*>(dst) + ((clip * n_tok + tok) * CPR + k) * 16) = v;
    }
}

// ---- out[c] += sum_r src[r][c]; grid (ceil(cols/256), row chunks of 512)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ src, int64_t ld, int rows, int cols,
                                                     float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const int r_begin = blockIdx.y * 512;
    int r_end = r_begin + 512;
    if (r_end > rows) r_end = rows;
    float s = 0.0f;
    for (int r = r_begin; r < r_end; ++r) s += elem_traits<T>::to_f32(src[(int64_t)r * ld + c]);
    unsafeAtomicAdd(out + c, s);
}

// ---- BCE with logits (mean), optional label mixup.  The loss is a sum in a FIXED order, so that the same logits give the same
// loss bit for bit: per-workgroup partials (per-thread strided sums, wave butterflies, four wave partials added by thread 0), then
// one wave adds the partials in index order.  (Rounds 1-3 added the workgroup partials with one atomic each: their order, and with it
// the last bit of the loss, changed from run to run -- seen as a 6e-8 difference between two evaluations of the same batch in
// tests/test_model_gpu.py, once in ~10 runs.)  Training (dlogits given): the partials are parked in the head of the dlogits buffer,
// which the gradient kernel overwrites afterwards -- no workspace, three small launches.  Loss only: one workgroup does it all.
__device__ __forceinline__ float bce_term(const float* __restrict__ z, const float* __restrict__ y, const int32_t* __restrict__ perm,
                                          const float* __restrict__ lam, int cols, int64_t i, float& t_out) {
    const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    float t = y[i];
    if (lam != nullptr) {
        const float l = lam[r];
        t = t * l + y[(int64_t)perm[r] * cols + c] * (1.0f - l);
    }
    t_out = t;
    const float v = z[i];
    // max(z,0) - z*y + log1p(exp(-|z|))   (ATen's numerically stable form)
    return fmaxf(v, 0.0f) - v * t + log1pf(expf(-fabsf(v)));
}
// partial[blockIdx] = sum of this workgroup's terms (grid-strided); with gridDim == 1 and `loss` given, the whole loss.
// Any workgroup size that is a multiple of 64 (the loss-only path takes 1024 threads: the one workgroup that keeps the
// summation order fixed without a scratch buffer then walks a 1024 x 519 validation batch in ~500 terms per thread).
__global__ __launch_bounds__(1024) void bce_partial_kernel(const float* __restrict__ z, const float* __restrict__ y,
                                                           const int32_t* __restrict__ perm, const float* __restrict__ lam,
                                                           int rows, int cols, float weight, float* __restrict__ partial,
                                                           float* __restrict__ loss) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);
    const int64_t total = (int64_t)rows * cols;
    const int nthr = (int)blockDim.x;
    float acc = 0.0f, t;
    for (int64_t i = (int64_t)blockIdx.x * nthr + threadIdx.x; i < total; i += (int64_t)gridDim.x * nthr)
        acc += bce_term(z, y, perm, lam, cols, i, t);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.0f;
        for (int w = 0; w < nthr / 64; ++w) tot += red[w];                             // fixed order
        if (loss != nullptr) unsafeAtomicAdd(loss, weight * tot / (float)total);     // (gridDim == 1: nothing else adds concurrently)
        else partial[blockIdx.x] = tot;
    }
}
// one wave: loss += weight * (partial[0] + partial[1] + ... in a fixed order) / total   (the caller's scalar accumulates the
// terms of a composite loss: launches of one stream are ordered)
__global__ __launch_bounds__(64) void bce_sum_kernel(const float* __restrict__ partial, int n, float scale, float* __restrict__ loss) {
    float acc = 0.0f;
    for (int i = threadIdx.x; i < n; i += 64) acc += partial[i];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) unsafeAtomicAdd(loss, scale * acc);
}
__global__ __launch_bounds__(256) void bce_grad_kernel(const float* __restrict__ z, const float* __restrict__ y,
                                                       const int32_t* __restrict__ perm, const float* __restrict__ lam,
                                                       int rows, int cols, float weight, float* __restrict__ dlogits) {
    const int64_t total = (int64_t)rows * cols;
    const float inv = 1.0f / (float)total;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        float t = y[i];
        if (lam != nullptr) {
            const float l = lam[r];
            t = t * l + y[(int64_t)perm[r] * cols + c] * (1.0f - l);
        }
        const float sg = 1.0f / (1.0f + expf(-z[i]));
        dlogits[i] = weight * (sg - t) * inv;
    }
}

__global__ __launch_bounds__(256) void sigmoid_mean_kernel(const float* __restrict__ z, int rows, int cols,
                                                           float* __restrict__ act) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.0f;
    for (int r = 0; r < rows; ++r) s += 1.0f / (1.0f + expf(-z[(int64_t)r * cols + c]));
    act[c] = s / (float)rows;
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, int64_t n, float alpha) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] *= alpha;
}

// x *= *alpha_dev (a device scalar: the upstream gradient of the loss; no host round trip)
__global__ __launch_bounds__(256) void scale_dev_kernel(float* __restrict__ x, int64_t n, const float* __restrict__ alpha_dev) {
    const float alpha = *alpha_dev;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] *= alpha;
}

// dst[r, c] = cast(src[r, c]) for c < cols, 0 for cols <= c < ld_dst  (fp32 rows -> zero-padded operand rows)
__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ src, int64_t ld_src,
                                                        void* __restrict__ dst, int64_t ld_dst, int rows, int cols,
                                                        int dtype) {
    const int64_t total = (int64_t)rows * ld_dst;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / ld_dst;
        const int c = (int)(i - r * ld_dst);
        const float v = c < cols ? src[r * ld_src + c] : 0.0f;
        if (dtype == MAEST_BF16) reinterpret_cast<bf16_t*>(dst)[i] = f2bf(v);
        else reinterpret_cast<float*>(dst)[i] = v;
    }
}

// ---- stochastic weight averaging over MANY parameters in one launch: avg += (w - avg) * inv_count
// (Lightning's StochasticWeightAveraging.avg_fn as used by helpers/swa_callback.py)
struct SwaTable {
    float* avg[CAST_MAX_ITEMS];
    const float* cur[CAST_MAX_ITEMS];
    int64_t numel[CAST_MAX_ITEMS];
    int block_begin[CAST_MAX_ITEMS + 1];   // prefix sum of 4096-element chunks
    int n;
};
__global__ __launch_bounds__(256) void swa_update_kernel(const SwaTable tab, float inv_count) {
    int it = 0;
    while (it + 1 < tab.n && (int)blockIdx.x >= tab.block_begin[it + 1]) ++it;
    const int64_t base = (int64_t)(blockIdx.x - tab.block_begin[it]) * 4096;
    float* __restrict__ a = tab.avg[it];
    const float* __restrict__ w = tab.cur[it];
    const int64_t n = tab.numel[it];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int64_t i = base + j * 256 + threadIdx.x;
        if (i < n) {
            const float av = a[i];
            a[i] = av + (w[i] - av) * inv_count;
        }
    }
}

__global__ __launch_bounds__(256) void affine_kernel(float* __restrict__ x, int64_t n, float add, float div) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        x[i] = (x[i] + add) / div;
}

}  // namespace maest

using namespace maest;

extern "C" int maest_transpose(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int rows, int cols,
                               int dtype, void* stream) {
    MAEST_REQUIRE(src && dst, "maest_transpose: null pointer");
    MAEST_REQUIRE(rows > 0 && cols > 0 && ld_dst >= rows && ld_src >= cols, "maest_transpose: bad shape");
    MAEST_REQUIRE(dtype == MAEST_F32 || dtype == MAEST_BF16, "maest_transpose: bad dtype");
    dim3 grid((unsigned)((ld_dst + 63) / 64), (cols + 63) / 64);
    if (dtype == MAEST_BF16)
        hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, dim3(256), 64 * 66 * 2, (hipStream_t)stream,
                           (const bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst, rows, cols);
    else
        hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 64 * 65 * 4, (hipStream_t)stream,
                           (const float*)src, ld_src, (float*)dst, ld_dst, rows, cols);
    return check_launch("maest_transpose");
}

extern "C" int maest_cast_weights(const float* src, void* dst, void* dst_t, int rows, int cols, int dtype,
                                  void* stream) {
    MAEST_REQUIRE(src && (dst || dst_t), "maest_cast_weights: null pointer");
    MAEST_REQUIRE(rows > 0 && cols > 0, "maest_cast_weights: bad shape");
    MAEST_REQUIRE(dtype == MAEST_F32 || dtype == MAEST_BF16, "maest_cast_weights: bad dtype");
    dim3 grid((rows + 63) / 64, (cols + 63) / 64);
    if (dtype == MAEST_BF16)
        hipLaunchKernelGGL(cast_weights_kernel<bf16_t>, grid, dim3(256), 64 * 65 * 4, (hipStream_t)stream, src,
                           (bf16_t*)dst, (bf16_t*)dst_t, rows, cols);
    else
        hipLaunchKernelGGL(cast_weights_kernel<float>, grid, dim3(256), 64 * 65 * 4, (hipStream_t)stream, src,
                           (float*)dst, (float*)dst_t, rows, cols);
    return check_launch("maest_cast_weights");
}

extern "C" int maest_cast_weights_multi(int n, const float* const* src, void* const* dst, void* const* dst_t,
                                        const int* rows, const int* cols, const int* scaled_rows, float row_scale, int dtype,
                                        void* stream) {
    MAEST_REQUIRE(n > 0 && src && dst && dst_t && rows && cols, "maest_cast_weights_multi: null pointer / n <= 0");
    MAEST_REQUIRE(dtype == MAEST_F32 || dtype == MAEST_BF16 || dtype == MAEST_SPLIT3_B, "maest_cast_weights_multi: bad dtype");
    for (int base = 0; base < n; base += CAST_MAX_ITEMS) {
        CastTable tab;
        tab.n = n - base < CAST_MAX_ITEMS ? n - base : CAST_MAX_ITEMS;
        int tiles = 0;
        for (int i = 0; i < tab.n; ++i) {
            const int k = base + i;
            MAEST_REQUIRE(src[k] && (dst[k] || dst_t[k]) && rows[k] > 0 && cols[k] > 0,
                          "maest_cast_weights_multi: bad item %d", k);
            tab.src[i] = src[k]; tab.dst[i] = dst[k]; tab.dst_t[i] = dst_t[k];
            tab.rows[i] = rows[k]; tab.cols[i] = cols[k];
            tab.scaled_rows[i] = scaled_rows ? scaled_rows[k] : 0;
            tab.tile_begin[i] = tiles;
            tiles += ((rows[k] + 63) / 64) * ((cols[k] + 63) / 64);
        }
        tab.tile_begin[tab.n] = tiles;
        tab.row_scale = row_scale;
        tab.split3b = dtype == MAEST_SPLIT3_B ? 1 : 0;
        if (tab.split3b)
            for (int i = 0; i < tab.n; ++i)
                MAEST_REQUIRE(tab.dst[i] && !tab.dst_t[i], "maest_cast_weights_multi: MAEST_SPLIT3_B wants dst and no dst_t (item %d)", base + i);
        if (dtype == MAEST_BF16 || dtype == MAEST_SPLIT3_B)
            hipLaunchKernelGGL(cast_weights_multi_kernel<bf16_t>, dim3(tiles), dim3(256), 64 * 65 * 4,
                               (hipStream_t)stream, tab);
        else
            hipLaunchKernelGGL(cast_weights_multi_kernel<float>, dim3(tiles), dim3(256), 64 * 65 * 4,
                               (hipStream_t)stream, tab);
    }
    return check_launch("maest_cast_weights_multi");
}

extern "C" int maest_colsum(const void* src, int64_t ld, int rows, int cols, int dtype, float* out, void* stream) {
    MAEST_REQUIRE(src && out, "maest_colsum: null pointer");
    MAEST_REQUIRE(rows > 0 && cols > 0 && ld >= cols, "maest_colsum: bad shape");
    MAEST_REQUIRE(dtype == MAEST_F32 || dtype == MAEST_BF16, "maest_colsum: bad dtype");
    dim3 grid((cols + 255) / 256, (rows + 511) / 512);
    if (dtype == MAEST_BF16)
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ld,
                           rows, cols, out);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, ld, rows,
                           cols, out);
    return check_launch("maest_colsum");
}

extern "C" int maest_bce_logits(const float* z, const float* y, const int32_t* perm, const float* lam, int rows,
                                int cols, float weight, float* loss, float* dlogits, void* stream) {
    MAEST_REQUIRE(z && y && loss, "maest_bce_logits: null pointer");
    MAEST_REQUIRE(rows > 0 && cols > 0, "maest_bce_logits: bad shape");
    MAEST_REQUIRE((perm == nullptr) == (lam == nullptr), "maest_bce_logits: perm and lam go together");
    const int64_t total = (int64_t)rows * cols;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 256) blocks = 256;
    if (dlogits == nullptr) {                          // loss only (evaluation): one workgroup of 16 waves, fixed order
        hipLaunchKernelGGL(bce_partial_kernel, dim3(1), dim3(1024), 64, (hipStream_t)stream, z, y, perm, lam, rows, cols, weight,
                           (float*)nullptr, loss);
    } else {
        hipLaunchKernelGGL(bce_partial_kernel, dim3(blocks), dim3(256), 64, (hipStream_t)stream, z, y, perm, lam, rows, cols, weight,
                           dlogits, (float*)nullptr);
        hipLaunchKernelGGL(bce_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)dlogits, blocks,
                           weight / (float)total, loss);
        hipLaunchKernelGGL(bce_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z, y, perm, lam, rows, cols, weight,
                           dlogits);
    }
    return check_launch("maest_bce_logits");
}

extern "C" int maest_sigmoid_mean(const float* z, int rows, int cols, float* act, void* stream) {
    MAEST_REQUIRE(z && act, "maest_sigmoid_mean: null pointer");
    MAEST_REQUIRE(rows > 0 && cols > 0, "maest_sigmoid_mean: bad shape");
    hipLaunchKernelGGL(sigmoid_mean_kernel, dim3((cols + 255) / 256), dim3(256), 0, (hipStream_t)stream, z, rows,
                       cols, act);
    return check_launch("maest_sigmoid_mean");
}

extern "C" int maest_scale_f32(float* x, int64_t n, float alpha, void* stream) {
    MAEST_REQUIRE(x && n >= 0, "maest_scale_f32: bad arguments");
    if (n == 0) return MAEST_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, alpha);
    return check_launch("maest_scale_f32");
}

extern "C" int maest_scale_dev_f32(float* x, int64_t n, const float* alpha_dev, void* stream) {
    MAEST_REQUIRE(x && alpha_dev && n >= 0, "maest_scale_dev_f32: bad arguments");
    if (n == 0) return MAEST_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(scale_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, alpha_dev);
    return check_launch("maest_scale_dev_f32");
}

extern "C" int maest_cast_rows(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int rows, int cols,
                               int dtype, void* stream) {
    MAEST_REQUIRE(src && dst, "maest_cast_rows: null pointer");
    MAEST_REQUIRE(rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= cols, "maest_cast_rows: bad shape");
    MAEST_REQUIRE(dtype == MAEST_F32 || dtype == MAEST_BF16, "maest_cast_rows: bad dtype");
    int64_t blocks = ((int64_t)rows * ld_dst + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(cast_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, ld_src, dst,
                       ld_dst, rows, cols, dtype);
    return check_launch("maest_cast_rows");
}

extern "C" int maest_affine_f32(float* x, int64_t n, float add, float div, void* stream) {
    MAEST_REQUIRE(x && n >= 0 && div != 0.0f, "maest_affine_f32: bad arguments");
    if (n == 0) return MAEST_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(affine_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, add, div);
    return check_launch("maest_affine_f32");
}

extern "C" int maest_swa_update_multi(int n, float* const* avg, const float* const* cur, const int64_t* numel,
                                      float inv_count, void* stream) {
    MAEST_REQUIRE(n > 0 && avg && cur && numel, "maest_swa_update_multi: null pointer / n <= 0");
    for (int base = 0; base < n; base += CAST_MAX_ITEMS) {
        SwaTable tab;
        tab.n = n - base < CAST_MAX_ITEMS ? n - base : CAST_MAX_ITEMS;
        int blocks = 0;
        for (int i = 0; i < tab.n; ++i) {
            const int k = base + i;
            MAEST_REQUIRE(avg[k] && cur[k] && numel[k] > 0, "maest_swa_update_multi: bad item %d", k);
            tab.avg[i] = avg[k]; tab.cur[i] = cur[k]; tab.numel[i] = numel[k];
            tab.block_begin[i] = blocks;
            blocks += (int)((numel[k] + 4095) / 4096);
        }
        tab.block_begin[tab.n] = blocks;
        hipLaunchKernelGGL(swa_update_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tab, inv_count);
    }
    return check_launch("maest_swa_update_multi");
}

extern "C" int maest_gather_head_rows(const void* src, int clips, int n_tok, int n_head, int dtype, void* dst, void* stream) {
    MAEST_REQUIRE(src && dst, "maest_gather_head_rows: null pointer");
    MAEST_REQUIRE(clips > 0 && n_head > 0 && n_tok >= n_head, "maest_gather_head_rows: bad shape clips=%d n_tok=%d n_head=%d",
                  clips, n_tok, n_head);
    MAEST_REQUIRE(dtype == MAEST_F32 || dtype == MAEST_BF16, "maest_gather_head_rows: bad dtype %d", dtype);
    const int64_t n = (int64_t)clips * n_head * 768 * (dtype == MAEST_BF16 ? 2 : 4) / 16;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (dtype == MAEST_BF16)
        hipLaunchKernelGGL(gather_head_rows_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)src, n_tok, n_head, (bf16_t*)dst, n);
    else
        hipLaunchKernelGGL(gather_head_rows_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const float*)src, n_tok, n_head, (float*)dst, n);
    return check_launch("maest_gather_head_rows");
}

extern "C" int maest_scatter_head_rows(const void* src, int clips, int n_tok, int n_head, int n_pad, int dtype, void* dst,
                                       void* stream) {
    MAEST_REQUIRE(src && dst, "maest_scatter_head_rows: null pointer");
    MAEST_REQUIRE(clips > 0 && n_head > 0 && n_pad >= n_head && n_tok >= n_pad,
                  "maest_scatter_head_rows: bad shape clips=%d n_tok=%d n_head=%d n_pad=%d", clips, n_tok, n_head, n_pad);
    MAEST_REQUIRE(dtype == MAEST_F32 || dtype == MAEST_BF16, "maest_scatter_head_rows: bad dtype %d", dtype);
    const int64_t n = (int64_t)clips * n_pad * 768 * (dtype == MAEST_BF16 ? 2 : 4) / 16;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (dtype == MAEST_BF16)
        hipLaunchKernelGGL(scatter_head_rows_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)src, n_tok, n_head, n_pad, (bf16_t*)dst, n);
    else
        hipLaunchKernelGGL(scatter_head_rows_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const float*)src, n_tok, n_head, n_pad, (float*)dst, n);
    return check_launch("maest_scatter_head_rows");
}
