// Patch-embedding operand builder, positional add / token assembly and their backward, plus the
// training-time input augmentations that the reference applies on the way in.
// (reference: PatchEmbed.forward models/maest.py:243-256 = Conv2d(1,768,k=16,s=10) :238-240;
//  time/freq positional add :645-675; structured patchout :684-687; flatten/transpose :769;
//  cls/dist tokens :785-796; mixup models/module.py:77-83; SpecMasking helpers/spec_masking.py:27-33.)
//
// The 16x16 / stride-10 convolution is evaluated as a GEMM [B*9*Tk, 256] x [768, 256]^T on the MFMA
// kernel (gemm.hip); this file produces its A operand straight from the (optionally mixed-up)
// spectrogram, ONLY for the time columns that survive structured patchout -- dropped columns are
// never computed, which is mathematically identical to computing then discarding them.
#include "common.h"

namespace maest {

constexpr int PE_D = 768;
constexpr int PE_K = 16;   // patch edge
constexpr int PE_S = 10;   // stride

// SpecMasking as a predicate of the operand load (helpers/spec_masking.py:27-33, applied per sample by the loader
// after normalisation and BEFORE the batch is mixed up: discogs/datamodule.py:140-152 then models/module.py:77-83):
// bit i of the result is set when time column tcol + i of clip b lies inside one of its time stripes; the whole row
// is dropped (all 16 bits) when frequency row `frow` lies inside one of its frequency stripes.  Stripes are
// (start, width) pairs, clamped like spec_mask_kernel below.
__device__ __forceinline__ uint32_t stripe_bits(const int32_t* __restrict__ t_stripes, int n_t,
                                                const int32_t* __restrict__ f_stripes, int n_f, int b, int frow,
                                                int tcol) {
    for (int k = 0; k < n_f; ++k) {
        int st = f_stripes[((int64_t)b * n_f + k) * 2];
        const int w = f_stripes[((int64_t)b * n_f + k) * 2 + 1];
        st = st < 0 ? 0 : st;
        if (frow >= st && frow < st + w) return 0xffffu;
    }
    uint32_t bits = 0;
    for (int k = 0; k < n_t; ++k) {
        int st = t_stripes[((int64_t)b * n_t + k) * 2];
        const int w = t_stripes[((int64_t)b * n_t + k) * 2 + 1];
        st = st < 0 ? 0 : st;
        int lo = st - tcol, hi = st + w - tcol;        // masked columns [lo, hi) relative to this patch row
        lo = lo < 0 ? 0 : lo;
        hi = hi > 16 ? 16 : hi;
        if (hi > lo) bits |= ((1u << (hi - lo)) - 1u) << lo;
    }
    return bits;
}

// one thread = one (patch row, ky): 16 contiguous input samples -> 16 contiguous operand elements
// XT: element type of x -- float, or _Float16 for the loader's half-precision mel batches (widened in the load)
template <typename XT>
__global__ __launch_bounds__(256) void patch_im2col_kernel(const XT* __restrict__ x, int B, int F, int T,
                                                           const int32_t* __restrict__ perm,
                                                           const float* __restrict__ lam,
                                                           const int32_t* __restrict__ tok_ft, int P,
                                                           const int32_t* __restrict__ t_stripes, int n_t,
                                                           const int32_t* __restrict__ f_stripes, int n_f,
                                                           void* __restrict__ out, int dtype, int stride_f, int stride_t) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)B * P * PE_K;
    if (gid >= total) return;
    const int ky = (int)(gid & 15);
    const int64_t prow = gid >> 4;
    const int j = (int)(prow % P);
    const int b = (int)(prow / P);
    const int f = tok_ft[2 * j];
    const int tcol = tok_ft[2 * j + 1] * stride_t;
    const int frow = f * stride_f + ky;
    const bool masked = n_t + n_f > 0;
    const XT* src = x + ((int64_t)b * F + frow) * T + tcol;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (float)src[i];
    if (masked) {
        const uint32_t bits = stripe_bits(t_stripes, n_t, f_stripes, n_f, b, frow, tcol);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (bits >> i) & 1u ? 0.0f : v[i];
    }
    if (lam != nullptr) {
        const float l = lam[b];
        const int b2 = perm[b];
        const XT* src2 = x + ((int64_t)b2 * F + frow) * T + tcol;
        float u[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) u[i] = (float)src2[i];
        if (masked) {
            const uint32_t bits = stripe_bits(t_stripes, n_t, f_stripes, n_f, b2, frow, tcol);
#pragma unroll
            for (int i = 0; i < 16; ++i) u[i] = (bits >> i) & 1u ? 0.0f : u[i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = v[i] * l + u[i] * (1.0f - l);
    }
    const int64_t o = prow * 256 + ky * 16;
    if (dtype == MAEST_BF16) {
        chunk16* dst = reinterpret_cast<chunk16*>(reinterpret_cast<bf16_t*>(out) + o);
        chunk16 c0, c1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c0[i] = pack_bf2(v[2 * i], v[2 * i + 1]);
            c1[i] = pack_bf2(v[8 + 2 * i], v[8 + 2 * i + 1]);
        }
        dst[0] = c0;
        dst[1] = c1;
    } else {
        float4* dst = reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + o);
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
    }
}

// x0[b, n, :]: a workgroup owns ONE token position n and a chunk of the batch.  The positional terms of that
// position (column tcol of time_pos [768, Tt], column f of freq_pos [768, Fg]: strided gathers) are fetched
// once per workgroup; the batch loop then streams float4 rows (the gather per element had made this kernel
// 4x slower than its bytes).  192 threads, 4 channels each.
constexpr int TA_BCHUNK = 32;
__global__ __launch_bounds__(192) void token_assemble_kernel(const float* __restrict__ patches,
                                                             const float* __restrict__ cls_token,
                                                             const float* __restrict__ dist_token,
                                                             const float* __restrict__ new_pos,
                                                             const float* __restrict__ freq_pos,
                                                             const float* __restrict__ time_pos, int Tt, int toffset,
                                                             const int32_t* __restrict__ tok_ft, int B, int Fg, int P,
                                                             float* __restrict__ x0) {
    const int Ntok = 2 + P;
    const int n = blockIdx.x;
    const int c = threadIdx.x * 4;
    const int b0 = blockIdx.y * TA_BCHUNK;
    const int b1 = b0 + TA_BCHUNK < B ? b0 + TA_BCHUNK : B;
    if (n < 2) {
        const float* tok = n == 0 ? cls_token : dist_token;
        float4 o;
        o.x = tok[c] + new_pos[n * PE_D + c];
        o.y = tok[c + 1] + new_pos[n * PE_D + c + 1];
        o.z = tok[c + 2] + new_pos[n * PE_D + c + 2];
        o.w = tok[c + 3] + new_pos[n * PE_D + c + 3];
        for (int b = b0; b < b1; ++b) *reinterpret_cast<float4*>(x0 + ((int64_t)b * Ntok + n) * PE_D + c) = o;
        return;
    }
    const int j = n - 2;
    const int f = tok_ft[2 * j];
    const int tcol = toffset + tok_ft[2 * j + 1];
    float tp[4], fp[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        tp[e] = time_pos[(int64_t)(c + e) * Tt + tcol];
        fp[e] = freq_pos[(int64_t)(c + e) * Fg + f];
    }
    for (int b = b0; b < b1; ++b) {
        const float4 p = *reinterpret_cast<const float4*>(patches + ((int64_t)b * P + j) * PE_D + c);
        float4 o;   // reference order: (conv + time_pos) + freq_pos   (maest.py:670,675)
        o.x = (p.x + tp[0]) + fp[0];
        o.y = (p.y + tp[1]) + fp[1];
        o.z = (p.z + tp[2]) + fp[2];
        o.w = (p.w + tp[3]) + fp[3];
        *reinterpret_cast<float4*>(x0 + ((int64_t)b * Ntok + n) * PE_D + c) = o;
    }
}

// grid (Ntok, batch slices): thread = 4 channels (16-byte accesses); each workgroup walks ITS slice of the batch with the
// loads of four clips in flight, and adds its partial sums to the (pre-zeroed) parameter gradients with atomics -- the
// time / frequency tables are reduced over tokens that way in any case.  (One workgroup per token walking all 256 clips
// with 4-byte loads was latency-bound: 222 us for 341 MB; two slices with 16-byte accesses: 154 us.)
constexpr int TAB_SLICES = 2;      // swept 1..32 at B = 256: 189 / 154 / 177 / 282 / 483 / 912 us -- the atomics of every extra slice cost 28 us
__global__ __launch_bounds__(192) void token_assemble_bwd_kernel(const float* __restrict__ dx0, int B, int Fg, int P,
                                                                 int Tt, int toffset,
                                                                 const int32_t* __restrict__ tok_ft,
                                                                 void* __restrict__ dpatches, int dtype,
                                                                 float* __restrict__ d_cls, float* __restrict__ d_dist,
                                                                 float* __restrict__ d_new_pos,
                                                                 float* __restrict__ d_freq_pos,
                                                                 float* __restrict__ d_time_pos) {
    const int Ntok = 2 + P;
    const int n = blockIdx.x;
    const int c = threadIdx.x * 4;
    const int per = (B + (int)gridDim.y - 1) / (int)gridDim.y;
    const int b0 = blockIdx.y * per;
    const int b1 = b0 + per < B ? b0 + per : B;
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const bool store = n >= 2 && dpatches != nullptr;
    auto one = [&](int b, const float4& g) {
        s[0] += g.x; s[1] += g.y; s[2] += g.z; s[3] += g.w;
        if (store) {
            const int64_t o = ((int64_t)b * P + (n - 2)) * PE_D + c;
            if (dtype == MAEST_BF16) {
                chunk8 v;
                v[0] = pack_bf2(g.x, g.y); v[1] = pack_bf2(g.z, g.w);
                *reinterpret_cast<chunk8*>(reinterpret_cast<bf16_t*>(dpatches) + o) = v;
            } else {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(dpatches) + o) = g;
            }
        }
    };
    const int64_t bstride = (int64_t)Ntok * PE_D;
    const float* src = dx0 + (int64_t)n * PE_D + c;
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
        float4 g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] = *reinterpret_cast<const float4*>(src + (b + u) * bstride);
#pragma unroll
        for (int u = 0; u < 4; ++u) one(b + u, g[u]);
    }
    for (; b < b1; ++b) one(b, *reinterpret_cast<const float4*>(src + b * bstride));
    if (b0 >= b1) return;
    if (n == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { unsafeAtomicAdd(d_cls + c + e, s[e]); unsafeAtomicAdd(d_new_pos + c + e, s[e]); }
    } else if (n == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { unsafeAtomicAdd(d_dist + c + e, s[e]); unsafeAtomicAdd(d_new_pos + PE_D + c + e, s[e]); }
    } else {
        const int j = n - 2;
        const int f = tok_ft[2 * j];
        const int tcol = toffset + tok_ft[2 * j + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsafeAtomicAdd(d_freq_pos + (int64_t)(c + e) * Fg + f, s[e]);
            unsafeAtomicAdd(d_time_pos + (int64_t)(c + e) * Tt + tcol, s[e]);
        }
    }
}

// zero stripes: grid (B, n_t + n_f)
__global__ __launch_bounds__(256) void spec_mask_kernel(float* __restrict__ x, int F, int T,
                                                        const int32_t* __restrict__ t_stripes, int n_t,
                                                        const int32_t* __restrict__ f_stripes, int n_f) {
    const int b = blockIdx.x, s = blockIdx.y;
    float* xb = x + (int64_t)b * F * T;
    if (s < n_t) {
        int st = t_stripes[((int64_t)b * n_t + s) * 2], w = t_stripes[((int64_t)b * n_t + s) * 2 + 1];
        if (st < 0) st = 0;
        if (st + w > T) w = T - st;
        for (int i = threadIdx.x; i < F * w; i += 256) xb[(int64_t)(i / w) * T + st + (i % w)] = 0.0f;
    } else {
        const int k = s - n_t;
        int st = f_stripes[((int64_t)b * n_f + k) * 2], w = f_stripes[((int64_t)b * n_f + k) * 2 + 1];
        if (st < 0) st = 0;
        if (st + w > F) w = F - st;
        for (int i = threadIdx.x; i < w * T; i += 256) xb[(int64_t)st * T + i] = 0.0f;
    }
}

// ---- on-disk mel chunks -> network input.
// Reference: DiscogsDataset.load_melspectrogram (discogs/dataset.py:69-140: raw float16 [frames, 96] rows,
// zero padding centred by np.roll(pad // 2), transpose to [96, T]) followed by the datamodule's norm_func
// (discogs/datamodule.py:126-136: (x - mean) / (2 std), evaluated by numpy IN float16: both constants are
// rounded to half, and so is each of the two results).  One workgroup = 64 frames x 96 bands of one clip,
// transposed through LDS so that the reads are whole 192-byte frame rows and the writes 256-byte time runs.
__global__ __launch_bounds__(256) void melfile_assemble_kernel(const uint16_t* __restrict__ frames,
                                                               const int64_t* __restrict__ row_start,
                                                               const int32_t* __restrict__ frames_read, int n_bands,
                                                               int T, int normalize, float norm_mean, float norm_div,
                                                               float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float (*tile)[97] = reinterpret_cast<float (*)[97]>(smem);   // [64][97]
    const int b = blockIdx.y, t0 = blockIdx.x * 64;
    const int nf = frames_read[b];
    const int shift = nf < T ? (T - nf) / 2 : 0;                 // np.roll(padded, pad // 2, axis=0)
    const _Float16 mean_h = (_Float16)norm_mean, div_h = (_Float16)norm_div;
    const uint16_t* src = frames + row_start[b] * n_bands;
    for (int i = threadIdx.x; i < 64 * n_bands; i += 256) {
        const int tl = i / n_bands, f = i - tl * n_bands;
        const int t = t0 + tl;
        float v = 0.0f;
        if (t < T) {
            int sidx = t - shift;
            if (sidx < 0) sidx += T;                             // rolled-around tail of the zero padding
            if (sidx < nf) {
                const uint16_t raw = src[(int64_t)sidx * n_bands + f];
                _Float16 h = __builtin_bit_cast(_Float16, raw);
                if (normalize) {
                    h = (_Float16)((float)h - (float)mean_h);    // half - half, rounded to half
                    h = (_Float16)((float)h / (float)div_h);     // half / half, rounded to half
                }
                v = (float)h;
            } else if (normalize) {
                _Float16 h = (_Float16)(0.0f - (float)mean_h);   // the zero padding is normalised too
                h = (_Float16)((float)h / (float)div_h);
                v = (float)h;
            }
        }
        tile[tl][f] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_bands * 64; i += 256) {
        const int f = i >> 6, tl = i & 63;
        if (t0 + tl < T) out[((int64_t)b * n_bands + f) * T + t0 + tl] = tile[tl][f];
    }
}

}  // namespace maest

using namespace maest;

extern "C" int maest_patch_im2col_strided(const void* x, int x_dtype, int B, int F, int T, int stride_f, int stride_t, const int32_t* perm,
                                          const float* lam, const int32_t* tok_ft, int P, const int32_t* t_stripes, int n_t,
                                          const int32_t* f_stripes, int n_f, void* out, int dtype, void* stream) {
    MAEST_REQUIRE(x && out && tok_ft, "maest_patch_im2col: null pointer");
    MAEST_REQUIRE(B > 0 && P > 0, "maest_patch_im2col: bad shape B=%d P=%d", B, P);
    MAEST_REQUIRE(F >= PE_K && T >= PE_K, "maest_patch_im2col: input %dx%d smaller than a patch", F, T);
    MAEST_REQUIRE(stride_f > 0 && stride_t > 0, "maest_patch_im2col: bad patch stride (%d, %d)", stride_f, stride_t);
    MAEST_REQUIRE((perm == nullptr) == (lam == nullptr), "maest_patch_im2col: perm and lam go together");
    MAEST_REQUIRE(dtype == MAEST_F32 || dtype == MAEST_BF16, "maest_patch_im2col: bad dtype");
    MAEST_REQUIRE(x_dtype == MAEST_F32 || x_dtype == MAEST_F16, "maest_patch_im2col: input must be fp32 or fp16");
    MAEST_REQUIRE(n_t >= 0 && n_f >= 0 && (n_t == 0 || t_stripes) && (n_f == 0 || f_stripes),
                  "maest_patch_im2col: bad stripe lists");
    const int64_t total = (int64_t)B * P * PE_K;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (x_dtype == MAEST_F16)
        hipLaunchKernelGGL(patch_im2col_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, B,
                           F, T, perm, lam, tok_ft, P, t_stripes, n_t, f_stripes, n_f, out, dtype, stride_f, stride_t);
    else
        hipLaunchKernelGGL(patch_im2col_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, B, F,
                           T, perm, lam, tok_ft, P, t_stripes, n_t, f_stripes, n_f, out, dtype, stride_f, stride_t);
    return check_launch("maest_patch_im2col");
}

extern "C" int maest_patch_im2col(const void* x, int x_dtype, int B, int F, int T, const int32_t* perm, const float* lam,
                                  const int32_t* tok_ft, int P, const int32_t* t_stripes, int n_t,
                                  const int32_t* f_stripes, int n_f, void* out, int dtype, void* stream) {
    return maest_patch_im2col_strided(x, x_dtype, B, F, T, PE_S, PE_S, perm, lam, tok_ft, P, t_stripes, n_t, f_stripes, n_f, out, dtype, stream);
}

extern "C" int maest_token_assemble(const float* patches, const float* cls_token, const float* dist_token,
                                    const float* new_pos, const float* freq_pos, const float* time_pos, int Fg, int Tt,
                                    int toffset, const int32_t* tok_ft, int B, int P, float* x0, void* stream) {
    MAEST_REQUIRE(patches && cls_token && dist_token && new_pos && freq_pos && time_pos && x0 && tok_ft,
                  "maest_token_assemble: null pointer");
    MAEST_REQUIRE(B > 0 && P > 0 && Fg > 0 && Tt > 0 && toffset >= 0, "maest_token_assemble: bad shape");
    hipLaunchKernelGGL(token_assemble_kernel, dim3(2 + P, (B + TA_BCHUNK - 1) / TA_BCHUNK), dim3(PE_D / 4), 0,
                       (hipStream_t)stream, patches, cls_token, dist_token, new_pos, freq_pos, time_pos, Tt, toffset,
                       tok_ft, B, Fg, P, x0);
    return check_launch("maest_token_assemble");
}

extern "C" int maest_token_assemble_bwd(const float* dx0, int B, int P, int Fg, int Tt, int toffset,
                                        const int32_t* tok_ft, void* dpatches, int dtype, float* d_cls, float* d_dist,
                                        float* d_new_pos, float* d_freq_pos, float* d_time_pos, void* stream) {
    MAEST_REQUIRE(dx0 && d_cls && d_dist && d_new_pos && d_freq_pos && d_time_pos && tok_ft,
                  "maest_token_assemble_bwd: null pointer");
    MAEST_REQUIRE(B > 0 && P > 0 && Fg > 0 && Tt > 0, "maest_token_assemble_bwd: bad shape");
    static_assert(PE_D == 192 * 4, "token_assemble_bwd_kernel: one thread per four channels");
    hipLaunchKernelGGL(token_assemble_bwd_kernel, dim3(2 + P, TAB_SLICES), dim3(192), 0, (hipStream_t)stream,
                       dx0, B, Fg, P, Tt, toffset, tok_ft, dpatches, dtype, d_cls, d_dist, d_new_pos, d_freq_pos,
                       d_time_pos);
    return check_launch("maest_token_assemble_bwd");
}

extern "C" int maest_spec_mask(float* x, int B, int F, int T, const int32_t* t_stripes, int n_t,
                               const int32_t* f_stripes, int n_f, void* stream) {
    MAEST_REQUIRE(x, "maest_spec_mask: null pointer");
    MAEST_REQUIRE(n_t >= 0 && n_f >= 0 && (n_t == 0 || t_stripes) && (n_f == 0 || f_stripes),
                  "maest_spec_mask: bad stripe lists");
    if (n_t + n_f == 0) return MAEST_OK;
    hipLaunchKernelGGL(spec_mask_kernel, dim3(B, n_t + n_f), dim3(256), 0, (hipStream_t)stream, x, F, T, t_stripes,
                       n_t, f_stripes, n_f);
    return check_launch("maest_spec_mask");
}

extern "C" int maest_melfile_assemble(const uint16_t* frames, const int64_t* row_start, const int32_t* frames_read,
                                      int B, int n_bands, int T, int normalize, float norm_mean, float norm_div,
                                      float* out, void* stream) {
    MAEST_REQUIRE(frames && row_start && frames_read && out, "maest_melfile_assemble: null pointer");
    MAEST_REQUIRE(B > 0 && T > 0 && n_bands > 0 && n_bands <= 96, "maest_melfile_assemble: bad shape B=%d T=%d bands=%d",
                  B, T, n_bands);
    hipLaunchKernelGGL(melfile_assemble_kernel, dim3((T + 63) / 64, B), dim3(256), 64 * 97 * 4, (hipStream_t)stream, frames,
                       row_start, frames_read, n_bands, T, normalize, norm_mean, norm_div, out);
    return check_launch("maest_melfile_assemble");
}
