// Second log-mel parameterisation: AugmentMelSTFT (reference: models/preprocess.py:17-128 -- 32 kHz, pre-emphasis
// [-0.97, 1] :82-84, torch.stft(n_fft = 1024, hop = 320, win = 800 non-periodic Hann, center = True) :85-94, power
// spectrum :95, 128 kaldi mel banks with fmin / fmax jitter :96-121, log(x + 1e-5) :123, (x + 4.5) / 5 :129).
// Named in north_star; dead code in the reference's MAEST path, so this is the same fused
// design as mel.hip instantiated for the other constants rather than a tuned kernel: one pass over HBM, a
// workgroup owns 32 consecutive frames of one clip, each wave transforms 8 of them in LDS with a 1024-point
// radix-4 complex FFT (5 stages, 4 butterflies per lane per stage; the input is real, bins 0..512 are used), the
// filterbank comes in band-sparse form from the host (it changes per call under fmin / fmax augmentation).
#include "common.h"

namespace maest {

constexpr int M2_NFFT = 1024;
constexpr int M2_HOP = 320;
constexpr int M2_NBINS = 513;
constexpr int M2_FPB = 32;            // frames per block
constexpr int M2_OUT_LD = M2_FPB + 1;
constexpr int M2_MAXBANDS = 128;

struct cplx2 {
    float re, im;
};
__device__ __forceinline__ cplx2 c2add(cplx2 a, cplx2 b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx2 c2sub(cplx2 a, cplx2 b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx2 c2mul(cplx2 a, cplx2 b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx2 c2mul_neg_i(cplx2 a) { return {a.im, -a.re}; }
__device__ __forceinline__ int rev4_1024(int k) {   // reverse the five base-4 digits of k
    return ((k & 3) << 8) | (((k >> 2) & 3) << 6) | (((k >> 4) & 3) << 4) | (((k >> 6) & 3) << 2) | ((k >> 8) & 3);
}

__global__ __launch_bounds__(256) void augment_mel_kernel(const float* __restrict__ wave_in, int S, int T,
                                                          const float* __restrict__ window,     // [1024], zero padded
                                                          const float* __restrict__ twiddle,    // [1024][2]
                                                          const int32_t* __restrict__ fb_start,
                                                          const int32_t* __restrict__ fb_len,
                                                          const float* __restrict__ fb_w, int fb_stride, int n_mels,
                                                          float pre0, float pre1, float log_eps, float norm_add,
                                                          float norm_div, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* tw = reinterpret_cast<float*>(smem);                       // [2048]
    float* otile = tw + 2048;                                         // [128][33]
    cplx2* zall = reinterpret_cast<cplx2*>(otile + M2_MAXBANDS * M2_OUT_LD);
    cplx2* z = zall + wv * M2_NFFT;                                   // per wave [1024]
    float* pw = reinterpret_cast<float*>(zall + 4 * M2_NFFT) + wv * 516;

    for (int i = threadIdx.x; i < 2048; i += 256) tw[i] = twiddle[i];
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * M2_FPB;
    const float* wsrc = wave_in + (int64_t)b * S;
    const int Sy = S - 1;                                             // length after the 2-tap pre-emphasis
    __syncthreads();

    for (int fi = 0; fi < M2_FPB / 4; ++fi) {
        const int tl = wv * (M2_FPB / 4) + fi;
        const int t = t0 + tl;
        // ---- pre-emphasis + framing (center = True, reflect padding of 512) + window
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int p = lane + 64 * j;
            int i = (t < T ? t : T - 1) * M2_HOP + p - M2_NFFT / 2;
            if (i < 0) i = -i;
            if (i >= Sy) i = 2 * (Sy - 1) - i;
            const float y = pre0 * wsrc[i] + pre1 * wsrc[i + 1];
            z[p] = {y * window[p], 0.0f};
        }
        __syncthreads();
        // ---- 1024-point complex FFT, radix-4 DIF
#pragma unroll
        for (int st = 0; st < 5; ++st) {
            const int L = M2_NFFT >> (2 * st);
            const int q = L >> 2;
            cplx2 y[4][4];
            int base[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = lane + 64 * u;
                const int blk = j / q, pos = j - blk * q;
                base[u] = blk * L + pos;
                const cplx2 a0 = z[base[u]], a1 = z[base[u] + q], a2 = z[base[u] + 2 * q], a3 = z[base[u] + 3 * q];
                const cplx2 b0 = c2add(a0, a2), b1 = c2sub(a0, a2), b2 = c2add(a1, a3), b3 = c2mul_neg_i(c2sub(a1, a3));
                const int tstep = (M2_NFFT / L) * pos;
                const cplx2 w1 = {tw[2 * tstep], tw[2 * tstep + 1]};
                const cplx2 w2 = {tw[4 * tstep], tw[4 * tstep + 1]};
                const cplx2 w3 = {tw[6 * tstep], tw[6 * tstep + 1]};
                y[u][0] = c2add(b0, b2);
                y[u][1] = c2mul(c2add(b1, b3), w1);
                y[u][2] = c2mul(c2sub(b0, b2), w2);
                y[u][3] = c2mul(c2sub(b1, b3), w3);
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                z[base[u]] = y[u][0]; z[base[u] + q] = y[u][1]; z[base[u] + 2 * q] = y[u][2]; z[base[u] + 3 * q] = y[u][3];
            }
            __syncthreads();
        }
        // ---- power spectrum of bins 0 .. 512
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int k = lane + 64 * j;
            if (k < M2_NBINS) {
                const cplx2 x = z[rev4_1024(k)];
                pw[k] = x.re * x.re + x.im * x.im;
            }
        }
        __syncthreads();
        // ---- mel projection, log, affine normalisation
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = lane + 64 * j;
            if (m < n_mels) {
                const int s0 = fb_start[m], n = fb_len[m];
                float acc = 0.0f;
                for (int i = 0; i < n; ++i) acc += pw[s0 + i] * fb_w[m * fb_stride + i];
                otile[m * M2_OUT_LD + tl] = (logf(acc + log_eps) + norm_add) / norm_div;
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n_mels * M2_FPB; i += 256) {
        const int m = i / M2_FPB, tl = i - m * M2_FPB;
        if (t0 + tl < T) out[((int64_t)b * n_mels + m) * T + t0 + tl] = otile[m * M2_OUT_LD + tl];
    }
}

}  // namespace maest

using namespace maest;

extern "C" int maest_augment_mel(const float* wave, int B, int S, const float* window, const float* twiddle,
                                 const int32_t* fb_start, const int32_t* fb_len, const float* fb_w, int fb_stride,
                                 int n_mels, float pre0, float pre1, float log_eps, float norm_add, float norm_div,
                                 float* out, void* stream) {
    MAEST_REQUIRE(wave && window && twiddle && fb_start && fb_len && fb_w && out, "maest_augment_mel: null pointer");
    MAEST_REQUIRE(B > 0 && S > M2_NFFT / 2 + 1, "maest_augment_mel: bad shape B=%d S=%d (reflect padding needs S > 513)", B, S);
    MAEST_REQUIRE(n_mels > 0 && n_mels <= M2_MAXBANDS && fb_stride > 0, "maest_augment_mel: bad filterbank n_mels=%d", n_mels);
    const int T = 1 + (S - 1) / M2_HOP;
    const int smem_bytes = (2048 + M2_MAXBANDS * M2_OUT_LD) * 4 + 4 * M2_NFFT * 8 + 4 * 516 * 4;
    static DeviceOnce once;
    ensure_dynamic_lds(once, &augment_mel_kernel, smem_bytes);
    dim3 grid((T + M2_FPB - 1) / M2_FPB, B);
    hipLaunchKernelGGL(augment_mel_kernel, grid, dim3(256), smem_bytes, (hipStream_t)stream, wave, S, T, window, twiddle,
                       fb_start, fb_len, fb_w, fb_stride, n_mels, pre0, pre1, log_eps, norm_add, norm_div, out);
    return check_launch("maest_augment_mel");
}
