// Fused log-mel front end: reflect-padded framing -> Hann window -> 512-point real FFT -> |.|^2 ->
// 96-band slaney mel projection -> log10(1 + 1e4 x) -> z-norm, one kernel, one HBM round trip.
// (reference: MelSpectrogram.forward models/helpers/melspectrogram.py:47-60 with the torchaudio
//  Spectrogram(n_fft=512, hop=256, power=2, center=True/reflect) :29-34 and MelScale(96, slaney) :36-42;
//  constants :16-24.)
//
// Algorithmically HBM-bound (0.88 MB / 10 s clip in+out vs ~7 MFLOP of FFT); as written it is instruction-issue bound
// (~800 VALU / LDS instructions per frame and wave: 0.25 ms for 256 clips = 0.11 of the HBM roofline, 1.0 M clips/s --
// 200x the rate the training step consumes them at).  A workgroup owns 64 consecutive frames
// of one clip so that the [96, T] output is written as 256-byte runs along T; each of its 4 waves
// transforms 16 frames, one at a time (the next frame's samples in flight), entirely in LDS and without block barriers: the 512 real samples are packed as 256
// complex points, transformed by 4 radix-4 DIF stages (one butterfly per lane per stage), unpacked
// to the 257-bin one-sided spectrum, and projected onto the mel bands with the filterbank stored in
// band-sparse form (each triangular band touches <= fb_stride consecutive bins).  All arithmetic is fp32.
#include "common.h"

namespace maest {

constexpr int MEL_NFFT = 512;
constexpr int MEL_HOP = 256;
constexpr int MEL_NBINS = 257;
constexpr int MEL_BANDS = 96;
constexpr int MEL_FRAMES_PER_BLOCK = 64;
constexpr int MEL_OUT_LD = MEL_FRAMES_PER_BLOCK + 1;
constexpr int MEL_WREG0 = 8, MEL_WREG1 = 16;   // filter weights kept in registers for bands 0..63 / 64..95

struct cplx {
    float re, im;
};
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx mul_neg_i(cplx a) { return {a.im, -a.re}; }   // a * (-i)
__device__ __forceinline__ int rev4_256(int k) {  // reverse the four base-4 digits of k
    return ((k & 3) << 6) | (((k >> 2) & 3) << 4) | (((k >> 4) & 3) << 2) | ((k >> 6) & 3);
}

// LDS hand-off between the lanes of ONE wave: a wave's DS instructions execute in order and all 64 lanes issue them
// together, so a write by one lane is visible to a later read by another lane of the same wave without a barrier;
// only the compiler has to keep the order (the host emulator, one thread per lane, maps this to a wave barrier).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// sample fetch of frame t: the 8 samples lane `lane` packs as complex points lane + 64 j.  INTERIOR (block-uniform: all 64
// frames of the block lie inside the clip and the clip starts 8-byte aligned): four plain 8-byte loads; otherwise
// clamped (frames >= T are computed on the last frame's data and not stored) and reflect-padded, sample by sample
template <bool INTERIOR>
__device__ __forceinline__ void mel_fetch(float (&x)[4][2], const float* __restrict__ wsrc, int t, int T, int S, int lane) {
    if (INTERIOR) {
        const float* p0 = wsrc + t * MEL_HOP - MEL_NFFT / 2 + 2 * lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 v = *reinterpret_cast<const float2*>(p0 + 128 * j);
            x[j][0] = v.x; x[j][1] = v.y;
        }
        return;
    }
    const int tc = t < T ? t : T - 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = 2 * (lane + 64 * j);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int i = tc * MEL_HOP + p + e - MEL_NFFT / 2;
            if (i < 0) i = -i;
            if (i >= S) i = 2 * (S - 1) - i;
            x[j][e] = wsrc[i];
        }
    }
}

__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ wave_in, int S, int T,
                                                     const float* __restrict__ window,
                                                     const float* __restrict__ twiddle,   // [512][2] exp(-2 pi i k / 512)
                                                     const int32_t* __restrict__ fb_start,
                                                     const int32_t* __restrict__ fb_len,
                                                     const float* __restrict__ fb_w, int fb_stride, float log_scale,
                                                     float norm_mean, float norm_2std, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* tw = reinterpret_cast<float*>(smem);                      // [1024]
    float* otile = tw + 1024;                                        // [96][65]
    cplx* z = reinterpret_cast<cplx*>(otile + MEL_BANDS * MEL_OUT_LD) + wv * 256;   // per wave [256]
    float* pw = reinterpret_cast<float*>(reinterpret_cast<cplx*>(otile + MEL_BANDS * MEL_OUT_LD) + 4 * 256) + wv * 260;

    const int b = blockIdx.y;
    const int t0 = blockIdx.x * MEL_FRAMES_PER_BLOCK;
    const float* wsrc = wave_in + (int64_t)b * S;
    // the wave works alone on its 16 frames (its own z / pw regions): no block barrier inside the frame loop; the
    // samples of frame fi + 1 are in flight while frame fi is transformed
    float x[4][2], win[4][2];
    const bool interior = t0 > 0 && (t0 + MEL_FRAMES_PER_BLOCK) * MEL_HOP + MEL_NFFT / 2 <= S && t0 + MEL_FRAMES_PER_BLOCK <= T &&
                          (reinterpret_cast<uintptr_t>(wsrc) & 7) == 0;      // block-uniform
    if (interior) mel_fetch<true>(x, wsrc, t0 + wv * 16, T, S, lane);
    else mel_fetch<false>(x, wsrc, t0 + wv * 16, T, S, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float2 w2 = *reinterpret_cast<const float2*>(window + 2 * (lane + 64 * j));
        win[j][0] = w2.x; win[j][1] = w2.y;
    }
    // this lane's two bands (lane, lane + 64): the first MEL_WREG0 / MEL_WREG1 filter weights live in registers (the
    // slaney bank's bands are 1..6 and 5..15 bins long), zero beyond the band; longer bands finish in a global-read loop
    int mb_start[2], mb_len[2];
    float mw0[MEL_WREG0], mw1[MEL_WREG1];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = lane + 64 * j;
        mb_start[j] = m < MEL_BANDS ? fb_start[m] : 0;
        mb_len[j] = m < MEL_BANDS ? fb_len[m] : 0;
    }
#pragma unroll
    for (int i = 0; i < MEL_WREG0; ++i) mw0[i] = i < mb_len[0] ? fb_w[lane * fb_stride + i] : 0.0f;
#pragma unroll
    for (int i = 0; i < MEL_WREG1; ++i) mw1[i] = i < mb_len[1] ? fb_w[(lane + 64) * fb_stride + i] : 0.0f;
    if (lane < 3) pw[MEL_NBINS + lane] = 0.0f;      // the padding the clamped reads below may touch
    for (int i = threadIdx.x; i < 1024; i += 256) tw[i] = twiddle[i];
    __syncthreads();

    for (int fi = 0; fi < 16; ++fi) {
        const int tl = wv * 16 + fi;       // frame within the block
        // ---- framing (center=True, reflect padding of 256 samples) + window + complex packing
#pragma unroll
        for (int j = 0; j < 4; ++j) z[lane + 64 * j] = {x[j][0] * win[j][0], x[j][1] * win[j][1]};
        if (fi + 1 < 16) {
            if (interior) mel_fetch<true>(x, wsrc, t0 + tl + 1, T, S, lane);
            else mel_fetch<false>(x, wsrc, t0 + tl + 1, T, S, lane);
        }
        wave_lds_sync();
        // ---- 256-point complex FFT, radix-4 DIF, 4 stages, one in-place butterfly per lane per stage
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int L = 256 >> (2 * st);
            const int q = L >> 2;
            const int blk = lane / q, pos = lane - blk * q;
            const int base = blk * L + pos;
            const cplx a0 = z[base], a1 = z[base + q], a2 = z[base + 2 * q], a3 = z[base + 3 * q];
            const cplx b0 = cadd(a0, a2), b1 = csub(a0, a2), b2 = cadd(a1, a3), b3 = mul_neg_i(csub(a1, a3));
            cplx y0 = cadd(b0, b2), y1 = cadd(b1, b3), y2 = csub(b0, b2), y3 = csub(b1, b3);
            const int tstep = (MEL_NFFT / L) * pos;   // exp(-2 pi i pos m / L) = tw[(512 / L) * pos * m]
            const cplx w1 = {tw[2 * tstep], tw[2 * tstep + 1]};
            const cplx w2 = {tw[4 * tstep], tw[4 * tstep + 1]};
            const cplx w3 = {tw[6 * tstep], tw[6 * tstep + 1]};
            y1 = cmul(y1, w1);
            y2 = cmul(y2, w2);
            y3 = cmul(y3, w3);
            z[base] = y0; z[base + q] = y1; z[base + 2 * q] = y2; z[base + 3 * q] = y3;   // the points this lane read
            wave_lds_sync();
        }
        // ---- unpack the real FFT: X[k] = E[k] + W^k O[k], power spectrum for k = 0..256
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int k = lane + 64 * j;
            if (k <= 256) {
                const cplx zk = z[rev4_256(k & 255)];
                cplx zc = z[rev4_256((256 - k) & 255)];
                zc.im = -zc.im;
                const cplx e = {0.5f * (zk.re + zc.re), 0.5f * (zk.im + zc.im)};
                const cplx d = {0.5f * (zk.re - zc.re), 0.5f * (zk.im - zc.im)};
                const cplx o = mul_neg_i(d);                                 // (Z[k] - conj Z[N-k]) / (2i)
                const cplx w = {tw[2 * k], tw[2 * k + 1]};                    // exp(-2 pi i k / 512)
                const cplx xk = cadd(e, cmul(o, w));
                pw[k] = xk.re * xk.re + xk.im * xk.im;
            }
        }
        wave_lds_sync();
        // ---- mel projection + logC + z-norm into the block's output tile
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = lane + 64 * j;
            if (m < MEL_BANDS) {
                const int s0 = mb_start[j], n = mb_len[j];
                float acc = 0.0f;
                if (j == 0) {
#pragma unroll
                    for (int i = 0; i < MEL_WREG0; ++i) acc += pw[s0 + i < MEL_NBINS ? s0 + i : MEL_NBINS] * mw0[i];
                    for (int i = MEL_WREG0; i < n; ++i) acc += pw[s0 + i] * fb_w[m * fb_stride + i];
                } else {
#pragma unroll
                    for (int i = 0; i < MEL_WREG1; ++i) acc += pw[s0 + i < MEL_NBINS ? s0 + i : MEL_NBINS] * mw1[i];
                    for (int i = MEL_WREG1; i < n; ++i) acc += pw[s0 + i] * fb_w[m * fb_stride + i];
                }
                const float lm = log10f(1.0f + acc * log_scale);
                otile[m * MEL_OUT_LD + tl] = (lm - norm_mean) / norm_2std;
            }
        }
        wave_lds_sync();       // pw / z are rewritten by the next frame
    }
    __syncthreads();
    // ---- coalesced store of the [96][64] tile
    for (int i = threadIdx.x; i < MEL_BANDS * MEL_FRAMES_PER_BLOCK; i += 256) {
        const int m = i >> 6, tl = i & 63;
        if (t0 + tl < T) out[((int64_t)b * MEL_BANDS + m) * T + t0 + tl] = otile[m * MEL_OUT_LD + tl];
    }
}

}  // namespace maest

using namespace maest;

extern "C" int maest_logmel(const float* wave, int B, int S, const float* window, const float* twiddle,
                            const int32_t* fb_start, const int32_t* fb_len, const float* fb_w, int fb_stride,
                            float log_scale, float norm_mean, float norm_2std, float* out, void* stream) {
    MAEST_REQUIRE(wave && window && twiddle && fb_start && fb_len && fb_w && out, "maest_logmel: null pointer");
    MAEST_REQUIRE(B > 0 && S > MEL_NFFT / 2, "maest_logmel: bad shape B=%d S=%d (reflect padding needs S > 256)", B, S);
    MAEST_REQUIRE(fb_stride > 0, "maest_logmel: bad fb_stride");
    const int T = 1 + S / MEL_HOP;
    const int smem_bytes = (1024 + MEL_BANDS * MEL_OUT_LD) * 4 + 4 * 256 * 8 + 4 * 260 * 4;
    dim3 grid((T + MEL_FRAMES_PER_BLOCK - 1) / MEL_FRAMES_PER_BLOCK, B);
    hipLaunchKernelGGL(logmel_kernel, grid, dim3(256), smem_bytes, (hipStream_t)stream, wave, S, T, window, twiddle,
                       fb_start, fb_len, fb_w, fb_stride, log_scale, norm_mean, norm_2std, out);
    return check_launch("maest_logmel");
}
