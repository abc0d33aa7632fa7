// Fused log-mel front end: reflect-padded framing -> Hann window -> 512-point real FFT -> |.|^2 ->
// 96-band slaney mel projection -> log10(1 + 1e4 x) -> z-norm, one kernel, one HBM round trip.
// (reference: MelSpectrogram.forward models/helpers/melspectrogram.py:47-60 with the torchaudio
//  Spectrogram(n_fft=512, hop=256, power=2, center=True/reflect) :29-34 and MelScale(96, slaney) :36-42;
//  constants :16-24.)
//
// Algorithmically HBM-bound (0.88 MB / 10 s clip in+out vs ~7 MFLOP of FFT); instruction-issue bound as written.  A
// workgroup owns 64 consecutive frames of one clip so that the [96, T] output is written as 256-byte runs along T; each of
// its 4 waves transforms 16 frames, FOUR at a time (16 lanes per frame, 16 complex points per lane; the next four frames'
// samples in flight), without block barriers: the 512 real samples are packed as 256 complex points and transformed as
// 256 = 16 x 16 -- two 16-point DFTs in registers around one exchange through LDS --, unpacked to the 257-bin one-sided
// spectrum, and projected onto the mel bands with the filterbank stored in band-sparse form (each triangular band touches
// <= fb_stride consecutive bins): the four frames' power spectra lie bin-major in LDS, a lane forms one band and one half
// band of all four frames from 16-byte reads with its weights in registers.  All arithmetic is fp32.  (Rounds 1-2: one
// frame per wave, four radix-4 stages through LDS, ~800 instructions per frame and wave = 0.11 of the HBM roofline; round 3:
// 16 x 16 with a per-frame projection, ~440 = 0.17; round 5: the projection over four frames at once, ~280 = 0.23, then the FFT
// stages and the unpack in packed fp32 instructions with explicit operand modifiers, ~165 = 0.31 -- DESIGN.md section 4.)
#include <type_traits>

#include "common.h"

namespace maest {

constexpr int MEL_NFFT = 512;
constexpr int MEL_HOP = 256;
constexpr int MEL_NBINS = 257;
constexpr int MEL_BANDS = 96;
constexpr int MEL_FRAMES_PER_BLOCK = 64;
constexpr int MEL_OUT_LD = MEL_FRAMES_PER_BLOCK + 1;

// A complex value is a (re, im) pair in ONE 64-bit register pair, and the arithmetic is the packed fp32 instruction set with its operand
// modifiers written out: `op_sel` / `op_sel_hi` choose which half of a source feeds the low / high result, `neg_lo` / `neg_hi` negate it --
// a multiplication by -i or a conjugation costs no instruction, a complex product two.  (Left to hipcc, the same source spent a third of
// the two 16-point DFT stages on moves that re-paired registers: 175 of 518 instructions per four frames.)  The host emulator takes the
// plain C++ twins, which perform the same IEEE operations.
typedef f32x2_t cplx;
#define MEL_RE(z) ((z)[0])
#define MEL_IM(z) ((z)[1])
#if defined(__AMDGCN__)
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { cplx d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { cplx d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
// a + (-i) t = (a.re + t.im, a.im - t.re)  /  a - (-i) t = (a.re - t.im, a.im + t.re)
__device__ __forceinline__ cplx cadd_rot(cplx a, cplx t) { cplx d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(t)); return d; }
__device__ __forceinline__ cplx csub_rot(cplx a, cplx t) { cplx d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(t)); return d; }
// a b = (a.re b.re - a.im b.im, a.re b.im + a.im b.re): the a.im products first (cmul_a), then a.re (b.re, b.im) added by one packed fma (cmul_b)
// (the two halves of a product as separate statements: asm statements keep their source order, so independent products are written
//  interleaved -- first halves, then second halves -- and no instruction waits on its predecessor)
template <bool NEG>
__device__ __forceinline__ cplx cmul_a(cplx a, cplx b) {         // NEG: the product with -b
    cplx t;
    if (NEG) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));  // (a.im b.im, -a.im b.re)
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    return t;
}
template <bool NEG>
__device__ __forceinline__ cplx cmul_b(cplx a, cplx b, cplx t) {
    cplx d;
    if (NEG) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(t));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(t));
    return d;
}
// a + conj(p) / a - conj(p)
__device__ __forceinline__ cplx cadd_conj(cplx a, cplx p) { cplx d; asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(p)); return d; }
__device__ __forceinline__ cplx csub_conj(cplx a, cplx p) { cplx d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(p)); return d; }
__device__ __forceinline__ cplx csqr2(cplx a) { cplx q; asm("v_pk_mul_f32 %0, %1, %1" : "=v"(q) : "v"(a)); return q; }      // (re^2, im^2)
#else
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return cplx{a[0] + b[0], a[1] + b[1]}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return cplx{a[0] - b[0], a[1] - b[1]}; }
__device__ __forceinline__ cplx cadd_rot(cplx a, cplx t) { return cplx{a[0] + t[1], a[1] - t[0]}; }
__device__ __forceinline__ cplx csub_rot(cplx a, cplx t) { return cplx{a[0] - t[1], a[1] + t[0]}; }
template <bool NEG>
__device__ __forceinline__ cplx cmul_a(cplx a, cplx b) { return NEG ? cplx{a[1] * b[1], -(a[1] * b[0])} : cplx{-(a[1] * b[1]), a[1] * b[0]}; }
template <bool NEG>
__device__ __forceinline__ cplx cmul_b(cplx a, cplx b, cplx t) {
    return NEG ? cplx{__builtin_fmaf(a[0], -b[0], t[0]), __builtin_fmaf(a[0], -b[1], t[1])}
               : cplx{__builtin_fmaf(a[0], b[0], t[0]), __builtin_fmaf(a[0], b[1], t[1])};
}
__device__ __forceinline__ cplx cadd_conj(cplx a, cplx p) { return cplx{a[0] + p[0], a[1] - p[1]}; }
__device__ __forceinline__ cplx csub_conj(cplx a, cplx p) { return cplx{a[0] - p[0], a[1] + p[1]}; }
__device__ __forceinline__ cplx csqr2(cplx a) { return cplx{a[0] * a[0], a[1] * a[1]}; }
#endif
__device__ __forceinline__ int rev4_256(int k) {  // reverse the four base-4 digits of k
    return ((k & 3) << 6) | (((k >> 2) & 3) << 4) | (((k >> 4) & 3) << 2) | ((k >> 6) & 3);
}

__device__ __forceinline__ float mel_log2(float x) { return __builtin_amdgcn_logf(x); }       // v_log_f32

// The value lane (16 - l) mod 16 of the same 16-lane row holds (l = lane mod 16) = mirror(next(x)): two DPP moves on the device -- rotate the
// row by one, mirror it: no LDS crossbar, no wait --, lane permutations under the host emulator.  (Two functions: a DPP move wants its source
// written two instructions earlier, so the callers issue them in batches.)
__device__ __forceinline__ float mel_row_next(float x, int lane) {          // the value of lane (l + 1) mod 16
#if defined(__AMDGCN__)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x12f /* row_ror:15 */, 0xf, 0xf, false));
#else
    return __shfl(x, (lane & 48) | ((lane + 1) & 15), 64);
#endif
}
__device__ __forceinline__ float mel_row_mirror(float x, int lane) {        // the value of lane 15 - l
#if defined(__AMDGCN__)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x140 /* row_mirror */, 0xf, 0xf, false));
#else
    return __shfl(x, (lane & 48) | (15 - (lane & 15)), 64);
#endif
}

__device__ __forceinline__ float mel_lane_xor1(float x) {                   // the value of lane ^ 1
#if defined(__AMDGCN__)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xb1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false));
#else
    return __shfl_xor(x, 1, 64);
#endif
}

// LDS hand-off between the lanes of ONE wave: a wave's DS instructions execute in order and all 64 lanes issue them
// together, so a write by one lane is visible to a later read by another lane of the same wave without a barrier;
// only the compiler has to keep the order (the host emulator, one thread per lane, maps this to a wave barrier).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- 16-point DFT in registers: v[n] (n = 4 a + b) -> Y[k] (k = c + 4 d) left at v[4 c + d] (base-4 digit reversal).  Four radix-4
// butterflies at a time, written operation by operation across the four (no statement depends on the one in front of it).
// ROT2 (second stage, third butterfly): its third input still owes a factor -i (the twiddle exp(-2 pi i 4 / 16)), taken by operand modifiers.
template <int S0, int ST, bool STAGE2>      // butterfly j works on v[S0 j + ST i], i = 0 .. 3
__device__ __forceinline__ void dft4x4(cplx (&v)[16]) {
    cplx b0[4], b1[4], b2[4], t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool rot = STAGE2 && j == 2;
        b0[j] = rot ? cadd_rot(v[S0 * j], v[S0 * j + 2 * ST]) : cadd(v[S0 * j], v[S0 * j + 2 * ST]);
        b2[j] = cadd(v[S0 * j + ST], v[S0 * j + 3 * ST]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool rot = STAGE2 && j == 2;
        b1[j] = rot ? csub_rot(v[S0 * j], v[S0 * j + 2 * ST]) : csub(v[S0 * j], v[S0 * j + 2 * ST]);
        t[j] = csub(v[S0 * j + ST], v[S0 * j + 3 * ST]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[S0 * j] = cadd(b0[j], b2[j]);
        v[S0 * j + 2 * ST] = csub(b0[j], b2[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[S0 * j + ST] = cadd_rot(b1[j], t[j]);
        v[S0 * j + 3 * ST] = csub_rot(b1[j], t[j]);
    }
}
__device__ __forceinline__ void dft16(cplx (&v)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R = 0.70710678118654752f;
    const cplx w1 = {C1, -S1}, w2 = {R, -R}, w3 = {S1, -C1}, w6 = {R, R};
    dft4x4<1, 4, false>(v);                            // over a: v[4 c + b] = T[c][b]
    // T[c][b] *= exp(-2 pi i b c / 16)   (b c = 4: -i, left to the second stage's butterfly; -(R, R) and -(C1, -S1) by operand modifiers)
    const cplx t5 = cmul_a<false>(v[5], w1), t9 = cmul_a<false>(v[9], w2), t13 = cmul_a<false>(v[13], w3), t6 = cmul_a<false>(v[6], w2);
    const cplx t14 = cmul_a<true>(v[14], w6), t7 = cmul_a<false>(v[7], w3), t11 = cmul_a<true>(v[11], w6), t15 = cmul_a<true>(v[15], w1);
    v[5] = cmul_b<false>(v[5], w1, t5); v[9] = cmul_b<false>(v[9], w2, t9); v[13] = cmul_b<false>(v[13], w3, t13); v[6] = cmul_b<false>(v[6], w2, t6);
    v[14] = cmul_b<true>(v[14], w6, t14); v[7] = cmul_b<false>(v[7], w3, t7); v[11] = cmul_b<true>(v[11], w6, t11); v[15] = cmul_b<true>(v[15], w1, t15);
    dft4x4<4, 1, true>(v);                             // over b: v[4 c + d] = Y[c + 4 d]
}
__device__ __forceinline__ constexpr int rev16(int k) { return 4 * (k & 3) + (k >> 2); }    // where dft16 leaves Y[k]

// the 16 complex points (= 32 samples) of frame t that lane l of a 16-lane group packs: z[16 n1 + l], n1 = 0..15.
// INTERIOR (block-uniform: all 64 frames of the block lie inside the clip and the clip starts 8-byte aligned): plain 8-byte
// loads, 16 lanes = one 128-byte line; otherwise clamped (frames >= T take the last frame's data and are not stored) and
// reflect-padded sample by sample
template <bool INTERIOR>
__device__ __forceinline__ void mel_fetch(cplx (&xin)[16], const float* __restrict__ wsrc, int t, int T, int S, int l) {
    if (INTERIOR) {
        const float* p0 = wsrc + t * MEL_HOP - MEL_NFFT / 2 + 2 * l;
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            xin[n1] = *reinterpret_cast<const cplx*>(p0 + 32 * n1);
        }
        return;
    }
    const int tc = t < T ? t : T - 1;
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        const int p = 2 * (16 * n1 + l);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int i = tc * MEL_HOP + p + e - MEL_NFFT / 2;
            if (i < 0) i = -i;
            if (i >= S) i = 2 * (S - 1) - i;
            xin[n1][e] = wsrc[i];
        }
    }
}

constexpr int MEL_XROW = 18 * 8;                  // exchange tile: 16 rows (k1) of 16 complex, pitch 18 (b128 reads conflict-free)
constexpr int MEL_XFRAME = 16 * MEL_XROW;         // 2304 B per frame
constexpr int MEL_PWBINS = 272;                   // 257 bins + 15 zero bins (a 16-weight register window starting at the last band's first bin)

// Round 3 form: a wave transforms FOUR frames at a time, 16 lanes per frame and 16 complex points per lane, as 256 = 16 x 16:
// a 16-point DFT over n1 in registers (twiddles of the 16-point transform are literals), the W256^(n2 k1) factors (per-lane
// registers), ONE exchange through LDS (lane n2 writes column n2, lane k1 reads row k1), a 16-point DFT over n2 in registers.
// The real-FFT unpack needs Z[256 - k], which lives in lane 16 - l of the same group: one lane permutation per value.  The
// mel projection runs over the four frames at once (a band and a half band per lane, filter weights in registers).  Per frame and
// wave ~165 instructions (counted in the code object: ~650 per four frames) against ~800 for four radix-4 stages through LDS with
// one frame per wave.
__global__ __launch_bounds__(256, 2) void logmel_kernel(const float* __restrict__ wave_in, int S, int T,
                                                     const float* __restrict__ window,
                                                     const float* __restrict__ twiddle,   // [512][2] exp(-2 pi i k / 512)
                                                     const int32_t* __restrict__ fb_start,
                                                     const int32_t* __restrict__ fb_len,
                                                     const float* __restrict__ fb_w, int fb_stride, float log_scale,
                                                     float norm_mean, float norm_2std, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l = lane & 15, grp = lane >> 4;
    float* wtab = reinterpret_cast<float*>(smem);                     // [512] Hann window
    float* tw256 = wtab + 512;                                        // [16 k1][16 l][2] exp(-2 pi i l k1 / 256): lane l reads at l + 16 k1 (immediate offsets)
    float* tw512 = tw256 + 512;                                       // [256][2] exp(-2 pi i k / 512): lane l reads at l + 16 k2
    float* otile = tw512 + 512;                                       // [96][65]
    char* xch = reinterpret_cast<char*>(otile + MEL_BANDS * MEL_OUT_LD) + wv * 4 * MEL_XFRAME;          // per wave: 4 frames
    // the power spectra of the four frames reuse the exchange tiles (2304 >= 1040 bytes per frame; a wave-level sync apart)
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * MEL_FRAMES_PER_BLOCK;
    const float* wsrc = wave_in + (int64_t)b * S;
    const bool interior = t0 > 0 && (t0 + MEL_FRAMES_PER_BLOCK) * MEL_HOP + MEL_NFFT / 2 <= S && t0 + MEL_FRAMES_PER_BLOCK <= T &&
                          (reinterpret_cast<uintptr_t>(wsrc) & 7) == 0;      // block-uniform
    cplx xin[16];          // the lane's 16 packed points as loaded: (even, odd) sample = (re, im)
    // this lane's band (lane) and half band (64 + lane / 2, weight slots 8 (lane & 1) .. + 7): first bin, length
    const int mbnd = 64 + (lane >> 1);
    const int sa = fb_start[lane], na = fb_len[lane] < fb_stride ? fb_len[lane] : fb_stride;
    const int sb = fb_start[mbnd] + 8 * (lane & 1), nb = fb_len[mbnd] < fb_stride ? fb_len[mbnd] : fb_stride;
    // log10(1 + s x) -> z-norm as one multiply-add behind the hardware's log2 (v_log_f32, 1 ulp; the argument is >= 1)
    const float out_mul = 0.30102999566398120f / norm_2std, out_add = -norm_mean / norm_2std;
    const float ls4 = 0.25f * log_scale;         // the power spectra below are kept times four
    // tables: 128 threads copy the window, 128 the 512-point twiddles (16 bytes each), every thread gathers one 256-point twiddle
    if (threadIdx.x < 128) *reinterpret_cast<f32x4_t*>(wtab + 4 * threadIdx.x) = *reinterpret_cast<const f32x4_t*>(window + 4 * threadIdx.x);
    else *reinterpret_cast<f32x4_t*>(tw512 + 4 * (threadIdx.x - 128)) = *reinterpret_cast<const f32x4_t*>(twiddle + 4 * (threadIdx.x - 128));
    *reinterpret_cast<cplx*>(tw256 + 2 * threadIdx.x) =
        *reinterpret_cast<const cplx*>(twiddle + 4 * (((threadIdx.x & 15) * (threadIdx.x >> 4)) & 255));       // exp(-2 pi i l k1 / 256) at [16 k1 + l]
    // the first 8 filter weights of this lane's band and its half band's 8 (zero beyond the band: the host pads its rows with zeros)
    float wa[8], wb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        wa[i] = i < na ? fb_w[lane * fb_stride + i] : 0.0f;
        wb[i] = 8 * (lane & 1) + i < nb ? fb_w[mbnd * fb_stride + 8 * (lane & 1) + i] : 0.0f;
    }
    __syncthreads();

    // (one copy of the frame loop per fetch form: with both forms inside one loop the prefetch registers met in phi copies, 32 moves per pass)
    auto frames = [&](auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        mel_fetch<INTERIOR>(xin, wsrc, t0 + wv * 16 + grp, T, S, l);
        for (int quad = 0; quad < 4; ++quad) {
            const int tl0 = wv * 16 + quad * 4;      // first of this wave's four frames within the block
            // ---- framing (center=True, reflect padding of 256 samples) + window + complex packing: v[n1] = z[16 n1 + l]
            cplx v[16];
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const cplx w2 = *reinterpret_cast<const cplx*>(wtab + 2 * (16 * n1 + l));
                v[n1] = xin[n1] * w2;
            }
            if (quad + 1 < 4) {                       // the next four frames' samples fly while these are transformed
                mel_fetch<INTERIOR>(xin, wsrc, t0 + tl0 + 4 + grp, T, S, l);
            }
            // ---- 256 = 16 x 16: DFT over n1, twiddle, exchange, DFT over n2.  (The lane's table values are read a stage ahead of their use:
            // the packed-math statements are asm, which hipcc does not move loads across)
            cplx tw[15];
#pragma unroll
            for (int k1 = 1; k1 < 16; ++k1) tw[k1 - 1] = *reinterpret_cast<const cplx*>(tw256 + 2 * (16 * k1 + l));      // W256^(l k1)
            dft16(v);
            char* xf = xch + grp * MEL_XFRAME;
            *reinterpret_cast<cplx*>(xf + l * 8) = v[rev16(0)];
#pragma unroll
            for (int g = 0; g < 3; ++g) {             // k1 = 1 + 5 g .. 5 + 5 g: five products at a time, first halves then second halves
                cplx ta[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) ta[i] = cmul_a<false>(v[rev16(1 + 5 * g + i)], tw[5 * g + i]);
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    *reinterpret_cast<cplx*>(xf + (1 + 5 * g + i) * MEL_XROW + l * 8) = cmul_b<false>(v[rev16(1 + 5 * g + i)], tw[5 * g + i], ta[i]);
            }
            wave_lds_sync();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4_t q = *reinterpret_cast<const f32x4_t*>(xf + l * MEL_XROW + j * 16);
                v[2 * j] = cplx{q[0], q[1]};
                v[2 * j + 1] = cplx{q[2], q[3]};
            }
            cplx wu[16];
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) wu[k2] = *reinterpret_cast<const cplx*>(tw512 + 2 * (l + 16 * k2));          // exp(-2 pi i (l + 16 k2) / 512)
            dft16(v);                                 // Z[l + 16 k2] at v[rev16(k2)]
            // ---- unpack the real FFT: X[k] = E[k] + W512^k O[k], power spectrum for k = l + 16 k2 (and bin 256 from lane 0)
            wave_lds_sync();                          // every lane has read its row: the tiles become the power spectra
            // power spectra of the wave's four frames, bin-major: pw[bin][frame] -- the projection below takes one bin of all four frames
            // with a single 16-byte read; the 64 lanes of a store cover 64 consecutive floats.  Bins 257 .. 271 are zero: a band's
            // register-resident weights run past its last bin (weight 0) and must meet finite values there.
            float* pw = reinterpret_cast<float*>(xch);
            if (lane < 4 * (MEL_PWBINS - MEL_NBINS)) pw[4 * MEL_NBINS + lane] = 0.0f;
                // Z[256 - k] sits in lane 16 - l, slot 15 - k2 (lane 0 pairs with itself: slot (16 - k2) mod 16): the partner values of all 16 slots
            // first (selects and lane moves: plain statements, scheduled by hipcc), then the packed arithmetic four slots at a time
            cplx pz[16];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {          // eight slots at a time: selects, then the rotations, then the mirrors
                float sr[8], si[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k2 = 8 * hf + i;
                    const cplx za = v[rev16((16 - k2) & 15)], zb = v[rev16(15 - k2)];      // (selected per component: a select between
                    sr[i] = l == 0 ? MEL_RE(za) : MEL_RE(zb);                               //  two array elements would pin v[] to scratch)
                    si[i] = l == 0 ? MEL_IM(za) : MEL_IM(zb);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) { sr[i] = mel_row_next(sr[i], lane); si[i] = mel_row_next(si[i], lane); }
#pragma unroll
                for (int i = 0; i < 8; ++i) pz[8 * hf + i] = cplx{mel_row_mirror(sr[i], lane), mel_row_mirror(si[i], lane)};
            }
            if (l == 0) {
                const float x256 = 2.0f * (MEL_RE(v[rev16(0)]) - MEL_IM(v[rev16(0)]));      // 2 X[256] = 2 (Re Z[0] - Im Z[0])
                pw[4 * 256 + grp] = x256 * x256;
            }
            // 2 X[k] = (Z[k] + conj Z[N-k]) - i (Z[k] - conj Z[N-k]) W512^k: the halves are left out (the power comes out times four,
            // exactly; the factor rides in the log's scale below)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                cplx e[4], d[4], t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { e[i] = cadd_conj(v[rev16(4 * g + i)], pz[4 * g + i]); d[i] = csub_conj(v[rev16(4 * g + i)], pz[4 * g + i]); }
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = cmul_a<false>(d[i], wu[4 * g + i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) d[i] = cmul_b<false>(d[i], wu[4 * g + i], t[i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) e[i] = cadd_rot(e[i], d[i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = csqr2(e[i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) pw[4 * (l + 16 * (4 * g + i)) + grp] = MEL_RE(t[i]) + MEL_IM(t[i]);
            }
            wave_lds_sync();
            // ---- mel projection + logC + z-norm into the block's output tile, the wave's four frames at once: lane L forms band L from
            // 8 bins (the slaney bank's bands 0 .. 63 are 1 .. 6 bins long) and one HALF of band 64 + L / 2 (8 of its 16 weight slots: bands
            // 64 .. 95 are 6 .. 15 bins long), a bin of the four frames per 16-byte LDS read, the weights in registers (wa / wb, loaded in
            // front of the frame loop); the halves meet through one lane exchange.  Longer bands finish in a global-read loop (never with
            // this bank).  (Rounds 3 - 5a: one frame at a time, 1.5 bands per lane, a 4-byte LDS read per product and its wait in front of
            // every multiply-add: 724 of the ~1800 instructions a wave spent on four frames.)
            {
                const f32x4_t* pwq = reinterpret_cast<const f32x4_t*>(xch);
                f32x4_t acc_a = {0.0f, 0.0f, 0.0f, 0.0f}, acc_b = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x4_t pa = pwq[sa + i], pb = pwq[sb + i];
                    acc_a = __builtin_elementwise_fma(pa, f32x4_t{wa[i], wa[i], wa[i], wa[i]}, acc_a);
                    acc_b = __builtin_elementwise_fma(pb, f32x4_t{wb[i], wb[i], wb[i], wb[i]}, acc_b);
                }
#pragma unroll 1
                for (int i = 8; i < na; ++i) acc_a += pwq[sa + i] * fb_w[lane * fb_stride + i];
#pragma unroll
                for (int f = 0; f < 4; ++f) acc_b[f] += mel_lane_xor1(acc_b[f]);
                if ((lane & 1) == 0)
#pragma unroll 1
                    for (int i = 16; i < nb; ++i) acc_b += pwq[sb + i] * fb_w[mbnd * fb_stride + i];
                const int tl = wv * 16 + quad * 4;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    otile[lane * MEL_OUT_LD + tl + f] = __builtin_fmaf(mel_log2(__builtin_fmaf(acc_a[f], ls4, 1.0f)), out_mul, out_add);
                    if ((lane & 1) == 0)
                        otile[mbnd * MEL_OUT_LD + tl + f] = __builtin_fmaf(mel_log2(__builtin_fmaf(acc_b[f], ls4, 1.0f)), out_mul, out_add);
                }
            }
            wave_lds_sync();       // pw / the exchange tile are rewritten by the next four frames
        }
    };
    if (interior) frames(std::true_type{});
    else frames(std::false_type{});
    __syncthreads();
    // ---- coalesced store of the [96][64] tile: thread = frame tid & 63 of bands (tid >> 6) + 4 j (256-byte runs along T)
    {
        const int tl = threadIdx.x & 63, m0 = threadIdx.x >> 6;
        if (t0 + tl < T) {
            float* dst = out + ((int64_t)b * MEL_BANDS + m0) * T + t0 + tl;
            const float* src = otile + m0 * MEL_OUT_LD + tl;
#pragma unroll
            for (int j = 0; j < MEL_BANDS / 4; ++j) dst[(int64_t)4 * j * T] = src[4 * j * MEL_OUT_LD];
        }
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
    const int smem_bytes = (512 + 512 + 512 + MEL_BANDS * MEL_OUT_LD) * 4 + 4 * 4 * MEL_XFRAME;
    dim3 grid((T + MEL_FRAMES_PER_BLOCK - 1) / MEL_FRAMES_PER_BLOCK, B);
    static DeviceOnce once;                       // 70 KiB of dynamic LDS: above the 64 KiB a kernel gets without the attribute
    ensure_dynamic_lds(once, &logmel_kernel, smem_bytes);
    hipLaunchKernelGGL(logmel_kernel, grid, dim3(256), smem_bytes, (hipStream_t)stream, wave, S, T, window, twiddle,
                       fb_start, fb_len, fb_w, fb_stride, log_scale, norm_mean, norm_2std, out);
    return check_launch("maest_logmel");
}
