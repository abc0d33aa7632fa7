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
#include "../../maest_amd/csrc/common.h"
__device__ long long* g_prof = nullptr;
#define STAMP(i) if (tid == 0 && blockIdx.x < 256) g_prof_local[i] = clock64();

namespace maest {

constexpr int G2_ROWB = 64;                 // bytes per row per K slice
constexpr int G2_TILE = 256 * G2_ROWB;      // 16384: one operand tile of one slice
constexpr int G2_STAGES = 4;
constexpr int G2_SMEM = G2_STAGES * 2 * G2_TILE;   // 131072
// s_waitcnt immediate (gfx9 encoding): vmcnt = N, expcnt / lgkmcnt = "no wait"
#define MAEST_WAIT_VMCNT(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (((N) >> 4) << 14) | 0x0F70)

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
};

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
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[nt][mt0 + mi][4 * g + e] + b4[e];
                    if (GMODE != 0) {
                        float gv, dv;
                        gelu_pair<EXACT>(v[e], gv, dv);
                        v[e] = GMODE == 1 ? gv : dv;
                    }
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
                for (int e = 0; e < 4; ++e) gelu_pair<EXACT>(acc[nt][mt0 + mi][4 * g + e] + b4[e], v[e], d[e]);
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

template <int OSZ, int MODE, int ROWS = Epi256<OSZ>::ROWS>
__device__ __forceinline__ void drain256(const char* smem, void* dst, int64_t ld, const void* aux, int64_t ld_aux,
                                         int mbase, int n0, int M, int N, int tid) {
    using E = Epi256<OSZ>;
#pragma unroll 4
    for (int c = tid; c < ROWS * E::CPR; c += 512) {
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
                    const float lo = bf2f((bf16_t)(vw & 0xffffu)) * bf2f((bf16_t)(rw & 0xffffu));
                    const float hi = bf2f((bf16_t)(vw >> 16)) * bf2f((bf16_t)(rw >> 16));
                    v[e] = pack_bf2(lo, hi);
                }
            }
        }
        *reinterpret_cast<chunk16*>(reinterpret_cast<char*>(dst) + ((int64_t)gm * ld + gn) * OSZ) = v;
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
            drain256<OSZ, 0, PR>(smem, p.C, p.ldc, nullptr, 0, mbase, n0, p.M, p.N, tid);
            drain256<OSZ, 0, PR>(smem + REGION, p.aux_out, p.ld_aux, nullptr, 0, mbase, n0, p.M, p.N, tid);
            __syncthreads();
        }
        return;
    }
#pragma unroll
    for (int ps = 0; ps < E::PASSES; ++ps) {
        const int pwm = ps / (4 / E::MT);
        const int mt0 = (ps % (4 / E::MT)) * E::MT;
        const int mbase = m0 + pwm * 128 + mt0 * 32;
        const bool mine = (wm == pwm);   // wave-uniform
        if (p.epi == MAEST_EPI_GELU) {
            if (p.aux_out != nullptr) {
                if (mine) stage256<OSZ, 2, EXACT>(smem, acc, p.bias, n0, p.N, mt0, wn, lane);
                __syncthreads();
                drain256<OSZ, 0>(smem, p.aux_out, p.ld_aux, nullptr, 0, mbase, n0, p.M, p.N, tid);
                __syncthreads();
            }
            if (mine) stage256<OSZ, 1, EXACT>(smem, acc, p.bias, n0, p.N, mt0, wn, lane);
            __syncthreads();
            drain256<OSZ, 0>(smem, p.C, p.ldc, nullptr, 0, mbase, n0, p.M, p.N, tid);
        } else {
            if (mine) stage256<OSZ, 0, EXACT>(smem, acc, p.bias, n0, p.N, mt0, wn, lane);
            __syncthreads();
            if (p.epi == MAEST_EPI_RESIDUAL)
                drain256<OSZ, 1>(smem, p.C, p.ldc, p.aux_in, p.ld_aux, mbase, n0, p.M, p.N, tid);
            else if (p.epi == MAEST_EPI_MUL)
                drain256<OSZ, 2>(smem, p.C, p.ldc, p.aux_in, p.ld_aux, mbase, n0, p.M, p.N, tid);
            else
                drain256<OSZ, 0>(smem, p.C, p.ldc, nullptr, 0, mbase, n0, p.M, p.N, tid);
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(512) void gemm_nt256_kernel(Gemm256Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;          // 0..7
    long long g_prof_local[6] = {0,0,0,0,0,0};
    STAMP(0)
    const int wm = wave >> 2, wn = wave & 3;
    const int h = lane >> 5;

    // persistent workgroups (one per CU): block b (XCD b % 8, observed) walks tiles
    //   round * gridDim + (b % 8) * (gridDim / 8) + b / 8
    // so that the workgroups of one XCD work on neighbouring tiles (same A panel) at the same time.
    const int nwg = p.tiles_m * p.tiles_n;
    const int per_xcd = gridDim.x >> 3;
    const int wg0 = gridDim.x >= 8 ? (int)((blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) : (int)blockIdx.x;
    for (int wg = wg0; wg < nwg; wg += gridDim.x) {
    const int tile_m = wg / p.tiles_n;
    const int tile_n = wg - tile_m * p.tiles_n;
    const int m0 = tile_m * 256, n0 = tile_n * 256;

    constexpr int ELT = (int)sizeof(T);
    constexpr int KS = G2_ROWB / ELT;        // 32 / 16 elements per slice
    const int nslices = p.K / KS;

    // LDS-DMA map: wave-instruction (wave, i) fills rows [(wave*2+i)*16, +16) of a tile; lane -> (row, position)
    const char* a_src[2];
    const char* b_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 16 + (lane >> 2);
        const int csrc = (lane & 3) ^ ((r >> 2) & 3);        // source-side swizzle
        int ra = m0 + r;
        if (ra > p.M - 1) ra = p.M - 1;
        int rb = n0 + r;
        if (rb > p.N - 1) rb = p.N - 1;
        a_src[i] = p.A + (int64_t)ra * p.lda * ELT + csrc * 16;
        b_src[i] = p.B + (int64_t)rb * p.ldb * ELT + csrc * 16;
    }
    const int dma_off = wave * 2 * 1024;

    f32x16_t acc[2][4];   // [nt][mt]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // half h2 (0/1) of the 4 LDS-DMA instructions this wave owes to slice s: one A piece + one B piece
    auto issue_half = [&](int s, int h2) {
        const int sc = s < nslices ? s : nslices - 1;        // past-the-end issues re-load the last slice into a dead
        char* la = smem + (s & (G2_STAGES - 1)) * 2 * G2_TILE + dma_off;   // buffer: keeps the vmcnt arithmetic uniform
        char* lb = la + G2_TILE;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[h2] + (int64_t)sc * G2_ROWB),
                                         (__attribute__((address_space(3))) void*)(la + h2 * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[h2] + (int64_t)sc * G2_ROWB),
                                         (__attribute__((address_space(3))) void*)(lb + h2 * 1024), 16, 0, 0);
    };
    auto issue = [&](int s) { issue_half(s, 0); issue_half(s, 1); };

    int a_off[4], b_off[2], a_swz[4], b_swz[2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int row = wm * 128 + mt * 32 + (lane & 31);
        a_off[mt] = row * G2_ROWB;
        a_swz[mt] = (row >> 2) & 3;
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int row = wn * 64 + nt * 32 + (lane & 31);
        b_off[nt] = row * G2_ROWB;
        b_swz[nt] = (row >> 2) & 3;
    }

    // ---- main loop: two phases per slice, LOAD (LDS -> fragment registers, issue the DMA of slice s+3) and
    // COMPUTE (16 MFMAs, no memory traffic), separated by raw barriers.  The second wave group (wm == 1)
    // runs ONE BARRIER BEHIND the first, so on every SIMD one wave is in COMPUTE while its partner is in
    // LOAD: the matrix pipe never waits for ds_read / DMA issue, and those never wait for the pipe.
    //   barrier 2s   : A: LOAD(s)     B: COMPUTE(s-1)
    //   barrier 2s+1 : A: COMPUTE(s)  B: LOAD(s)
    // Slice s is resident before anybody reads it: every wave ends LOAD(s-1) with vmcnt(8) (slices s+1, s+2
    // may still fly) and the barrier(s) in between publish that to the other group.  DMA of slice s+3
    // overwrites buffer (s-1)&3, whose last reader (B's LOAD(s-1)) finished before barrier 2s.
    STAMP(1)
    issue(0);
    issue(1);
    issue(2);
    MAEST_WAIT_VMCNT(8);
    __builtin_amdgcn_s_barrier();
    STAMP(2)
    if (wm == 1) __builtin_amdgcn_s_barrier();          // stagger (wave-uniform)
    chunk16 fa[2][4], fb[2][2];
    for (int s = 0; s < nslices; ++s) {
        const char* la = smem + (s & (G2_STAGES - 1)) * 2 * G2_TILE;
        const char* lb = la + G2_TILE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int kc = 2 * ks + h;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                fa[ks][mt] = *reinterpret_cast<const chunk16*>(la + a_off[mt] + ((kc ^ a_swz[mt]) << 4));
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                fb[ks][nt] = *reinterpret_cast<const chunk16*>(lb + b_off[nt] + ((kc ^ b_swz[nt]) << 4));
        }
        issue(s + 3);
        __builtin_amdgcn_s_waitcnt(0x0078);   // vmcnt(8) lgkmcnt(0): slice s+1 resident, fragments in registers
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) mma_chunk<T>(acc[nt][mt], fb[ks][nt], fa[ks][mt]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();          // un-stagger
    MAEST_WAIT_VMCNT(0);   // drain the past-the-end loads before LDS is reused
    __syncthreads();   // every wave is done with the operand buffers: LDS becomes the C staging area
    STAMP(3)
    if (p.out_dtype == MAEST_BF16) epilogue256<2, sizeof(T) == 4>(smem, acc, p, m0, n0, wm, wn, lane, tid);
    else epilogue256<4, sizeof(T) == 4>(smem, acc, p, m0, n0, wm, wn, lane, tid);
    STAMP(4)
    if (tid == 0 && blockIdx.x < 256 && wg == wg0) { for (int i = 0; i < 5; ++i) g_prof[blockIdx.x * 8 + i] = g_prof_local[i]; }
    }   // tile loop (epilogue256 ends with a barrier: LDS is free for the next tile's DMA)
}

template <typename T>
static int launch256(Gemm256Params& p, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt256_kernel<T>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G2_SMEM);
        attr_done = true;
    }
    int grid = p.tiles_m * p.tiles_n;
    if (grid > 256) grid = 256;          // persistent: one workgroup per CU
    if (grid >= 8) grid &= ~7;           // keep the per-XCD tile walk regular
    hipLaunchKernelGGL(gemm_nt256_kernel<T>, dim3(grid), dim3(512), G2_SMEM, stream, p);
    return check_launch("maest_gemm_nt(256)");
}

// Called by maest_gemm_nt for large, 16-byte-friendly problems.  Returns -1 when the shape does not qualify.
int gemm_nt256_try(const void* A, int64_t lda, const void* B, int64_t ldb, int in_dtype, void* C, int64_t ldc,
                   int out_dtype, int M, int N, int K, const float* bias, int epi, const void* aux_in, void* aux_out,
                   int64_t ld_aux, hipStream_t stream) {
    if (epi == MAEST_EPI_ATOMIC) return -1;
    if (M < 512 || N < 256 || (N % 256) != 0) return -1;
    if ((K * (in_dtype == MAEST_BF16 ? 2 : 4)) % G2_ROWB != 0) return -1;
    Gemm256Params p;
    p.A = (const char*)A; p.B = (const char*)B; p.C = C;
    p.bias = bias; p.aux_in = aux_in; p.aux_out = aux_out;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ld_aux = ld_aux;
    p.M = M; p.N = N; p.K = K; p.out_dtype = out_dtype; p.epi = epi;
    p.tiles_m = (M + 255) / 256;
    p.tiles_n = N / 256;
    return in_dtype == MAEST_BF16 ? launch256<bf16_t>(p, stream) : launch256<float>(p, stream);
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
    const v4i16b_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16b_t*)(p));
    const v4i16b_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16b_t*)(p + 4 * 512));
    const chunk8 l2 = __builtin_bit_cast(chunk8, lo), h2 = __builtin_bit_cast(chunk8, hi);   // no repacking
    chunk16 c;
    c[0] = l2[0]; c[1] = l2[1]; c[2] = h2[0]; c[3] = h2[1];
    return c;
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

struct GemmTn256Params {
    const char* A;
    const char* B;
    float* C;
    float* colsum;
    int64_t lda, ldb, ldc;
    int M, N, K;
    int tiles_m, tiles_n;
    int k_slices_per_split;
};

template <typename T>
__global__ __launch_bounds__(512) void gemm_tn256_kernel(GemmTn256Params p) {
    using C = Tn256<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;

    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int tile_i = wg / p.tiles_n;
    const int tile_j = wg - tile_i * p.tiles_n;
    const int i0 = tile_i * 256, j0 = tile_j * 256;

    const int total_slices = p.K / C::KS;
    const int s_begin = blockIdx.y * p.k_slices_per_split;
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
    float cs[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // bias-gradient partials: column lane&31 of i-block a, this lane's k half
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    const bool do_colsum = (p.colsum != nullptr) && (tile_j == 0) && (wn == 0);   // wave-uniform

    auto issue = [&](int s) {
        const int sc = s < nslices ? s : nslices - 1;
        char* la = smem + (s & (G2_STAGES - 1)) * 2 * G2_TILE + dma_off;
        char* lb = la + G2_TILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + sc * a_step),
                                             (__attribute__((address_space(3))) void*)(la + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + sc * b_step),
                                             (__attribute__((address_space(3))) void*)(lb + i * 1024), 16, 0, 0);
        }
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
#pragma unroll
            for (int a = 0; a < 4; ++a) fa[ks][a] = frag_tn256<T>(la, ks, wm * 128 + a * 32, lane);
#pragma unroll
            for (int b = 0; b < 2; ++b) fb[ks][b] = frag_tn256<T>(lb, ks, wn * 64 + b * 32, lane);
        }
        issue(s + 3);
        __builtin_amdgcn_s_waitcnt(0x0078);   // vmcnt(8) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) mma_chunk<T>(acc[a][b], fa[ks][a], fb[ks][b]);   // D rows = i, cols = j
        }
        __builtin_amdgcn_s_setprio(0);
        if (do_colsum) {   // the fragments already hold A[k][i] for (i = lane&31, 8 or 4 k's): sum them on the VALU
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t w = fa[ks][a][e];
                        if (C::ELT == 2) cs[a] += bf2f((bf16_t)(w & 0xffffu)) + bf2f((bf16_t)(w >> 16));
                        else cs[a] += u2f(w);
                    }
        }
        __builtin_amdgcn_s_barrier();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
    MAEST_WAIT_VMCNT(0);

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
    if (do_colsum) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float tot = cs[a] + __shfl_xor(cs[a], 32, 64);     // merge the two k halves
            if (lane < 32) unsafeAtomicAdd(p.colsum + i0 + wm * 128 + a * 32 + lane, tot);
        }
    }
}

template <typename T>
static int launch_tn256(GemmTn256Params& p, int split_k, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn256_kernel<T>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G2_SMEM);
        attr_done = true;
    }
    hipLaunchKernelGGL(gemm_tn256_kernel<T>, dim3(p.tiles_m * p.tiles_n, split_k), dim3(512), G2_SMEM, stream, p);
    return check_launch("maest_gemm_tn(256)");
}

// Called by maest_gemm_tn; returns -1 when the shape does not qualify.  split_k <= 0 = automatic.
int gemm_tn256_try(const void* A, int64_t lda, const void* B, int64_t ldb, int dtype, float* C, int64_t ldc, int M,
                   int N, int K, float* colsum, int split_k, hipStream_t stream) {
    const int ks = dtype == MAEST_BF16 ? Tn256<bf16_t>::KS : Tn256<float>::KS;
    if ((M % 256) != 0 || (N % 256) != 0 || (K % ks) != 0 || K < 8 * ks) return -1;
    if ((int64_t)M * N < (int64_t)12 * 65536) return -1;   // few output tiles: the 128x128 kernel splits K finer
    GemmTn256Params p;
    p.A = (const char*)A; p.B = (const char*)B; p.C = C; p.colsum = colsum;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.tiles_m = M / 256;
    p.tiles_n = N / 256;
    const int total = K / ks;
    const int tiles = p.tiles_m * p.tiles_n;
    if (split_k <= 0) split_k = (256 + tiles - 1) / tiles;   // one workgroup per CU
    if (split_k > total / 4) split_k = total / 4 > 0 ? total / 4 : 1;
    p.k_slices_per_split = (total + split_k - 1) / split_k;
    split_k = (total + p.k_slices_per_split - 1) / p.k_slices_per_split;
    return dtype == MAEST_BF16 ? launch_tn256<bf16_t>(p, split_k, stream) : launch_tn256<float>(p, split_k, stream);
}

}  // namespace maest
namespace maest { void set_error(const char*, ...) {} int check_launch(const char*) { return hipGetLastError() != hipSuccess; } }
#include <stdio.h>
#include <vector>
int main() {
    const int M = 74240, N = 3072;
    for (int K : {64, 768}) {
        void *A, *B, *C; long long* prof;
        hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2);
        hipMalloc(&prof, 256 * 8 * 8);
        hipMemset(A, 0, (size_t)M * K * 2); hipMemset(B, 0, (size_t)N * K * 2);
        hipMemcpyToSymbol(HIP_SYMBOL(g_prof), &prof, sizeof(prof));
        for (int it = 0; it < 3; ++it)
            maest::gemm_nt256_try(A, K, B, K, MAEST_BF16, C, N, MAEST_BF16, M, N, K, nullptr, MAEST_EPI_NONE, nullptr, nullptr, 0, 0);
        hipDeviceSynchronize();
        std::vector<long long> h(256 * 8);
        hipMemcpy(h.data(), prof, 256 * 8 * 8, hipMemcpyDeviceToHost);
        double s[4] = {0, 0, 0, 0};
        for (int b = 0; b < 256; ++b) for (int i = 0; i < 4; ++i) s[i] += (double)(h[b * 8 + i + 1] - h[b * 8 + i]);
        printf("K=%d  avg cycles (first tile of each WG): setup %.0f  prologue(fill+wait) %.0f  main loop %.0f  epilogue %.0f\n", K, s[0] / 256, s[1] / 256, s[2] / 256, s[3] / 256);
    }
    return 0;
}

