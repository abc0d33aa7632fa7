// MFMA GEMM with fused epilogues for the MAEST ViT linears (reference: nn.Linear call sites
// models/maest.py:353,355,361,376 (qkv / proj), :197-199,203-206 (fc1 / GELU / fc2), :572 (head),
// the im2col form of nn.Conv2d :238-240, and their autograd dgrad / wgrad).
//
//   C[M,N] = epilogue( sum_k A[m,k] * B[n,k] )           ("NT": both operands k-contiguous)
//
// Design (gfx950): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave
// = 2x2 MFMA 32x32 tiles, 64 fp32 accumulators per lane); K is walked in 128-BYTE slices per
// row (64 bf16 or 32 fp32) so the bf16 perf path and the fp32 parity path share every address
// computation.  Global -> registers -> LDS staging with the next slice's loads issued before the
// MFMAs of the current one (double-buffered LDS, one barrier per slice).  LDS rows are padded
// 128 -> 144 bytes: ds_read_b128 of 16 consecutive rows then covers all 64 banks exactly once.
// Workgroup ids are remapped so that each XCD sweeps a contiguous range of tiles (n fastest)
// and re-reads the A panel / the weights from its own L2.
#include "common.h"

namespace maest {

constexpr int GEMM_BM = 128;
constexpr int GEMM_BN = 128;
constexpr int GEMM_ROWB = 128;   // payload bytes per tile row per K slice
constexpr int GEMM_PITCH = 144;  // padded LDS row pitch (bytes)
constexpr int GEMM_TILE_BYTES = GEMM_BM * GEMM_PITCH;      // 18432
constexpr int GEMM_SMEM_BYTES = 4 * GEMM_TILE_BYTES;       // A,B x 2 buffers = 73728

struct GemmParams {
    const char* A;
    const char* B;
    void* C;
    const float* bias;
    const void* aux_in;
    void* aux_out;
    int64_t lda, ldb, ldc, ld_aux;  // in elements
    int M, N, K;
    int out_dtype;   // MAEST_F32 / MAEST_BF16
    int epi;         // MAEST_EPI_*
    int tiles_m, tiles_n;
    int k_slices_per_split;  // K slices handled by one blockIdx.y
};

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

    f32x16_t acc[2][2];
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
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma_chunk<T>(acc[i][j], fa[i], fb[j]);
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

    // ---------------------------------------------------------------- epilogue
    const int col_l = lane & 31;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + col_l;
        if (col >= p.N) continue;
        const float bias = (p.bias != nullptr && blockIdx.y == 0) ? p.bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + frag_row(r, lane);
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bias;
                const int64_t ci = (int64_t)row * p.ldc + col;
                const int64_t xi = (int64_t)row * p.ld_aux + col;
                switch (p.epi) {
                    case MAEST_EPI_NONE:
                        break;
                    case MAEST_EPI_GELU:
                        if (p.aux_out != nullptr) {
                            if (p.out_dtype == MAEST_BF16) reinterpret_cast<bf16_t*>(p.aux_out)[xi] = f2bf(v);
                            else reinterpret_cast<float*>(p.aux_out)[xi] = v;
                        }
                        v = gelu_f(v);
                        break;
                    case MAEST_EPI_RESIDUAL:
                        v += reinterpret_cast<const float*>(p.aux_in)[xi];
                        break;
                    case MAEST_EPI_DGELU: {
                        const float pre = (p.out_dtype == MAEST_BF16)
                                              ? bf2f(reinterpret_cast<const bf16_t*>(p.aux_in)[xi])
                                              : reinterpret_cast<const float*>(p.aux_in)[xi];
                        v *= gelu_grad_f(pre);
                        break;
                    }
                    case MAEST_EPI_ATOMIC:
                        unsafeAtomicAdd(reinterpret_cast<float*>(p.C) + ci, v);
                        continue;
                    default:
                        break;
                }
                if (p.out_dtype == MAEST_BF16) reinterpret_cast<bf16_t*>(p.C)[ci] = f2bf(v);
                else reinterpret_cast<float*>(p.C)[ci] = v;
            }
        }
    }
}

template <typename T>
static int launch_gemm(GemmParams& p, int split_k, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<T>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_SMEM_BYTES);
        attr_done = true;
    }
    dim3 grid(p.tiles_m * p.tiles_n, split_k, 1);
    hipLaunchKernelGGL(gemm_nt_kernel<T>, grid, dim3(256), GEMM_SMEM_BYTES, stream, p);
    return check_launch("maest_gemm_nt");
}

}  // namespace maest

using namespace maest;

extern "C" int maest_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, int in_dtype,
                             void* C, int64_t ldc, int out_dtype, int M, int N, int K,
                             const float* bias, int epi, const void* aux_in, void* aux_out,
                             int64_t ld_aux, int split_k, void* stream) {
    MAEST_REQUIRE(A && B && C, "maest_gemm_nt: null operand");
    MAEST_REQUIRE(M > 0 && N > 0 && K > 0, "maest_gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
    MAEST_REQUIRE(in_dtype == MAEST_F32 || in_dtype == MAEST_BF16, "maest_gemm_nt: bad in_dtype %d", in_dtype);
    MAEST_REQUIRE(out_dtype == MAEST_F32 || out_dtype == MAEST_BF16, "maest_gemm_nt: bad out_dtype %d", out_dtype);
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
    MAEST_REQUIRE(epi != MAEST_EPI_DGELU || aux_in, "maest_gemm_nt: dgelu epilogue needs aux_in");
    GemmParams p;
    p.A = (const char*)A; p.B = (const char*)B; p.C = C;
    p.bias = bias; p.aux_in = aux_in; p.aux_out = aux_out;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ld_aux = ld_aux;
    p.M = M; p.N = N; p.K = K;
    p.out_dtype = out_dtype; p.epi = epi;
    p.tiles_m = (M + GEMM_BM - 1) / GEMM_BM;
    p.tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
    const int total = K / ks;
    if (split_k > total) split_k = total;
    p.k_slices_per_split = (total + split_k - 1) / split_k;
    split_k = (total + p.k_slices_per_split - 1) / p.k_slices_per_split;
    return in_dtype == MAEST_BF16 ? launch_gemm<bf16_t>(p, split_k, (hipStream_t)stream)
                                  : launch_gemm<float>(p, split_k, (hipStream_t)stream);
}
