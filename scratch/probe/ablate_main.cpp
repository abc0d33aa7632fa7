// times gemm_nt256 (256x256 kernel) main loop with ingredients removed (see MAEST_ABLATE_* in gemm256.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
namespace maest {
int gemm_nt256_try(const void* A, int64_t lda, const void* B, int64_t ldb, int in_dtype, void* C, int64_t ldc,
                   int out_dtype, int M, int N, int K, const float* bias, int epi, const void* aux_in, void* aux_out,
                   int64_t ld_aux, hipStream_t stream);
}
namespace maest {
int gemm_tn256_try(const void* A, int64_t lda, const void* B, int64_t ldb, int dtype, float* C, int64_t ldc, int M,
                   int N, int K, float* colsum, int split_k, hipStream_t stream);
}
static int tn_main(const char* tag) {
    const int M = 3072, N = 768;
    void *A, *B; float* C;
    const int Kmax = 74240 * 2;
    hipMalloc(&A, (size_t)Kmax * M * 2); hipMalloc(&B, (size_t)Kmax * N * 2); hipMalloc(&C, (size_t)M * N * 4);
    hipMemset(A, 0, (size_t)Kmax * M * 2); hipMemset(B, 0, (size_t)Kmax * N * 2); hipMemset(C, 0, (size_t)M * N * 4);
    for (int K : {7168, 74240, 148480}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 2; ++i) maest::gemm_tn256_try(A, M, B, N, 1, C, N, M, N, K, nullptr, 0, 0);
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) maest::gemm_tn256_try(A, M, B, N, 1, C, N, M, N, K, nullptr, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("TN %s K=%d: %.3f ms  %.1f TF/s-equivalent\n", tag, K, ms, 2.0 * M * N * K / ms / 1e9);
    }
    return 0;
}
int main(int argc, char** argv) {
    if (argc > 2) return tn_main(argv[1]);
    const int M = 74240, N = 3072;
    void *A, *B, *C;
    hipMalloc(&A, (size_t)M * 6144 * 2); hipMalloc(&B, (size_t)N * 6144 * 2); hipMalloc(&C, (size_t)M * N * 2);
    hipMemset(A, 0, (size_t)M * 6144 * 2); hipMemset(B, 0, (size_t)N * 6144 * 2);
    for (int K : {768, 1536, 3072, 6144}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 2; ++i) maest::gemm_nt256_try(A, K, B, K, 1, C, N, 1, M, N, K, nullptr, 0, nullptr, nullptr, 0, 0);
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) maest::gemm_nt256_try(A, K, B, K, 1, C, N, 1, M, N, K, nullptr, 0, nullptr, nullptr, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%s K=%d: %.3f ms  %.1f TF/s-equivalent\n", argv[1], K, ms, 2.0 * M * N * K / ms / 1e9);
    }
    return 0;
}
