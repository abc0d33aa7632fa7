// times gemm_nt256w (the full-line 256x256 NT kernel) with ingredients of its main loop removed (MAEST_ABLATE_* hooks
// in gemm256.hip), on RANDOM bf16 operands (zero-filled operands run at a higher clock and flatter the numbers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
namespace maest {
int gemm_nt256_try(const void* A, int64_t lda, const void* B, int64_t ldb, int in_dtype, void* C, int64_t ldc,
                   int out_dtype, int M, int N, int K, const float* bias, int epi, const void* aux_in, void* aux_out,
                   int64_t ld_aux, hipStream_t stream, float* rowdot, int ntok);
}
static void fill(void* dev, size_t n) {
    std::vector<uint16_t> h(n);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const float f = ((float)(s >> 40) / 8388608.0f) - 1.0f;      // uniform [-1, 1)
        uint32_t u; memcpy(&u, &f, 4);
        h[i] = (uint16_t)((u + 0x8000u) >> 16);
    }
    hipMemcpy(dev, h.data(), n * 2, hipMemcpyHostToDevice);
}
int main(int argc, char** argv) {
    const char* tag = argc > 1 ? argv[1] : "?";
    struct Shape { int M, N, K; } shapes[] = {{65536, 4096, 4096}, {74240, 768, 3072}, {74240, 2304, 768}, {74240, 768, 768},
                                          {65536, 768, 3072}, {65536, 768, 768}, {65536, 2304, 768}, {65536, 3072, 768}, {74240, 3072, 768}};
    void *A, *B, *C;
    hipMalloc(&A, (size_t)74240 * 4096 * 2); hipMalloc(&B, (size_t)4096 * 4096 * 2); hipMalloc(&C, (size_t)74240 * 4096 * 2);
    fill(A, (size_t)74240 * 4096); fill(B, (size_t)4096 * 4096);
    for (auto sh : shapes) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) maest::gemm_nt256_try(A, sh.K, B, sh.K, 1, C, sh.N, 1, sh.M, sh.N, sh.K, nullptr, 0, nullptr, nullptr, 0, 0, nullptr, 0);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) maest::gemm_nt256_try(A, sh.K, B, sh.K, 1, C, sh.N, 1, sh.M, sh.N, sh.K, nullptr, 0, nullptr, nullptr, 0, 0, nullptr, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-28s M=%6d N=%5d K=%5d: %8.3f ms  %7.1f TF/s-equivalent\n", tag, sh.M, sh.N, sh.K, ms, 2.0 * sh.M * sh.N * sh.K / ms / 1e9);
    }
    return 0;
}
