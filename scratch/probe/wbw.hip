// write-bandwidth probe: how fast can 256 CUs push C tiles to HBM, by store pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int NT> __global__ __launch_bounds__(512) void fill(u4* p, size_t n16) {
    size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
    size_t stride = (size_t)gridDim.x * 512;
    u4 v = {1u, 2u, 3u, (uint32_t)i};
    for (; i < n16; i += stride) {
        if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
    }
}
// one WG = one 256 x (256 bf16 = 512 B) tile of a [M][N] bf16 matrix; 512 threads, 32 lanes per row
template <int NT> __global__ __launch_bounds__(512) void tile(char* p, int M, int N) {
    int tn = N / 256;
    int tm = blockIdx.x / tn, tc = blockIdx.x % tn;
    char* base = p + ((size_t)tm * 256) * N * 2 + (size_t)tc * 512;
    u4 v = {1u, 2u, 3u, (uint32_t)threadIdx.x};
    int lane32 = threadIdx.x & 31, r0 = threadIdx.x >> 5;   // 16 rows per pass
    for (int r = r0; r < 256; r += 16) {
        u4* d = (u4*)(base + (size_t)r * N * 2 + lane32 * 16);
        if (NT) __builtin_nontemporal_store(v, d); else *d = v;
    }
}
__global__ __launch_bounds__(512) void copyk(const u4* __restrict__ s, u4* __restrict__ d, size_t n16) {
    size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
    size_t stride = (size_t)gridDim.x * 512;
    for (; i < n16; i += stride) __builtin_nontemporal_store(s[i], d + i);
}
__global__ __launch_bounds__(512) void readk(const u4* __restrict__ s, u4* __restrict__ d, size_t n16) {
    size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
    size_t stride = (size_t)gridDim.x * 512;
    u4 a = {0, 0, 0, 0};
    for (; i < n16; i += stride) a += s[i];
    if (a.x == 0x12345678u) d[0] = a;
}
#define T(name, bytes, ...)                                                              \
    do {                                                                                 \
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);                     \
        for (int w = 0; w < 2; ++w) { __VA_ARGS__; }                                     \
        hipEventRecord(e0);                                                              \
        for (int w = 0; w < 10; ++w) { __VA_ARGS__; }                                    \
        hipEventRecord(e1); hipEventSynchronize(e1);                                     \
        float ms; hipEventElapsedTime(&ms, e0, e1);                                      \
        printf("%-40s %8.1f us  %7.2f TB/s\n", name, ms * 100, (bytes) / (ms / 10 * 1e-3) / 1e12); \
    } while (0)
int main() {
    size_t bytes = (size_t)143360 * 2304 * 2;   // the qkv output at B=256, N=560: 660 MB
    char *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 1, bytes);
    size_t n16 = bytes / 16;
    for (int g : {256, 512, 1024, 2048, 8192}) {
        char nm[64];
        snprintf(nm, 64, "fill plain grid=%d", g); T(nm, bytes, (fill<0><<<g, 512>>>((u4*)a, n16)));
        snprintf(nm, 64, "fill nt    grid=%d", g); T(nm, bytes, (fill<1><<<g, 512>>>((u4*)a, n16)));
    }
    int M = 143360, N = 2304;
    T("tile plain (560x9 tiles)", bytes, (tile<0><<<(M / 256) * (N / 256), 512>>>(a, M, N)));
    T("tile nt", bytes, (tile<1><<<(M / 256) * (N / 256), 512>>>(a, M, N)));
    N = 768; 
    T("tile nt N=768", (size_t)M * N * 2, (tile<1><<<(M / 256) * (N / 256), 512>>>(a, M, N)));
    T("copy nt", 2 * bytes, (copyk<<<2048, 512>>>((u4*)a, (u4*)b, n16)));
    T("read", bytes, (readk<<<2048, 512>>>((u4*)a, (u4*)b, n16)));
    return 0;
}
