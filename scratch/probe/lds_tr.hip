// LDS read throughput: ds_read_b64_tr_b16 (TN fragment pattern) vs ds_read_b128 (NT fragment pattern)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef short v4i16 __attribute__((ext_vector_type(4)));
template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k(int iters, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 32768 / 4; i += NW * 64) ((uint32_t*)smem)[i] = i;
    __syncthreads();
    uint32_t acc = 0;
    const int h = lane >> 5, g16 = (lane >> 4) & 1, q = lane & 15;
    int off[12];
    for (int f = 0; f < 6; ++f) {
        if (MODE == 0) {        // TN pattern: tile [32 tokens][256 features] bf16, 512 B rows, swizzled chunks
            const int iblk = (wave % 4) * 64 % 256 + (f % 4) * 32 % 256;
            for (int ks = 0; ks < 2; ++ks) {
                const int row = ks * 16 + 8 * h + (q >> 2);
                const int cb = (iblk + 16 * g16 + 4 * (q & 3)) * 2;
                const int pb = ((((cb >> 4) ^ ((row & 3) << 2))) << 4) | (cb & 15);
                off[f * 2 + ks] = row * 512 + pb;
            }
        } else {                // NT pattern: tile [256 rows][64 B], chunk swizzle
            const int row = ((wave * 32 + f * 32) % 256) + (lane & 31);
            for (int ks = 0; ks < 2; ++ks) off[f * 2 + ks] = row * 64 + (((2 * ks + h) ^ ((row >> 2) & 3)) << 4);
        }
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < 12; ++f) {
            if (MODE == 0) {
                const char* p = smem + off[f];
                v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(p));
                v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(p + 4 * 512));
                acc ^= (uint32_t)lo[0] ^ (uint32_t)hi[3];
            } else {
                u4 v = *(const u4*)(smem + off[f]);
                acc ^= v[0] ^ v[3];
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }
    if (acc == 0x1234567u) sink[0] = acc;
}
template <int MODE, int NW>
void run(const char* name, uint32_t* sink) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NW><<<256, NW * 64, 32768>>>(100, sink);
    hipEventRecord(e0);
    k<MODE, NW><<<256, NW * 64, 32768>>>(iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)iters * 12 * (MODE == 0 ? 2 : 1) * NW;     // wave-instructions per CU
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-34s waves/CU=%d: %.2f clk per wave-instruction, %.1f B/clk/CU\n", name, NW, cyc / instr,
           instr * (MODE == 0 ? 512 : 1024) / cyc);
}
int main() {
    uint32_t* sink; hipMalloc(&sink, 64);
    run<0, 4>("ds_read_b64_tr_b16 (TN pattern)", sink);
    run<0, 8>("ds_read_b64_tr_b16 (TN pattern)", sink);
    run<1, 4>("ds_read_b128 (NT pattern)", sink);
    run<1, 8>("ds_read_b128 (NT pattern)", sink);
    return 0;
}
