// probe: semantics of ds_read_b64_tr_b16 on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4i16 __attribute__((ext_vector_type(4)));
__global__ void probe(const int* __restrict__ lane_addr, uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;   // value = element index
    __syncthreads();
    const int off = lane_addr[threadIdx.x];   // element offset supplied by this lane
    v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(lds + off));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)r[j];
}
int main() {
    int h_addr[64]; uint16_t h_out[256];
    int *d_addr; uint16_t* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int test = 0; test < 3; ++test) {
        for (int l = 0; l < 64; ++l) {
            int g = l >> 4, q = l & 15;
            if (test == 0) h_addr[l] = g * 64 + q * 4;                   // contiguous 4x16 block per group
            if (test == 1) h_addr[l] = g * 1000 + (q / 4) * 100 + (q % 4) * 4;  // row stride 100 elems, 4 rows x 16 cols
            if (test == 2) h_addr[l] = l * 4;                                // fully linear over the wave
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("test %d\n", test);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d addr %4d -> %4d %4d %4d %4d\n", l, h_addr[l], h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]);
        }
    }
    return 0;
}
