// How many clocks does a CU's store path take per store / atomic INSTRUCTION, by width and pattern?  ("about one per 64 clk,
// whatever its width".)  One 512-thread workgroup per CU, every wave issues REPS instructions of one kind back to back into a private
// 64 KiB window per wave (so it stays in L2 and HBM bandwidth is not what is measured); time per instruction and CU from the shader
// clock of wave 0.   hipcc --offload-arch=gfx950 -O3 store_issue.hip -o store_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
constexpr int REPS = 256;
// KIND 0: dwordx4, lane-linear (1 KiB = 8 whole lines per instruction)   1: dwordx2 lane-linear (512 B)   2: dword lane-linear (256 B)
//      3: fp32 atomic add, lane-linear (256 B)   4: dwordx4, a row of 128 B per lane PAIR (32 lines touched)   5: dwordx2, row per lane pair
//      6: dwordx4 non-temporal, lane-linear
template <int KIND> __global__ __launch_bounds__(512) void k(char* buf, unsigned long long* out, int S) {     // S KiB window per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* win = buf + ((size_t)blockIdx.x * 8 + wave) * 65536;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 8
    for (int i = 0; i < REPS; ++i) {
        if (KIND == 0) *(u4*)(win + (i & (S - 1)) * 1024 + lane * 16) = u4{1u, 2u, 3u, (uint32_t)i};
        else if (KIND == 6) __builtin_nontemporal_store(u4{1u, 2u, 3u, (uint32_t)i}, (u4*)(win + (i & (S - 1)) * 1024 + lane * 16));
        else if (KIND == 1) *(u2*)(win + (i & (2 * S - 1)) * 512 + lane * 8) = u2{1u, (uint32_t)i};
        else if (KIND == 2) *(uint32_t*)(win + (i & (4 * S - 1)) * 256 + lane * 4) = (uint32_t)i;
        else if (KIND == 3) unsafeAtomicAdd((float*)(win + (i & (4 * S - 1)) * 256 + lane * 4), 1.0f);
        else if (KIND == 4) *(u4*)(win + (lane & 31) * 128 + (lane >> 5) * 16 + (i & 3) * 32 + ((i >> 2) & (S / 4 - 1)) * 4096) = u4{1u, 2u, 3u, (uint32_t)i};
        else if (KIND == 5) *(u2*)(win + (lane & 31) * 128 + (lane >> 5) * 8 + (i & 7) * 16 + ((i >> 3) & (S / 4 - 1)) * 4096) = u2{1u, (uint32_t)i};
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
template <int KIND> static void run(const char* name, char* buf, unsigned long long* out, int bytes, int S) {
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, buf, out, S);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, buf, out, S);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
    // s_memtime counts at 100 MHz on this part: convert with the kernel's wall time instead -> report both
    const double instr = 8.0 * REPS;      // per CU
    printf("%-58s %8.1f us   %7.1f ns per instruction and CU   %6.1f B/ns per CU  (memtime ticks per instr %.2f)\n", name, ms * 1e3,
           ms * 1e6 / instr, bytes / (ms * 1e6 / instr), avg / instr);
}
int main() {
    char* buf; unsigned long long* out;
    hipMalloc(&buf, (size_t)256 * 8 * 65536); hipMalloc(&out, 256 * 8);
    hipMemset(buf, 0, (size_t)256 * 8 * 65536);
    for (int S : {64, 4}) {
        printf("-- %d KiB window per wave = %d MiB over the chip (%s)\n", S, S * 8 * 256 / 1024, S == 64 ? "beyond the L2s: streams to MALL / HBM" : "L2-resident");
        run<0>("dwordx4 lane-linear (8 whole lines / instr)", buf, out, 1024, S);
        run<6>("dwordx4 lane-linear, non-temporal", buf, out, 1024, S);
        run<1>("dwordx2 lane-linear (4 lines)", buf, out, 512, S);
        run<2>("dword lane-linear (2 lines)", buf, out, 256, S);
        run<3>("global_atomic_add_f32 lane-linear (2 lines)", buf, out, 256, S);
        run<4>("dwordx4, 32 rows touched per instr (16-byte row pieces)", buf, out, 1024, S);
        run<5>("dwordx2, 32 rows touched per instr (8-byte row pieces)", buf, out, 512, S);
    }
    return 0;
}
