// Runtime half of the SIMT lockstep emulator (see include/hip/hip_runtime.h).  TEST ONLY.
#include <hip/hip_runtime.h>

thread_local dim3 threadIdx;
thread_local dim3 blockIdx;
dim3 blockDim;
dim3 gridDim;
// Kernels declare `extern __shared__ char smem[]` at block scope inside namespace maest, which (with
// __shared__ defined away) names maest::smem: one global array is the LDS of the block being run.
namespace maest {
alignas(256) char smem[160 * 1024];
}
using maest::smem;

namespace emu {
pthread_barrier_t block_bar;
Wave* waves = nullptr;
thread_local int tid_linear = 0;

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads % 64 != 0) {
        fprintf(stderr, "emu: block size %d is not a multiple of the wave size 64\n", nthreads);
        abort();
    }
    const int nwaves = nthreads / 64;
    blockDim = block;
    gridDim = grid;
    waves = new Wave[nwaves];
    for (int w = 0; w < nwaves; ++w) pthread_barrier_init(&waves[w].bar, nullptr, 64);
    pthread_barrier_init(&block_bar, nullptr, nthreads);
    std::vector<std::thread> pool;
    pool.reserve(nthreads);
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                // poison LDS between blocks so stale-data bugs are visible
                memset(smem, 0x7f, sizeof(smem));
                pool.clear();
                for (int t = 0; t < nthreads; ++t) {
                    pool.emplace_back([=, &body]() {
                        tid_linear = t;
                        threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                        blockIdx = dim3(bx, by, bz);
                        body();
                    });
                }
                for (auto& th : pool) th.join();
            }
    pthread_barrier_destroy(&block_bar);
    for (int w = 0; w < nwaves; ++w) pthread_barrier_destroy(&waves[w].bar);
    delete[] waves;
    waves = nullptr;
}
}  // namespace emu
