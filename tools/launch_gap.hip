// Back-to-back launch period of near-empty kernels (not part of the product): does a large dynamic LDS
// request or a 512-register wave make the dispatch itself slow?
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/launch_gap tools/launch_gap.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_small(unsigned* out) {
    extern __shared__ unsigned char lds[];
    if (out && threadIdx.x == 0 && blockIdx.x == 0xffffff) out[0] = lds[0];
}
__global__ __launch_bounds__(256, 1) void k_bigregs(unsigned* out) {
    extern __shared__ unsigned char lds[];
    asm volatile("v_mov_b32 v255, 0\n v_accvgpr_write_b32 a255, v255" ::: "v255", "a255");
    if (out && threadIdx.x == 0 && blockIdx.x == 0xffffff) out[0] = lds[0];
}

template <typename K>
void run(const char* name, K k, int grid, size_t lds) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, (unsigned*)nullptr);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, (unsigned*)nullptr);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s grid %5d  lds %7zu B: %.2f us per launch (%s)\n", name, grid, lds, ms * 10, hipGetErrorString(hipGetLastError()));
}

int main() {
    for (int grid : {256, 2048})
        for (size_t lds : {(size_t)0, (size_t)65536, (size_t)150528, (size_t)163840}) {
            run("small", k_small, grid, lds);
            run("512 regs", k_bigregs, grid, lds);
        }
    return 0;
}
