// Sustained MFMA rate under the power limit (not part of the product): register-only loops of
// v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 on random or zero operands, one or two
// waves per SIMD.  Answers: which shape does more flops per joule on this part, and how much of the
// scan kernels' clock loss is the matrix pipe alone.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_power tools/mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__device__ inline unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// bf16 pairs of magnitude ~0.01..0.06 with random sign and mantissa (what unit-norm 768-d rows look like)
__device__ inline bf16x8_t frag(unsigned seed, int zero) {
    u32x4_t w;
    for (int i = 0; i < 4; ++i) {
        const unsigned r = mix(seed * 4 + i);
        w[i] = zero ? 0u : ((0x3c003c00u + (r & 0x03ff03ffu)) | (r & 0x80008000u));
    }
    return __builtin_bit_cast(bf16x8_t, w);
}

// SHAPE 0: 32x32x16, 4 accumulator chains; A fragment shared by two consecutive MFMAs (as in k_scan_qreg2)
// SHAPE 1: 16x16x32, 8 accumulator chains; A fragment shared by four consecutive MFMAs
template <int SHAPE>
__global__ __launch_bounds__(256) void k_power(float* out, int iters, int zero) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8_t a[8], b[16];
    for (int i = 0; i < 8; ++i) a[i] = frag(tid * 64 + i, zero);
    for (int i = 0; i < 16; ++i) b[i] = frag(tid * 64 + 32 + i, zero);
    float r = 0.f;
    if (SHAPE == 0) {
        f32x16_t c[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                c[(2 * i) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[2 * i], c[(2 * i) & 3], 0, 0, 0);
                c[(2 * i + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[2 * i + 1], c[(2 * i + 1) & 3], 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) r += c[i][j];
    } else {
        f32x4_t c[8];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) c[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    c[(4 * i + j) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[(4 * i + j) & 15], c[(4 * i + j) & 7], 0, 0, 0);
            }
        }
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) r += c[i][j];
    }
    if (r == 12345.678f) out[tid] = r;
}

template <int SHAPE>
void run(const char* name, int waves_per_simd, int zero, float* out) {
    const int grid = 256 * waves_per_simd, iters = 40000;
    // flops per wave per iteration: SHAPE 0: 16 x 32768; SHAPE 1: 32 x 16384 -- the same
    const double flop = (double)grid * 4 * iters * 16.0 * 32768.0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) k_power<SHAPE><<<grid, 256>>>(out, iters, zero);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k_power<SHAPE><<<grid, 256>>>(out, iters, zero);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    // issue-limited time at clock f: SHAPE 0: 16 MFMA x 8 passes x 4 clk = 512 clk per iteration per wave
    const double clk_per_iter = 512.0 * waves_per_simd;
    printf("%-10s waves/SIMD=%d %-6s %8.3f ms  %7.1f TFLOP/s  (clock if issue-bound: %.2f GHz)  %s\n", name, waves_per_simd,
           zero ? "zero" : "random", ms, flop / ms / 1e9, iters * clk_per_iter / ms / 1e6, hipGetErrorString(hipGetLastError()));
}

int main() {
    float* out; hipMalloc(&out, 1 << 24);
    for (int zero = 0; zero < 2; ++zero)
        for (int w = 1; w <= 2; ++w) {
            run<0>("32x32x16", w, zero, out);
            run<1>("16x16x32", w, zero, out);
        }
    return 0;
}
