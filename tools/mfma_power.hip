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

// SHAPE 0: 32x32x16, 4 accumulator chains of 16 registers; SHAPE 1: 16x16x32, 16 accumulator chains of 4 registers.
// Both loops are ONE asm statement per iteration (16 / 32 MFMAs back to back, accumulators and operands pinned in
// VGPRs as "+v" / "v" operands), so the compiler can add neither v_accvgpr copies nor nops between the MFMAs -- round
// 2's plain-HIP version of the 16x16x32 arm carried 88 v_accvgpr_write + 31 v_accvgpr_read per 32 MFMAs and reached
// 1.63 PF on ZEROS, i.e. it was not issue-bound and its random-data ranking of the two shapes was void.
// Same flops per iteration: 16 x 32768 = 32 x 16384 MACs x 2.
template <int SHAPE>
__global__ __launch_bounds__(256) void k_power(float* out, int iters, int zero) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8_t a[4], b[4];
    for (int i = 0; i < 4; ++i) a[i] = frag(tid * 64 + i, zero);
    for (int i = 0; i < 4; ++i) b[i] = frag(tid * 64 + 32 + i, zero);
    float r = 0.f;
    if (SHAPE == 0) {
        f32x16_t c0, c1, c2, c3;
        for (int j = 0; j < 16; ++j) c0[j] = c1[j] = c2[j] = c3[j] = 0.f;
        for (int it = 0; it < iters; ++it) {
            asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %4, %8, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %9, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %5, %10, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %5, %11, %3\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %6, %8, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %6, %9, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %7, %10, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %7, %11, %3\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %4, %10, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %11, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %5, %8, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %5, %9, %3\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %6, %10, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %6, %11, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %7, %8, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %7, %9, %3"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
        }
        for (int j = 0; j < 16; ++j) r += c0[j] + c1[j] + c2[j] + c3[j];
    } else {
        f32x4_t c[16];
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) c[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#define M16(ci, ai, bi) "v_mfma_f32_16x16x32_bf16 %" #ci ", %" #ai ", %" #bi ", %" #ci "\n\t"
            asm volatile(  // every A fragment feeds four consecutive MFMAs (as in k_scan_q16), 16 independent chains
                M16(0, 16, 20) M16(1, 16, 21) M16(2, 16, 22) M16(3, 16, 23) M16(4, 17, 20) M16(5, 17, 21) M16(6, 17, 22) M16(7, 17, 23)
                M16(8, 18, 20) M16(9, 18, 21) M16(10, 18, 22) M16(11, 18, 23) M16(12, 19, 20) M16(13, 19, 21) M16(14, 19, 22) M16(15, 19, 23)
                M16(0, 17, 20) M16(1, 17, 21) M16(2, 17, 22) M16(3, 17, 23) M16(4, 18, 20) M16(5, 18, 21) M16(6, 18, 22) M16(7, 18, 23)
                M16(8, 19, 20) M16(9, 19, 21) M16(10, 19, 22) M16(11, 19, 23) M16(12, 16, 20) M16(13, 16, 21) M16(14, 16, 22) M16(15, 16, 23)
                ""
                : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]),
                  "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15])
                : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
#undef M16
        }
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) r += c[i][j];
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
