// MFMA peak microbenchmark on gfx950: 16x16x32 bf16, NACC independent accumulators, W waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 0.001f + i); b[i] = (__bf16)(seed * 0.5f + i); }
    f32x4_t acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = f32x4_t{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int wgs_per_cu, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, grid = 256 * wgs_per_cu;
    k<NACC><<<grid, 256>>>(d, iters, 1.0f); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<grid, 256>>>(d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * NACC * 16384.0;
    printf("NACC=%d waves/SIMD=%d: %.3f ms  %.0f TFLOP/s\n", NACC, wgs_per_cu, ms, flops / ms / 1e9);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<4>(1, d); run<16>(1, d); run<16>(2, d); run<32>(2, d); run<16>(4, d);
    return 0;
}
