// LDS fragment read + MFMA inner-loop microbenchmark (no global traffic): which loop shape reaches the MFMA peak?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;
__device__ __forceinline__ int swz(int row, int chunk) {
    return (row >> 1) * 256 + ((((row & 1) << 3) | (chunk ^ ((row >> 1) & 7))) << 4);
}
// MODE 0: read 8 frags, 16 MFMA per ksub (as in k_conv_igemm_dma).  MODE 1: software pipelined (next ksub's frags are
// read before this ksub's MFMAs).  MODE 2: MODE 1 + s_setprio around the MFMAs.
template <int MODE, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 * 512 rows * 128 B = 128 KB like the 256x256 tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 131072 / 4; i += NWAVES * 64) reinterpret_cast<float*>(smem)[i] = (float)(i & 7);
    __syncthreads();
    const int wave_m = wave % 4, wave_n = (wave / 4) % 4;
    const int frow = lane & 15, fk = lane >> 4;
    f32x4_t acc[4][4];
    for (int c = 0; c < 4; ++c) for (int p = 0; p < 4; ++p) acc[c][p] = f32x4_t{0, 0, 0, 0};
    const char* sx = smem;
    const char* sw = smem + 256 * 128;
    auto load = [&](int buf, int ksub, bf16x8_t (&fx)[4], bf16x8_t (&fw)[4]) {
        const char* bx = sx + buf * 65536;
        const char* bw = sw + buf * 65536;
#pragma unroll
        for (int p = 0; p < 4; ++p) fx[p] = *reinterpret_cast<const bf16x8_t*>(bx + swz(wave_m * 64 + p * 16 + frow, ksub * 4 + fk));
#pragma unroll
        for (int c = 0; c < 4; ++c) fw[c] = *reinterpret_cast<const bf16x8_t*>(bw + swz(wave_n * 64 + c * 16 + frow, ksub * 4 + fk));
    };
    auto mma = [&](bf16x8_t (&fx)[4], bf16x8_t (&fw)[4]) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[c], fx[p], acc[c][p], 0, 0, 0);
    };
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int ksub = 0; ksub < 2; ++ksub) {
                bf16x8_t fx[4], fw[4];
                load(it & 1, ksub, fx, fw);
                mma(fx, fw);
            }
        }
    } else {
        bf16x8_t ax[4], aw[4], bx[4], bw[4];
        load(0, 0, ax, aw);
        for (int it = 0; it < iters; ++it) {
            load(it & 1, 1, bx, bw);
            if (MODE == 2) __builtin_amdgcn_s_setprio(1);
            mma(ax, aw);
            if (MODE == 2) __builtin_amdgcn_s_setprio(0);
            load((it + 1) & 1, 0, ax, aw);
            if (MODE == 2) __builtin_amdgcn_s_setprio(1);
            mma(bx, bw);
            if (MODE == 2) __builtin_amdgcn_s_setprio(0);
        }
    }
    float s = 0;
    for (int c = 0; c < 4; ++c) for (int p = 0; p < 4; ++p) s += acc[c][p][0] + acc[c][p][2];
    out[blockIdx.x * NWAVES * 64 + tid] = s;
}
template <int MODE, int NWAVES> void run(float* d, int wgs_per_cu) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, grid = 256 * wgs_per_cu;
    const size_t lds = wgs_per_cu == 1 ? 131072 : 65536;
    hipFuncSetAttribute((const void*)k<MODE, NWAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    k<MODE, NWAVES><<<grid, NWAVES * 64, 131072 / wgs_per_cu>>>(d, 10); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, NWAVES><<<grid, NWAVES * 64, 131072 / wgs_per_cu>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * NWAVES * iters * 32 * 16384.0;
    printf("MODE=%d waves/WG=%d WG/CU=%d: %.3f ms  %.0f TFLOP/s\n", MODE, NWAVES, wgs_per_cu, ms, flops / ms / 1e9);
    (void)lds;
}
int main() {
    float* d; hipMalloc(&d, 256 * 2 * 1024 * 4);
    run<0, 16>(d, 1); run<1, 16>(d, 1); run<2, 16>(d, 1);
    run<0, 8>(d, 1); run<1, 8>(d, 1); run<2, 8>(d, 1);
    run<0, 4>(d, 2); run<1, 4>(d, 2);
    return 0;
}
