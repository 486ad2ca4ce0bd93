// Gate for "feed the un-pooled gradient to the weight-gradient kernel as a 2:4 structured-sparse operand" (VERDICT r3, item 2):
// what does v_smfmac_f32_16x16x64_bf16 buy over two v_mfma_f32_16x16x32_bf16 on gfx950, (a) bare, operands in registers, and
// (b) in the fragment-read mix of k_conv3x3_wgrad_patch (per 64-deep k-step and wave: 4 A fragments = the dY tiles, 9 B
// fragments = tap-shifted X tiles, 36 accumulator tiles; B is dense in both forms and dominates the LDS reads; the sparse form
// reads A compressed, half the bytes).  Random operands, 256 workgroups x 8 waves, no global traffic, in-kernel clock stamps.
// build: hipcc -O3 --offload-arch=gfx950 tools_dev/ubench_smfmac.hip -o tools_dev/ubench_smfmac
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(__bf16)))) __bf16 bf16x16_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;

// MODE 0 dense bare, 1 sparse bare, 2 dense LDS-fed, 3 sparse LDS-fed
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* stamps, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 64 KB of random bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned seed = 1234567u + tid * 7919u + blockIdx.x * 104729u;
    for (int i = tid; i < 65536 / 4; i += 512) {
        seed = seed * 1664525u + 1013904223u;
        const unsigned a = 0x3f000000u | ((seed >> 9) & 0x007f0000u) | ((seed & 1u) << 31);          // bf16 in +-[0.5, 1)
        seed = seed * 1664525u + 1013904223u;
        const unsigned b = 0x3f00u | ((seed >> 25) & 0x7fu) | (((seed >> 3) & 1u) << 15);
        reinterpret_cast<unsigned*>(smem)[i] = a | b;
    }
    __syncthreads();
    f32x4_t acc[4][9];
    for (int c = 0; c < 4; ++c) for (int t = 0; t < 9; ++t) acc[c][t] = f32x4_t{0, 0, 0, 0};
    const char* base = smem + (wave * 64 + lane) * 16;          // conflict-free: consecutive lanes, consecutive 16-byte slots
    const int idx = 0x4444 * 0 + 0xe4e4;                        // any valid 2-of-4 pattern (pairs 0,1 / 2,3 ...)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (MODE <= 1) {
        bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(base), a1 = *reinterpret_cast<const bf16x8_t*>(base + 8192);
        bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(base + 16384), b1 = *reinterpret_cast<const bf16x8_t*>(base + 24576);
        bf16x16_t bb;
        for (int j = 0; j < 8; ++j) { bb[j] = b0[j]; bb[8 + j] = b1[j]; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (MODE == 0) {
                        acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc[c][t], 0, 0, 0);
                        acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc[c][t], 0, 0, 0);
                    } else {
                        acc[c][t] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a0, bb, acc[c][t], idx, 0, 0);
                    }
                }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            const char* p = base + ((it & 3) << 10) * 0;
            bf16x8_t fa[4][2];
            bf16x8_t fb[9][2];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                fa[c][0] = *reinterpret_cast<const bf16x8_t*>(p + ((c * 2 * 8192) & 65535));
                if (MODE == 2) fa[c][1] = *reinterpret_cast<const bf16x8_t*>(p + (((c * 2 + 1) * 8192) & 65535));
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                fb[t][0] = *reinterpret_cast<const bf16x8_t*>(p + (((8 + t * 2) * 8192 + t * 512) & 65535));
                fb[t][1] = *reinterpret_cast<const bf16x8_t*>(p + (((9 + t * 2) * 8192 + t * 512) & 65535));
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (MODE == 2) {
                        acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[c][0], fb[t][0], acc[c][t], 0, 0, 0);
                        acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[c][1], fb[t][1], acc[c][t], 0, 0, 0);
                    } else {
                        bf16x16_t bb;
#pragma unroll
                        for (int j = 0; j < 8; ++j) { bb[j] = fb[t][0][j]; bb[8 + j] = fb[t][1][j]; }
                        acc[c][t] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(fa[c][0], bb, acc[c][t], idx, 0, 0);
                    }
                }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
    float s = 0;
    for (int c = 0; c < 4; ++c) for (int t = 0; t < 9; ++t) s += acc[c][t][0] + acc[c][t][3];
    out[blockIdx.x * 512 + tid] = s;
}

template <int MODE> void run(float* d, unsigned long long* st, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, grid = 256;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    k<MODE><<<grid, 512, 65536>>>(d, st, 100); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, 512, 65536>>>(d, st, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[512]; hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    double clk = 0; for (int i = 0; i < 256; ++i) clk += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; clk /= 256;
    const double steps = (double)grid * 8 * iters;                 // (wave, k-step) pairs: each = 36 tiles x 16x16x64 MACs
    const double dense_flops = steps * 36 * 2.0 * 16 * 16 * 64;
    const double cyc_per_step = (double)h[0] / iters;
    printf("%-34s %8.3f ms  dense-equivalent %7.0f TFLOP/s  clock %.2f GHz  %6.0f cycles per k-step and wave (%.1f per tile)\n", what, ms,
           dense_flops / ms / 1e9, clk, cyc_per_step, cyc_per_step / 36);
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    unsigned long long* st; hipMalloc(&st, 512 * 8);
    run<0>(d, st, "dense 2 x mfma 16x16x32, bare");
    run<1>(d, st, "sparse smfmac 16x16x64, bare");
    run<2>(d, st, "dense, LDS-fed (8 A + 18 B reads)");
    run<3>(d, st, "sparse, LDS-fed (4 A + 18 B reads)");
    return 0;
}
