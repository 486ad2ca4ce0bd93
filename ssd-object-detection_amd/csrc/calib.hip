// Measurement aid (no reference counterpart, never on the product path): what THIS device sustains for the inner loop
// of the convolution kernels -- bf16 16x16x32 MFMAs fed by ds_read_b128 fragment reads from LDS, random operands, no
// global memory traffic.  bench.py runs it next to the train step so that step times measured on different boxes of the
// pool can be normalised (MI355X devices differ by ~12 % on exactly this kind of loop, MI355X_MICROARCH.md "DVFS give-back"
// item 5) and reports the in-kernel clock from s_memtime / s_memrealtime stamps around the loop (item 6).
#include "common.h"
#include "conv_common.h"

namespace {

constexpr int CAL_WAVES = 8;                 // one 512-thread workgroup per CU: two waves per SIMD, as k_conv3x3_p512
constexpr int CAL_LDS = 131072;

__device__ __forceinline__ unsigned cal_hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// a wave owns a 64 x 64 tile of a 256 x 256 block (16 accumulator tiles); per k sub-step 8 fragment reads and 16 MFMAs,
// the next sub-step's fragments requested before the current MFMAs (the schedule of the patch kernels)
__global__ __launch_bounds__(CAL_WAVES * 64) void k_mfma_calibration(unsigned long long* __restrict__ stamps, float* __restrict__ sink,
                                                                     int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // random bf16 in +-[0.5, 4): sign and mantissa random, exponent 126..128
    for (int i = tid; i < CAL_LDS / 4; i += CAL_WAVES * 64) {
        const unsigned h = cal_hash((unsigned)i * 2654435761u + blockIdx.x * 97u);
        auto bf = [](unsigned r) { return ((r & 1u) << 15) | ((126u + ((r >> 1) % 3u)) << 7) | ((r >> 3) & 0x7fu); };
        reinterpret_cast<unsigned*>(smem)[i] = bf(h) | (bf(h >> 16) << 16);
    }
    __syncthreads();
    const int wave_m = wave & 3, wave_n = wave >> 2;
    const int frow = lane & 15, fk = lane >> 4;
    f32x4_t acc[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[c][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto load = [&](int buf, int ksub, bf16x8_t (&fx)[4], bf16x8_t (&fw)[4]) {
        const char* bx = smem + buf * 65536;
        const char* bw = bx + 256 * 128;
#pragma unroll
        for (int p = 0; p < 4; ++p) fx[p] = *reinterpret_cast<const bf16x8_t*>(bx + swz(wave_m * 64 + p * 16 + frow, ksub * 4 + fk));
#pragma unroll
        for (int c = 0; c < 4; ++c) fw[c] = *reinterpret_cast<const bf16x8_t*>(bw + swz(wave_n * 128 + c * 16 + frow, ksub * 4 + fk));
    };
    auto mma = [&](bf16x8_t (&fx)[4], bf16x8_t (&fw)[4]) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[c], fx[p], acc[c][p], 0, 0, 0);
    };
    bf16x8_t ax[4], aw[4], bx[4], bw[4];
    load(0, 0, ax, aw);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        load(it & 1, 1, bx, bw);
        mma(ax, aw);
        load((it + 1) & 1, 0, ax, aw);
        mma(bx, bw);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < 4; ++p) s += acc[c][p][0] + acc[c][p][1] + acc[c][p][2] + acc[c][p][3];
    sink[blockIdx.x * (CAL_WAVES * 64) + tid] = s;           // keeps the loop alive; never read
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

OnceLds g_once_cal;

}  // namespace

extern "C" {

int ssd_dev_mfma_calibration_workgroups(void) { return 256; }

double ssd_dev_mfma_calibration_flops(int iters) {
    // per iteration and wave: 2 sub-steps x 16 MFMAs x (16 x 16 x 32 x 2) flop
    return 256.0 * CAL_WAVES * (double)iters * 32.0 * 16384.0;
}

int ssd_dev_mfma_calibration(int iters, void* stamps, void* sink, void* stream) {
    if (iters <= 0 || !stamps || !sink) return SSD_ERR_VALUE;
    if (ensure_lds(g_once_cal, (const void*)k_mfma_calibration, CAL_LDS) != 0) return SSD_ERR_LAUNCH;
    hipLaunchKernelGGL(k_mfma_calibration, dim3(256), dim3(CAL_WAVES * 64), CAL_LDS, (hipStream_t)stream,
                       (unsigned long long*)stamps, (float*)sink, iters);
    return ssd_launch_status();
}

}  // extern "C"
