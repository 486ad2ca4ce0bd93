// 3x3 / stride 1 / pad 1 convolution FORWARD in block-scaled fp8 (OCP e4m3, one E8M0 scale per 32 channels: the MX format) on
// v_mfma_scale_f32_16x16x128_f8f6f4 -- BASELINE configs[4] ("SSD512 ... fp8 MFMA convs") for the layers with >= 256 input
// channels.  The reference has no counterpart (fp32 TensorFlow convolutions, models/ssd_model.py:86-93 for these layers): parity
// is against the fp32 restatement on the SAME dequantised operands (exact up to fp32 summation order) and, as the stated
// quantisation error, against the fp32 convolution of the bf16 operands (tests/test_fp8_gpu.py).
//
// Operand layout of the instruction, probed on the device (tools_dev/mfma_scale_layout_check.hip + the one-hot channel sweep of
// tools_dev/dbg_fp8.py): lane (gq = lane >> 4, li = lane & 15) holds row / column li; its bytes 0..15 are k = 16 gq + i, its bytes
// 16..31 are k = 64 + 16 gq + (i - 16) -- two 64-deep halves, as the bf16 instructions' k-steps -- and the scale byte of lane group s
// (selected by opsel from a 32-bit register) scales the MX block k = 32 s .. 32 s + 31, which is spread over the first halves
// of lane groups 2 (s & 1) .. +1 or their second halves.  So a lane reads 16-byte chunks gq and 4 + gq of its 128-byte row.
//
// Kernel = the 128 x 128 implicit GEMM of k_conv_igemm_dma (conv.hip) at one byte per element: a k-step is one tap x 128 channels
// (rows of 128 B, LDS-DMA with the same XOR-swizzled image, two LDS buffers, one barrier per step), the scale bytes of a step
// travel as 4-byte LDS-DMA pieces next to the tiles, one matrix instruction per 16 x 16 tile and step (bf16: two).
#include "common.h"
#include <hip/hip_bf16.h>
#include "conv_common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef int v8i __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned f8_ld4(const char* __restrict__ p, const char* __restrict__ other) { (void)other; return *reinterpret_cast<const unsigned*>(p); }

constexpr int F8_TILE = 128 * 128;           // operand tile of a k-step: 128 rows x 128 B
constexpr int F8_BUF = 2 * F8_TILE + 2 * 512;   // A | B | scales A [128][4] | scales B [128][4]
constexpr int F8_LDS = 2 * F8_BUF;           // 66 KB

// one 32-channel block per thread: scale = 2^ceil(log2(amax / 448)) as an E8M0 byte, elements = OCP e4m3 of x / scale
__global__ __launch_bounds__(256) void k_quant_mx_fp8(const bf16_raw* __restrict__ x, unsigned char* __restrict__ q,
                                                      unsigned char* __restrict__ scale, long long nblocks) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nblocks; i += (long long)gridDim.x * 256) {
        const uint4* src = reinterpret_cast<const uint4*>(x + i * 32);
        float v[32];
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint4 w = src[j];
            const unsigned ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[j * 8 + 2 * k] = __uint_as_float(ws[k] << 16);
                v[j * 8 + 2 * k + 1] = __uint_as_float(ws[k] & 0xffff0000u);
            }
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) amax = fmaxf(amax, fabsf(v[j]));
        int e = 0;                                            // smallest e with amax * 2^-e <= 448 (e4m3's largest finite value)
        if (amax > 0.f) {
            int ex;
            const float m = frexpf(amax / 448.f, &ex);       // amax / 448 = m * 2^ex, m in [0.5, 1)
            e = (m == 0.5f) ? ex - 1 : ex;
            e = e < -127 ? -127 : (e > 127 ? 127 : e);
        }
        const float inv = ldexpf(1.f, -e);
        unsigned out[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int w = 0;
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * j] * inv, v[4 * j + 1] * inv, w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * j + 2] * inv, v[4 * j + 3] * inv, w, true);
            out[j] = (unsigned)w;
        }
        uint4* dst = reinterpret_cast<uint4*>(q + i * 32);
        dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
        dst[1] = make_uint4(out[4], out[5], out[6], out[7]);
        scale[i] = (unsigned char)(e + 127);
    }
}

__global__ __launch_bounds__(256) void k_conv3x3_mxfp8(const unsigned char* __restrict__ x, const unsigned char* __restrict__ xs,
                                                       const unsigned char* __restrict__ w, const unsigned char* __restrict__ wsc,
                                                       ConvGeom g, Epilogue ep) {
    // g: x [B,H,W,C] bytes, C % 128 == 0; w [N][9][C] bytes; xs [B*H*W][C/32], wsc [N][9][C/32] scale bytes
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave & 1, wave_n = wave >> 1;
    const int ntn = (g.N + 127) / 128;
    const int mt = blockIdx.x / ntn, n0 = (blockIdx.x % ntn) * 128, m0 = mt * 128;
    const int cb = g.C >> 5;                                  // scale bytes per pixel / per (filter, tap)
    const int csteps = g.C >> 7;                              // k-steps per tap

    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)g.M * (unsigned)g.C, 0x00020000);
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (unsigned)g.N * 9u * (unsigned)g.C, 0x00020000);
    const __amdgpu_buffer_rsrc_t xsres = __builtin_amdgcn_make_buffer_rsrc((void*)xs, 0, (unsigned)g.M * (unsigned)cb, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsres = __builtin_amdgcn_make_buffer_rsrc((void*)wsc, 0, (unsigned)g.N * 9u * (unsigned)cb, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;

    // tile pieces: instruction i (16 per operand, 4 per wave: i = wave + 4 j) covers tile rows 8i..8i+7; lane L -> row
    // 8i + 2 (L >> 4) + ((L >> 3) & 1), 16-byte chunk (L & 7) ^ ((4i + (L >> 4)) & 7)   [the image swz() reads]
    int py[4], px[4], pb[4];                                  // pixel of the staged activation row (py < 0: beyond M)
    unsigned wrow[4], cchunk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = wave + 4 * j;
        const int r = 8 * i + 2 * (lane >> 4) + ((lane >> 3) & 1);
        cchunk[j] = (unsigned)((lane & 7) ^ ((4 * i + (lane >> 4)) & 7)) * 16u;
        const int m = m0 + r;
        const bool mv = m < g.M;
        const int mm = mv ? m : 0;
        const int b = fdiv(mm, g.d_hw);
        const int rem = mm - b * g.d_hw.d;
        const int oy = fdiv(rem, g.d_w);
        py[j] = mv ? oy : -(1 << 20);
        px[j] = rem - oy * g.d_w.d;
        pb[j] = b * g.H * g.W;
        const int n = n0 + r;
        wrow[j] = n < g.N ? (unsigned)n * 9u * (unsigned)g.C : 0xffffffffu;
    }
    // scale pieces: 4 bytes per row and step; waves 0,1 bring the activation rows 64 wave + lane, waves 2,3 the filter rows
    const int srow = (wave & 1) * 64 + lane;
    int sy = 0, sx = 0, sb = 0;
    unsigned swrow = 0xffffffffu;
    if (wave < 2) {
        const int m = m0 + srow;
        const bool mv = m < g.M;
        const int mm = mv ? m : 0;
        const int b = fdiv(mm, g.d_hw);
        const int rem = mm - b * g.d_hw.d;
        const int oy = fdiv(rem, g.d_w);
        sy = mv ? oy : -(1 << 20); sx = rem - oy * g.d_w.d; sb = b * g.H * g.W;
    } else {
        const int n = n0 + srow;
        swrow = n < g.N ? (unsigned)n * 9u * (unsigned)cb : 0xffffffffu;
    }
    auto issue = [&](int step, int buf) {
        const int tap = step / csteps, cs = step - tap * csteps;
        const int kh = tap / 3 - 1, kw = tap % 3 - 1;
        char* base = smem + buf * F8_BUF;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = wave + 4 * j;
            const int iy = py[j] + kh, ix = px[j] + kw;
            const bool ok = (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
            const unsigned off = (unsigned)(pb[j] + iy * g.W + ix) * (unsigned)g.C + (unsigned)cs * 128u + cchunk[j];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void*)(base + i * 1024), 16, ok ? off : OOB, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = wave + 4 * j;
            const unsigned off = wrow[j] + (unsigned)tap * (unsigned)g.C + (unsigned)cs * 128u + cchunk[j];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (lds_void*)(base + F8_TILE + i * 1024), 16, wrow[j] != 0xffffffffu ? off : OOB, 0, 0, 0);
        }
        if (wave < 2) {
            const int iy = sy + kh, ix = sx + kw;
            const bool ok = (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
            const unsigned off = (unsigned)(sb + iy * g.W + ix) * (unsigned)cb + (unsigned)cs * 4u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsres, (lds_void*)(base + 2 * F8_TILE + wave * 256), 4, ok ? off : OOB, 0, 0, 0);
        } else {
            const unsigned off = swrow + (unsigned)tap * (unsigned)cb + (unsigned)cs * 4u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsres, (lds_void*)(base + 2 * F8_TILE + 512 + (wave - 2) * 256), 4, swrow != 0xffffffffu ? off : OOB, 0, 0, 0);
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[c][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int nsteps = 9 * csteps;
    const int gq = lane >> 4, li = lane & 15;
    issue(0, 0);
    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + 1 < nsteps) issue(st + 1, cur ^ 1);
        const char* base = smem + cur * F8_BUF;
        v8i fx[4], fw[4];
        int sxv[4], swv[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = wave_m * 64 + p * 16 + li;
            const uint4 lo = lds_ld16_scoped(base + swz(row, gq), smem), hi = lds_ld16_scoped(base + swz(row, 4 + gq), smem);
            fx[p] = v8i{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
            sxv[p] = (int)(f8_ld4(base + 2 * F8_TILE + row * 4, smem) >> (8 * gq));
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int row = wave_n * 64 + c * 16 + li;
            const uint4 lo = lds_ld16_scoped(base + F8_TILE + swz(row, gq), smem), hi = lds_ld16_scoped(base + F8_TILE + swz(row, 4 + gq), smem);
            fw[c] = v8i{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
            swv[c] = (int)(f8_ld4(base + 2 * F8_TILE + 512 + row * 4, smem) >> (8 * gq));
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int p = 0; p < 4; ++p)
                acc[c][p] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw[c], fx[p], acc[c][p], 0, 0, 0, swv[c], 0, sxv[p]);
    }
    int mrow[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int m = m0 + wave_m * 64 + p * 16 + (lane & 15);
        mrow[p] = m < g.M ? m : -1;
    }
    conv_epilogue_rows<EPI_FWD, 4, 4>(acc, g, ep, mrow, n0 + wave_n * 64, lane);
}

OnceLds g_f8_once;

}  // namespace

extern "C" {

int ssd_quantize_mx_fp8(const void* x_bf16, void* q, void* scale, long long n, void* stream) {
    if (!x_bf16 || !q || !scale || n <= 0 || (n & 31)) return SSD_ERR_VALUE;
    const long long nb = n / 32;
    const long long gridl = (nb + 255) / 256;
    hipLaunchKernelGGL(k_quant_mx_fp8, dim3((unsigned)(gridl > 16384 ? 16384 : gridl)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const bf16_raw*>(x_bf16), static_cast<unsigned char*>(q), static_cast<unsigned char*>(scale), nb);
    return ssd_launch_status();
}

int ssd_conv3x3_fwd_mxfp8(const void* x8, const void* xscale, const void* w8, const void* wscale, const float* bias, void* y, int B,
                          int H, int W, int Cin, int Cout, int relu, void* stream) {
    if (!x8 || !xscale || !w8 || !wscale || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SSD_ERR_VALUE;
    if (Cin % 128 || Cout % 8) return SSD_ERR_UNSUPPORTED;
    if ((long long)B * H * W * Cin >= (1ll << 31) || (long long)Cout * 9 * Cin >= (1ll << 31)) return SSD_ERR_UNSUPPORTED;
    const ConvGeom g = make_geom(B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1);
    Epilogue ep = {};
    ep.bias = bias; ep.relu = relu; ep.out = static_cast<bf16_raw*>(y); ep.ldo = Cout;
    if (ensure_lds(g_f8_once, reinterpret_cast<const void*>(k_conv3x3_mxfp8), F8_LDS) != 0) return SSD_ERR_LAUNCH;
    const unsigned grid = (unsigned)(((g.M + 127) / 128) * ((Cout + 127) / 128));
    hipLaunchKernelGGL(k_conv3x3_mxfp8, dim3(grid), dim3(256), F8_LDS, (hipStream_t)stream, static_cast<const unsigned char*>(x8),
                       static_cast<const unsigned char*>(xscale), static_cast<const unsigned char*>(w8),
                       static_cast<const unsigned char*>(wscale), g, ep);
    return ssd_launch_status();
}

}  // extern "C"
