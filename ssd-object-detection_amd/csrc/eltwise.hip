// Element-wise and pooling pieces of the ResNet-50 trunk (BASELINE configs[4]: SSD512 with a ResNet-50 backbone; the reference
// itself hard-codes a VGG trunk, models/ssd_model.py:46,75-97, so there is no reference counterpart -- semantics are Keras /
// TensorFlow's: Add + ReLU of a residual block, MaxPooling2D(3, strides=2, padding="same") behind the stem, and their
// tape.gradient).  All HBM-bound streams, 16 bytes per lane.
#include "common.h"
#include <hip/hip_bf16.h>

namespace {

typedef unsigned short bf16_raw;

__device__ __forceinline__ float bf2f_(unsigned v16) { return __uint_as_float(v16 << 16); }
__device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
}

// out = relu(a + b)  (residual add of a bottleneck block; bf16 in / out, the sum in fp32, one rounding)
__global__ __launch_bounds__(256) void k_add_relu(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ out,
                                                  long long nvec) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const uint4 x = a[i], y = b[i];
        const unsigned xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
        unsigned r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float lo = fmaxf(bf2f_(xs[k] & 0xffffu) + bf2f_(ys[k] & 0xffffu), 0.f);
            const float hi = fmaxf(bf2f_(xs[k] >> 16) + bf2f_(ys[k] >> 16), 0.f);
            r[k] = pack2(lo, hi);
        }
        out[i] = make_uint4(r[0], r[1], r[2], r[3]);
    }
}

// out (+)= g where act > 0, else (+)= 0   (the ReLU of a residual block's output, on the identity-skip branch)
__global__ __launch_bounds__(256) void k_relu_mask(const uint4* __restrict__ g, const uint4* __restrict__ act, uint4* __restrict__ out,
                                                   int accumulate, long long nvec) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const uint4 gv = g[i], av = act[i];
        uint4 ov = accumulate ? out[i] : make_uint4(0, 0, 0, 0);
        const unsigned gs[4] = {gv.x, gv.y, gv.z, gv.w}, as[4] = {av.x, av.y, av.z, av.w}, os[4] = {ov.x, ov.y, ov.z, ov.w};
        unsigned r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float glo = bf2f_(as[k] & 0xffffu) > 0.f ? bf2f_(gs[k] & 0xffffu) : 0.f;
            const float ghi = bf2f_(as[k] >> 16) > 0.f ? bf2f_(gs[k] >> 16) : 0.f;
            r[k] = pack2(bf2f_(os[k] & 0xffffu) + glo, bf2f_(os[k] >> 16) + ghi);
        }
        out[i] = make_uint4(r[0], r[1], r[2], r[3]);
    }
}

// 3x3 / stride 2 max pooling, explicit (TF "SAME") padding: y[b][oy][ox][c] = max over the window rows 2 oy - pt + {0,1,2},
// columns 2 ox - pl + {0,1,2} inside the map.  code: one nibble per element = index 3 dy + dx of the FIRST maximum, 15 = the
// maximum is <= 0 (no gradient through the ReLU in front) -- the convention of the 2x2 pooling's codes.  One thread per output
// pixel and 8 channels.
__global__ __launch_bounds__(256) void k_maxpool3x3s2_fwd(const bf16_raw* __restrict__ x, bf16_raw* __restrict__ y,
                                                          unsigned* __restrict__ code, int B, int H, int W, int C, int Ho, int Wo,
                                                          int pt, int pl) {
    const int c8 = C >> 3;
    const long long total = (long long)B * Ho * Wo * c8;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int cc = (int)(idx % c8);
        long long r = idx / c8;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float best[8];
        unsigned pos[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; pos[k] = 15u; }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int iy = 2 * oy - pt + dy, ix = 2 * ox - pl + dx;
                if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
                const uint4 v = *reinterpret_cast<const uint4*>(x + (((long long)b * H + iy) * W + ix) * C + cc * 8);
                const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float f = (k & 1) ? __uint_as_float(w[k >> 1] & 0xffff0000u) : __uint_as_float(w[k >> 1] << 16);
                    if (f > best[k]) { best[k] = f; pos[k] = (unsigned)(3 * dy + dx); }
                }
            }
        unsigned o4[4], cw = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) o4[k] = (__float_as_uint(best[2 * k]) >> 16) | (__float_as_uint(best[2 * k + 1]) & 0xffff0000u);
#pragma unroll
        for (int k = 0; k < 8; ++k) cw |= (best[k] > 0.f ? pos[k] : 15u) << (4 * k);
        *reinterpret_cast<uint4*>(y + idx * 8) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
        code[idx] = cw;
    }
}

// Backward: windows overlap (an input pixel belongs to up to four of them), so the gradient is GATHERED per input pixel --
// dx[p] = sum of dy over the windows whose recorded winner is p, in fixed window order: deterministic, no atomics.
__global__ __launch_bounds__(256) void k_maxpool3x3s2_bwd(const unsigned* __restrict__ code, const bf16_raw* __restrict__ dy,
                                                          bf16_raw* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo,
                                                          int pt, int pl) {
    const int c8 = C >> 3;
    const long long total = (long long)B * H * W * c8;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int cc = (int)(idx % c8);
        long long r = idx / c8;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int b = (int)(r / H);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // windows (oy, ox) with 2 oy - pt <= iy <= 2 oy - pt + 2
        const int oy_lo = (iy + pt - 2 + 1) >> 1, oy_hi = (iy + pt) >> 1;      // ceil((iy + pt - 2) / 2) .. floor((iy + pt) / 2)
        const int ox_lo = (ix + pl - 2 + 1) >> 1, ox_hi = (ix + pl) >> 1;
        for (int oy = max(oy_lo, 0); oy <= min(oy_hi, Ho - 1); ++oy)
            for (int ox = max(ox_lo, 0); ox <= min(ox_hi, Wo - 1); ++ox) {
                const unsigned me = (unsigned)(3 * (iy - (2 * oy - pt)) + (ix - (2 * ox - pl)));
                const long long o = (((long long)b * Ho + oy) * Wo + ox) * c8 + cc;
                const unsigned cw = code[o];
                const uint4 g = *reinterpret_cast<const uint4*>(dy + o * 8);
                const unsigned w[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (((cw >> (4 * k)) & 15u) != me) continue;
                    acc[k] += (k & 1) ? __uint_as_float(w[k >> 1] & 0xffff0000u) : __uint_as_float(w[k >> 1] << 16);
                }
            }
        *reinterpret_cast<uint4*>(dx + idx * 8) = make_uint4(pack2(acc[0], acc[1]), pack2(acc[2], acc[3]), pack2(acc[4], acc[5]), pack2(acc[6], acc[7]));
    }
}

inline unsigned grid_for(long long items) {
    const long long g = (items + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" {

int ssd_add_relu_fwd(const void* a, const void* b, void* out, long long n, void* stream) {
    if (!a || !b || !out || n <= 0 || (n & 7)) return SSD_ERR_VALUE;
    hipLaunchKernelGGL(k_add_relu, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, static_cast<const uint4*>(a),
                       static_cast<const uint4*>(b), static_cast<uint4*>(out), n / 8);
    return ssd_launch_status();
}

int ssd_relu_mask_bwd(const void* g, const void* act, void* out, int accumulate, long long n, void* stream) {
    if (!g || !act || !out || n <= 0 || (n & 7)) return SSD_ERR_VALUE;
    hipLaunchKernelGGL(k_relu_mask, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, static_cast<const uint4*>(g),
                       static_cast<const uint4*>(act), static_cast<uint4*>(out), accumulate, n / 8);
    return ssd_launch_status();
}

int ssd_maxpool3x3s2_fwd(const void* x, void* y, void* code, int B, int H, int W, int C, int Ho, int Wo, int pad_t, int pad_l,
                         void* stream) {
    if (!x || !y || !code || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || Ho <= 0 || Wo <= 0 || pad_t < 0 || pad_l < 0 || pad_t > 2 || pad_l > 2)
        return SSD_ERR_VALUE;
    if (2 * (Ho - 1) - pad_t >= H || 2 * (Wo - 1) - pad_l >= W) return SSD_ERR_VALUE;      // every window touches the map
    hipLaunchKernelGGL(k_maxpool3x3s2_fwd, dim3(grid_for((long long)B * Ho * Wo * (C >> 3))), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const bf16_raw*>(x), static_cast<bf16_raw*>(y), static_cast<unsigned*>(code), B, H, W, C, Ho, Wo, pad_t, pad_l);
    return ssd_launch_status();
}

int ssd_maxpool3x3s2_bwd(const void* code, const void* dy, void* dx, int B, int H, int W, int C, int Ho, int Wo, int pad_t, int pad_l,
                         void* stream) {
    if (!code || !dy || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || Ho <= 0 || Wo <= 0 || pad_t < 0 || pad_l < 0 || pad_t > 2 || pad_l > 2)
        return SSD_ERR_VALUE;
    hipLaunchKernelGGL(k_maxpool3x3s2_bwd, dim3(grid_for((long long)B * H * W * (C >> 3))), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const unsigned*>(code), static_cast<const bf16_raw*>(dy), static_cast<bf16_raw*>(dx), B, H, W, C, Ho, Wo, pad_t, pad_l);
    return ssd_launch_status();
}

}  // extern "C"
