// Optimizer step for gfx950 (MI355X): per-tensor gradient clipping + Adam over ONE flat fp32 buffer.
//
// Replaces, for all 64 parameter tensors at once, the per-tensor loop of _train_step:
//   tf.clip_by_norm(g, 0.01)            models/ssd_model.py:249   g * clip / max(||g||_2, clip)
//   mean over micro-batches / ranks     :251-256                  (grad_scale = 1/count)
//   Adam.apply_gradients                :258-260; hyper-parameters tools/train.py:42-51, config/default.yml:20-24
// Parameters, gradients and both Adam moments live in flat fp32 buffers; every tensor starts on a
// multiple of OPT_BLOCK elements, so each block of OPT_BLOCK elements belongs to exactly one tensor
// (block -> tensor id table).  All reductions are fixed-order (deterministic).
#include "common.h"
#include <hip/hip_bf16.h>

namespace {

constexpr int OPT_BLOCK = 1024;              // elements per block = 256 threads x float4

__device__ __forceinline__ unsigned short f2bf_bits(float f) {
    const __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<const unsigned short*>(&h);
}

// partial[b] = sum of squares of block b (double)
__global__ __launch_bounds__(256) void k_sqnorm_partial(const float* __restrict__ g, long long n, double* __restrict__ partial) {
    __shared__ double s_red[4];
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    double s = 0.0;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4*>(g + i);
        s = (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    } else {
        for (long long k = i; k < n; ++k) s += (double)g[k] * g[k];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int lo = __shfl_xor(__double2loint(s), off);
        const int hi = __shfl_xor(__double2hiint(s), off);
        s += __hiloint2double(hi, lo);
    }
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// scale[t] = clip / max(||g_t||, clip): one workgroup per tensor; every thread sums a strided subset of the
// tensor's block partials in index order, then a fixed-shape tree combines them (deterministic).
__global__ __launch_bounds__(256) void k_clip_scale(const double* __restrict__ partial, const int* __restrict__ tensor_block_off,
                                                    int ntensors, float clip, float* __restrict__ scale,
                                                    float* __restrict__ norms) {
    __shared__ double s_red[256];
    const int t = blockIdx.x;
    if (t >= ntensors) return;
    const int b0 = tensor_block_off[t], b1 = tensor_block_off[t + 1];
    double s = 0.0;
    for (int b = b0 + threadIdx.x; b < b1; b += 256) s += partial[b];
    s_red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) s_red[threadIdx.x] += s_red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float nrm = (float)sqrt(s_red[0]);
        if (norms) norms[t] = nrm;
        scale[t] = clip > 0.f ? clip / fmaxf(nrm, clip) : 1.f;
    }
}

// g *= scale[tensor(block)]   (in place; used before the data-parallel all-reduce)
__global__ __launch_bounds__(256) void k_apply_scale(float* __restrict__ g, long long n, const int* __restrict__ block_tensor,
                                                     const float* __restrict__ scale) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const float sc = scale[block_tensor[blockIdx.x]];
    float4 v = *reinterpret_cast<float4*>(g + i);
    v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
    *reinterpret_cast<float4*>(g + i) = v;
}

// acc (+)= g * scale[tensor(block)]   (micro-batch accumulation of clipped gradients, models/ssd_model.py:251-255)
__global__ __launch_bounds__(256) void k_accumulate(float* __restrict__ acc, const float* __restrict__ g, long long n,
                                                    const int* __restrict__ block_tensor, const float* __restrict__ scale,
                                                    int first) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const float sc = scale ? scale[block_tensor[blockIdx.x]] : 1.f;
    const float4 gv = *reinterpret_cast<const float4*>(g + i);
    float4 a = first ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<float4*>(acc + i);
    a.x += gv.x * sc; a.y += gv.y * sc; a.z += gv.z * sc; a.w += gv.w * sc;
    *reinterpret_cast<float4*>(acc + i) = a;
}

// Keras Adam: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr_t m / (sqrt(v) + eps), lr_t given.
// g is first multiplied by grad_scale * (scale ? scale[tensor] : 1).  Also refreshes the bf16 copy.
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, unsigned short* __restrict__ p_bf16, long long n,
                                              const int* __restrict__ block_tensor, const float* __restrict__ scale,
                                              float grad_scale, float lr_t, float b1, float b2, float eps) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const float sc = grad_scale * (scale ? scale[block_tensor[blockIdx.x]] : 1.f);
    const float4 gv = *reinterpret_cast<const float4*>(g + i);
    float4 pv = *reinterpret_cast<float4*>(p + i);
    float4 mv = *reinterpret_cast<float4*>(m + i);
    float4 vv = *reinterpret_cast<float4*>(v + i);
    const float gs[4] = {gv.x * sc, gv.y * sc, gv.z * sc, gv.w * sc};
    float* pp = reinterpret_cast<float*>(&pv);
    float* mm = reinterpret_cast<float*>(&mv);
    float* vq = reinterpret_cast<float*>(&vv);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mm[k] = b1 * mm[k] + (1.f - b1) * gs[k];
        vq[k] = b2 * vq[k] + (1.f - b2) * gs[k] * gs[k];
        pp[k] -= lr_t * mm[k] / (sqrtf(vq[k]) + eps);
    }
    *reinterpret_cast<float4*>(p + i) = pv;
    *reinterpret_cast<float4*>(m + i) = mv;
    *reinterpret_cast<float4*>(v + i) = vv;
    if (p_bf16)
        *reinterpret_cast<uint2*>(p_bf16 + i) = make_uint2((unsigned)f2bf_bits(pp[0]) | ((unsigned)f2bf_bits(pp[1]) << 16),
                                                           (unsigned)f2bf_bits(pp[2]) | ((unsigned)f2bf_bits(pp[3]) << 16));
}

// plain SGD (tools/train.py:44-45 accepts name == "sgd"): p -= lr * g
__global__ __launch_bounds__(256) void k_sgd(float* __restrict__ p, const float* __restrict__ g, unsigned short* __restrict__ p_bf16,
                                             long long n, const int* __restrict__ block_tensor, const float* __restrict__ scale,
                                             float grad_scale, float lr) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const float sc = grad_scale * (scale ? scale[block_tensor[blockIdx.x]] : 1.f);
    const float4 gv = *reinterpret_cast<const float4*>(g + i);
    float4 pv = *reinterpret_cast<float4*>(p + i);
    pv.x -= lr * gv.x * sc; pv.y -= lr * gv.y * sc; pv.z -= lr * gv.z * sc; pv.w -= lr * gv.w * sc;
    *reinterpret_cast<float4*>(p + i) = pv;
    if (p_bf16)
        *reinterpret_cast<uint2*>(p_bf16 + i) = make_uint2((unsigned)f2bf_bits(pv.x) | ((unsigned)f2bf_bits(pv.y) << 16),
                                                           (unsigned)f2bf_bits(pv.z) | ((unsigned)f2bf_bits(pv.w) << 16));
}

}  // namespace

extern "C" {

int ssd_opt_block_elems(void) { return OPT_BLOCK; }

int ssd_grad_clip_scales(const float* grad, long long n, const int32_t* tensor_block_off, int ntensors, float clip,
                         double* partial, float* scale, float* norms, void* stream) {
    if (!grad || !tensor_block_off || !partial || !scale || n <= 0 || n % OPT_BLOCK || ntensors <= 0) return SSD_ERR_VALUE;
    const unsigned nb = (unsigned)(n / OPT_BLOCK);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sqnorm_partial, dim3(nb), dim3(256), 0, s, grad, n, partial);
    hipLaunchKernelGGL(k_clip_scale, dim3(ntensors), dim3(256), 0, s, partial, tensor_block_off, ntensors, clip,
                       scale, norms);
    return ssd_launch_status();
}

int ssd_grad_apply_scale(float* grad, long long n, const int32_t* block_tensor, const float* scale, void* stream) {
    if (!grad || !block_tensor || !scale || n <= 0 || n % OPT_BLOCK) return SSD_ERR_VALUE;
    hipLaunchKernelGGL(k_apply_scale, dim3((unsigned)(n / OPT_BLOCK)), dim3(256), 0, (hipStream_t)stream, grad, n,
                       block_tensor, scale);
    return ssd_launch_status();
}

int ssd_grad_accumulate(float* acc, const float* grad, long long n, const int32_t* block_tensor, const float* scale,
                        int first, void* stream) {
    if (!acc || !grad || n <= 0 || n % OPT_BLOCK || (scale && !block_tensor)) return SSD_ERR_VALUE;
    hipLaunchKernelGGL(k_accumulate, dim3((unsigned)(n / OPT_BLOCK)), dim3(256), 0, (hipStream_t)stream, acc, grad, n,
                       block_tensor, scale, first);
    return ssd_launch_status();
}

int ssd_adam_step(float* param, const float* grad, float* m, float* v, void* param_bf16, long long n,
                  const int32_t* block_tensor, const float* scale, float grad_scale, float lr_t, float beta1, float beta2,
                  float eps, void* stream) {
    if (!param || !grad || !m || !v || n <= 0 || n % OPT_BLOCK || (scale && !block_tensor)) return SSD_ERR_VALUE;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)(n / OPT_BLOCK)), dim3(256), 0, (hipStream_t)stream, param, grad, m, v,
                       static_cast<unsigned short*>(param_bf16), n, block_tensor, scale, grad_scale, lr_t, beta1, beta2, eps);
    return ssd_launch_status();
}

int ssd_sgd_step(float* param, const float* grad, void* param_bf16, long long n, const int32_t* block_tensor,
                 const float* scale, float grad_scale, float lr, void* stream) {
    if (!param || !grad || n <= 0 || n % OPT_BLOCK || (scale && !block_tensor)) return SSD_ERR_VALUE;
    hipLaunchKernelGGL(k_sgd, dim3((unsigned)(n / OPT_BLOCK)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       static_cast<unsigned short*>(param_bf16), n, block_tensor, scale, grad_scale, lr);
    return ssd_launch_status();
}

}  // extern "C"
