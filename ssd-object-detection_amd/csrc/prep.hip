// Input-side preprocessing on the device: what the reference does per image on the host
// between the decoded JPEG and the network input,
//   image / 255                      data_loaders/coco/make_dataset.py:117
//   cv2.resize(image, (300, 300))    data_loaders/ssd/make_dataset.py:40      (INTER_LINEAR, float image)
//   (x - 0.5) * 2                    models/ssd_model.py:214
// and per box
//   xy += wh / 2                     data_loaders/coco/make_dataset.py:132    (COCO top-left -> centre)
//   box /= [w, h, w, h]              data_loaders/ssd/make_dataset.py:43-44
// for a ragged batch of uint8 RGB images in one launch each.  cv2 is not installable here: the bilinear rule below
// restates OpenCV's published INTER_LINEAR for float images (half-pixel centres, coefficient 1-f / f in float32,
// horizontal pass then vertical pass, edge columns/rows clamped); oracle/ssd_oracle.py:resize_bilinear is its CPU twin.
#include "common.h"

namespace {

typedef unsigned short bf16_raw;

__device__ __forceinline__ bf16_raw f2bf_rn(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);                        // round to nearest even (inputs are finite)
    return (bf16_raw)(u >> 16);
}

// source coordinate of destination index d: cv2's  fx = (float)((d + 0.5) * scale - 0.5),  s = floor(fx), f = fx - s,
// clamped so that s, s+1 stay inside [0, n-1]
__device__ __forceinline__ void src_coord(int d, double scale, int n, int& s0, int& s1, float& f) {
    float fx = (float)__dadd_rn(__dmul_rn((double)d + 0.5, scale), -0.5);   // product and sum rounded separately, as on the host
    int s = (int)floorf(fx);
    fx -= (float)s;
    if (s < 0) { s = 0; fx = 0.f; }
    if (s >= n - 1) { s = n - 1; fx = 0.f; }
    s0 = s;
    s1 = s + 1 < n ? s + 1 : n - 1;
    f = fx;
}

__global__ void k_image_resize_prep(const unsigned char* __restrict__ src, const long long* __restrict__ src_off,
                                    const int* __restrict__ src_hw, bf16_raw* __restrict__ out, int S, int normalize) {
    // u8 / 255 as the reference's float64 division rounded to float32 (TensorSpec float32): 256 possible values, one
    // per thread of the block
    __shared__ float lut[256];
    lut[threadIdx.x] = (float)((double)threadIdx.x / 255.0);
    __syncthreads();
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * S) return;
    const int dy = i / S, dx = i - dy * S;
    const int H = src_hw[2 * b], W = src_hw[2 * b + 1];
    const unsigned char* img = src + src_off[b];
    int x0, x1, y0, y1;
    float fx, fy;
    src_coord(dx, (double)W / (double)S, W, x0, x1, fx);
    src_coord(dy, (double)H / (double)S, H, y0, y1, fy);
    const float ax0 = 1.f - fx, ay0 = 1.f - fy;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float p00 = lut[img[((long long)y0 * W + x0) * 3 + c]];
        const float p01 = lut[img[((long long)y0 * W + x1) * 3 + c]];
        const float p10 = lut[img[((long long)y1 * W + x0) * 3 + c]];
        const float p11 = lut[img[((long long)y1 * W + x1) * 3 + c]];
        // separate multiplies and adds (no contraction), horizontal pass first
        const float h0 = __fadd_rn(__fmul_rn(p00, ax0), __fmul_rn(p01, fx));
        const float h1 = __fadd_rn(__fmul_rn(p10, ax0), __fmul_rn(p11, fx));
        float r = __fadd_rn(__fmul_rn(h0, ay0), __fmul_rn(h1, fy));
        if (normalize) r = __fmul_rn(__fadd_rn(r, -0.5f), 2.f);
        v[c] = r;
    }
    *reinterpret_cast<uint4*>(out + ((long long)b * S * S + i) * 8) =
        make_uint4((unsigned)f2bf_rn(v[0]) | ((unsigned)f2bf_rn(v[1]) << 16), (unsigned)f2bf_rn(v[2]), 0u, 0u);
}

__global__ void k_box_prep(const float4* __restrict__ box, const int* __restrict__ gt_off, const int* __restrict__ src_hw,
                           float4* __restrict__ out, int B, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int lo = 0, hi = B;                                     // image of box i: gt_off[b] <= i < gt_off[b+1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (gt_off[mid] <= i) lo = mid; else hi = mid;
    }
    const float h = (float)src_hw[2 * lo], w = (float)src_hw[2 * lo + 1];
    float4 v = box[i];
    // float32 arithmetic in the reference's order: centre first (in pixels), then the division
    v.x = __fadd_rn(v.x, __fdiv_rn(v.z, 2.f));
    v.y = __fadd_rn(v.y, __fdiv_rn(v.w, 2.f));
    out[i] = make_float4(__fdiv_rn(v.x, w), __fdiv_rn(v.y, h), __fdiv_rn(v.z, w), __fdiv_rn(v.w, h));
}

}  // namespace

extern "C" {

int ssd_image_resize_prep(const void* src, const int64_t* src_off, const int32_t* src_hw, void* out, int B, int S,
                          int normalize, void* stream) {
    if (!src || !src_off || !src_hw || !out || B <= 0 || S <= 0) return SSD_ERR_VALUE;
    hipLaunchKernelGGL(k_image_resize_prep, dim3((unsigned)((S * S + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const unsigned char*>(src), reinterpret_cast<const long long*>(src_off), src_hw,
                       static_cast<bf16_raw*>(out), S, normalize);
    return ssd_launch_status();
}

int ssd_box_prep(const float* box_tlwh, const int32_t* gt_off, const int32_t* src_hw, float* box_out, int B, int total_gt,
                 void* stream) {
    if (B <= 0 || total_gt < 0 || !gt_off || !src_hw) return SSD_ERR_VALUE;
    if (total_gt == 0) return SSD_OK;
    if (!box_tlwh || !box_out) return SSD_ERR_VALUE;
    hipLaunchKernelGGL(k_box_prep, dim3((unsigned)((total_gt + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(box_tlwh), gt_off, src_hw, reinterpret_cast<float4*>(box_out), B, total_gt);
    return ssd_launch_status();
}

}  // extern "C"
