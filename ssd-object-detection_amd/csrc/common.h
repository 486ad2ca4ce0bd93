// Shared helpers for the gfx950 SSD hot-path library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ssd_hip.h"

#define SSD_ABI_VERSION 1

static inline int ssd_launch_status() {
    return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

static inline size_t ssd_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// 64-lane wavefront helpers (gfx950: wave64 only)
#define SSD_WAVE 64

// Development overrides (ssd_dev_knob, include/ssd_hip.h); defined in conv.hip.  Never set by the product.
int ssd_knob(const char* name, int dflt);

// Persistent pointwise-convolution GEMM (pwgemm.hip), called from conv.hip's dispatch.  geom / epilogue: the ConvGeom / Epilogue
// of conv_common.h (both translation units include that header).
bool ssd_pw_gemm_serves(int epi, const void* geom, const void* epilogue);
int ssd_pw_gemm_launch(int epi, const void* x, const void* w, const void* geom, const void* epilogue, void* ws, size_t ws_bytes,
                       void* stream);

// fixed-order sum of weight-gradient slabs (k_wgrad_reduce2 / k_wgrad_reduce_wide, conv.hip): dW = sum over ns splits
void ssd_launch_wgrad_reduce(hipStream_t s, const float* slab_w, long long sw, long long nw, float* dw, const float* slab_b,
                             long long sb, int nb, float* db, int ns);
