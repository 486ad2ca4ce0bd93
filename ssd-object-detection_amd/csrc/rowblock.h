// Shared device helpers: dtype conversion and coalesced staging of a block of logit rows into LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>

constexpr int RB_WG = 256;                   // workgroup size the helpers assume

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __hip_bfloat16 from_f32<__hip_bfloat16>(float v) { return __float2bfloat16(v); }

// Stage `count` contiguous elements starting at src (16-byte aligned) into LDS as float, coalesced
// 16 B per lane, STAGE_DEPTH loads in flight per lane before the first LDS write.
constexpr int STAGE_DEPTH = 12;
template <typename T>
__device__ __forceinline__ void stage_block(const T* __restrict__ src, size_t count, float* lds) {
    constexpr int PER = 16 / sizeof(T);
    const size_t nvec = count / PER;
    const uint4* v = reinterpret_cast<const uint4*>(src);
    for (size_t base = 0; base < nvec; base += (size_t)STAGE_DEPTH * RB_WG) {
        uint4 raw[STAGE_DEPTH];
        const size_t last = nvec ? nvec - 1 : 0;
#pragma unroll
        for (int j = 0; j < STAGE_DEPTH; ++j) {
            const size_t i = base + (size_t)j * RB_WG + threadIdx.x;
            raw[j] = v[i < last ? i : last];               // unconditional: keeps `raw` in registers, the loads in flight together
        }
#pragma unroll
        for (int j = 0; j < STAGE_DEPTH; ++j) {
            const size_t i = base + (size_t)j * RB_WG + threadIdx.x;
            if (i >= nvec) continue;
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<float4*>(lds + i * 4) = make_float4(__uint_as_float(raw[j].x), __uint_as_float(raw[j].y),
                                                                      __uint_as_float(raw[j].z), __uint_as_float(raw[j].w));
            } else {
                const unsigned w[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
                float f[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f[2 * k] = __uint_as_float(w[k] << 16);
                    f[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
                }
                *reinterpret_cast<float4*>(lds + i * 8) = make_float4(f[0], f[1], f[2], f[3]);
                *reinterpret_cast<float4*>(lds + i * 8 + 4) = make_float4(f[4], f[5], f[6], f[7]);
            }
        }
    }
    for (size_t i = nvec * PER + threadIdx.x; i < count; i += RB_WG) lds[i] = to_f32<T>(src[i]);
}


// The same staging split in two, for a loop that requests the NEXT block's elements (stage_load) before it computes on the
// current one and writes them to LDS (stage_store) after the following barrier: the global loads are in flight during the
// computation.  NV = ceil(count / (16 / sizeof(T)) / RB_WG) registers of 16 bytes per lane.
// Every lane loads unconditionally (a lane past the end re-reads the last chunk, which stage_store then ignores): with a
// conditional assignment the compiler keeps `raw` in scratch memory and waits for each load before it issues the next one --
// eleven serialised memory round trips per block instead of eleven loads in flight (k_score_decode: 94 us -> see DESIGN.md).
template <typename T, int NV>
__device__ __forceinline__ void stage_load(const T* __restrict__ src, size_t count, uint4 (&raw)[NV]) {
    constexpr int PER = 16 / sizeof(T);
    const size_t nvec = count / PER;
    const uint4* v = reinterpret_cast<const uint4*>(src);
    if (nvec == 0) return;                      // fewer elements than one 16-byte chunk: stage_store takes them one by one
    const size_t last = nvec - 1;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const size_t i = (size_t)j * RB_WG + threadIdx.x;
        raw[j] = v[i < last ? i : last];
    }
}

template <typename T, int NV>
__device__ __forceinline__ void stage_store(const T* __restrict__ src, size_t count, const uint4 (&raw)[NV], float* lds) {
    constexpr int PER = 16 / sizeof(T);
    const size_t nvec = count / PER;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const size_t i = (size_t)j * RB_WG + threadIdx.x;
        if (i >= nvec) continue;
        if constexpr (sizeof(T) == 4) {
            // (component-wise: a whole-uint4 copy out of `raw` is lowered as a memory copy and pins the array in scratch)
            *reinterpret_cast<float4*>(lds + i * 4) = make_float4(__uint_as_float(raw[j].x), __uint_as_float(raw[j].y),
                                                                  __uint_as_float(raw[j].z), __uint_as_float(raw[j].w));
        } else {
            const unsigned w[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
            float f[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f[2 * k] = __uint_as_float(w[k] << 16);
                f[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
            }
            *reinterpret_cast<float4*>(lds + i * 8) = make_float4(f[0], f[1], f[2], f[3]);
            *reinterpret_cast<float4*>(lds + i * 8 + 4) = make_float4(f[4], f[5], f[6], f[7]);
        }
    }
    for (size_t i = nvec * PER + threadIdx.x; i < count; i += RB_WG) lds[i] = to_f32<T>(src[i]);   // ragged tail (last block only)
}
