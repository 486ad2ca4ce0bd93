// Types, geometry, epilogues and host helpers shared by the convolution translation units (conv.hip, sparse.hip, calib.hip).
#pragma once
#include <atomic>
#include <type_traits>
#include "common.h"
#include <hip/hip_bf16.h>

namespace {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4_t;
typedef unsigned short bf16_raw;

constexpr int WG = 256;

struct FastDiv {                             // exact n / d for 0 <= n < 2^31
    unsigned mg;
    int sh;                                  // sh < 0: d == 1
    int d;
};

inline FastDiv make_fastdiv(int d) {
    FastDiv f;
    f.d = d;
    if (d <= 1) { f.mg = 0; f.sh = -1; return f; }
    int l = 0;
    while ((1ll << l) < d) ++l;
    const int s = 31 + l;
    f.mg = (unsigned)(((1ull << s) + (unsigned long long)d - 1) / (unsigned long long)d);
    f.sh = s - 32;
    return f;
}

__device__ __forceinline__ int fdiv(int n, const FastDiv& f) {
    return f.sh < 0 ? n : (int)(__umulhi((unsigned)n, f.mg) >> f.sh);
}

struct ConvGeom {
    int B, H, W, C;                          // source tensor (NHWC), C % 8 == 0
    int Ho, Wo, N;                           // destination spatial dims and channel count (GEMM N)
    int KH, KW;
    int mul, div, pad_t, pad_l;              // source coordinate = (o*mul + k - pad) / div
    int M;                                   // B*Ho*Wo
    int nchunks;                             // KH*KW*C/8  (16-byte k chunks)
    int ldw;                                 // weight row stride (elements) = KH*KW*C
    int cpt;                                 // chunks per tap = C/8
    FastDiv d_hw, d_w;                       // divide by Ho*Wo and by Wo
    FastDiv d_h1;                            // divide by H + 1 (row strip of k_conv3x3_patch32)
    // stride-2 data gradient: destination pixels are enumerated parity class by parity class (py,px), each class
    // padded to whole tiles, so that a tile has ONE parity and the taps that cannot hit it are skipped outright
    int s2;                                  // 1: parity-class enumeration active (div == 2, cpt % 8 == 0)
    int cls_n[4];                            // pixels per class (B * cls_h[py] * cls_w[px]), class = 2*py + px
    int cls_h[2], cls_w[2];
    int ablate;                              // dev only (SSD_ABLATE): 1 no DMA after the prologue, 2 no wait/barrier, 4 no MFMA
    int dma32;                               // source tensor and weights below 4 GB: LDS-DMA through buffer descriptors (32-bit byte offsets)
};

enum { EPI_FWD = 0, EPI_HEAD = 1, EPI_DGRAD = 2 };
#ifndef SSD_DMA_SLOT
#define SSD_DMA_SLOT 0                        // 0: issue the next tile's DMA before the MFMAs, 1: between the two MFMA blocks
#endif
#ifndef SSD_PT256
#define SSD_PT256 4                           // pixel tiles per wave of the 256x256 implicit-GEMM tile (8 = 8 waves of 128x64: no faster)
#endif

struct Epilogue {
    const float* bias;                       // [N] or null                       (FWD, HEAD)
    int relu;                                //                                    (FWD)
    bf16_raw* out;                           // [M][ldo]                           (FWD, DGRAD)
    int ldo;
    const bf16_raw* mask_src;                // DGRAD: zero where mask_src <= 0 (ReLU backward), may be null
    int accumulate;                          // DGRAD: out += result
    // HEAD: columns [0,n_loc) -> loc, [n_loc, n_loc+n_conf) -> conf, in the reference's concatenated layout
    bf16_raw* loc;
    bf16_raw* conf;
    int n_loc, n_conf;                       // per pixel: n*4 and n*C
    int anchors_total;                       // A
    int level_off;                           // first anchor of this level
    int per_cell;                            // n
    int classes;                             // C
    // split-K: when slab != null the kernel stores raw fp32 partial sums to slab[split][M][N] and k_igemm_finalize
    // applies the epilogue to their fixed-order sum
    float* slab;
    int ksplit;
    // FWD, 16x16-block kernels only: also write the 2x2 / stride-2 max pooling of this layer's output (and the winner
    // codes of k_maxpool_fwd_argmax), computed from the staged tile: saves the pooling kernel's re-read of the output
    bf16_raw* pool_out;                      // [B][pool_h][pool_w][N] or null
    unsigned* pool_code;                     // [B][pool_h][pool_w][N/8]
    int pool_h, pool_w;
    // DGRAD, kernels with the staged epilogue only: the result is the gradient of a POOLED map; instead of storing it (out
    // may be null) route it through the 2x2 / stride-2 max pooling behind it (the winner codes of k_maxpool_fwd_argmax) and
    // store the full-resolution gradient -- k_maxpool_bwd_argmax computed in the epilogue, without the round trip of the
    // pooled gradient through HBM and without the launch
    const unsigned* up_code;                 // [B][Ho][Wo][N/8] or null
    bf16_raw* up_out;                        // [B][up_h][up_w][N]
    int up_h, up_w;
    // ReLU sign bits, one byte per (pixel, 8 channels): bit k = channel 8c + k of the pixel is > 0.  FWD (kernels with the
    // staged store, k_conv0_fwd): written next to the activation.  DGRAD: read INSTEAD of mask_src -- the data gradients
    // spent 13-130 us each on re-reading whole bf16 activations for their sign (16x the bytes, exposed at the store tail)
    unsigned char* relu_bits;                // FWD out [M][N/8] or null
    const unsigned char* mask_bits;          // DGRAD in [M][N/8] or null
};

// bit k set <=> bf16 element k of the 16-byte chunk is > 0
__device__ __forceinline__ unsigned relu_bits8(const uint4& v) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    unsigned b = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        b |= (__uint_as_float(w[k] << 16) > 0.f ? 1u : 0u) << (2 * k);
        b |= (__uint_as_float(w[k] & 0xffff0000u) > 0.f ? 2u : 0u) << (2 * k);
    }
    return b;
}
// word mask of two sign bits: bit `pos` keeps the low bf16, bit `pos + 1` the high one (two sign-extending bit-field extracts
// and one byte permute; written with selects it was six vector instructions per word in every data-gradient epilogue)
__device__ __forceinline__ unsigned keep_mask2(unsigned bits, int pos) {
    const unsigned lo = (unsigned)__builtin_amdgcn_sbfe((int)bits, pos, 1), hi = (unsigned)__builtin_amdgcn_sbfe((int)bits, pos + 1, 1);
    return __builtin_amdgcn_perm(hi, lo, 0x07060100u);      // bytes 0, 1 of lo; bytes 2, 3 of hi
}
// zero the elements of a chunk whose bit is clear
__device__ __forceinline__ uint4 gate_bits8(uint4 v, unsigned b) {
    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] &= keep_mask2(b, 2 * k);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// two floats -> two bf16 in one word (low half = a): ONE v_cvt_pk_bf16_f32 (the same round-to-nearest-even instruction that
// f2bf() lowers to, so results are identical); composed from two f2bf() the compiler emitted two conversions, a shift and an or
typedef __bf16 ssd_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float ssd_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    const ssd_f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ssd_bf16x2_t));
}
__device__ __forceinline__ float bf2f(bf16_raw v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ bf16_raw f2bf(float f) {
    const __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<const bf16_raw*>(&h);
}

// LDS slot of 16-byte chunk `chunk` (0..7) of tile row `row`: two rows share one 256-byte bank row,
// slots XOR-swizzled so that 16 consecutive rows reading the same chunk hit 16 different slots.
__device__ __forceinline__ int swz(int row, int chunk) {
    return (row >> 1) * 256 + ((((row & 1) << 3) | (chunk ^ ((row >> 1) & 7))) << 4);
}

// Epilogue shared by the implicit-GEMM kernels: lane holds channels n_base + (lane>>4)*4 + {0..3} of pixel
// m_base + (lane&15) for every (channel tile, pixel tile) of its wave.
// Store 4 consecutive output channels n..n+3 of output pixel m (bias already added to v).
template <int EPI>
__device__ __forceinline__ void epi_store(float (&v)[4], int m, int n, bool full, const ConvGeom& g, const Epilogue& ep) {
    if constexpr (EPI == EPI_FWD) {
        if (ep.relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        bf16_raw* o = ep.out + (long long)m * ep.ldo + n;
        if (full) {
            *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf16x2(v[0], v[1]),
                                                      pack_bf16x2(v[2], v[3]));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (n + j < g.N) o[j] = f2bf(v[j]);
        }
    } else if constexpr (EPI == EPI_DGRAD) {
        bf16_raw* o = ep.out + (long long)m * ep.ldo + n;
        const bf16_raw* ms = ep.mask_src ? ep.mask_src + (long long)m * ep.ldo + n : nullptr;
        if (full) {
            if (ep.accumulate) {
                const uint2 old = *reinterpret_cast<const uint2*>(o);
                v[0] += __uint_as_float(old.x << 16); v[1] += __uint_as_float(old.x & 0xffff0000u);
                v[2] += __uint_as_float(old.y << 16); v[3] += __uint_as_float(old.y & 0xffff0000u);
            }
            if (ms) {
                const uint2 mk = *reinterpret_cast<const uint2*>(ms);
                if (!(__uint_as_float(mk.x << 16) > 0.f)) v[0] = 0.f;
                if (!(__uint_as_float(mk.x & 0xffff0000u) > 0.f)) v[1] = 0.f;
                if (!(__uint_as_float(mk.y << 16) > 0.f)) v[2] = 0.f;
                if (!(__uint_as_float(mk.y & 0xffff0000u) > 0.f)) v[3] = 0.f;
            }
            *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf16x2(v[0], v[1]),
                                                      pack_bf16x2(v[2], v[3]));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (n + j >= g.N) continue;
                float r = v[j];
                if (ep.accumulate) r += bf2f(o[j]);
                if (ms && !(bf2f(ms[j]) > 0.f)) r = 0.f;
                o[j] = f2bf(r);
            }
        }
    } else {  // EPI_HEAD: scatter into loc [B][A][4] and conf [B][A][classes]
        const int b = fdiv(m, g.d_hw);
        const int pix = m - b * g.d_hw.d;
        const long long anchor0 = (long long)b * ep.anchors_total + ep.level_off + (long long)pix * ep.per_cell;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nn = n + j;
            if (nn >= ep.n_loc + ep.n_conf) continue;
            if (nn < ep.n_loc) ep.loc[anchor0 * 4 + nn] = f2bf(v[j]);
            else ep.conf[anchor0 * ep.classes + (nn - ep.n_loc)] = f2bf(v[j]);
        }
    }
}

__device__ __forceinline__ void load_bias4(const Epilogue& ep, int n, int N, bool full, float (&b4)[4]) {
    b4[0] = b4[1] = b4[2] = b4[3] = 0.f;
    if (!ep.bias) return;
    if (full) {
        const float4 bv = *reinterpret_cast<const float4*>(ep.bias + n);
        b4[0] = bv.x; b4[1] = bv.y; b4[2] = bv.z; b4[3] = bv.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (n + j < N) b4[j] = ep.bias[n + j];
    }
}

template <int EPI, int CT, int PT>
__device__ __forceinline__ void conv_epilogue_rows(f32x4_t (&acc)[CT][PT], const ConvGeom& g, const Epilogue& ep,
                                                   const int (&mrow)[PT], int nbase, int lane) {
    // mrow[p]: flat output pixel of this lane for pixel tile p (-1: outside); nbase: first channel of the wave
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int n = nbase + c * 16 + (lane >> 4) * 4;
        if (n >= g.N) continue;
        const bool full = n + 3 < g.N;
        if (ep.slab) {                           // split-K partial sums (uniform branch)
            float* sl = ep.slab + (long long)blockIdx.y * g.M * g.N;
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const int m = mrow[p];
                if (m < 0) continue;
                float* o = sl + (long long)m * g.N + n;
                if (full && !(g.N & 3)) {                 // 16-byte aligned: one store
                    *reinterpret_cast<float4*>(o) = make_float4(acc[c][p][0], acc[c][p][1], acc[c][p][2], acc[c][p][3]);
                    continue;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) if (n + j < g.N) o[j] = acc[c][p][j];
            }
            continue;
        }
        float bias4[4];
        if constexpr (EPI == EPI_DGRAD) { bias4[0] = bias4[1] = bias4[2] = bias4[3] = 0.f; }
        else load_bias4(ep, n, g.N, full, bias4);
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int m = mrow[p];
            if (m < 0) continue;
            float v[4] = {acc[c][p][0] + bias4[0], acc[c][p][1] + bias4[1], acc[c][p][2] + bias4[2], acc[c][p][3] + bias4[3]};
            epi_store<EPI>(v, m, n, full, g, ep);
        }
    }
}

// Epilogue through LDS: the accumulator layout (a lane holds 4 channels of 16 different pixels) gives 8-byte stores in
// 32-byte runs (2-byte scattered stores for the head layout); staging the [BM px][BN ch] bf16 tile in LDS first turns
// them into whole 16-byte chunks of contiguous rows (heads: 64 consecutive elements per wave-instruction).
// Tile image: BN*2-byte rows, 16-byte chunk index XORed with the row so that both the 8-byte fragment writes and the
// row-wise reads are (nearly) conflict-free.  Caller guarantees: all waves are past their last LDS read and no LDS-DMA
// is in flight.  row_to_m(row) -> flat output pixel or -1.
template <int EPI>
__device__ __forceinline__ bool staged_ok(const ConvGeom& g, const Epilogue& ep) {
    if (ep.slab) return false;
    if constexpr (EPI == EPI_FWD) return (g.N & 7) == 0 && (ep.ldo & 7) == 0;
    if constexpr (EPI == EPI_DGRAD) return (g.N & 7) == 0 && (ep.ldo & 7) == 0;
    return true;                                // EPI_HEAD
}

struct NoPool { __device__ long long operator()(int, int) const { return -1; } };

// LDS accesses that carry an alias scope (the __restrict__ pair, inlined).  The compiler's waitcnt pass puts s_waitcnt vmcnt(0)
// in front of every LDS access WITHOUT scope information while an LDS-DMA is in flight -- in a persistent kernel that
// prefetches its next block by DMA (k_conv3x3_c64b) that is a wait for the next block's patch in the middle of the epilogue.
// The kernels order DMA and LDS accesses themselves (counted s_waitcnt + barrier); elsewhere the scope changes nothing.
__device__ __forceinline__ uint4 lds_ld16_scoped(const char* __restrict__ p, const char* __restrict__ other) {
    (void)other;
    return *reinterpret_cast<const uint4*>(p);
}
__device__ __forceinline__ void lds_st16_scoped(char* __restrict__ p, const char* __restrict__ other, uint4 v) {
    (void)other;
    *reinterpret_cast<uint4*>(p) = v;
}
__device__ __forceinline__ void lds_st8_scoped(char* __restrict__ p, const char* __restrict__ other, uint2 v) {
    (void)other;
    *reinterpret_cast<uint2*>(p) = v;
}

// Second half of the staged epilogue: the [BM px][BN ch] bf16 tile image in LDS (layout above; FWD: bias and ReLU already
// applied) leaves as whole 16-byte chunks of contiguous rows; DGRAD applies accumulate / ReLU mask here, FWD the fused pooling.
// Caller: a barrier between the last write of the image and this call.
template <int EPI, int BM, int BN, int NT, typename RowMap, typename PoolMap, bool DG_BATCHED = true>
__device__ __forceinline__ void staged_store(char* smem, const ConvGeom& g, const Epilogue& ep, int n0, int tid, RowMap row_to_m,
                                             PoolMap pool_index) {
    constexpr int CPR = BN / 8;                 // 16-byte chunks per tile row
    if constexpr (EPI == EPI_HEAD) {
        // A pixel's loc run (per_cell x 4) and conf run (per_cell x classes) are contiguous in the outputs but only 2-byte
        // aligned in general.  When the anchor counts are even (all six SSD levels: 4 or 6 per cell, even level offsets and
        // anchor total) every run starts 4-byte aligned and has even length: two channels per store, 64 consecutive channel
        // pairs of one pixel per wave-instruction -- half the store instructions of the element-wise form below
        const bool pairs = !((ep.per_cell | ep.level_off | ep.anchors_total | ep.n_loc | ep.n_conf) & 1);
        if (pairs) {
            constexpr int ITER2 = BM * BN / (2 * NT);
            for (int it = 0; it < ITER2; ++it) {
                const int idx = it * NT + tid;
                const int row = idx / (BN / 2), col = (idx - row * (BN / 2)) * 2;
                const int n = n0 + col;
                const int m = row_to_m(row);
                if (m < 0 || n >= ep.n_loc + ep.n_conf) continue;
                const unsigned val = *reinterpret_cast<const unsigned*>(smem + row * (BN * 2) + ((((col >> 3) ^ row) & (CPR - 1)) << 4) + (col & 7) * 2);
                const int b = fdiv(m, g.d_hw);
                const int pix = m - b * g.d_hw.d;
                const long long anchor0 = (long long)b * ep.anchors_total + ep.level_off + (long long)pix * ep.per_cell;
                if (n < ep.n_loc) *reinterpret_cast<unsigned*>(ep.loc + anchor0 * 4 + n) = val;
                else *reinterpret_cast<unsigned*>(ep.conf + anchor0 * ep.classes + (n - ep.n_loc)) = val;
            }
            return;
        }
        // element-wise, 64 consecutive channels of one pixel per wave-instruction
        constexpr int ITER = BM * BN / NT;
        for (int it = 0; it < ITER; ++it) {
            const int idx = it * NT + tid;
            const int row = idx / BN, col = idx - row * BN;
            const int n = n0 + col;
            const int m = row_to_m(row);
            if (m < 0 || n >= ep.n_loc + ep.n_conf) continue;
            const bf16_raw val = *reinterpret_cast<const bf16_raw*>(smem + row * (BN * 2) + ((((col >> 3) ^ row) & (CPR - 1)) << 4) + (col & 7) * 2);
            const int b = fdiv(m, g.d_hw);
            const int pix = m - b * g.d_hw.d;
            const long long anchor0 = (long long)b * ep.anchors_total + ep.level_off + (long long)pix * ep.per_cell;
            if (n < ep.n_loc) ep.loc[anchor0 * 4 + n] = val;
            else ep.conf[anchor0 * ep.classes + (n - ep.n_loc)] = val;
        }
    } else {
        constexpr int ITER = BM * CPR / NT;
        // forward with fused pooling and no consumer of the full-resolution map (out == nullptr): only the pooled map leaves
        const bool store_full = EPI != EPI_FWD || ep.out != nullptr;
        if constexpr (EPI == EPI_DGRAD && DG_BATCHED) {
            // Data gradient: a chunk needs up to three global reads before it can be stored (the value to accumulate onto, the
            // ReLU mask, the pooling codes).  Chunk by chunk -- load, wait, store, next -- every iteration was a full memory
            // round trip that also waited for the previous chunk's store (s_waitcnt vmcnt counts stores too): 16 serialised
            // round trips per 512-pixel tile with the matrix cores idle.  Here the reads of DG_BATCH chunks are issued
            // together, unconditionally (a chunk outside the map reads element 0 and is dropped), then the chunks are
            // finished: one round trip per batch.  (The registers are free: the accumulators went to LDS above.)
            constexpr int DG_BATCH = ITER % 4 == 0 ? 4 : (ITER % 2 == 0 ? 2 : 1);
            const long long nb8 = g.N >> 3;
            for (int it0 = 0; it0 < ITER; it0 += DG_BATCH) {
                int mm[DG_BATCH], rr[DG_BATCH], cc[DG_BATCH];
                long long oo[DG_BATCH];
                bool ok[DG_BATCH];
#pragma unroll
                for (int j = 0; j < DG_BATCH; ++j) {
                    const int idx = (it0 + j) * NT + tid;
                    rr[j] = idx / CPR; cc[j] = idx - rr[j] * CPR;
                    const int n = n0 + cc[j] * 8;
                    mm[j] = row_to_m(rr[j]);
                    ok[j] = mm[j] >= 0 && n < g.N;
                    oo[j] = ok[j] ? (long long)mm[j] * ep.ldo + n : 0;
                }
                uint4 old[DG_BATCH], mk[DG_BATCH];
                unsigned mb[DG_BATCH], cw[DG_BATCH];
#pragma unroll
                for (int j = 0; j < DG_BATCH; ++j) { old[j] = mk[j] = make_uint4(0, 0, 0, 0); mb[j] = 0xffu; cw[j] = 0u; }
                if (ep.accumulate) {
#pragma unroll
                    for (int j = 0; j < DG_BATCH; ++j) old[j] = *reinterpret_cast<const uint4*>(ep.out + oo[j]);
                }
                if (ep.mask_src) {
#pragma unroll
                    for (int j = 0; j < DG_BATCH; ++j) mk[j] = *reinterpret_cast<const uint4*>(ep.mask_src + oo[j]);
                } else if (ep.mask_bits) {
#pragma unroll
                    for (int j = 0; j < DG_BATCH; ++j) mb[j] = ep.mask_bits[ok[j] ? (long long)mm[j] * nb8 + ((n0 >> 3) + cc[j]) : 0];
                }
                if (ep.up_code) {
#pragma unroll
                    for (int j = 0; j < DG_BATCH; ++j) cw[j] = ep.up_code[ok[j] ? (long long)mm[j] * nb8 + ((n0 >> 3) + cc[j]) : 0];
                }
#pragma unroll
                for (int j = 0; j < DG_BATCH; ++j) {
                    if (!ok[j]) continue;
                    const int row = rr[j], ch = cc[j], m = mm[j];
                    const int n = n0 + ch * 8;
                    uint4 v = lds_ld16_scoped(smem + row * (BN * 2) + (((ch ^ row) & (CPR - 1)) << 4), smem);
                    if (ep.accumulate) {                          // out += result (two gradients meet at a feature map)
                        auto add2 = [](unsigned a, unsigned b) {
                            const float lo = __uint_as_float(a << 16) + __uint_as_float(b << 16);
                            const float hi = __uint_as_float(a & 0xffff0000u) + __uint_as_float(b & 0xffff0000u);
                            return pack_bf16x2(lo, hi);
                        };
                        v.x = add2(v.x, old[j].x); v.y = add2(v.y, old[j].y); v.z = add2(v.z, old[j].z); v.w = add2(v.w, old[j].w);
                    }
                    if (ep.mask_src) {                            // ReLU backward: zero where the forward activation was <= 0
                        auto gate = [](unsigned val, unsigned m2) {
                            if (!(__uint_as_float(m2 << 16) > 0.f)) val &= 0xffff0000u;
                            if (!(__uint_as_float(m2 & 0xffff0000u) > 0.f)) val &= 0x0000ffffu;
                            return val;
                        };
                        v.x = gate(v.x, mk[j].x); v.y = gate(v.y, mk[j].y); v.z = gate(v.z, mk[j].z); v.w = gate(v.w, mk[j].w);
                    } else if (ep.mask_bits) {
                        v = gate_bits8(v, mb[j]);
                    }
                    if (ep.up_code) {                             // un-pool: the four positions of the window, winner or zero
                        const int b = fdiv(m, g.d_hw);
                        const int rem = m - b * g.d_hw.d;
                        const int oy = fdiv(rem, g.d_w), ox = rem - oy * g.d_w.d;
                        const unsigned gw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int iy = 2 * oy + (q >> 1), ix = 2 * ox + (q & 1);
                            if (iy >= ep.up_h || ix >= ep.up_w) continue;
                            unsigned w4[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const unsigned lo = ((cw[j] >> (8 * k)) & 15u) == (unsigned)q ? 0x0000ffffu : 0u;
                                const unsigned hi = ((cw[j] >> (8 * k + 4)) & 15u) == (unsigned)q ? 0xffff0000u : 0u;
                                w4[k] = gw[k] & (lo | hi);
                            }
                            *reinterpret_cast<uint4*>(ep.up_out + (((long long)b * ep.up_h + iy) * ep.up_w + ix) * g.N + n) =
                                make_uint4(w4[0], w4[1], w4[2], w4[3]);
                        }
                        if (!ep.out) continue;
                    }
                    *reinterpret_cast<uint4*>(ep.out + oo[j]) = v;
                }
            }
        } else
#pragma unroll 4
        for (int it = 0; it < (store_full ? ITER : 0); ++it) {
            const int idx = it * NT + tid;
            const int row = idx / CPR, ch = idx - row * CPR;
            const int n = n0 + ch * 8;
            const int m = row_to_m(row);
            if (m < 0 || n >= g.N) continue;
            uint4 v = lds_ld16_scoped(smem + row * (BN * 2) + (((ch ^ row) & (CPR - 1)) << 4), smem);
            const long long o = (long long)m * ep.ldo + n;
            if constexpr (EPI == EPI_DGRAD) {
                if (ep.accumulate) {                          // out += result (two gradients meet at a feature map)
                    const uint4 old = *reinterpret_cast<const uint4*>(ep.out + o);
                    auto add2 = [](unsigned a, unsigned b) {
                        const float lo = __uint_as_float(a << 16) + __uint_as_float(b << 16);
                        const float hi = __uint_as_float(a & 0xffff0000u) + __uint_as_float(b & 0xffff0000u);
                        return pack_bf16x2(lo, hi);
                    };
                    v.x = add2(v.x, old.x); v.y = add2(v.y, old.y); v.z = add2(v.z, old.z); v.w = add2(v.w, old.w);
                }
                if (ep.mask_src) {                            // ReLU backward: zero where the forward activation was <= 0
                    const uint4 mk = *reinterpret_cast<const uint4*>(ep.mask_src + o);
                    auto gate = [](unsigned val, unsigned m2) {
                        if (!(__uint_as_float(m2 << 16) > 0.f)) val &= 0xffff0000u;
                        if (!(__uint_as_float(m2 & 0xffff0000u) > 0.f)) val &= 0x0000ffffu;
                        return val;
                    };
                    v.x = gate(v.x, mk.x); v.y = gate(v.y, mk.y); v.z = gate(v.z, mk.z); v.w = gate(v.w, mk.w);
                } else if (ep.mask_bits) {
                    v = gate_bits8(v, ep.mask_bits[(long long)m * (g.N >> 3) + (n >> 3)]);
                }
            }
            if constexpr (EPI == EPI_FWD) {
                if (ep.relu_bits) ep.relu_bits[(long long)m * (g.N >> 3) + (n >> 3)] = (unsigned char)relu_bits8(v);
            }
            if constexpr (EPI == EPI_DGRAD) {
                if (ep.up_code) {                             // un-pool: the four positions of the window, winner or zero
                    const int b = fdiv(m, g.d_hw);
                    const int rem = m - b * g.d_hw.d;
                    const int oy = fdiv(rem, g.d_w), ox = rem - oy * g.d_w.d;
                    const unsigned cw = ep.up_code[(long long)m * (g.N >> 3) + (n >> 3)];
                    const unsigned gw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int iy = 2 * oy + (q >> 1), ix = 2 * ox + (q & 1);
                        if (iy >= ep.up_h || ix >= ep.up_w) continue;
                        unsigned w4[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned lo = ((cw >> (8 * k)) & 15u) == (unsigned)q ? 0x0000ffffu : 0u;
                            const unsigned hi = ((cw >> (8 * k + 4)) & 15u) == (unsigned)q ? 0xffff0000u : 0u;
                            w4[k] = gw[k] & (lo | hi);
                        }
                        *reinterpret_cast<uint4*>(ep.up_out + (((long long)b * ep.up_h + iy) * ep.up_w + ix) * g.N + n) =
                            make_uint4(w4[0], w4[1], w4[2], w4[3]);
                    }
                    if (!ep.out) continue;
                }
            }
            *reinterpret_cast<uint4*>(ep.out + o) = v;
        }
        if constexpr (EPI == EPI_FWD && !std::is_same<PoolMap, NoPool>::value) {
            // fused 2x2 pooling of a 16-wide block tile (rows = 16 y + x, BM / 16 rows): pooled pixel (py, px) <- tile rows
            // of (2py+dy, 2px+dx)
            if (ep.pool_out) {
                for (int idx = tid; idx < (BM / 4) * CPR; idx += NT) {
                    const int pp = idx / CPR, ch = idx - pp * CPR;
                    const int py = pp >> 3, px = pp & 7;
                    const int n = n0 + ch * 8;
                    const long long po = pool_index(py, px);
                    if (po < 0 || n >= g.N) continue;
                    if (ep.relu) {
                        // after ReLU every candidate is >= +0 (v_max_f32 returns +0 for max(-0, +0)), so bf16 bit patterns
                        // order like the values: packed 16-bit integer max / compare, two channels per instruction
                        // (the float form below cost 25 % of block1_conv2's forward time)
                        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                        uint4 cand[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int row = (2 * py + (q >> 1)) * 16 + 2 * px + (q & 1);
                            cand[q] = make_uint4(0, 0, 0, 0);      // a position outside the map never wins: 0 only ties with 0 = dead
                            if (row_to_m(row) >= 0) cand[q] = lds_ld16_scoped(smem + row * (BN * 2) + (((ch ^ row) & (CPR - 1)) << 4), smem);
                        }
                        // (the packed 16-bit instructions are written out: from vector-typed C++ the compiler turned every
                        //  min(x, 1) into a compare + select per half -- 218 vector instructions per thread for this stage,
                        //  a quarter of the MFMA time of a 64-channel block; ~18 per word now)
                        unsigned o4[4], cw = 0;
                        const unsigned one = 0x00010001u, four = 0x00040004u;
                        auto pk_max = [](unsigned x, unsigned y) { unsigned r; asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
                        auto pk_min = [](unsigned x, unsigned y) { unsigned r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
                        auto pk_mad = [](unsigned x, unsigned y, unsigned z) { unsigned r; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z)); return r; };
                        auto pk_sub = [](unsigned x, unsigned y) { unsigned r; asm("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned w0 = (&cand[0].x)[k], w1 = (&cand[1].x)[k], w2 = (&cand[2].x)[k], w3 = (&cand[3].x)[k];
                            const unsigned m = pk_max(pk_max(w0, w1), pk_max(w2, w3));
                            const unsigned ne0 = pk_min(w0 ^ m, one), ne1 = pk_min(w1 ^ m, one), ne2 = pk_min(w2 ^ m, one);
                            const unsigned t = pk_mad(ne1, ne2, ne1);            // ne1 (1 + ne2)
                            const unsigned first = pk_mad(ne0, t, ne0);          // index of the first candidate equal to the max
                            const unsigned alive = pk_min(m, one);
                            const unsigned cu = pk_mad(alive, pk_sub(first, four), four);   // 4 = no winner (max is 0)
                            o4[k] = m;
                            cw |= ((cu & 0xfu) | ((cu >> 12) & 0xf0u)) << (8 * k);
                        }
                        *reinterpret_cast<uint4*>(ep.pool_out + po * g.N + n) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
                        ep.pool_code[po * (g.N >> 3) + (n >> 3)] = cw;
                        continue;
                    }
                    float best[8];
                    unsigned pos[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; pos[k] = 4u; }
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            const int row = (2 * py + dy) * 16 + 2 * px + dx;
                            if (row_to_m(row) < 0) continue;
                            const uint4 v = lds_ld16_scoped(smem + row * (BN * 2) + (((ch ^ row) & (CPR - 1)) << 4), smem);
                            const unsigned wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const float f = (k & 1) ? __uint_as_float(wds[k >> 1] & 0xffff0000u) : __uint_as_float(wds[k >> 1] << 16);
                                if (f > best[k]) { best[k] = f; pos[k] = (unsigned)(2 * dy + dx); }
                            }
                        }
                    unsigned o4[4], cw = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o4[k] = (__float_as_uint(best[2 * k]) >> 16) | (__float_as_uint(best[2 * k + 1]) & 0xffff0000u);
#pragma unroll
                    for (int k = 0; k < 8; ++k) cw |= (best[k] > 0.f ? pos[k] : 4u) << (4 * k);
                    *reinterpret_cast<uint4*>(ep.pool_out + po * g.N + n) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
                    ep.pool_code[po * (g.N >> 3) + (n >> 3)] = cw;
                }
            }
        }
    }
}

// DG_BATCHED = false: the data gradient's chunk-by-chunk store loop (fewer registers: for a kernel at its register limit whose
// data gradient normally leaves through its own path, k_conv3x3_c64b)
template <int EPI, int BM, int BN, int CT, int PT, int NT, typename RowMap, typename PoolMap, bool DG_BATCHED = true>
__device__ __forceinline__ void staged_epilogue(f32x4_t (&acc)[CT][PT], char* smem, const ConvGeom& g, const Epilogue& ep,
                                                int n0, int wrow0, int wcol0, int tid, RowMap row_to_m, PoolMap pool_index) {
    constexpr int CPR = BN / 8;                 // 16-byte chunks per tile row
    const int lane = tid & 63;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int col = wcol0 + c * 16 + (lane >> 4) * 4;
        const int n = n0 + col;
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI != EPI_DGRAD) { if (n < g.N) load_bias4(ep, n, g.N, n + 3 < g.N, b4); }
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int row = wrow0 + p * 16 + (lane & 15);
            float v[4] = {acc[c][p][0] + b4[0], acc[c][p][1] + b4[1], acc[c][p][2] + b4[2], acc[c][p][3] + b4[3]};
            if constexpr (EPI == EPI_FWD) {
                if (ep.relu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                }
            }
            *reinterpret_cast<uint2*>(smem + row * (BN * 2) + ((((col >> 3) ^ row) & (CPR - 1)) << 4) + (col & 4) * 2) =
                make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
    }
    __syncthreads();
    staged_store<EPI, BM, BN, NT, RowMap, PoolMap, DG_BATCHED>(smem, g, ep, n0, tid, row_to_m, pool_index);
}

// callers without a pooling stage
template <int EPI, int BM, int BN, int CT, int PT, int NT, typename RowMap>
__device__ __forceinline__ void staged_epilogue(f32x4_t (&acc)[CT][PT], char* smem, const ConvGeom& g, const Epilogue& ep,
                                                int n0, int wrow0, int wcol0, int tid, RowMap row_to_m) {
    staged_epilogue<EPI, BM, BN, CT, PT, NT>(acc, smem, g, ep, n0, wrow0, wcol0, tid, row_to_m, NoPool{});
}

// Kernels that need more than 64 KB of dynamic LDS are registered once per device (idempotent; a racing second thread
// repeats the same call).  No other process-wide state exists in this library.
struct OnceLds { std::atomic<unsigned> done{0}; };
inline int ensure_lds(OnceLds& o, const void* fn, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    const unsigned bit = 1u << (dev & 31);
    if (o.done.load(std::memory_order_acquire) & bit) return 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -1;
    o.done.fetch_or(bit, std::memory_order_release);
    return 0;
}

inline ConvGeom make_geom(int B, int H, int W, int C, int Ho, int Wo, int N, int KH, int KW, int mul, int div, int pad_t,
                   int pad_l) {
    ConvGeom g;
    g.B = B; g.H = H; g.W = W; g.C = C; g.Ho = Ho; g.Wo = Wo; g.N = N; g.KH = KH; g.KW = KW;
    g.mul = mul; g.div = div; g.pad_t = pad_t; g.pad_l = pad_l;
    g.M = B * Ho * Wo;
    g.nchunks = KH * KW * C / 8;
    g.ldw = KH * KW * C;
    g.cpt = C / 8;
    g.d_hw = make_fastdiv(Ho * Wo);
    g.d_w = make_fastdiv(Wo);
    g.d_h1 = make_fastdiv(H + 1);
#ifdef SSD_DEV_ABLATE                        // timing-only ablations (they change results): development builds only
    g.ablate = ssd_knob("SSD_ABLATE", 0);
#else
    g.ablate = 0;
#endif
    g.dma32 = ((long long)B * H * W * C < (1ll << 31) - 16 && (long long)N * g.ldw < (1ll << 31) - 16) ? 1 : 0;
    g.s2 = 0;
    const int s2on = ssd_knob("SSD_DGRAD_S2", 1);
    if (div == 2 && s2on && g.cpt % 8 == 0) {
        g.s2 = 1;
        for (int p = 0; p < 2; ++p) { g.cls_h[p] = (Ho + 1 - p) / 2; g.cls_w[p] = (Wo + 1 - p) / 2; }
        for (int c = 0; c < 4; ++c) g.cls_n[c] = B * g.cls_h[c >> 1] * g.cls_w[c & 1];
    }
    return g;
}

}  // namespace
