// Convolution stack for gfx950 (MI355X): NHWC bf16 implicit-GEMM convolutions on MFMA.
//
// Replaces the TensorFlow kernels behind SSDObjectDetectionModel._build (models/ssd_model.py:74-171)
// and their autodiff (tape.gradient, :248): Conv2D 3x3/1x1 (+bias, +ReLU) forward, data gradient
// and weight gradient, and 2x2 max-pooling forward/backward.  TF "SAME" padding is asymmetric for
// stride 2 (pad_before = pad_total/2, the remainder after) and is passed explicitly as (pad_t, pad_l).
//
//   k_conv_igemm   forward AND data-gradient: out[m][n] = sum_k A[m][k] * Wm[n][k].
//                  A is gathered on the fly from an NHWC tensor: GEMM row m = output pixel (b,oy,ox),
//                  k = (tap, channel); the source pixel of a tap is (o*mul + k - pad) / div, taken only
//                  when divisible and in range (zero otherwise).  Forward: mul = stride, div = 1.
//                  Data gradient: the same kernel run on dY with the spatially flipped, transposed
//                  weights ([Cin][kh][kw][Cout], made by k_weight_transpose), mul = 1, div = stride,
//                  pad = k-1-pad: one code path for stride 1 and 2.
//                  Tile 128 pixels x BN channels x 64 (k); 4 waves (2x2), each 64 x BN/2 as 16x16x32
//                  bf16 MFMAs with weights as the A operand, so a lane ends up with 4 consecutive output
//                  channels of one pixel (8-byte NHWC stores).  Global->register->LDS staging, one tile
//                  ahead (loads issued before the MFMAs of the current tile, LDS written after them, one
//                  barrier per k-step), XOR-swizzled 16-byte slots so ds_read_b128 is conflict-free.
//   k_conv_wgrad   dW[co][(tap,ci)] = sum_pixels dY[pix][co] * X[src(pix,tap)][ci]: both operands are
//                  "k-major" in memory, so tiles are staged as [pixel][channel] and the fragments are
//                  fetched with the transposing LDS read ds_read_b64_tr_b16.  Split over pixel ranges
//                  (grid.z); fp32 partial slabs are summed in fixed order (deterministic) by
//                  k_wgrad_reduce.  The bias gradient rides along as an extra MFMA against a B fragment
//                  of ones.
#include <atomic>
#include <climits>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include "common.h"
#include <hip/hip_bf16.h>
#include <stdlib.h>

namespace {

}  // namespace
#include "conv_common.h"
namespace {

// split-K finalize: out = epilogue(sum over splits of slab[split][m][n..n+3] + bias)
template <int EPI>
__global__ void k_igemm_finalize(ConvGeom g, Epilogue ep) {
    const int n4 = (g.N + 3) >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)g.M * n4) return;
    const int m = (int)(idx / n4), n = (int)(idx - (long long)m * n4) * 4;
    const bool full = n + 3 < g.N;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (full && !(g.N & 3)) {                     // 16-byte aligned slab rows: one load per split (same order of additions)
        for (int z = 0; z < ep.ksplit; ++z) {
            const float4 t = *reinterpret_cast<const float4*>(ep.slab + ((long long)z * g.M + m) * g.N + n);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        }
    } else {
        for (int z = 0; z < ep.ksplit; ++z) {
            const float* sl = ep.slab + ((long long)z * g.M + m) * g.N + n;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (n + j < g.N) v[j] += sl[j];
        }
    }
    if constexpr (EPI != EPI_DGRAD) {
        float b4[4];
        load_bias4(ep, n, g.N, false, b4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += b4[j];
    }
    epi_store<EPI>(v, m, n, full, g, ep);
}

// ------------------------------------------------------------------------------------------------
// Implicit GEMM, LDS-DMA version: tiles go global -> LDS directly (global_load_lds_dwordx4: per-lane
// gather address, wave-uniform 1 KiB LDS destination, no VGPR staging, no ds_write).  The swizzled LDS
// image is produced by choosing which chunk each lane fetches: wave-instruction i fills bank rows
// 4i..4i+3 = tile rows 8i..8i+7; lane L owns slot L of that KiB.  Padding / out-of-range chunks are
// fetched from a 16-byte zero block.  Schedule per k-step (two LDS buffers, ONE barrier):
//   wait vmcnt(0) + barrier  -> tile ks has landed everywhere and everybody is done reading tile ks-1
//   issue the DMA of tile ks+1 into the other buffer (in flight during this step's MFMAs)
//   read fragments of tile ks, 64 MFMAs.
__device__ __attribute__((aligned(16))) const unsigned g_zero16[4] = {0u, 0u, 0u, 0u};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// (a device function, not a call inside the kernel's lambda: with the builtin written there clang's HOST pass silently emitted
//  no stub for any instantiation of k_conv_igemm_dma -- an undefined symbol at load time, no diagnostic)
__device__ __forceinline__ void dma16_buf(__amdgpu_buffer_rsrc_t r, char* lds, unsigned off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds, 16, off, 0, 0, 0);
}

template <int BM, int BN, int EPI, int PT>
__global__ __launch_bounds__((BM / (16 * PT)) * (BN >= 128 ? BN / 64 : 2) * 64) void k_conv_igemm_dma(
    const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ w, ConvGeom g, Epilogue ep) {
    constexpr int WAVES_M = BM / (16 * PT);     // wave tile: 16*PT pixels x 16*CT channels
    constexpr int WAVES_N = BN >= 128 ? BN / 64 : 2;
    constexpr int NW = WAVES_M * WAVES_N;       // waves per workgroup (4, 8 or 16)
    constexpr int CT = BN / (16 * WAVES_N);     // 16-wide channel tiles per wave (4, or 2 for BN = 64)
    constexpr int XI = BM / 8 / NW;             // activation DMA instructions per wave per k-step
    constexpr int WI = BN / 8 / NW;             // weight DMA instructions per wave per k-step
    static_assert(XI >= 1 && WI >= 1, "tile too small for the wave count");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto s_x = [&](int buf) { return smem + buf * ((BM + BN) * 128); };
    auto s_w = [&](int buf) { return smem + buf * ((BM + BN) * 128) + BM * 128; };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave % WAVES_M, wave_n = wave / WAVES_M;
    // XCD-aware tile order: workgroup L runs on XCD L % 8 (round-robin dispatch); the N-tiles of one pixel tile are
    // consecutive on ONE XCD so that the gathered activation tile is fetched into that XCD's L2 once.
    const int ntn = (g.N + BN - 1) / BN;
    int ntm = (g.M + BM - 1) / BM;
    if (g.s2) ntm = (g.cls_n[0] + BM - 1) / BM + (g.cls_n[1] + BM - 1) / BM + (g.cls_n[2] + BM - 1) / BM + (g.cls_n[3] + BM - 1) / BM;
    const int kx = blockIdx.x >> 3;
    const int mt = (kx / ntn) * 8 + (blockIdx.x & 7);
    if (mt >= ntm) return;
    int m0 = mt * BM;
    const int n0 = (kx % ntn) * BN;
    // parity class of this tile (stride-2 data gradient only)
    int cls = 0, cls_count = g.M, cls_py = 0, cls_px = 0, cls_hw = 1, cls_wd = 1;
    if (g.s2) {
        int t0 = 0;
        for (cls = 0; cls < 3; ++cls) {
            const int ntc = (g.cls_n[cls] + BM - 1) / BM;
            if (mt < t0 + ntc) break;
            t0 += ntc;
        }
        m0 = (mt - t0) * BM;
        cls_count = g.cls_n[cls];
        cls_py = cls >> 1; cls_px = cls & 1;
        cls_wd = g.cls_w[cls_px];
        cls_hw = g.cls_h[cls_py] * cls_wd;
    }
    // destination pixel (b, oy, ox) of tile-local row index m (class-local when s2)
    auto decode = [&](int m, int& bb, int& oy, int& ox) {
        if (g.s2) {
            bb = m / cls_hw;
            const int rem = m - bb * cls_hw;
            const int a = rem / cls_wd;
            oy = 2 * a + cls_py;
            ox = 2 * (rem - a * cls_wd) + cls_px;
        } else {
            bb = fdiv(m, g.d_hw);
            const int rem = m - bb * g.d_hw.d;
            oy = fdiv(rem, g.d_w);
            ox = rem - oy * g.d_w.d;
        }
    };

    // DMA ownership: wave-instruction i = (wave&1) + 2*((wave>>1) + (NW/2)*j) covers tile rows 8i..8i+7 (i has the
    // wave's parity, so a lane fetches the same k-chunk for all of its rows);
    // lane L -> row 8i + 2*(L>>4) + ((L>>3)&1), chunk (L&7) ^ ((row>>1)&7) = (L&7) ^ (4*(wave&1) + (L>>4))
    const int rl = 2 * (lane >> 4) + ((lane >> 3) & 1);
    const int slot = (lane & 7) ^ (4 * (wave & 1) + (lane >> 4));
    // per staged row: (ybase, xbase) = source coordinate of tap (0,0) times div, rowpix = flat source pixel of it
    // (div == 1).  An invalid row gets ybase far outside so that every tap fails the range test.
    int ybase[XI], xbase[XI], ibase[XI];
#pragma unroll
    for (int j = 0; j < XI; ++j) {
        const int i = (wave & 1) + 2 * ((wave >> 1) + (NW / 2) * j);
        const int m = m0 + 8 * i + rl;
        const bool mv = m < cls_count;
        int b, oy, ox;
        decode(mv ? m : 0, b, oy, ox);
        ybase[j] = mv ? oy * g.mul - g.pad_t : -(1 << 20);
        xbase[j] = ox * g.mul - g.pad_l;
        ibase[j] = b * g.H * g.W;              // < 2^31 (checked on the host)
    }
    unsigned wrow[WI];                          // element offset of the weight row, ~0u if out of range (weights < 4G elements)
#pragma unroll
    for (int j = 0; j < WI; ++j) {
        const int i = (wave & 1) + 2 * ((wave >> 1) + (NW / 2) * j);
        const int n = n0 + 8 * i + rl;
        wrow[j] = n < g.N ? (unsigned)n * (unsigned)g.ldw : ~0u;
    }
    int tap = slot / g.cpt, cc = slot - tap * g.cpt;
    int kh = tap / g.KW, kw = tap - kh * g.KW;
    int q = slot;
    const int dmask = g.div - 1, dshift = g.div > 1 ? 1 : 0;   // div is 1 or 2

    // Tensors below 4 GB (every layer of the SSD networks): LDS-DMA through buffer descriptors -- 32-bit byte offsets, a lane
    // whose chunk is padding / out of range gets an offset beyond the buffer (the hardware writes zeros).  The 64-bit
    // per-lane address form below it (global_load_lds) moves the same bytes 4-5 % slower (measured on the weight-gradient patch
    // kernel, round 3) and remains for larger tensors.
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, g.dma32 ? (unsigned)g.B * g.H * g.W * g.C * 2u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, g.dma32 ? (unsigned)g.N * (unsigned)g.ldw * 2u : 0u, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    auto issue_dma = [&](int buf) {
        const bool kvalid = q < g.nchunks;
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            const int i = (wave & 1) + 2 * ((wave >> 1) + (NW / 2) * j);
            const int ny = ybase[j] + kh, nx = xbase[j] + kw;
            const int iy = ny >> dshift, ix = nx >> dshift;
            // unsigned compares fold the >= 0 tests; the element offset fits 32 bits (tensor < 4G elements, host check)
            const bool ok = kvalid && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W && (((ny | nx) & dmask) == 0);
            const unsigned off = (unsigned)(ibase[j] + iy * g.W + ix) * (unsigned)g.C + (unsigned)(cc * 8);
            if (g.dma32) {
                dma16_buf(xres, s_x(buf) + i * 1024, ok ? off * 2u : OOB);
            } else {
                const bf16_raw* src = ok ? x + off : reinterpret_cast<const bf16_raw*>(g_zero16);
                __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(s_x(buf) + i * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < WI; ++j) {
            const int i = (wave & 1) + 2 * ((wave >> 1) + (NW / 2) * j);
            const bool wok = kvalid && wrow[j] != ~0u;
            if (g.dma32) {
                dma16_buf(wres, s_w(buf) + i * 1024, wok ? (wrow[j] + (unsigned)(q * 8)) * 2u : OOB);
            } else {
                const bf16_raw* src = wok ? w + (wrow[j] + (unsigned)(q * 8)) : reinterpret_cast<const bf16_raw*>(g_zero16);
                __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(s_w(buf) + i * 1024), 16, 0, 0);
            }
        }
        q += 8;
        cc += 8;
        while (cc >= g.cpt) {
            cc -= g.cpt;
            if (++kw == g.KW) { kw = 0; ++kh; }
        }
    };

    f32x4_t acc[CT][PT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p) acc[c][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    int nks = (g.nchunks + 7) >> 3;
    // stride-2 data gradient: only the taps whose parity matches the tile's class contribute; steps are addressed
    // absolutely as (valid tap vt, sub-step) instead of incrementally
    int vtaps = 0, nvt = 0;                     // 4 bits per valid tap index
    const int spt = g.cpt >> 3;                 // k-steps per tap (s2 only)
    if (g.s2) {
        for (int t = 0; t < g.KH * g.KW; ++t) {
            const int th = t / g.KW, tw = t - th * g.KW;
            if ((((cls_py - g.pad_t + th) | (cls_px - g.pad_l + tw)) & 1) == 0) { vtaps |= t << (4 * nvt); ++nvt; }
        }
        nks = nvt * spt;
    }
    auto set_step = [&](int step) {             // s2: position the k state on valid step `step`
        const int vt = step / spt, sub = step - vt * spt;
        const int t = (vtaps >> (4 * vt)) & 15;
        kh = t / g.KW; kw = t - kh * g.KW;
        cc = sub * 8 + slot;
        q = t * g.cpt + cc;
    };
    // split-K: this workgroup covers k-steps [ks0, ks1)
    int ks0 = 0, ks1 = nks;
    if (ep.slab) {
        const int per = (nks + ep.ksplit - 1) / ep.ksplit;
        ks0 = min(nks, (int)blockIdx.y * per);
        ks1 = min(nks, ks0 + per);
        if (!g.s2) {                              // position the incremental k state on step ks0
            q = ks0 * 8 + slot;
            tap = q / g.cpt; cc = q - tap * g.cpt;
            kh = tap / g.KW; kw = tap - kh * g.KW;
        }
    }
    if (g.s2 && ks0 < ks1) set_step(ks0);
    if (ks0 < ks1) issue_dma(0);
    const int frow = lane & 15, fk = lane >> 4;
    for (int ks = ks0; ks < ks1; ++ks) {
        const int cur = (ks - ks0) & 1;
        if (!(g.ablate & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
#pragma unroll
        for (int ksub = 0; ksub < 2; ++ksub) {
            // the next tile's DMA is issued between the two MFMA blocks: its address arithmetic and issue slots then
            // overlap with MFMAs already in flight instead of sitting in front of them
            if (ksub == SSD_DMA_SLOT && ks + 1 < ks1 && !(g.ablate & 1)) {
                if (g.s2) set_step(ks + 1);
                issue_dma(cur ^ 1);
            }
            bf16x8_t fx[PT], fw[CT];
#pragma unroll
            for (int p = 0; p < PT; ++p)
                fx[p] = *reinterpret_cast<const bf16x8_t*>(s_x(cur) + swz(wave_m * (16 * PT) + p * 16 + frow, ksub * 4 + fk));
#pragma unroll
            for (int c = 0; c < CT; ++c)
                fw[c] = *reinterpret_cast<const bf16x8_t*>(s_w(cur) + swz(wave_n * (16 * CT) + c * 16 + frow, ksub * 4 + fk));
            if (g.ablate & 4) {
#pragma unroll
                for (int p = 0; p < PT; ++p) asm volatile("" ::"v"(fx[p]));
#pragma unroll
                for (int c = 0; c < CT; ++c) asm volatile("" ::"v"(fw[c]));
            } else {
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int p = 0; p < PT; ++p)
                        acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[c], fx[p], acc[c][p], 0, 0, 0);
            }
        }
    }
    if (staged_ok<EPI>(g, ep) && !(g.ablate & 8)) {
        auto row_to_m = [&](int row) {
            const int m = m0 + row;
            if (m >= cls_count) return -1;
            if (!g.s2) return m;
            int b, oy, ox;
            decode(m, b, oy, ox);
            return (b * g.Ho + oy) * g.Wo + ox;
        };
        staged_epilogue<EPI, BM, BN, CT, PT, NW * 64>(acc, smem, g, ep, n0, wave_m * (16 * PT), wave_n * (16 * CT), tid, row_to_m);
        return;
    }
    int mrow[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int m = m0 + wave_m * (16 * PT) + p * 16 + (lane & 15);
        mrow[p] = m < cls_count ? m : -1;
        if (g.s2 && mrow[p] >= 0) {
            int b, oy, ox;
            decode(m, b, oy, ox);
            mrow[p] = (b * g.Ho + oy) * g.Wo + ox;
        }
    }
    conv_epilogue_rows<EPI, CT, PT>(acc, g, ep, mrow, n0 + wave_n * (16 * CT), lane);
}

// ------------------------------------------------------------------------------------------------
// Implicit GEMM, 256x256x64 tile, 8 waves, eight-phase ping-pong schedule (the wide stride-1/2 layers without
// parity classes or split-K).  Waves form two groups of four (one wave of each group per SIMD); the groups run
// half a phase apart, so that on every SIMD one wave is in its MFMA cluster while its partner issues LDS reads and
// the LDS-DMA of a later tile.  A k-tile is staged as four half-tiles of 16 KB (W0, X0, W1, X1: the 32 channels /
// 64 pixels of quadrant-row h of every wave); a phase stages ONE half-tile (2 DMA instructions per wave) and
// multiplies ONE quadrant (64 px x 32 ch x k64 = 16 MFMAs per wave).  Phase p issues half-tile p + 7 of the stream
// (W0 X0 W1 X1 per tile), i.e. three half-tiles of tile t+2 are still in flight across the barriers when tile t+1
// is first read: the only wait on the DMA counter is a counted vmcnt(6) in the last phase of a tile.
//   hazards: a half-tile is read one phase after the wait that retires it (RAW); it is re-staged two phases after
//   its last LDS read, or one phase after for W0 whose reads are retired (lgkmcnt) before the reading phase's
//   first barrier (WAR).  Both hold for either group under the half-phase stagger.
template <int HX, int HW>
__device__ __forceinline__ void mma_quadrant(f32x4_t (&acc)[4][8], const bf16x8_t (&fx)[4][2], const bf16x8_t (&fw)[2][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int p = 0; p < 4; ++p)
                acc[HW * 2 + c][HX * 4 + p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[c][ks], fx[p][ks], acc[HW * 2 + c][HX * 4 + p], 0, 0, 0);
    // hipcc otherwise lets part of the cluster drift below the phase's closing barrier, in among the partner wave's
    // turn: tie the eight accumulators to this point (no instruction is emitted)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int p = 0; p < 4; ++p) asm volatile("" : "+v"(acc[HW * 2 + c][HX * 4 + p]));
}

__device__ __forceinline__ void keep_frags(const bf16x8_t (&fx)[4][2], const bf16x8_t (&fw)[2][2]) {   // dev ablation only
#pragma unroll
    for (int p = 0; p < 4; ++p) { asm volatile("" ::"v"(fx[p][0])); asm volatile("" ::"v"(fx[p][1])); }
#pragma unroll
    for (int c = 0; c < 2; ++c) { asm volatile("" ::"v"(fw[c][0])); asm volatile("" ::"v"(fw[c][1])); }
}

// ABL: compile-time development ablations (1 no DMA in the loop, 2 no barriers, 4 no MFMA, 16 no fragment reads); 0 in production
template <int EPI, int ABL>
__global__ __launch_bounds__(512) void k_conv_igemm_8ph(const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ w,
                                                        ConvGeom g, Epilogue ep) {
    constexpr int BM = 256, BN = 256;
    constexpr int HALF = 128 * 128;             // one half-tile image: 128 rows x 64 k bf16
    constexpr int OFF_W0 = 0, OFF_X0 = HALF, OFF_W1 = 2 * HALF, OFF_X1 = 3 * HALF, BUF = 4 * HALF;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;    // wave tile: pixels wr*128.., channels wc*64..; group = wr
    const int ntn = (g.N + BN - 1) / BN;
    const int ntm = (g.M + BM - 1) / BM;
    const int kx = blockIdx.x >> 3;
    const int mt = (kx / ntn) * 8 + (blockIdx.x & 7);          // XCD-aware order, as k_conv_igemm_dma
    if (mt >= ntm) return;
    const int m0 = mt * BM, n0 = (kx % ntn) * BN;

    // DMA ownership inside a half-tile (16 one-KiB pieces = 8 rows each): wave-instruction i = (wave&1) + 2*((wave>>1) + 4j)
    const int rl = 2 * (lane >> 4) + ((lane >> 3) & 1);
    const int slot = (lane & 7) ^ (4 * (wave & 1) + (lane >> 4));
    int ybase[2][2], xbase[2][2], ibase[2][2];
    unsigned wrow[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 8 * ((wave & 1) + 2 * ((wave >> 1) + 4 * j)) + rl;       // row of the half-tile image
            const int m = m0 + (r >> 6) * 128 + h * 64 + (r & 63);
            const bool mv = m < g.M;
            const int mm = mv ? m : 0;
            const int b = fdiv(mm, g.d_hw);
            const int rem = mm - b * g.d_hw.d;
            const int oy = fdiv(rem, g.d_w);
            const int ox = rem - oy * g.d_w.d;
            ybase[h][j] = mv ? oy * g.mul - g.pad_t : -(1 << 20);
            xbase[h][j] = ox * g.mul - g.pad_l;
            ibase[h][j] = b * g.H * g.W;
            const int n = n0 + (r >> 5) * 64 + h * 32 + (r & 31);
            wrow[h][j] = n < g.N ? (unsigned)n * (unsigned)g.ldw : ~0u;
        }
    // k state of the activation stream (tap and channel chunk of this lane's slot) and of the weight stream
    int tap = slot / g.cpt, cc = slot - tap * g.cpt;
    int kh = tap / g.KW, kw = tap - kh * g.KW;
    int xq = slot, wq = slot;
    const int dmask = g.div - 1, dshift = g.div > 1 ? 1 : 0;

    // LDS-DMA through buffer descriptors: 32-bit byte offsets, and a lane whose chunk is padding / out of range gets
    // an offset beyond the buffer, for which the hardware writes zeros to the lane's LDS slot (no zero block, no branch)
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)g.B * g.H * g.W * g.C * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (unsigned)g.N * (unsigned)g.ldw * 2u, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    auto issue_x = [&](int off, int h) {
        const bool kvalid = xq < g.nchunks;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = (wave & 1) + 2 * ((wave >> 1) + 4 * j);
            const int ny = ybase[h][j] + kh, nx = xbase[h][j] + kw;
            const int iy = ny >> dshift, ix = nx >> dshift;
            const bool ok = kvalid && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W && (((ny | nx) & dmask) == 0);
            const unsigned o = ((unsigned)(ibase[h][j] + iy * g.W + ix) * (unsigned)g.C + (unsigned)(cc * 8)) * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void*)(smem + off + i * 1024), 16, ok ? o : OOB, 0, 0, 0);
        }
    };
    auto advance_x = [&]() {                    // cpt >= 8 (host check): at most one tap boundary per k-tile
        xq += 8;
        cc += 8;
        const bool wrap = cc >= g.cpt;
        cc -= wrap ? g.cpt : 0;
        kw += wrap ? 1 : 0;
        const bool roll = kw == g.KW;
        kw = roll ? 0 : kw;
        kh += roll ? 1 : 0;
    };
    auto issue_w = [&](int off, int h) {
        const bool kvalid = wq < g.nchunks;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = (wave & 1) + 2 * ((wave >> 1) + 4 * j);
            const unsigned o = (wrow[h][j] + (unsigned)(wq * 8)) * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (lds_void*)(smem + off + i * 1024), 16,
                                                     (kvalid && wrow[h][j] != ~0u) ? o : OOB, 0, 0, 0);
        }
    };

    f32x4_t acc[4][8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < 8; ++p) acc[c][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nt = (g.nchunks + 7) >> 3;
    // prologue: tile 0 completely, W0 X0 W1 of tile 1.  Stages beyond the last tile are issued all the same: every
    // lane is then out of range, zeros land in a half-tile nobody reads any more, and the DMA count per phase stays 2.
    issue_w(OFF_W0, 0); issue_x(OFF_X0, 0); issue_w(OFF_W1, 1); issue_x(OFF_X1, 1);
    wq += 8;
    advance_x();
    issue_w(BUF + OFF_W0, 0); issue_x(BUF + OFF_X0, 0); issue_w(BUF + OFF_W1, 1);
    wq += 8;
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // group 1 runs half a phase behind group 0

    const int frow = lane & 15, fk = lane >> 4;
    const int xf0 = swz(wr * 64 + frow, fk), xf1 = swz(wr * 64 + frow, 4 + fk);     // + p * 2048
    const int wf0 = swz(wc * 32 + frow, fk), wf1 = swz(wc * 32 + frow, 4 + fk);     // + c * 2048
    auto ldf = [&](int off) {
        if constexpr (ABL & 16) return bf16x8_t{};
        else return *reinterpret_cast<const bf16x8_t*>(smem + off);
    };
    bf16x8_t fx[4][2] = {}, fw0[2][2] = {}, fw1[2][2] = {};

#define SSD_PHASE_MMA(HX_, HW_, FW_)                                                                                \
    if constexpr (!(ABL & 2)) __builtin_amdgcn_s_barrier();                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    __builtin_amdgcn_s_setprio(1);                                                                                  \
    if constexpr (!(ABL & 4)) mma_quadrant<HX_, HW_>(acc, fx, FW_);                                                      \
    else keep_frags(fx, FW_);                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    __builtin_amdgcn_s_setprio(0);                                                                                  \
    if constexpr (!(ABL & 2)) __builtin_amdgcn_s_barrier();                                                              \
    __builtin_amdgcn_sched_barrier(0);

    for (int t = 0; t < nt; ++t) {
        const int cb = (t & 1) * BUF, nb = BUF - cb;
        // ---- phase 0: quadrant (X0, W0); stage X1 of tile t+1
#pragma unroll
        for (int c = 0; c < 2; ++c) { fw0[c][0] = ldf(cb + OFF_W0 + wf0 + c * 2048); fw0[c][1] = ldf(cb + OFF_W0 + wf1 + c * 2048); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 4; ++p) { fx[p][0] = ldf(cb + OFF_X0 + xf0 + p * 2048); fx[p][1] = ldf(cb + OFF_X0 + xf1 + p * 2048); }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");      // the W0 reads are done: W0 may be re-staged next phase
        if constexpr (!(ABL & 1)) issue_x(nb + OFF_X1, 1);
        advance_x();
        SSD_PHASE_MMA(0, 0, fw0)
        // ---- phase 1: quadrant (X0, W1); stage W0 of tile t+2
#pragma unroll
        for (int c = 0; c < 2; ++c) { fw1[c][0] = ldf(cb + OFF_W1 + wf0 + c * 2048); fw1[c][1] = ldf(cb + OFF_W1 + wf1 + c * 2048); }
        if constexpr (!(ABL & 1)) issue_w(cb + OFF_W0, 0);
        SSD_PHASE_MMA(0, 1, fw1)
        // ---- phase 2: quadrant (X1, W1); stage X0 of tile t+2
#pragma unroll
        for (int p = 0; p < 4; ++p) { fx[p][0] = ldf(cb + OFF_X1 + xf0 + p * 2048); fx[p][1] = ldf(cb + OFF_X1 + xf1 + p * 2048); }
        if constexpr (!(ABL & 1)) issue_x(cb + OFF_X0, 0);
        SSD_PHASE_MMA(1, 1, fw1)
        // ---- phase 3: quadrant (X1, W0); stage W1 of tile t+2; all of tile t+1 has landed once only the last three
        // half-tiles (6 DMA instructions of this wave) are still in flight
        if constexpr (!(ABL & 1)) issue_w(cb + OFF_W1, 1);
        wq += 8;
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        SSD_PHASE_MMA(1, 0, fw0)
    }
#undef SSD_PHASE_MMA
    if (wr == 0) __builtin_amdgcn_s_barrier();          // balance the stagger

    // Epilogue through LDS (free now): with one workgroup per CU nothing would hide a scattered store tail
    if (staged_ok<EPI>(g, ep) && !(g.ablate & 8)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the (all-zero) stages issued past the last tile
        auto row_to_m = [&](int row) { const int m = m0 + row; return m < g.M ? m : -1; };
        staged_epilogue<EPI, 256, 256, 4, 8, 512>(acc, smem, g, ep, n0, wr * 128, wc * 64, tid, row_to_m);
        return;
    }
    int mrow[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int m = m0 + wr * 128 + p * 16 + (lane & 15);
        mrow[p] = m < g.M ? m : -1;
    }
    conv_epilogue_rows<EPI, 4, 8>(acc, g, ep, mrow, n0 + wc * 64, lane);
}

// ------------------------------------------------------------------------------------------------
// Halo patch of a 16x16 block of output pixels (3x3 / stride 1 / pad 1 kernels below)
constexpr int PATCH_W = 18;
constexpr int PATCH_PIX = PATCH_W * PATCH_W;               // 324

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with an LDS-resident input patch (forward, and data gradient with the transposed
// weights): the halo patch of a block is brought into LDS once per channel chunk and all nine taps read their MFMA
// fragments from it at shifted addresses; only the weight slice of a tap streams.  32-channel chunks, linear padded images, two workgroups per CU.
//   * patch image: one 96-byte row per halo pixel (64 B of channels + 32 B pad).  The pitch makes every shifted
//     ds_read_b128 conflict-free WITHOUT an address swizzle, so a fragment address is a per-lane base + immediate
//     (the swizzled 128-byte rows of the first form cost ~10 VALU per read and were 2-way conflicted for 3 of 4 shifts);
//   * weight slice of a tap: [BN][32 k] in 64-byte rows, chunk XORed with (-(row >> 2)) & 3 (conflict-free, fixed rows);
//   * all DMA through buffer descriptors: per-lane byte offsets are computed once, the (tap, chunk) part is a scalar
//     offset, invalid / padding lanes are out of range (zeros): no vector arithmetic per DMA in the loop;
//   * 80 KB of LDS (two patch buffers, two weight buffers) and <= 128 VGPRs: two workgroups per CU cover each other's
//     prologue, barriers and store tail; the next chunk's patch is prefetched at the first tap of the current one and
//     left in flight across the barrier (counted vmcnt).
// LDS fragment read that carries an alias scope (see lds_read_tr16_scoped): keeps the compiler's waitcnt pass from ordering
// it behind LDS-DMA requests that are in flight for OTHER buffers.
__device__ __forceinline__ bf16x8_t lds_read_b128_scoped(const char* __restrict__ p, const char* __restrict__ other) {
    (void)other;
    return *reinterpret_cast<const bf16x8_t*>(p);
}
constexpr int P32_PITCH = 96;
constexpr int P32_PATCH = 32 * 1024;                       // 324 px x 96 B = 31104 B, rounded to 32 DMA instructions

template <int BN, int EPI, bool FLAT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_conv3x3_patch32(
    const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ w, ConvGeom g, Epilogue ep, int tiles_x, int tiles_y, int nblocks, int rowflat) {
    constexpr int CT = BN / 32;
    constexpr int PT = 4;
    constexpr int WBYTES = BN * 64;                          // weight slice of one tap
    constexpr int OFF_W = 2 * P32_PATCH;
    constexpr int OFF_DUMMY = OFF_W + 2 * WBYTES;            // BN = 64: waves 4-7 have no weight rows, their DMA lands here
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave & 3, wave_n = wave >> 2;
    // two block shapes.  flat == 0: a 16x16 block of one image, halo patch 18x18 (tap pitch Q = 18).
    // flat != 0 (narrow maps): the batch is one long strip of positions, image rows padded to Q = W + 2 (a zero column
    // on either side) and images separated by ONE zero row; a block is 256 consecutive positions, its patch those plus
    // Q + 1 on either side, tap (kh,kw) is position + kh*Q + kw.  Pad positions are computed and dropped: 7 % of the
    // work at 38x38 and 14 % at 19x19, where 16x16 blocks would waste 37 % and 65 %.
    // XCD-aware order (workgroup L runs on XCD L % 8): the channel tiles of one pixel block are consecutive on ONE
    // XCD, so the block's patch is fetched into that L2 once
    const int ntn = (g.N + BN - 1) / BN;
    const int kx = blockIdx.x >> 3;
    const int pblock = (kx / ntn) * 8 + (blockIdx.x & 7);
    if (pblock >= nblocks) return;
    const int n0 = (kx % ntn) * BN;
    constexpr bool flat = FLAT;
    const int Q = FLAT ? g.W + 2 : PATCH_W;
    const int img = (g.H + 1) * Q;                          // flat positions per image
    int b = 0, y0 = 0, x0 = 0, f0 = 0;
    if constexpr (FLAT) {
        f0 = pblock * 256;
    } else {
        int t = pblock;
        const int tx = t % tiles_x; t /= tiles_x;
        x0 = tx * 16;
        if (rowflat) { y0 = t * 16; }                       // row of the strip of all images (below), not of one image
        else { const int ty = t % tiles_y; b = t / tiles_y; y0 = ty * 16; }
    }
    // rowflat (wide maps whose height is not a multiple of 16): the rows of all images form one strip, images separated
    // by ONE zero row, and a block is 16 consecutive strip rows x 16 columns -- it may straddle two images, the shared
    // zero row is the bottom padding of one and the top padding of the other.  75 rows per image cost 76 instead of 80.
    // block row r (-1 .. 16 for the halo) -> row index into [B * H], -1 = padding / outside
    auto image_row = [&](int r) {
        if (!rowflat) { const int y = y0 + r; return (unsigned)y < (unsigned)g.H ? b * g.H + y : -1; }
        const int R = y0 + r;
        if (R < 0) return -1;
        const int bb = fdiv(R, g.d_h1), yy = R - bb * (g.H + 1);
        return (bb < g.B && yy < g.H) ? bb * g.H + yy : -1;
    };
    // flat position -> source pixel (element offset / C) or -1
    auto flat_pixel = [&](int f) {
        if (f < 0) return -1;
        const int bb = f / img;
        const int r = f - bb * img;
        const int yy = r / Q, xx = r - yy * Q - 1;
        return (bb < g.B && yy < g.H && (unsigned)xx < (unsigned)g.W) ? (bb * g.H + yy) * g.W + xx : -1;
    };

    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)g.B * g.H * g.W * g.C * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (unsigned)g.N * (unsigned)g.ldw * 2u, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    // patch DMA: instruction i = wave + 8j (j < 4) fills slots 64i .. 64i+63; slot q -> pixel q / 6, 16-byte piece q % 6
    unsigned pvo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = (wave + 8 * j) * 64 + lane;
        const int pp = q / 6, sl = q - pp * 6;
        int pix;
        if (flat) {
            pix = pp < 256 + 2 * (Q + 1) ? flat_pixel(f0 - (Q + 1) + pp) : -1;
        } else {
            const int py = pp / PATCH_W, px = pp - py * PATCH_W;
            const int ix = x0 - 1 + px;
            const int ir = pp < PATCH_PIX ? image_row(py - 1) : -1;
            pix = (ir >= 0 && (unsigned)ix < (unsigned)g.W) ? ir * g.W + ix : -1;
        }
        pvo[j] = (sl < 4 && pix >= 0) ? ((unsigned)pix * (unsigned)g.C + (unsigned)(sl * 8)) * 2u : OOB;
    }
    // weight DMA: instruction i = wave (< BN/16) fills rows 16i .. 16i+15; lane L -> row 16i + (L>>2), physical piece L & 3
    // BN = 64: the slices of TWO taps are staged per step (image rows 0..63 = the first tap, 64..127 = the second: the same
    // 128-row image and the same 80 KB of LDS as BN = 128), see the loop below
    unsigned wvo;
    {
        const int row = 16 * wave + (lane >> 2);
        const int piece = (lane & 3) ^ ((-(row >> 2)) & 3);
        const int n = n0 + (BN == 64 ? (row & 63) : row);
        wvo = (row < (BN == 64 ? 128 : BN) && n < g.N) ? ((unsigned)n * (unsigned)g.ldw + (unsigned)(piece * 8)) * 2u : OOB;
    }
    const int wdst = wave < BN / 16 ? wave * 1024 : -1;     // -1: dummy
    const int nchunk = g.C >> 5;

    // The kernel runs at the 128-register limit of four waves per SIMD, and the compiler kept three of the four patch offsets
    // in scratch, re-loading each one right in front of its DMA: "wait for the reload" is s_waitcnt vmcnt(0), which also waits
    // for the patch DMA issued just before to LAND -- the four requests of a chunk went out one memory round trip apart.
    // Here the three offsets are parked in scratch on purpose and fetched TOGETHER ahead of the first request of a chunk
    // (they are only needed once per chunk, when the fragment registers are free): one round trip, four requests back to back.
    unsigned pstash[3];
    pstash[0] = pvo[1]; pstash[1] = pvo[2]; pstash[2] = pvo[3];
    asm volatile("" ::"v"(&pstash[0]) : "memory");          // the address escapes: the array stays in (scratch) memory
    const unsigned pvo0 = pvo[0];
    auto dma_patch = [&](int chunk, int buf) {
        const unsigned o1 = pstash[0], o2 = pstash[1], o3 = pstash[2];
        char* dst = smem + buf * P32_PATCH + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void*)dst, 16, pvo0, chunk * 64, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void*)(dst + 8 * 1024), 16, o1, chunk * 64, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void*)(dst + 16 * 1024), 16, o2, chunk * 64, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void*)(dst + 24 * 1024), 16, o3, chunk * 64, 0, 0);
    };
    auto dma_w = [&](int chunk, int tap, int buf) {
        char* dst = wdst >= 0 ? smem + OFF_W + buf * WBYTES + wdst : smem + OFF_DUMMY;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (lds_void*)dst, 16, wvo, (tap * g.C + chunk * 32) * 2, 0, 0);
    };

    f32x4_t acc[CT][PT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p) acc[c][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fk = lane >> 4;
    const int xbase = (flat ? 4 * wave_m * 16 + frow : 4 * wave_m * PATCH_W + frow) * P32_PITCH + fk * 16;
    constexpr int prow = (FLAT ? 16 : PATCH_W) * P32_PITCH;  // bytes between the pixel tiles p, p+1 of a wave
    int wbase;
    {
        const int row = wave_n * (16 * CT) + frow;
        wbase = OFF_W + row * 64 + ((fk ^ ((-(row >> 2)) & 3)) << 4);
    }
    auto ldf = [&](int addr) { return *reinterpret_cast<const bf16x8_t*>(smem + addr); };

    if constexpr (BN == 64) {
        // 64 output channels: a wave has 8 MFMAs per tap, and with one barrier per tap the loop ran at 0.24 of the MFMA peak
        // (block2_conv1's data gradient).  Two taps per step: 16 MFMAs between barriers, five steps per chunk instead of
        // nine.  Waves 0-3 request the first tap's slice of a step, waves 4-7 the second's (out of range for the missing
        // tenth tap: zeros, never read), so every wave issues exactly one weight request per step and the counted waits of
        // the single-tap form carry over.
        constexpr int WB2 = 2 * WBYTES;
        auto dma_w2 = [&](int chunk, int step, int buf) {
            const int tap = 2 * step + (wave >> 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (lds_void*)(smem + OFF_W + buf * WB2 + wave * 1024), 16,
                                                     tap <= 8 ? wvo : OOB, ((tap <= 8 ? tap : 8) * g.C + chunk * 32) * 2, 0, 0);
        };
        dma_patch(0, 0);
        dma_w2(0, 0, 0);
        for (int chunk = 0; chunk < nchunk; ++chunk) {
            const int pb = (chunk & 1) * P32_PATCH;
            const bool next_chunk = chunk + 1 < nchunk;
#pragma unroll
            for (int st = 0; st < 5; ++st) {
                const int sidx = chunk * 5 + st;
                const int wb = (sidx & 1) * WB2;
                if (st == 1 && next_chunk) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (!(g.ablate & 16)) {
                    if (st < 4) dma_w2(chunk, st + 1, (sidx + 1) & 1);
                    else if (next_chunk) dma_w2(chunk + 1, 0, (sidx + 1) & 1);
                }
                if (st == 0 && next_chunk && !(g.ablate & 32)) dma_patch(chunk + 1, (chunk + 1) & 1);
                bf16x8_t fx[2][PT], fw[2][CT];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int tap = 2 * st + t;
                    if (tap > 8) continue;
                    const int tapoff = pb + xbase + ((tap / 3) * Q + tap % 3) * P32_PITCH;
#pragma unroll
                    for (int p = 0; p < PT; ++p) fx[t][p] = ldf(tapoff + p * prow);
#pragma unroll
                    for (int c = 0; c < CT; ++c) fw[t][c] = ldf(wb + wbase + t * WBYTES + c * 1024);
                }
                __builtin_amdgcn_s_setprio(0);               // (as in the 128-channel form below)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (2 * st + t > 8) continue;
#pragma unroll
                    for (int c = 0; c < CT; ++c)
#pragma unroll
                        for (int p = 0; p < PT; ++p)
                            acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[t][c], fx[t][p], acc[c][p], 0, 0, 0);
                }
                __builtin_amdgcn_s_setprio(3);
            }
        }
    } else {
    dma_patch(0, 0);
    dma_w(0, 0, 0);
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const int pb = (chunk & 1) * P32_PATCH;
        const bool next_chunk = chunk + 1 < nchunk;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int wb = ((chunk + tap) & 1) * WBYTES;    // step s = 9 chunk + tap: s & 1 == (chunk + tap) & 1
            // this step's weight slice has landed (the patch prefetch issued after it at tap 0 may still be in flight)
            // ... and this wave's LDS reads of the previous step are retired (lgkmcnt) BEFORE the barrier: the weight
            // buffer they came from is re-staged right after it (a read merely issued before the barrier can still be
            // queued in the LDS when a fast L2-hit DMA of another wave lands)
            if (tap == 1 && next_chunk) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (!(g.ablate & 16)) {                          // dev: 16 = no weight DMA in the loop, 32 = no patch DMA in the loop
                if (tap < 8) dma_w(chunk, tap + 1, ((chunk + tap + 1) & 1));
                else if (next_chunk) dma_w(chunk + 1, 0, ((chunk + tap + 1) & 1));
            }
            if (tap == 0 && next_chunk && !(g.ablate & 32)) dma_patch(chunk + 1, (chunk + 1) & 1);
            bf16x8_t fx[PT], fw[CT];
            const int tapoff = pb + xbase + ((tap / 3) * Q + tap % 3) * P32_PITCH;
#pragma unroll
            for (int p = 0; p < PT; ++p) fx[p] = ldf(tapoff + p * prow);
#pragma unroll
            for (int c = 0; c < CT; ++c) fw[c] = ldf(wb + wbase + c * 1024);
            // the MFMA block runs at LOW priority, everything else of a step (wait, barrier, DMA issue, fragment reads) at high:
            // the short instructions of one workgroup's step slip in between the MFMAs of the other's (+3 % on every layer)
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int p = 0; p < PT; ++p)
                    acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[c], fx[p], acc[c][p], 0, 0, 0);
            __builtin_amdgcn_s_setprio(3);
        }
    }
    }
    if (staged_ok<EPI>(g, ep) && !(g.ablate & 8)) {          // tile image [256 block pixels][BN] in the (now idle) patch buffers
        auto row_to_m = [&](int row) {
            if (flat) return flat_pixel(f0 + row);
            const int ir = image_row(row >> 4), xx = x0 + (row & 15);
            return (ir >= 0 && xx < g.Wo) ? ir * g.Wo + xx : -1;
        };
        auto pool_index = [&](int py, int px) -> long long {
            if (flat || rowflat) return -1;
            const int gy = (y0 >> 1) + py, gx = (x0 >> 1) + px;
            return (gy < ep.pool_h && gx < ep.pool_w) ? ((long long)b * ep.pool_h + gy) * ep.pool_w + gx : -1;
        };
        staged_epilogue<EPI, 256, BN, CT, PT, 512>(acc, smem, g, ep, n0, wave_m * 64, wave_n * (16 * CT), tid, row_to_m, pool_index);
        return;
    }
    int mrow[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        if (flat) {
            mrow[p] = flat_pixel(f0 + (4 * wave_m + p) * 16 + (lane & 15));
        } else {
            const int ir = image_row(4 * wave_m + p), xx = x0 + (lane & 15);
            mrow[p] = (ir >= 0 && xx < g.Wo) ? ir * g.Wo + xx : -1;
        }
    }
    conv_epilogue_rows<EPI, CT, PT>(acc, g, ep, mrow, n0 + wave_n * (16 * CT), lane);
}

// ------------------------------------------------------------------------------------------------
// Third form of the LDS-patch kernel: 512 pixels x 128 channels per workgroup, ONE workgroup per CU (an alternative to
// k_conv3x3_patch32<128>, bit-identical results: same chunk / tap accumulation order).
// Twice the pixels per weight slice: 113 KB from L2 per 1152 MFMAs instead of 93 KB per 576, and a wave's tile is
// 128 px x 64 channels (12 fragment reads per 32 MFMAs instead of 8 per 16).  With a single workgroup on the CU nobody
// covers a stall, so latency is taken out of the loop instead:
//   * weight slices live in a ring of FOUR tap slots: the slice of step s + 3 is requested at step s and made visible
//     (vmcnt + barrier) at the top of step s + 2;
//   * the next chunk's halo patch is requested in the first four taps of the current chunk (two DMA instructions per wave
//     and tap) and is first read at the last tap;
//   * vmcnt is counted per tap (the tap loop is unrolled and the number of DMA instructions a wave issues per tap is fixed:
//     table N(T) below), so no wait includes a younger request;
//   * step s runs on fragments that were read from LDS during step s - 1 (second register set), so its MFMAs start right
//     after the barrier.
// Measured: within +-10 % of k_conv3x3_patch32 layer by layer (faster on the deep narrow maps,
// slower at 150x150) -- with either kernel the matrix pipes are busy ~60 % of the time and the waves are parked at the
// per-tap wait / barrier for most of the rest.
constexpr int P5_PIX = 34 * PATCH_W;                        // 612 halo pixels of a 32 x 16 block
constexpr int P5_NDMA = (P5_PIX * 6 + 63) / 64;             // 58 one-KiB DMA instructions
constexpr int P5_PATCH = P5_NDMA * 1024;                    // 59392 B per buffer
constexpr int P5_WSLOT = 128 * 64;                          // weight slice of one tap: [128 co][32 k]
constexpr int P5_OFF_W = 2 * P5_PATCH;
constexpr int P5_OFF_DUMMY = P5_OFF_W + 4 * P5_WSLOT;       // landing zone of the DMA instructions beyond the patch
constexpr int P5_LDS = P5_OFF_DUMMY + 1024;

template <int EPI, bool FLAT>
__global__ __launch_bounds__(512) void k_conv3x3_p512(const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ w, ConvGeom g,
                                                      Epilogue ep, int tiles_x, int tiles_y, int nblocks, int rowflat) {
    constexpr int BN = 128, CT = 4, PT = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave & 3, wave_n = wave >> 2;        // pixel rows 8 wave_m .. +7 of the block, channels 64 wave_n .. +63
    const int ntn = (g.N + BN - 1) / BN;
    const int kx = blockIdx.x >> 3;
    const int pblock = (kx / ntn) * 8 + (blockIdx.x & 7);   // channel tiles of one pixel block consecutive on ONE XCD
    if (pblock >= nblocks) return;
    const int n0 = (kx % ntn) * BN;
    const int Q = FLAT ? g.W + 2 : PATCH_W;
    const int img = (g.H + 1) * Q;
    int b = 0, y0 = 0, x0 = 0, f0 = 0;
    if constexpr (FLAT) {
        f0 = pblock * 512;
    } else {
        int t = pblock;
        const int tx = t % tiles_x; t /= tiles_x;
        x0 = tx * 16;
        if (rowflat) { y0 = t * 32; }
        else { const int ty = t % tiles_y; b = t / tiles_y; y0 = ty * 32; }
    }
    auto image_row = [&](int r) {                           // block row (-1 .. 32) -> row of [B * H] or -1, as in k_conv3x3_patch32
        if (!rowflat) { const int y = y0 + r; return (unsigned)y < (unsigned)g.H ? b * g.H + y : -1; }
        const int R = y0 + r;
        if (R < 0) return -1;
        const int bb = fdiv(R, g.d_h1), yy = R - bb * (g.H + 1);
        return (bb < g.B && yy < g.H) ? bb * g.H + yy : -1;
    };
    auto flat_pixel = [&](int f) {
        if (f < 0) return -1;
        const int bb = f / img;
        const int r = f - bb * img;
        const int yy = r / Q, xx = r - yy * Q - 1;
        return (bb < g.B && yy < g.H && (unsigned)xx < (unsigned)g.W) ? (bb * g.H + yy) * g.W + xx : -1;
    };
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)g.B * g.H * g.W * g.C * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (unsigned)g.N * (unsigned)g.ldw * 2u, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    // patch DMA: instruction i = wave + 8 j (j < 8; i >= 58 lands in the dummy zone) fills slots 64 i ..: slot q -> pixel q / 6,
    // 16-byte piece q % 6 (pieces 4, 5 are the row padding)
    unsigned pvo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int q = (wave + 8 * j) * 64 + lane;
        const int pp = q / 6, sl = q - pp * 6;
        int pix;
        if constexpr (FLAT) {
            pix = pp < 512 + 2 * (Q + 1) ? flat_pixel(f0 - (Q + 1) + pp) : -1;
        } else {
            const int py = pp / PATCH_W, px = pp - py * PATCH_W;
            const int ix = x0 - 1 + px;
            const int ir = pp < P5_PIX ? image_row(py - 1) : -1;
            pix = (ir >= 0 && (unsigned)ix < (unsigned)g.W) ? ir * g.W + ix : -1;
        }
        pvo[j] = (sl < 4 && pix >= 0 && wave + 8 * j < P5_NDMA) ? ((unsigned)pix * (unsigned)g.C + (unsigned)(sl * 8)) * 2u : OOB;
    }
    unsigned wvo;                                           // weight DMA: wave i fills rows 16 i .. 16 i + 15 of a slot
    {
        const int row = 16 * wave + (lane >> 2);
        const int piece = (lane & 3) ^ ((-(row >> 2)) & 3);
        const int n = n0 + row;
        wvo = n < g.N ? ((unsigned)n * (unsigned)g.ldw + (unsigned)(piece * 8)) * 2u : OOB;
        if (g.ablate & 16) wvo = OOB;                       // dev: weight requests fetch nothing (same instruction count)
    }
    const int nchunk = g.C >> 5;
    const int nstep = nchunk * 9;
    auto dma_piece = [&](int chunk, int buf, int j, bool live) {
        char* dst = wave + 8 * j < P5_NDMA ? smem + buf * P5_PATCH + (wave + 8 * j) * 1024 : smem + P5_OFF_DUMMY;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void*)dst, 16, live ? pvo[j] : OOB, chunk * 64, 0, 0);
    };
    auto dma_w = [&](int step) {                            // slice of step (chunk = step / 9, tap = step % 9) into ring slot step & 3
        const int st = step < nstep ? step : nstep - 1;     // past the end: a harmless refetch (the count per tap stays fixed)
        const int ch = st / 9, tp = st - ch * 9;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (lds_void*)(smem + P5_OFF_W + (step & 3) * P5_WSLOT + wave * 1024), 16, wvo,
                                                 (tp * g.C + ch * 32) * 2, 0, 0);
    };
    f32x4_t acc[CT][PT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p) acc[c][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fk = lane >> 4;
    const int xbase = (FLAT ? 8 * wave_m * 16 + frow : 8 * wave_m * PATCH_W + frow) * P32_PITCH + fk * 16;
    constexpr int prow = (FLAT ? 16 : PATCH_W) * P32_PITCH;
    int wbase;
    {
        const int row = wave_n * 64 + frow;
        wbase = P5_OFF_W + row * 64 + ((fk ^ ((-(row >> 2)) & 3)) << 4);
    }
    auto ldf = [&](int addr) { return lds_read_b128_scoped(smem + addr, smem); };
    auto load_x = [&](bf16x8_t (&fx)[PT], int pb, int tap) {
        const int tapoff = pb + xbase + ((tap / 3) * Q + tap % 3) * P32_PITCH;
#pragma unroll
        for (int p = 0; p < PT; ++p) fx[p] = ldf(tapoff + p * prow);
    };
    auto load_w = [&](bf16x8_t (&fw)[CT], int step) {
        const int wb = (step & 3) * P5_WSLOT;
#pragma unroll
        for (int c = 0; c < CT; ++c) fw[c] = ldf(wb + wbase + c * 1024);
    };
    // prologue: the whole first patch and the first three weight slices; fragments of step 0
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_piece(0, 0, j, true);
    dma_w(0); dma_w(1); dma_w(2);
    bf16x8_t fxa[PT], fxb[PT], fwa[CT], fwb[CT];
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");         // patch 0 and slice 0 (slices 1, 2 are younger)
    __builtin_amdgcn_s_barrier();
    load_x(fxa, 0, 0);
    load_w(fwa, 0);
    // nine taps per chunk: the register set of a step follows the parity of the GLOBAL step, so chunks go in pairs (C % 64 == 0)
    auto run_chunk = [&](int chunk, auto par_tag) {
        constexpr int PAR = decltype(par_tag)::value;
        const int pb = (chunk & 1) * P5_PATCH;
        const bool next_chunk = chunk + 1 < nchunk && !(g.ablate & 32);   // dev 32: patch requests after the first fetch nothing
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // Step s runs on fragments read during step s - 1; at its top it makes slice s + 1 (and, at tap 8, the next patch)
            // visible, requests slice s + 3 and two patch pieces (taps 0-3), and reads the fragments of step s + 1 under its
            // own MFMAs.  DMA instructions a wave has issued AFTER slice s + 1 (requested two steps ago, first of its step):
            //   T:    0  1  2  3  4  5  6  7  8
            //   N(T): 1  3  5  5  5  3  1  1  1       (first chunk: the prologue issued patch, slices 0, 1, 2 in that order: same)
            // All patch pieces of the NEXT chunk are requested by tap 3, i.e. older than everything tap 8 leaves in flight.
            constexpr int NT[9] = {1, 3, 5, 5, 5, 3, 1, 1, 1};
            if (NT[tap] == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
            else if (NT[tap] == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int step = chunk * 9 + tap;
            dma_w(step + 3);
            if (tap < 4) {
                dma_piece(chunk + 1, (chunk + 1) & 1, 2 * tap, next_chunk);
                dma_piece(chunk + 1, (chunk + 1) & 1, 2 * tap + 1, next_chunk);
            }
            bf16x8_t (&fxc)[PT] = ((tap + PAR) & 1) ? fxb : fxa;
            bf16x8_t (&fxn)[PT] = ((tap + PAR) & 1) ? fxa : fxb;
            bf16x8_t (&fwc)[CT] = ((tap + PAR) & 1) ? fwb : fwa;
            bf16x8_t (&fwn)[CT] = ((tap + PAR) & 1) ? fwa : fwb;
            // fragments of the next step (tap 8: tap 0 of the next chunk, from the other patch buffer; the very last step
            // re-reads its own)
            if (tap < 8) load_x(fxn, pb, tap + 1);
            else load_x(fxn, next_chunk ? (P5_PATCH - pb) : pb, next_chunk ? 0 : 8);
            load_w(fwn, step + 1 < nstep ? step + 1 : step);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int p = 0; p < PT; ++p)
                    acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwc[c], fxc[p], acc[c][p], 0, 0, 0);
            // Issue order inside the step: one fragment read after every two MFMAs, the last eight MFMAs without.  Left alone the scheduler either
            // sinks the reads to just before their first use in the NEXT step (the MFMAs then start by waiting for LDS), or,
            // pinned in front of the MFMAs, all eight waves burst 96 KB of reads at the LDS right after the barrier and the
            // in-order MFMAs queue behind them: measured 0.59 MFMA-busy at 2.3 GHz with or without any memory traffic.
#define SSD_P5_GROUP(NM_) __builtin_amdgcn_sched_group_barrier(0x008, NM_, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0)
            SSD_P5_GROUP(2); SSD_P5_GROUP(2); SSD_P5_GROUP(2); SSD_P5_GROUP(2); SSD_P5_GROUP(2); SSD_P5_GROUP(2);
            SSD_P5_GROUP(2); SSD_P5_GROUP(2); SSD_P5_GROUP(2); SSD_P5_GROUP(2); SSD_P5_GROUP(2); SSD_P5_GROUP(2);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);   // the last eight MFMAs cover the latency of the last reads
#undef SSD_P5_GROUP
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int chunk = 0; chunk < nchunk; chunk += 2) {
        run_chunk(chunk, std::integral_constant<int, 0>{});
        run_chunk(chunk + 1, std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the refetches past the end have landed before LDS is reused
    if (staged_ok<EPI>(g, ep)) {                             // tile image [512 block pixels][128] over the patch buffers + ring
        auto row_to_m = [&](int row) {
            if constexpr (FLAT) return flat_pixel(f0 + row);
            const int ir = image_row(row >> 4), xx = x0 + (row & 15);
            return (ir >= 0 && xx < g.Wo) ? ir * g.Wo + xx : -1;
        };
        auto pool_index = [&](int py, int px) -> long long {
            if (FLAT || rowflat) return -1;
            const int gy = (y0 >> 1) + py, gx = (x0 >> 1) + px;
            return (gy < ep.pool_h && gx < ep.pool_w) ? ((long long)b * ep.pool_h + gy) * ep.pool_w + gx : -1;
        };
        staged_epilogue<EPI, 512, BN, CT, PT, 512>(acc, smem, g, ep, n0, wave_m * 128, wave_n * 64, tid, row_to_m, pool_index);
        return;
    }
    int mrow[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        if constexpr (FLAT) {
            mrow[p] = flat_pixel(f0 + (8 * wave_m + p) * 16 + (lane & 15));
        } else {
            const int ir = image_row(8 * wave_m + p), xx = x0 + (lane & 15);
            mrow[p] = (ir >= 0 && xx < g.Wo) ? ir * g.Wo + xx : -1;
        }
    }
    conv_epilogue_rows<EPI, CT, PT>(acc, g, ep, mrow, n0 + wave_n * 64, lane);
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 with 64 input and 64 output channels (block1_conv2 forward and data gradient): 1.5-2.2 GB of
// traffic for 425 GFLOP, i.e. close to the HBM bound, and in the general patch kernel the [64][576] weight matrix
// (72 KB) was re-streamed for every 16x16 block: more than half of the CU's L2 ingest.  Here the weights live in
// REGISTERS for the whole kernel (a wave owns 32 output channels: 2 tiles x 9 taps x 2 k-halves = 36 A fragments,
// 144 VGPRs), workgroups are persistent, the only LDS traffic is the halo patch (64 channels, 160-byte rows: conflict-
// free and immediate-addressed) which is prefetched one block ahead, and the nine taps of a block run without a
// single barrier.  The result leaves through the shared staged epilogue.
constexpr int C64_PITCH = 160;

// ------------------------------------------------------------------------------------------------
// Timing the pieces of a first form of this kernel (16x16 blocks, one 8-wave workgroup per CU) on MI355X showed its
// fragment reads (72 ds_read_b128 per wave and block = 4.6k LDS cycles per block at 128 B/clk) to cost as much as its
// MFMAs (4.6k cycles) without overlapping them, and the epilogue (another 30 % of the time) to run with the matrix cores
// idle because the CU holds a single workgroup.  Here
//   * a fragment is read ONCE per patch row and reused by every tap that touches it: the wave walks the 6 patch rows of
//     its 4 output rows, and the fragment (row r, dx, k-half) feeds the MFMAs of (dy, p = r - dy) for all valid dy --
//     36 reads per wave and block instead of 72, issued two steps ahead of their use;
//   * workgroups are 4 waves on an 8 x 16 block and two of them share a CU (74 KB of LDS, 256 VGPRs per wave): they
//     drift apart, so one's epilogue (LDS turn-around, coalesced stores, fused pooling) runs under the other's MFMAs.
constexpr int C64B_ROWS = 8;                               // block = 8 rows x 16 columns
constexpr int C64B_PATCH_PIX = (C64B_ROWS + 2) * PATCH_W;  // 180 halo pixels
constexpr int C64B_NDMA = (C64B_PATCH_PIX * 10 + 63) / 64; // 29 wave-instructions of 64 x 16 B
constexpr int C64B_PATCH = C64B_NDMA * 1024;               // 29696 B per buffer
constexpr int C64B_STAGE = 2 * C64B_PATCH;                 // [128 px][64 ch] bf16 staging tile (16 KB)
constexpr int C64B_FRAG_DEPTH = 3;                          // fragment reads run two steps ahead (four and six: measured, no faster)
constexpr int C64B_BIAS = C64B_STAGE + 128 * 128;          // forward: the 64 biases (the accumulators start from them)
constexpr int C64B_LDS = C64B_BIAS + 256;

// W0 form (block1_conv2's data gradient + block1_conv1's weight gradient in one kernel).  The gradient this layer produces has
// exactly one consumer -- the weight gradient of the 3-channel first layer, dW0[co][tap][ci] = sum_px dX[px][co] * img[px +
// tap][ci] -- and that consumer was a pure HBM stream (k_conv0_wgrad: 737 MB read, 187 us at batch 64) behind a 737 MB
// write here.  The masked bf16 tile is already in LDS for the store stage: instead of storing it, the four waves multiply
// it with the 10 x 18 image halo patch (channels 0..3 of the 16-byte pixel, fetched by 4-byte LDS-DMA next to the 64-channel
// patch): wave w owns output channels 16 w .. +15, three column tiles of (4 taps x 4 channels) -- the pad channel of the
// centre tap carries ones, i.e. the bias gradient --, four k-steps of 32 pixels: 12 MFMAs and 32 transposing LDS reads per
// wave and block next to the 144 MFMAs of the convolution.  The running sums live in LDS (the kernel has no register to
// spare), one [64][72] + [64] fp32 slab per workgroup leaves at the end; nothing else is written.
struct C64W0 {
    const bf16_raw* img;                                    // [B][H][W][8] bf16 (ssd_image_prep), channels 3..7 zero
    float* slab_w;                                          // [gridDim.x][64][72]
    float* slab_b;                                          // [gridDim.x][64]
};
constexpr int C64B_IMG = 6 * 256;                           // 180 halo pixels x 8 bytes = six 4-byte DMA instructions
constexpr int C64B_W0_IMG = 2 * C64B_PATCH;                 // (the staging tile reuses the patch buffer just consumed)
constexpr int C64B_W0_SUMS = C64B_W0_IMG + 2 * C64B_IMG;
constexpr int C64B_W0_LDS = C64B_W0_SUMS + 4 * 3 * 64 * 16; // [wave][column tile][lane] f32x4

typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
__device__ __forceinline__ s16x4_t lds_read_tr16_scoped(const char* __restrict__ p, const char* __restrict__ other) {
    (void)other;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
}

template <int EPI, bool W0 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_conv3x3_c64b(const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ w, ConvGeom g, Epilogue ep, int tiles_x, int tiles_y,
                    C64W0 w0) {
    static_assert(!W0 || EPI == EPI_DGRAD, "the fused first-layer weight gradient belongs to the data gradient");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave & 1, wave_n = wave >> 1;        // output rows 4 wave_m + p, channels 32 wave_n + 16 c
    const int nblocks = g.B * tiles_x * tiles_y;
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)g.B * g.H * g.W * 64u * 2u, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    constexpr int NJ = (C64B_NDMA + 3) / 4;                 // DMA instructions per wave: i = wave + 4 j
    const __amdgpu_buffer_rsrc_t ires = __builtin_amdgcn_make_buffer_rsrc((void*)(W0 ? w0.img : x), 0, (unsigned)g.B * g.H * g.W * 16u, 0x00020000);

    int pcode[NJ];                                          // py << 16 | px << 8 | byte offset of the piece, -1 = padding
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = (wave + 4 * j) * 64 + lane;
        const int pp = q / 10, sl = q - pp * 10;
        const int py = pp / PATCH_W, px = pp - py * PATCH_W;
        pcode[j] = (sl < 8 && pp < C64B_PATCH_PIX && wave + 4 * j < C64B_NDMA) ? ((py << 16) | (px << 8) | (sl * 16)) : -1;
    }
    auto issue_patch = [&](int t, int buf) {
        int r = t;
        const int tx = r % tiles_x; r /= tiles_x;
        const int ty = r % tiles_y;
        const int b = r / tiles_y;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (wave + 4 * j < C64B_NDMA) {
                const int iy = ty * C64B_ROWS - 1 + (pcode[j] >> 16), ix = tx * 16 - 1 + ((pcode[j] >> 8) & 255);
                const bool ok = pcode[j] >= 0 && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
                const unsigned off = (unsigned)((b * g.H + iy) * g.W + ix) * 128u + (unsigned)(pcode[j] & 255);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void*)(smem + buf * C64B_PATCH + (wave + 4 * j) * 1024), 16,
                                                         ok ? off : OOB, 0, 0, 0);
            }
        }
        if constexpr (W0) {                                 // image halo: dword q = pixel q / 2, channels 2 (q & 1) .. +1
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (wave + 4 * j < 6) {
                    const int q = (wave + 4 * j) * 64 + lane, pp = q >> 1;
                    const int py = pp / PATCH_W, px = pp - py * PATCH_W;
                    const int iy = ty * C64B_ROWS - 1 + py, ix = tx * 16 - 1 + px;
                    const bool ok = pp < C64B_PATCH_PIX && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
                    const unsigned off = (unsigned)((b * g.H + iy) * g.W + ix) * 16u + (unsigned)(q & 1) * 4u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ires, (lds_void*)(smem + C64B_W0_IMG + buf * C64B_IMG + (wave + 4 * j) * 256), 4,
                                                             ok ? off : OOB, 0, 0, 0);
                }
            }
        }
    };
    // XCD-aware order: workgroup i runs on XCD i % 8; give every XCD one contiguous range of blocks and let its workgroups
    // walk it side by side, so that the halo rows/columns shared by neighbouring blocks meet in that XCD's L2
    const bool xcd_order = (gridDim.x & 7) == 0 && !(g.ablate & 64);
    const int per_xcd = (nblocks + 7) >> 3, slots = gridDim.x >> 3;
    auto block_of = [&](int i) {                            // the i-th block of this workgroup, -1 = done
        if (!xcd_order) { const int t = blockIdx.x + i * gridDim.x; return t < nblocks ? t : -1; }
        const int lin = i * slots + ((int)blockIdx.x >> 3);
        const int t = ((int)blockIdx.x & 7) * per_xcd + lin;
        return (lin < per_xcd && t < nblocks) ? t : -1;
    };
    if (block_of(0) >= 0) issue_patch(block_of(0), 0);

    const int frow = lane & 15, fk = lane >> 4;
    bf16x8_t fw[2][9][2];                                   // as in k_conv3x3_c64
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int n = wave_n * 32 + c * 16 + frow;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (n < g.N) v = *reinterpret_cast<const uint4*>(w + (unsigned)n * 576u + (unsigned)(tap * 64 + ks * 32 + fk * 8));
                fw[c][tap][ks] = *reinterpret_cast<const bf16x8_t*>(&v);
            }
    }
    const int xbase = (4 * wave_m * PATCH_W + frow) * C64_PITCH + fk * 16;
    // forward: the bias is the accumulators' initial value (from LDS, written once here; the first barrier of the block loop
    // publishes it) -- fetched in the epilogue it was a global round trip per block with the matrix cores idle
    if constexpr (EPI == EPI_FWD) {
        if (tid < 64) reinterpret_cast<float*>(smem + C64B_BIAS)[tid] = (ep.bias && tid < g.N) ? ep.bias[tid] : 0.f;
    }
    if constexpr (W0) {                                     // this wave's running sums (no other wave touches them)
        f32x4_t* sums = reinterpret_cast<f32x4_t*>(smem + C64B_W0_SUMS) + wave * 3 * 64 + lane;
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) sums[nt * 64] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    // W0: mkn[p] = the four sign bytes of this wave's 32 channels at the lane's pixel of accumulator row p of block t (block row
    // 4 wave_m + p, column lane & 15); 0 outside the map
    unsigned mkn[4] = {0u, 0u, 0u, 0u};
    auto load_masks = [&](int t) {
        int r = t;
        const int tx = r % tiles_x; r /= tiles_x;
        const int ty = r % tiles_y;
        const int b = r / tiles_y;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int y = ty * C64B_ROWS + 4 * wave_m + p, xx = tx * 16 + (lane & 15);
            mkn[p] = 0u;
            if (y < g.Ho && xx < g.Wo)
                mkn[p] = *reinterpret_cast<const unsigned*>(ep.mask_bits + (long long)((b * g.Ho + y) * g.Wo + xx) * 8 + wave_n * 4);
        }
    };
    if constexpr (W0) { if (block_of(0) >= 0) load_masks(block_of(0)); }

    int it = 0, prev_st = 0;
    for (int t = block_of(0); t >= 0; t = block_of(++it)) {
        const int cur = it & 1;
        // this block's patch (issued one block ago) is older than the epilogue stores issued since: wait for all but those
        // (prev_st = wave-uniform lower bound of the store instructions the previous epilogue issued)
        if (g.ablate & 2) {}                                // (timing experiment, development builds: do not wait for the patch)
        else if (prev_st >= 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else if (prev_st == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
        else if (prev_st == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else if (prev_st == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        else if (prev_st == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else if (prev_st == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        __builtin_amdgcn_s_setprio(0);
        int r = t;
        const int tx = r % tiles_x; r /= tiles_x;
        const int ty = r % tiles_y;
        const int b = r / tiles_y;
        const int y0 = ty * C64B_ROWS, x0 = tx * 16;
        // data gradient: the ReLU mask (the forward activation at the output position) is fetched NOW, ahead of the next
        // patch, and is in registers long before the epilogue needs it -- fetched there, every block paid a full memory
        // round trip with the matrix cores idle (300 us of the 630 us this kernel took)
        // (only the sign-bit form of the mask is pre-fetched -- one register per chunk; a bf16 mask_src takes the shared staged
        //  epilogue, which reads it there: holding four 16-byte masks across the MFMA phase cost 12 more registers in a kernel
        //  that sits at the 256-register limit and spilled 20, each reload in the epilogue draining the store queue)
        //  The four chunks of a thread sit in ONE column of the block, two rows apart: their pixel indices are recomputed where
        //  they are used instead of being held across the MFMA phase.)
        unsigned mk[4];
        // chunk i of thread `t`: block row 2 i + (t >> 7), column (t >> 3) & 15, channels 8 (t & 7) .. +7; -1 = outside the map
        auto chunk_pixel = [&](int t, int i) {
            const int y = y0 + 2 * i + (t >> 7), xx = x0 + ((t >> 3) & 15);
            return (y < g.Ho && xx < g.Wo) ? (b * g.Ho + y) * g.Wo + xx : -1;
        };
        const bool premask = EPI == EPI_DGRAD && ep.mask_bits != nullptr && !(g.ablate & 128);
        if constexpr (W0) {
            // W0 form: the mask is applied to the accumulators themselves (the masked tile feeds the weight-gradient MFMAs from
            // LDS: no second pass over it) and travels one block ahead like the patch: the wait at the top of the loop has
            // covered it, the epilogue waits for no memory operation at all
#pragma unroll
            for (int p = 0; p < 4; ++p) mk[p] = mkn[p];
        } else if constexpr (EPI == EPI_DGRAD) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = chunk_pixel(tid, i);
                mk[i] = 0u;
                if (premask && m >= 0) mk[i] = ep.mask_bits[(long long)m * 8 + (tid & 7)];
            }
        }
        const int tnext = block_of(it + 1);
        if (tnext >= 0 && !(g.ablate & 1)) issue_patch(tnext, cur ^ 1);
        if constexpr (W0) { if (tnext >= 0) load_masks(tnext); }
        const int pb = cur * C64B_PATCH + xbase;
        f32x4_t acc[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4_t a0 = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if constexpr (EPI == EPI_FWD)
                a0 = *reinterpret_cast<const f32x4_t*>(smem + C64B_BIAS + (wave_n * 32 + c * 16 + (lane >> 4) * 4) * 4);
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[c][p] = a0;
        }
        // step s = (patch row rr = s / 6, dx = (s / 2) % 3, k-half = s & 1)
        auto frag = [&](int s2) {
            return *reinterpret_cast<const bf16x8_t*>(smem + pb + ((s2 / 6) * PATCH_W + (s2 / 2) % 3) * C64_PITCH + (s2 & 1) * 64);
        };
        // fragment ring: the read of step s2 + FD - 1 is issued at the start of step s2 (into the slot step s2 - 1 used)
        constexpr int FD = C64B_FRAG_DEPTH;
        bf16x8_t fr[FD];
#pragma unroll
        for (int i = 0; i < FD - 1; ++i) fr[i] = frag(i);
#pragma unroll
        for (int s2 = 0; s2 < 36; ++s2) {
            if (s2 + FD - 1 < 36) fr[(s2 + FD - 1) % FD] = frag(s2 + FD - 1);
            const int rr = s2 / 6, dx = (s2 / 2) % 3, ks = s2 & 1;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int p = rr - dy;
                if (p >= 0 && p < 4) {
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[c][dy * 3 + dx][ks], fr[s2 % FD], acc[c][p], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);              // keep the reads FD - 1 steps ahead, no further (144 VGPRs of weights)
        }
        // From here to the barrier at the top of the next block this wave runs at high priority: its epilogue (conversions, LDS
        // turn, stores / the fused weight gradient) is a few hundred short instructions that were being starved of issue slots
        // by the OTHER workgroup's stream of 144 MFMAs on the same SIMD -- the two workgroups of a CU did not overlap, they took
        // turns (section 4: "the second workgroup hides only half").  W0 form 480 -> 446 us, forward 410 -> 403 us.
        __builtin_amdgcn_s_setprio(3);
        if (g.ablate & 8) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int p = 0; p < 4; ++p) asm volatile("" :: "v"(acc[c][p]));
            prev_st = 0;
            continue;
        }
        auto row_to_m = [&](int row) {
            const int y = y0 + (row >> 4), xx = x0 + (row & 15);
            return (y < g.Ho && xx < g.Wo) ? (b * g.Ho + y) * g.Wo + xx : -1;
        };
        auto pool_index = [&](int py, int px) -> long long {
            const int gy = (y0 >> 1) + py, gx = (x0 >> 1) + px;
            return (gy < ep.pool_h && gx < ep.pool_w) ? ((long long)b * ep.pool_h + gy) * ep.pool_w + gx : -1;
        };
        if constexpr (W0) {
            // (the host only launches this form with sign bits: premask holds)
            char* st = smem + cur * C64B_PATCH;             // staging tile [128 px][64 ch] over the patch every wave is done with
            // (bare barriers: __syncthreads() would also wait for the next block's patch, the LDS-DMA in flight)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            int t2 = tid;
            asm volatile("" : "+v"(t2));                     // (opaque copy: indices are recomputed here, not held across the MFMA phase)
            {
                const int l3 = t2 & 63;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int col = wave_n * 32 + c * 16 + (l3 >> 4) * 4;
                    const int sh = 8 * (2 * c + (l3 >> 5)) + 4 * ((l3 >> 4) & 1);   // this lane's four sign bits within mk[p]
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int row = wave_m * 64 + p * 16 + (l3 & 15);
                        const unsigned bits = mk[p] >> sh;
                        const unsigned k0 = keep_mask2(bits, 0), k1 = keep_mask2(bits, 2);
                        lds_st8_scoped(st + row * 128 + ((((col >> 3) ^ row) & 7) << 4) + (col & 4) * 2, smem,
                            make_uint2((pack_bf16x2(acc[c][p][0], acc[c][p][1])) & k0,
                                       (pack_bf16x2(acc[c][p][2], acc[c][p][3])) & k1));
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // dW0 tile of this wave: D[co = 16 wave + 4 (lane >> 4) + j][column = lane & 15] over the 128 pixels of the block.
            // MFMA k index 8 g + e of k-step s  <->  block pixel 32 s + 8 g + e (row 2 s + (g >> 1), column 8 (g & 1) + e); both
            // operands come through the transposing read: lane 4 q + p of a 16-lane group supplies the address of k row q (+ 4
            // for the second half), elements 4 p .. 4 p + 3 of the 16 columns.
            int l2 = t2 & 63;
            const int gq = l2 >> 4, q = (l2 >> 2) & 3, p4 = l2 & 3;
            int db[2], ib[3];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = 8 * gq + q + 4 * h;
                db[h] = r * 128 + ((((2 * wave + (p4 >> 1)) ^ r) & 7) << 4) + (p4 & 1) * 8;
            }
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {                // column 4 p + ci of tile nt = tap 4 nt + p, image channel ci
                int tap = 4 * nt + p4;
                if (tap > 8) tap = 8;                       // (columns of taps 9..11 are never stored)
                ib[nt] = (((gq >> 1) + tap / 3) * PATCH_W + 8 * (gq & 1) + q + tap % 3) * 8;
            }
            const char* im = smem + C64B_W0_IMG + cur * C64B_IMG;
            f32x4_t* sums = reinterpret_cast<f32x4_t*>(smem + C64B_W0_SUMS) + wave * 3 * 64 + l2;
            f32x4_t aw[3];
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                const uint4 u = lds_ld16_scoped(reinterpret_cast<const char*>(sums + nt * 64), smem);
                aw[nt] = f32x4_t{__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
            }
            bf16x8_t ones;
#pragma unroll
            for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8_t fd;
                reinterpret_cast<s16x4_t*>(&fd)[0] = lds_read_tr16_scoped(st + db[0] + s * 4096, smem);
                reinterpret_cast<s16x4_t*>(&fd)[1] = lds_read_tr16_scoped(st + db[1] + s * 4096, smem);
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    bf16x8_t fp;
                    reinterpret_cast<s16x4_t*>(&fp)[0] = lds_read_tr16_scoped(im + ib[nt] + s * (2 * PATCH_W * 8), smem);
                    reinterpret_cast<s16x4_t*>(&fp)[1] = lds_read_tr16_scoped(im + ib[nt] + s * (2 * PATCH_W * 8) + 32, smem);
                    if (nt == 1) fp = (l2 & 15) == 3 ? ones : fp;   // centre tap, pad channel: the bias gradient
                    aw[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd, fp, aw[nt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int nt = 0; nt < 3; ++nt)
                lds_st16_scoped(reinterpret_cast<char*>(sums + nt * 64), smem,
                                make_uint4(__float_as_uint(aw[nt][0]), __float_as_uint(aw[nt][1]), __float_as_uint(aw[nt][2]), __float_as_uint(aw[nt][3])));
            prev_st = 0;                                    // nothing was stored: only the DMA is in flight
            continue;
        }
        // The staging tile has its own LDS here and the barrier at the top of the block loop already separates one block's
        // reads of it from the next block's writes: ONE barrier per epilogue, and a bare one -- __syncthreads() also waits for
        // every LDS-DMA in flight (s_waitcnt vmcnt(0)), i.e. for the next block's patch, in the middle of the epilogue.
        if constexpr (EPI == EPI_FWD) {
            char* st = smem + C64B_STAGE;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int col = wave_n * 32 + c * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int row = wave_m * 64 + p * 16 + (lane & 15);
                    float v[4] = {acc[c][p][0], acc[c][p][1], acc[c][p][2], acc[c][p][3]};
                    if (ep.relu) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                    }
                    lds_st8_scoped(st + row * 128 + ((((col >> 3) ^ row) & 7) << 4) + (col & 4) * 2, smem,
                        make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            int t2 = tid;
            asm volatile("" : "+v"(t2));                     // (opaque copy: the store stage recomputes its indices instead of holding them across the MFMA phase)
            staged_store<EPI_FWD, 128, 64, 256, decltype(row_to_m), decltype(pool_index)>(st, g, ep, 0, t2, row_to_m, pool_index);
        } else if (premask) {
            if constexpr (EPI == EPI_DGRAD) {               // staged_epilogue's data-gradient path with the mask already here
                char* st = smem + C64B_STAGE;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int col = wave_n * 32 + c * 16 + (lane >> 4) * 4;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int row = wave_m * 64 + p * 16 + (lane & 15);
                        lds_st8_scoped(st + row * 128 + ((((col >> 3) ^ row) & 7) << 4) + (col & 4) * 2, smem,
                            make_uint2(pack_bf16x2(acc[c][p][0], acc[c][p][1]),
                                       pack_bf16x2(acc[c][p][2], acc[c][p][3])));
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                int t2 = tid;
                asm volatile("" : "+v"(t2));                 // (opaque copy: keeps the compiler from carrying the indices over from above)
                // the masks are consumed HERE, in uniform control flow: left to the divergent chunks below, the wait for them was
                // re-issued at every join as s_waitcnt vmcnt(0) -- which also waits for the chunk store just issued
                asm volatile("" : "+v"(mk[0]), "+v"(mk[1]), "+v"(mk[2]), "+v"(mk[3]));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int idx = i * 256 + t2, row = idx >> 3, ch = idx & 7;
                    const int m = chunk_pixel(t2, i);
                    if (m < 0) continue;
                    uint4 v = lds_ld16_scoped(st + row * 128 + (((ch ^ row) & 7) << 4), smem);
                    v = gate_bits8(v, mk[i]);
                    *reinterpret_cast<uint4*>(ep.out + (long long)m * ep.ldo + ch * 8) = v;
                }
            }
        } else {
            staged_epilogue<EPI, 128, 64, 2, 4, 256, decltype(row_to_m), decltype(pool_index), false>(
                acc, smem + C64B_STAGE, g, ep, 0, wave_m * 64, wave_n * 32, tid, row_to_m, pool_index);
        }
        // store instructions of that epilogue with at least one active lane (its loop: iteration i, wave w covers the
        // 8 pixels x = 8 (w & 1) .. +7 of block row 2 i + (w >> 1)): a lower bound of what this wave issued after the DMA
        prev_st = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) prev_st += (y0 + 2 * i + (wave >> 1) < g.Ho && x0 + 8 * (wave & 1) < g.Wo) ? 1 : 0;
        if (EPI == EPI_FWD && !ep.out) prev_st = 0;          // pool-only: no full-resolution stores
        // fused pooling: wave w stores pooled row (y0 >> 1) + w, columns (x0 >> 1) .. + 7 (pooled map, then the codes)
        if (EPI == EPI_FWD && ep.pool_out && (y0 >> 1) + wave < ep.pool_h && (x0 >> 1) < ep.pool_w) prev_st += 2;
    }
    if constexpr (W0) {                                     // this workgroup's slab, in the layout of k_conv0_wgrad's
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        const float* sm = reinterpret_cast<const float*>(smem + C64B_W0_SUMS);
        auto tile_value = [&](int co, int tap, int ci) {    // D element (row co & 15, column 4 (tap & 3) + ci) of tile (co >> 4, tap >> 2)
            const int l = 16 * ((co & 15) >> 2) + 4 * (tap & 3) + ci;
            return sm[(((co >> 4) * 3 + (tap >> 2)) * 64 + l) * 4 + (co & 3)];
        };
        for (int idx = tid; idx < 64 * 72 + 64; idx += 256) {
            if (idx < 64 * 72) {
                const int co = idx / 72, col = idx - co * 72;
                w0.slab_w[(long long)blockIdx.x * (64 * 72) + idx] = (col & 7) < 3 ? tile_value(co, col >> 3, col & 7) : 0.f;
            } else if (w0.slab_b) {
                w0.slab_b[(long long)blockIdx.x * 64 + (idx - 64 * 72)] = tile_value(idx - 64 * 72, 4, 3);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// First layer, forward: 3x3 / stride 1 / pad 1, 8 input channels (3 image channels + padding), 64 output channels,
// bias + ReLU.  33 us of MFMA work against 830 MB of traffic: the kernel is organised around the stores.  A workgroup
// owns a 16x16 block: the 18x18 halo patch is 5 KB (one 16-byte pixel per DMA lane), the whole [64][72] weight matrix
// lives in registers as MFMA A fragments (k = 4 taps x 8 channels per MFMA, 3 MFMAs cover the 9 taps), every wave
// computes two 16-pixel rows and turns each [16 px][64 ch] result through its private 2 KB of LDS so that it leaves as
// two fully coalesced 1 KiB stores (16 consecutive NHWC pixels are contiguous).
__global__ __launch_bounds__(512) void k_conv0_fwd(const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ w,
                                                   const float* __restrict__ bias, bf16_raw* __restrict__ out, ConvGeom g,
                                                   int relu, int tiles_x, int tiles_y, unsigned char* __restrict__ relu_bits) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 6 * 1024 + 8 * 2048];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblocks = g.B * tiles_x * tiles_y;
    // persistent: blocks blockIdx.x, + gridDim.x, ...; the next block's patch is in flight while this one is computed
    auto issue_patch = [&](int t, int buf) {
        if (wave < 6) {                                     // one pixel per lane, row-major 18x18
            int r = t;
            const int tx = r % tiles_x; r /= tiles_x;
            const int ty = r % tiles_y;
            const int b = r / tiles_y;
            const int pp = wave * 64 + lane;
            const int py = pp / PATCH_W, px = pp - py * PATCH_W;
            const int iy = ty * 16 - 1 + py, ix = tx * 16 - 1 + px;
            const bool ok = pp < PATCH_PIX && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
            const bf16_raw* src = ok ? x + (unsigned)((b * g.H + iy) * g.W + ix) * 8u : reinterpret_cast<const bf16_raw*>(g_zero16);
            __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(smem + buf * 6144 + wave * 1024), 16, 0, 0);
        }
    };
    if ((int)blockIdx.x < nblocks) issue_patch(blockIdx.x, 0);
    // weights as A fragments: lane (row = lane & 15, k chunk fk = lane >> 4) of tile c, k-step ks holds
    // w[c*16 + row][tap = 4 ks + fk][0..7]
    const int frow = lane & 15, fk = lane >> 4;
    bf16x8_t fw[4][3];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int tap = 4 * ks + fk, co = c * 16 + frow;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (tap < 9 && co < g.N) v = *reinterpret_cast<const uint4*>(w + (unsigned)co * 72u + (unsigned)tap * 8u);
            fw[c][ks] = *reinterpret_cast<const bf16x8_t*>(&v);
        }
    float b4[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int n = c * 16 + fk * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) b4[c][j] = (bias && n + j < g.N) ? bias[n + j] : 0.f;
    }
    // pixel fragment of k-step ks: lane (pixel = lane & 15, fk) reads the patch pixel shifted by tap 4 ks + fk
    int toff[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        int tap = 4 * ks + fk;
        if (tap > 8) tap = 8;                               // weights of the missing taps are zero
        toff[ks] = ((tap / 3) * PATCH_W + tap % 3 + frow) * 16;
    }
    char* stage = smem + 2 * 6144 + wave * 2048;
    int it = 0, prev_st = 0;
    for (int t = blockIdx.x; t < nblocks; t += gridDim.x, ++it) {
        const int cur = it & 1;
        // the patch DMA of this block is older than the stores this wave issued since: leave those in flight.  prev_st
        // counts only the store instructions that had at least one active lane (wave-uniform), i.e. a lower bound
        if (prev_st >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (prev_st == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (prev_st == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (prev_st == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + (int)gridDim.x < nblocks) issue_patch(t + gridDim.x, cur ^ 1);
        int r = t;
        const int tx = r % tiles_x; r /= tiles_x;
        const int ty = r % tiles_y;
        const int b = r / tiles_y;
        const int y0 = ty * 16, x0 = tx * 16;
        const char* patch = smem + cur * 6144;
        prev_st = 0;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) prev_st += (y0 + 2 * wave + p < g.Ho && x0 + 8 * h < g.Wo) ? 1 : 0;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int row = 2 * wave + p;                   // block row of this pixel tile
            bf16x8_t fx[3];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) fx[ks] = *reinterpret_cast<const bf16x8_t*>(patch + row * (PATCH_W * 16) + toff[ks]);
            f32x4_t acc[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[c][ks], fx[ks], acc[c], 0, 0, 0);
            }
            // lane holds channels c*16 + fk*4 + {0..3} of pixel frow: bias, ReLU, pack, into the wave's staging tile
            // [16 px][128 B] with the 16-byte chunk index XORed by (px & 7)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v[4] = {acc[c][0] + b4[c][0], acc[c][1] + b4[c][1], acc[c][2] + b4[c][2], acc[c][3] + b4[c][3]};
                if (relu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                }
                // (scoped: a plain LDS store here waits for the next block's patch DMA and, with it, for every store of the
                //  previous row -- s_waitcnt vmcnt(0) twice per block in a kernel that lives on its stores)
                lds_st8_scoped(stage + frow * 128 + (((c * 2 + (fk >> 1)) ^ (frow & 7)) << 4) + (fk & 1) * 8, smem,
                    make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave wrote it: no barrier needed
            const int y = y0 + row;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int idx = h * 64 + lane;
                const int px = idx >> 3, ch = idx & 7;
                const uint4 v = lds_ld16_scoped(stage + px * 128 + ((ch ^ (px & 7)) << 4), smem);
                const int xx = x0 + px;
                if (y < g.Ho && xx < g.Wo) {        // N == 64 (host check): every chunk of the pixel is stored
                    *reinterpret_cast<uint4*>(out + ((unsigned)((b * g.Ho + y) * g.Wo + xx) * 64u + (unsigned)(ch * 8))) = v;
                    if (relu_bits) relu_bits[(unsigned)((b * g.Ho + y) * g.Wo + xx) * 8u + (unsigned)ch] = (unsigned char)relu_bits8(v);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging tile is reused by the second row
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient.  grid (col tiles, co tiles, splits).  Per step 64 pixels.
constexpr int WG_LD = 288;                   // LDS row stride (bytes) of a [pixel][128 ch] tile: 256 + 32 pad

__device__ __forceinline__ void conv_wgrad_block(const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ dy,
                                                 float* __restrict__ slab_w, float* __restrict__ slab_b, const ConvGeom& g,
                                                 int m_per_split, char* smem, const int bidx, const int bidy, const int bidz) {
    // g: source = x dims (B,H,W,C), destination = dy dims (Ho,Wo,N); mul = stride, div = 1
    constexpr int TILE = 64 * WG_LD;
    auto s_dy = [&](int buf) { return smem + buf * (2 * TILE); };
    auto s_x = [&](int buf) { return smem + buf * (2 * TILE) + TILE; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave & 1, wave_n = wave >> 1;       // m: co, n: (tap,ci) columns
    const int col0 = bidx * 128, co0 = bidy * 128;
    const int ktot = g.ldw;                                 // KH*KW*C columns
    const int m_begin = bidz * m_per_split;
    const int m_end = min(g.M, m_begin + m_per_split);

    const int cslot = tid & 15, prow = tid >> 4;           // 16-byte column chunk, pixel row (+16j)
    // this thread's X column chunk -> (tap, channel)
    const int qx = (col0 >> 3) + cslot;
    const bool xcol_ok = qx < g.nchunks;
    const int tapx = xcol_ok ? qx / g.cpt : 0;
    const int ccx = qx - tapx * g.cpt;
    const int khx = tapx / g.KW, kwx = tapx - khx * g.KW;
    const int co_chunk = co0 + cslot * 8;
    const bool dycol_ok = co_chunk < g.N;

    uint4 rdy[4], rxx[4];
    auto load_tiles = [&](int mstep) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mstep + prow + 16 * j;
            const bool mok = m < m_end;
            rdy[j] = make_uint4(0, 0, 0, 0);
            rxx[j] = make_uint4(0, 0, 0, 0);
            if (mok && dycol_ok) rdy[j] = *reinterpret_cast<const uint4*>(dy + ((long long)m * g.N + co_chunk));
            if (mok && xcol_ok) {
                const int b = fdiv(m, g.d_hw);
                const int rem = m - b * g.d_hw.d;
                const int oy = fdiv(rem, g.d_w);
                const int ox = rem - oy * g.d_w.d;
                const int iy = oy * g.mul - g.pad_t + khx, ix = ox * g.mul - g.pad_l + kwx;
                if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                    rxx[j] = *reinterpret_cast<const uint4*>(x + ((((long long)b * g.H + iy) * g.W + ix) * g.C + ccx * 8));
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<uint4*>(s_dy(buf) + (prow + 16 * j) * WG_LD + cslot * 16) = rdy[j];
            *reinterpret_cast<uint4*>(s_x(buf) + (prow + 16 * j) * WG_LD + cslot * 16) = rxx[j];
        }
    };

    f32x4_t acc[4][4];
    f32x4_t accb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        accb[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = slab_b != nullptr && bidx == 0 && wave_n == 0;
    bf16x8_t ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;

    const int nsteps = (m_end - m_begin + 63) / 64;
    if (nsteps > 0) {
        load_tiles(m_begin);
        store_tiles(0);
    }
    __syncthreads();
    // transposing read: lane (16-group gq, index i) supplies the address of k-row (i>>2), columns 4*(i&3)..+3
    const int gq = lane >> 4, li = lane & 15;
    const int tr_row = li >> 2, tr_col = (li & 3) * 4;
    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        const bool more = st + 1 < nsteps;
        if (more) load_tiles(m_begin + (st + 1) * 64);
#pragma unroll
        for (int ksub = 0; ksub < 2; ++ksub) {
            bf16x8_t fa[4], fb[4];
            const int krow = ksub * 32 + gq * 8 + tr_row;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const char* base = s_dy(cur) + krow * WG_LD + (wave_m * 64 + a * 16 + tr_col) * 2;
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base));
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base + 4 * WG_LD));
                union { s16x4_t h[2]; bf16x8_t v; } u;
                u.h[0] = lo; u.h[1] = hi;
                fa[a] = u.v;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const char* base = s_x(cur) + krow * WG_LD + (wave_n * 64 + c * 16 + tr_col) * 2;
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base));
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base + 4 * WG_LD));
                union { s16x4_t h[2]; bf16x8_t v; } u;
                u.h[0] = lo; u.h[1] = hi;
                fb[c] = u.v;
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[c], acc[a][c], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int a = 0; a < 4; ++a) accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], ones, accb[a], 0, 0, 0);
            }
        }
        if (more) store_tiles(cur ^ 1);
        __syncthreads();
    }
    // partial tile -> slab[z][co][col]  (D[row = co (lane>>4)*4+j][col = lane&15])
    float* out = slab_w + (long long)bidz * g.N * ktot;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = col0 + wave_n * 64 + c * 16 + (lane & 15);
            if (col >= ktot) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + wave_m * 64 + a * 16 + (lane >> 4) * 4 + j;
                if (co < g.N) out[(long long)co * ktot + col] = acc[a][c][j];
            }
        }
    if (do_bias && (lane & 15) == 0) {
        float* ob = slab_b + (long long)bidz * g.N;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + wave_m * 64 + a * 16 + (lane >> 4) * 4 + j;
                if (co < g.N) ob[co] = accb[a][j];
            }
    }
}

__global__ __launch_bounds__(WG) void k_conv_wgrad(const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ dy,
                                                   float* __restrict__ slab_w, float* __restrict__ slab_b, ConvGeom g,
                                                   int m_per_split) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv_wgrad_block(x, dy, slab_w, slab_b, g, m_per_split, smem, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several small layers' weight gradients in ONE launch (ssd_conv2d_bwd_weight_batched): block b belongs to the layer whose block
// range holds it and runs exactly the block of k_conv_wgrad it would have been there -- same slabs, same sums, bit for bit.
constexpr int WGB_MAX = 8;
struct WgradBatchItem {
    const bf16_raw* x;
    const bf16_raw* dy;
    float* slab_w;
    float* slab_b;
    ConvGeom g;
    int mps, ctiles, mtiles, blk0;
};
struct WgradBatchArgs {
    int count;
    WgradBatchItem it[WGB_MAX];
};
__global__ __launch_bounds__(WG) void k_conv_wgrad_batched(WgradBatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int l = 0;
#pragma unroll
    for (int k = 1; k < WGB_MAX; ++k) l += (k < a.count && (int)blockIdx.x >= a.it[k].blk0) ? 1 : 0;
    const WgradBatchItem& it = a.it[l];
    const int local = (int)blockIdx.x - it.blk0;
    const int per_split = it.ctiles * it.mtiles;
    const int bz = local / per_split, r = local - bz * per_split;
    const int by = r / it.ctiles, bx = r - by * it.ctiles;
    conv_wgrad_block(it.x, it.dy, it.slab_w, it.slab_b, it.g, it.mps, smem, bx, by, bz);
}

// ... and their slab sums in one launch: k_wgrad_reduce2's arithmetic per layer
struct ReduceBatchItem {
    const float* slab_w;
    const float* slab_b;
    float* dw;
    float* db;
    long long sw, nw, sb;
    int nb, ns, blk0;
    unsigned nbw;
};
struct ReduceBatchArgs {
    int count;
    ReduceBatchItem it[WGB_MAX];
};
__global__ __launch_bounds__(256) void k_wgrad_reduce2_batched(ReduceBatchArgs a) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < WGB_MAX; ++k) l += (k < a.count && (int)blockIdx.x >= a.it[k].blk0) ? 1 : 0;
    const ReduceBatchItem& it = a.it[l];
    const unsigned local = blockIdx.x - (unsigned)it.blk0;
    if (local < it.nbw) {
        const long long i = ((long long)local * 256 + threadIdx.x) * 4;
        if (i >= it.nw) return;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int z = 0; z < it.ns; ++z) {
            const float4 v = *reinterpret_cast<const float4*>(it.slab_w + (long long)z * it.sw + i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(it.dw + i) = s;
    } else {
        const int i = (int)(local - it.nbw) * 256 + threadIdx.x;
        if (i >= it.nb) return;
        float s = 0.f;
        for (int z = 0; z < it.ns; ++z) s += it.slab_b[(long long)z * it.sb + i];
        it.db[i] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient as a plain GEMM over pixels, 256 output channels x 256 (tap, ci) columns per workgroup, 64 pixels
// per step (the 1x1 and the strided layers with >= 256 channels; the 3x3 / stride-1 layers use the patch kernel below).
//   dW[co][col] += sum_m dY[m][co] * Xcol[m][col]
// Both tiles are [pixel][256 channels] images (512-byte rows) filled by buffer LDS-DMA (a lane whose pixel / column is
// padding gets an out-of-range offset = zeros) and read with transposing LDS reads.  The 32-byte column groups of a
// row are XORed with (row & 7): the eight consecutive rows of a half-wave read then hit eight bank groups, and since
// the key has period 8 every read address is a per-lane base + immediate.  Eight waves (2 x 4), each 128 co x 64 cols
// = 32 accumulator tiles, 64 MFMAs per step; two LDS buffers (128 KB), one barrier per step.
// Transposing LDS read whose access carries an alias scope (the __restrict__ pair, inlined).  The compiler's waitcnt pass
// orders an LDS read behind every in-flight LDS-DMA (s_waitcnt vmcnt(0)) unless the read has scope information; without it
// the first fragment read of a step waited for the NEXT tile's DMA issued just before, i.e. the double buffer never
// overlapped a transfer with the MFMAs.  The kernels order DMA and reads themselves (s_waitcnt vmcnt + barrier per step).
// (lds_read_tr16_scoped: defined in front of k_conv3x3_c64b)

constexpr int WT_TILE = 64 * 512;                          // one [64 px][256 ch] image
// STAGES = 2: two 64-pixel buffers, one step ahead, vmcnt(0) + barrier per step.  STAGES = 4 (round 4): four 32-pixel stages
// (one MFMA k-sub-step each), three stages of DMA in flight behind a COUNTED vmcnt -- the step no longer waits for the DMA it
// has just issued, and a stage's latency hides under three sub-steps of MFMAs instead of two.
template <int STAGES>
__global__ __launch_bounds__(512) void k_conv_wgrad_tile(const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ dy,
                                                         float* __restrict__ slab_w, float* __restrict__ slab_b, ConvGeom g,
                                                         int m_per_split, int nsplit, int cout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave & 1, wave_n = wave >> 1;           // 128 channels x 64 columns per wave
    // XCD-aware order when the splits are a multiple of 8: all tiles of one pixel split run consecutively on ONE XCD
    // (workgroup L -> XCD L % 8) and share its rows through that L2; otherwise plain order, which keeps every XCD busy
    const int ktot = g.ldw;
    const int ctiles = (ktot + 255) >> 8, mtiles = (cout + 255) >> 8, tiles = ctiles * mtiles;
    int split, tile;
    if ((nsplit & 7) == 0) {
        const int kx = blockIdx.x >> 3;
        split = (kx / tiles) * 8 + (blockIdx.x & 7);
        tile = kx % tiles;
    } else {
        split = blockIdx.x / tiles;
        tile = blockIdx.x - split * tiles;
    }
    if (split >= nsplit) return;
    const int bx = tile % ctiles, by = tile / ctiles;
    const int col0 = bx * 256, co0 = by * 256;
    const int m_begin = split * m_per_split;
    const int m_end = min(g.M, m_begin + m_per_split);

    // DMA: instruction i (= wave + 8j, j < 4) fills tile rows 2i, 2i+1; lane L -> row 2i + (L>>5), physical chunk L & 31
    const __amdgpu_buffer_rsrc_t dyres = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (unsigned)g.M * (unsigned)g.N * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)g.B * g.H * g.W * g.C * 2u, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    const int drow = lane >> 5;
    int rowj[4];
    unsigned dycol[4], xcol[4];                                // byte offset of the lane's chunk inside a pixel row, OOB if padding
    int xkh[4], xkw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 2 * (wave + 8 * j) + drow;
        rowj[j] = row;
        const int pc = lane & 31;                              // physical 16-byte chunk
        const int lg = ((pc >> 1) & 8) | (((pc >> 1) ^ row) & 7);   // logical 32-byte group
        const int ch = (lg * 2 + (pc & 1)) * 8;                // channel / column of the tile
        dycol[j] = co0 + ch < g.N ? (unsigned)(co0 + ch) * 2u : OOB;
        const int q = (col0 + ch) >> 3;
        if (q < g.nchunks) {
            const int tap = q / g.cpt;
            xcol[j] = (unsigned)(q - tap * g.cpt) * 16u;
            xkh[j] = tap / g.KW;
            xkw[j] = tap - xkh[j] * g.KW;
        } else {
            xcol[j] = OOB; xkh[j] = 0; xkw[j] = 0;
        }
    }
    const bool pointwise = g.KH == 1 && g.KW == 1 && g.mul == 1 && g.pad_t == 0 && g.pad_l == 0;   // source pixel = output pixel
    constexpr int NJ = STAGES == 4 ? 2 : 4;                      // DMA instructions per wave, operand and stage
    constexpr int XOFF = STAGES == 4 ? WT_TILE / 2 : WT_TILE;     // the x image of a stage starts here
    auto issue_dma = [&](int mstep, int buf) {
        char* base = smem + buf * (2 * XOFF);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int m = mstep + rowj[j];
            const bool mok = m < m_end;
            const unsigned od = (unsigned)m * (unsigned)g.N * 2u + dycol[j];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dyres, (lds_void*)(base + (wave + 8 * j) * 1024), 16,
                                                     (mok && dycol[j] != OOB) ? od : OOB, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int m = mstep + rowj[j];
            bool ok = m < m_end && xcol[j] != OOB;
            unsigned pix;
            if (pointwise) {
                pix = (unsigned)m;
            } else {
                const int mm = ok ? m : 0;
                const int b = fdiv(mm, g.d_hw);
                const int rem = mm - b * g.d_hw.d;
                const int oy = fdiv(rem, g.d_w);
                const int ox = rem - oy * g.d_w.d;
                const int iy = oy * g.mul - g.pad_t + xkh[j], ix = ox * g.mul - g.pad_l + xkw[j];
                ok = ok && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
                pix = (unsigned)((b * g.H + iy) * g.W + ix);
            }
            const unsigned ox_ = pix * (unsigned)g.C * 2u + xcol[j];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void*)(base + XOFF + (wave + 8 * j) * 1024), 16,
                                                     ok ? ox_ : OOB, 0, 0, 0);
        }
    };

    f32x4_t acc[8][4];
    f32x4_t accb[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        accb[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = slab_b != nullptr && bx == 0 && wave_n == 0;
    bf16x8_t ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;

    // MFMA k index <-> tile row: sub-step ksub, lane group gq, `half`: row = 32 ksub + 16 (gq>>1) + 8 half + 4 (gq&1) + (li>>2)
    const int gq = lane >> 4, li = lane & 15;
    const int kk0 = (gq >> 1) * 16 + (gq & 1) * 4 + (li >> 2);
    const int key = kk0 & 7;
    int abase[8], bbase[4];
#pragma unroll
    for (int a = 0; a < 8; ++a) abase[a] = kk0 * 512 + ((wave_m * 8 + (a ^ key)) << 5) + (li & 3) * 8;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int gl = wave_n * 4 + c;
        bbase[c] = XOFF + kk0 * 512 + (((gl & 8) | ((gl & 7) ^ key)) << 5) + (li & 3) * 8;
    }
    auto rd = [&](int addr) { return lds_read_tr16_scoped(smem + addr, smem); };

    const int nsteps = STAGES == 4 ? (m_end - m_begin + 31) / 32 : (m_end - m_begin + 63) / 64;
    if constexpr (STAGES == 4) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
            if (p < nsteps) issue_dma(m_begin + p * 32, p);
    } else {
        if (nsteps > 0) issue_dma(m_begin, 0);
    }
    auto run = [&](auto bias_tag) {
        constexpr bool BIAS = decltype(bias_tag)::value;
        for (int st = 0; st < nsteps; ++st) {
            if constexpr (STAGES == 4) {
                // stage st has landed when at most the two younger stages' 2 * NJ instructions each are still in flight
                const int younger = min(2, nsteps - 1 - st);
                if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                    // ... for every wave; and stage st - 1's buffer is free (raw barrier:
                asm volatile("" ::: "memory");                   //  __syncthreads() would wait for vmcnt(0) first)
                if (st + 3 < nsteps) issue_dma(m_begin + (st + 3) * 32, (st + 3) & 3);
                const int boff = (st & 3) * (2 * XOFF);
                bf16x8_t fb[4], fa[8];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int half = 0; half < 2; ++half)
                        reinterpret_cast<s16x4_t*>(&fb[c])[half] = rd(bbase[c] + boff + half * 4096);
#pragma unroll
                for (int a = 0; a < 8; ++a)
#pragma unroll
                    for (int half = 0; half < 2; ++half)
                        reinterpret_cast<s16x4_t*>(&fa[a])[half] = rd(abase[a] + boff + half * 4096);
#pragma unroll
                for (int a = 0; a < 8; ++a) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[c], acc[a][c], 0, 0, 0);
                    if constexpr (BIAS) accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], ones, accb[a], 0, 0, 0);
                }
                continue;
            }
            const int cur = st & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (st + 1 < nsteps) issue_dma(m_begin + (st + 1) * 64, cur ^ 1);
            const int boff = cur * (2 * WT_TILE);
            int ab[8], bb[4];
#pragma unroll
            for (int a = 0; a < 8; ++a) ab[a] = abase[a] + boff;
#pragma unroll
            for (int c = 0; c < 4; ++c) bb[c] = bbase[c] + boff;
#pragma unroll
            for (int ksub = 0; ksub < 2; ++ksub) {
                bf16x8_t fb[4], fa[8];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int half = 0; half < 2; ++half)
                        reinterpret_cast<s16x4_t*>(&fb[c])[half] = rd(bb[c] + ksub * 16384 + half * 4096);
#pragma unroll
                for (int a = 0; a < 8; ++a)
#pragma unroll
                    for (int half = 0; half < 2; ++half)
                        reinterpret_cast<s16x4_t*>(&fa[a])[half] = rd(ab[a] + ksub * 16384 + half * 4096);
#pragma unroll
                for (int a = 0; a < 8; ++a) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[c], acc[a][c], 0, 0, 0);
                    if constexpr (BIAS) accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], ones, accb[a], 0, 0, 0);
                }
            }
        }
    };
    if (do_bias) run(std::true_type{}); else run(std::false_type{});

    float* out = slab_w + (long long)split * g.N * ktot;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = col0 + wave_n * 64 + c * 16 + (lane & 15);
            if (col >= ktot) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + wave_m * 128 + a * 16 + (lane >> 4) * 4 + j;
                if (co < g.N) out[(long long)co * ktot + col] = acc[a][c][j];
            }
        }
    if (do_bias && (lane & 15) == 0) {
        float* ob = slab_b + (long long)split * g.N;
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + wave_m * 128 + a * 16 + (lane >> 4) * 4 + j;
                if (co < g.N) ob[co] = accb[a][j];
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of a 3x3 / stride 1 / pad 1 convolution with LDS-resident tiles ("patch" form).
// A workgroup owns 64 output channels x 64 input channels (one channel chunk) x all nine taps, and walks a range
// of 16x16 output-pixel blocks.  Per block it brings in the dY tile [256 px][64 co] and the 18x18 halo patch of X
// [324 px][64 ci] ONCE (LDS-DMA, double-buffered) and accumulates
//     dW[co][t][ci] += sum_px dY[px][co] * X[px + shift(t)][ci]          for the nine taps t
// with both operands fetched by transposing LDS reads (the patch at tap-shifted addresses).  The 144 accumulator
// tiles (4 co-tiles x 9 taps x 4 ci-tiles) are dealt to eight waves, 18 each (2 co-tiles x 9 taps x 1 ci-tile), so
// that all four SIMDs carry the same MFMA load (one wave per tap left one SIMD with 3 waves and the others with 2).
// ~128 MACs per byte brought into the CU, versus 32 for the generic 128x128 tile that re-stages X for every tap.
// Block geometry (template): BH output rows x 8*BW8 output columns.  The block's pixels are consumed as "pair groups"
// of 16 (two rows x eight columns); a k-step (32 pixels) takes two of them; an odd count is padded with an all-zero
// dY group.  Three shapes cover the SSD300 maps: 16x16 (300, 150, 75), 6x40 (38) and 10x24 (19).
template <int BH, int BW8>
struct WpGeom {
    static constexpr int BW = BW8 * 8;
    static constexpr int PW = BW + 2, PH = BH + 2;             // halo patch
    static constexpr int PPIX = PW * PH;
    static constexpr int P_INSTR = (PPIX * 8 + 63) / 64;       // one-KiB DMA instructions (8 pixels each)
    static constexpr int P_BYTES = P_INSTR * 1024;
    static constexpr int NPG = (BH / 2) * BW8;                 // pair groups
    static constexpr int KS = (NPG + 1) / 2;                   // k-steps per block
    static constexpr int DY_PIX = KS * 32;
    static constexpr int DY_BYTES = DY_PIX * 128;
    static constexpr int DY_INSTR = DY_PIX / 8;
    static constexpr int BUF = DY_BYTES + P_BYTES;             // one buffer: dY tile + X patch
    static_assert(BH % 2 == 0 && DY_INSTR % 8 == 0, "block shape");
};

// 32-byte column group permutation of a 128-byte pixel row.  A half-wave of a transposing read touches 8 CONSECUTIVE
// pixel rows (any alignment: taps shift them); (p & 1, (p >> 1) & 3) then takes all eight values, i.e. the eight
// 32-byte pieces fall into eight different bank groups.  The key has period 8 in p, which is what lets the reader
// keep eight per-lane base addresses and reach every (k-step, tap, half) with an immediate offset.
__device__ __forceinline__ int wp_key(int px) { return (px >> 1) & 3; }

template <int BH, int BW8>
__global__ __launch_bounds__(512) void k_conv3x3_wgrad_patch(const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ dy,
                                                             float* __restrict__ slab_w, float* __restrict__ slab_b,
                                                             ConvGeom g, int tiles_x, int tiles_y, int tiles_per_split,
                                                             int nsplit, int cout, int single_buf, int xg) {
    // g: source = x (B,H,W,C), destination = dy (Ho=H, Wo=W, N = ldy)
    using G = WpGeom<BH, BW8>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    // eight waves, two per SIMD (waves w and w + 4 share one), with UNEQUAL roles: the "loader" waves 0-3 issue all of
    // the next block's LDS-DMA (an LDS-DMA instruction holds its wave for 100-200 cycles, ~2700 cycles per block) and own
    // LTAPS = 4 taps x four co-tiles = 16 accumulator tiles of input-channel tile ct = w; their partners 4-7 issue no
    // DMA and own the other five taps = 20 tiles.  While a loader is stuck in its DMA issue the partner keeps the SIMD's MFMA
    // pipe busy; with equal shares and everybody issuing DMA both waves of a SIMD stalled together (measured: 459 us
    // with, 340 us without the DMA, same clock, the difference all in barrier waits).
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wave & 3, role = wave >> 2;              // role 0: loader, first taps; role 1: the rest
    // work units (split, co tile, ci chunk), chunk fastest.  XCD-aware order (workgroup L runs on XCD L % 8): `xg`
    // consecutive units -- the channel groups that walk the SAME pixel blocks -- sit on one XCD, so that a dY tile /
    // X patch is fetched into that L2 once instead of once per group (xg from the host: a divisor of the unit count
    // per split, or a multiple of it, that still leaves every XCD with work)
    const int nchunk = g.C >> 6, cotiles = (cout + 63) >> 6;
    const int nunits = nchunk * cotiles * nsplit;
    const int j = blockIdx.x >> 3;
    int id = ((j / xg) * 8 + (blockIdx.x & 7)) * xg + j % xg;
    if (id >= nunits) return;
    const int chunk = id % nchunk; id /= nchunk;
    const int cot = id % cotiles;
    const int split = id / cotiles;
    const int co0 = cot * 64, ci0 = chunk * 64;
    const int ntiles = g.B * tiles_x * tiles_y;
    const int t_begin = split * tiles_per_split, t_end = min(ntiles, t_begin + tiles_per_split);

    // (buffer-descriptor DMA: 32-bit byte offsets, a lane outside the map / the tile gets an out-of-range offset and the
    //  hardware writes zeros -- no 64-bit address arithmetic and no zero block)
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)g.B * g.H * g.W * g.C * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t dyres = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (unsigned)g.B * g.Ho * g.Wo * g.N * 2u, 0x00020000);
    constexpr unsigned WP_OOB = 0xfffffff0u;
    // DMA ownership: instruction i of the dY tile / of the patch goes to loader wave i % 4
    auto issue_dma = [&](int t, int buf) {
        int r = t;
        const int tx = r % tiles_x; r /= tiles_x;
        const int ty = r % tiles_y;
        const int b = r / tiles_y;
        const int y0 = ty * BH, x0 = tx * G::BW;
        char* base = smem + buf * G::BUF;
#pragma unroll
        for (int j = 0; j < G::DY_INSTR / 4; ++j) {
            const int i = ct + 4 * j;                       // 8 pixel slots x 128 B per instruction
            const int kk = 8 * i + (lane >> 3), sl = lane & 7;
            const int c16 = (((sl >> 1) ^ wp_key(kk)) << 1) | (sl & 1);
            const int pg = kk >> 4;                         // pair group -> (row pair, column group)
            const int rp = pg / BW8, xg = pg - rp * BW8;
            const int y = y0 + 2 * rp + ((kk >> 3) & 1), xx = x0 + xg * 8 + (kk & 7);
            const int co = co0 + c16 * 8;
            const bool ok = pg < G::NPG && y < g.Ho && xx < g.Wo && co < g.N;
            const unsigned off = ((unsigned)((b * g.Ho + y) * g.Wo + xx) * (unsigned)g.N + (unsigned)co) * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dyres, (lds_void*)(base + i * 1024), 16, ok ? off : WP_OOB, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < (G::P_INSTR + 3) / 4; ++j) {
            const int i = ct + 4 * j;
            if (i < G::P_INSTR) {
                const int pp = 8 * i + (lane >> 3), sl = lane & 7;
                const int c16 = (((sl >> 1) ^ wp_key(pp)) << 1) | (sl & 1);
                const int py = pp / G::PW, px = pp - py * G::PW;
                const int iy = y0 - 1 + py, ix = x0 - 1 + px;
                const bool ok = pp < G::PPIX && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
                const unsigned off = ((unsigned)((b * g.H + iy) * g.W + ix) * (unsigned)g.C + (unsigned)(ci0 + c16 * 8)) * 2u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void*)(base + G::DY_BYTES + i * 1024), 16, ok ? off : WP_OOB, 0, 0, 0);
            }
        }
    };

    constexpr int LTAPS = 4;                                // taps of a loader wave; its partner takes the other 9 - LTAPS (3 | 6 measured the same)
    f32x4_t acc[4][9 - LTAPS];                              // [co tile][tap of this role]
    f32x4_t accb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        accb[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t5 = 0; t5 < 9 - LTAPS; ++t5) acc[a][t5] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = slab_b != nullptr && chunk == 0 && ct == 0 && role == 1;
    bf16x8_t ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;

    // two LDS buffers: the next block's DMA is in flight during this block's MFMAs
    if (t_begin < t_end && role == 0) issue_dma(t_begin, 0);
    // MFMA k index <-> block pixel: k-step ks, `half` select pair group pg = 2ks + half = (row pair rp, column group xg);
    // within it lane group gq and the lane select row 2rp + (gq>>1), column 8xg + 4*(gq&1) + (li>>2), so that the 8 rows
    // of a half-wave instruction are consecutive pixels.  All address arithmetic is hoisted: the dY row of a lane is
    // base + immediate, and a patch row is one of eight per-lane bases (pixel offset mod 8) + immediate.
    const int gq = lane >> 4, li = lane & 15;
    int abase[4];
    {
        const int kk0 = (gq >> 1) * 8 + (gq & 1) * 4 + (li >> 2);
#pragma unroll
        for (int a = 0; a < 4; ++a) abase[a] = kk0 * 128 + ((a ^ wp_key(kk0)) << 5) + (li & 3) * 8;
    }
    int gbase[8];
    {
        const int p0 = (gq >> 1) * G::PW + (gq & 1) * 4 + (li >> 2);
#pragma unroll
        for (int r = 0; r < 8; ++r) gbase[r] = G::DY_BYTES + (p0 + r) * 128 + ((ct ^ wp_key(p0 + r)) << 5) + (li & 3) * 8;
    }
    typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
    // fragments of one k-step: 8 dY reads + 2 patch reads per tap (all with immediate offsets)
    struct Frag { bf16x8_t fa[4]; bf16x8_t fb[9 - LTAPS]; };
    // the block loop exists per role (and with / without the bias MFMAs) so that its body has no branch: the k-steps of
    // a block are one basic block, software-pipelined by hand (fragments of k-step ks+1 are read during the MFMAs of ks)
    auto run = [&](auto role_tag, auto bias_tag) {
        constexpr int ROLE = decltype(role_tag)::value;
        constexpr bool BIAS = decltype(bias_tag)::value;
        constexpr int TAP0 = ROLE == 0 ? 0 : LTAPS, NTAP = ROLE == 0 ? LTAPS : 9 - LTAPS;
        auto load_frag = [&](Frag& f, const int (&ab)[4], const int (&gb)[8], auto ks_tag) {
            constexpr int ks = decltype(ks_tag)::value;
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    reinterpret_cast<s16x4_t*>(&f.fa[a])[half] =
                        lds_read_tr16_scoped(smem + ab[a] + (ks * 2 + half) * 2048, smem);
#pragma unroll
            for (int t5 = 0; t5 < NTAP; ++t5)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int tap = TAP0 + t5;
                    const int pg = (ks * 2 + half) < G::NPG ? (ks * 2 + half) : 0;   // padding group: any finite data (dY is zero there)
                    const int rp = pg / BW8, xg = pg - rp * BW8;
                    const int ctap = (2 * rp + tap / 3) * G::PW + xg * 8 + (tap % 3);   // compile-time pixel offset
                    reinterpret_cast<s16x4_t*>(&f.fb[t5])[half] =
                        lds_read_tr16_scoped(smem + gb[ctap & 7] + (ctap >> 3) * 1024, smem);
                }
        };
        for (int t = t_begin; t < t_end; ++t) {
            const int cur = (t - t_begin) & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if constexpr (ROLE == 0) { if (t + 1 < t_end) issue_dma(t + 1, cur ^ 1); }
            const int boff = cur * G::BUF;
            int ab[4], gb[8];
#pragma unroll
            for (int a = 0; a < 4; ++a) ab[a] = abase[a] + boff;
#pragma unroll
            for (int r = 0; r < 8; ++r) gb[r] = gbase[r] + boff;
            Frag f0, f1;
            auto mma = [&](const Frag& f) {
#pragma unroll
                for (int t5 = 0; t5 < NTAP; ++t5)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        acc[a][t5] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.fa[a], f.fb[t5], acc[a][t5], 0, 0, 0);
                if constexpr (BIAS) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.fa[a], ones, accb[a], 0, 0, 0);
                }
            };
            // interleave: one MFMA, then one LDS read of the next k-step
            auto weave = [&]() {
                constexpr int NR = 8 + 2 * NTAP, NM = 4 * NTAP;          // reads of the next k-step, MFMAs of this one
                constexpr int PAIRS = NR < NM ? NR : NM;
                if constexpr (NR > PAIRS) {                             // more reads than MFMAs: the surplus goes first
                    __builtin_amdgcn_sched_group_barrier(0x100, NR - PAIRS, 0);
                }
#pragma unroll
                for (int i = 0; i < PAIRS; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                constexpr int REST = NM - PAIRS + (BIAS ? 4 : 0);
                if constexpr (REST > 0) __builtin_amdgcn_sched_group_barrier(0x008, REST, 0);
            };
            static_assert(G::KS == 8, "the hand-unrolled pipeline below assumes eight k-steps per block");
            load_frag(f0, ab, gb, std::integral_constant<int, 0>{});
            // the woven fragment-read / MFMA stream of a block runs at LOW priority, the top of the block (wait, barrier, the
            // loaders' DMA issue, address set-up, first fragments) at high: the two waves of a SIMD are in different phases for
            // most of a block (loader / partner), and the one in its MFMAs no longer holds the other one up: +3-5 % per layer
            __builtin_amdgcn_s_setprio(0);
            load_frag(f1, ab, gb, std::integral_constant<int, 1>{}); mma(f0); weave();
            load_frag(f0, ab, gb, std::integral_constant<int, 2>{}); mma(f1); weave();
            load_frag(f1, ab, gb, std::integral_constant<int, 3>{}); mma(f0); weave();
            load_frag(f0, ab, gb, std::integral_constant<int, 4>{}); mma(f1); weave();
            load_frag(f1, ab, gb, std::integral_constant<int, 5>{}); mma(f0); weave();
            load_frag(f0, ab, gb, std::integral_constant<int, 6>{}); mma(f1); weave();
            load_frag(f1, ab, gb, std::integral_constant<int, 7>{}); mma(f0); weave();
            mma(f1);
            __builtin_amdgcn_s_setprio(3);
        }
    };
    if (role == 0) run(std::integral_constant<int, 0>{}, std::false_type{});
    else if (do_bias) run(std::integral_constant<int, 1>{}, std::true_type{});
    else run(std::integral_constant<int, 1>{}, std::false_type{});
    // slab[split][co][tap][ci]  (dW layout [Cout][kh][kw][Cin], rows = ldy channels)
    const int ktot = g.ldw;
    float* out = slab_w + (long long)split * g.N * ktot;
    const int tap0 = role == 0 ? 0 : LTAPS, ntap = role == 0 ? LTAPS : 9 - LTAPS;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int t5 = 0; t5 < 9 - LTAPS; ++t5) {
            if (t5 >= ntap) continue;
            const int col = (tap0 + t5) * g.C + ci0 + ct * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + a * 16 + (lane >> 4) * 4 + j;
                if (co < g.N) out[(long long)co * ktot + col] = acc[a][t5][j];
            }
        }
    if (do_bias && (lane & 15) == 0) {
        float* ob = slab_b + (long long)split * g.N;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + a * 16 + (lane >> 4) * 4 + j;
                if (co < g.N) ob[co] = accb[a][j];
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the first layer (3x3 / stride 1 / pad 1, 8 input channels = 3 image channels + padding, <= 64
// output channels): dW[co][tap][ch] = sum_px dY[px][co] * X[px + shift(tap)][ch], 72 columns.  The work is reading dY
// once (HBM-bound); a workgroup walks 16x16-pixel blocks (dY tile 32 KB + an 18x18 x 16-byte halo patch, LDS-DMA,
// double-buffered), wave w multiplies k-step w (32 pixels) of every block: 4 channel tiles x 5 column tiles (a column
// tile = two taps x 8 channels) = 20 MFMAs; the eight partial sums are added in wave order at the end.
constexpr int W0_DY = 256 * 128;                           // dY tile bytes
constexpr int W0_PATCH = 6 * 1024;                         // 324 px x 16 B = 5184 B -> 6 DMA instructions
constexpr int W0_BUF = W0_DY + W0_PATCH;

__global__ __launch_bounds__(512) void k_conv0_wgrad(const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ dy,
                                                     float* __restrict__ slab_w, float* __restrict__ slab_b, ConvGeom g,
                                                     int tiles_x, int tiles_y, int tiles_per_split, int cout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x;
    const int ntiles = g.B * tiles_x * tiles_y;
    const int t_begin = split * tiles_per_split, t_end = min(ntiles, t_begin + tiles_per_split);

    auto issue_dma = [&](int t, int buf) {
        int r = t;
        const int tx = r % tiles_x; r /= tiles_x;
        const int ty = r % tiles_y;
        const int b = r / tiles_y;
        const int y0 = ty * 16, x0 = tx * 16;
        char* base = smem + buf * W0_BUF;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                       // dY: slot layout of k_conv3x3_wgrad_patch<16, 2>
            const int i = wave + 8 * j;
            const int kk = 8 * i + (lane >> 3), sl = lane & 7;
            const int c16 = (((sl >> 1) ^ wp_key(kk)) << 1) | (sl & 1);
            const int pg = kk >> 4;
            const int y = y0 + 2 * (pg >> 1) + ((kk >> 3) & 1), xx = x0 + (pg & 1) * 8 + (kk & 7);
            const int co = c16 * 8;
            const bool ok = y < g.Ho && xx < g.Wo && co < g.N;
            const bf16_raw* src = ok ? dy + ((unsigned)((b * g.Ho + y) * g.Wo + xx) * (unsigned)g.N + (unsigned)co)
                                     : reinterpret_cast<const bf16_raw*>(g_zero16);
            __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(base + i * 1024), 16, 0, 0);
        }
        if (wave < 6) {                                     // patch: one 16-byte pixel per lane, row-major 18x18
            const int pp = wave * 64 + lane;
            const int py = pp / PATCH_W, px = pp - py * PATCH_W;
            const int iy = y0 - 1 + py, ix = x0 - 1 + px;
            const bool ok = pp < PATCH_PIX && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
            const bf16_raw* src = ok ? x + (unsigned)((b * g.H + iy) * g.W + ix) * 8u : reinterpret_cast<const bf16_raw*>(g_zero16);
            __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(base + W0_DY + wave * 1024), 16, 0, 0);
        }
    };

    f32x4_t acc[4][5];
    f32x4_t accb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        accb[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 5; ++c) acc[a][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    bf16x8_t ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;

    // this wave's k-step: block rows 2w, 2w+1.  MFMA k index <-> pixel as in k_conv3x3_wgrad_patch
    const int gq = lane >> 4, li = lane & 15;
    const int kk0 = (gq >> 1) * 8 + (gq & 1) * 4 + (li >> 2);
    int abase[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) abase[a] = wave * 4096 + kk0 * 128 + ((a ^ wp_key(kk0)) << 5) + (li & 3) * 8;
    // patch pixel of tap (0,0) for half 0: row 2w + (gq>>1), column 4 (gq&1) + (li>>2); column tile t: lanes with
    // (li & 2) == 0 read tap 2t, the others tap 2t+1 (tap 9 does not exist: its columns are never stored)
    const int p0 = (2 * wave + (gq >> 1)) * PATCH_W + (gq & 1) * 4 + (li >> 2);
    int bbase[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        int tap = 2 * t + ((li >> 1) & 1);
        if (tap > 8) tap = 8;
        bbase[t] = W0_DY + (p0 + (tap / 3) * PATCH_W + tap % 3) * 16 + (li & 1) * 8;
    }
    typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
    auto rd = [&](int addr) { return lds_read_tr16_scoped(smem + addr, smem); };

    if (t_begin < t_end) issue_dma(t_begin, 0);
    for (int t = t_begin; t < t_end; ++t) {
        const int cur = (t - t_begin) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < t_end) issue_dma(t + 1, cur ^ 1);
        const int boff = cur * W0_BUF;
        bf16x8_t fa[4], fb[5];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int a = 0; a < 4; ++a) reinterpret_cast<s16x4_t*>(&fa[a])[half] = rd(boff + abase[a] + half * 2048);
#pragma unroll
            for (int c = 0; c < 5; ++c) reinterpret_cast<s16x4_t*>(&fb[c])[half] = rd(boff + bbase[c] + half * 8 * 16);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[c], acc[a][c], 0, 0, 0);
            accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], ones, accb[a], 0, 0, 0);
        }
    }
    // sum the eight waves' partial tiles in wave order (fixed order: reproducible), through LDS
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);            // [24 tiles][64 lanes][4]
    for (int w = 0; w < 8; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    f32x4_t v;
                    if (c < 5) v = acc[a][c < 5 ? c : 0]; else v = accb[a];
                    f32x4_t* slot = reinterpret_cast<f32x4_t*>(red + ((a * 6 + c) * 64 + lane) * 4);
                    if (w > 0) { const f32x4_t o = *slot; v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
                    *slot = v;
                }
            }
        }
        __syncthreads();
    }
    // slab[split][co][72]; tile (a, c): D row = co a*16 + (lane>>4)*4 + j, column c*16 + (lane&15)
    const int ktot = g.ldw;                                 // 72
    for (int idx = tid; idx < 24 * 64; idx += 512) {
        const int tile = idx >> 6, l = idx & 63;
        const int a = tile / 6, c = tile - a * 6;
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(red + idx * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = a * 16 + (l >> 4) * 4 + j;
            if (co >= g.N) continue;
            if (c < 5) {
                const int col = c * 16 + (l & 15);
                if (col < ktot) slab_w[((long long)split * g.N + co) * ktot + col] = v[j];
            } else if (slab_b != nullptr && (l & 15) == 0) {
                slab_b[(long long)split * g.N + co] = v[j];
            }
        }
    }
}

// weights and bias in one launch: blocks [0, nbw) reduce the first nw elements of the weight slab (stride sw per split)
// four at a time, the remaining blocks the nb bias elements (stride sb)
__global__ __launch_bounds__(256) void k_wgrad_reduce2(const float* __restrict__ slab_w, long long sw, long long nw,
                                                       float* __restrict__ dw, const float* __restrict__ slab_b, long long sb,
                                                       int nb, float* __restrict__ db, int nsplit, unsigned nbw) {
    if (blockIdx.x < nbw) {
        const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
        if (i >= nw) return;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int z = 0; z < nsplit; ++z) {
            const float4 v = *reinterpret_cast<const float4*>(slab_w + (long long)z * sw + i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(dw + i) = s;
    } else {
        const int i = (int)(blockIdx.x - nbw) * 256 + threadIdx.x;
        if (i >= nb) return;
        float s = 0.f;
        for (int z = 0; z < nsplit; ++z) s += slab_b[(long long)z * sb + i];
        db[i] = s;
    }
}

// many splits, few outputs (first layer: 512 splits of 4.6 K values): 16 threads per float4 of the output, thread g adds
// splits g, g+16, ... in order, then the 16 partial sums are added in order: fixed summation order, 16x the parallelism
__global__ __launch_bounds__(256) void k_wgrad_reduce_wide(const float* __restrict__ slab_w, long long sw, long long nw,
                                                          float* __restrict__ dw, const float* __restrict__ slab_b, long long sb,
                                                          int nb, float* __restrict__ db, int nsplit, unsigned nbw) {
    __shared__ float4 part[256];
    const int o = threadIdx.x & 15, grp = threadIdx.x >> 4;
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (blockIdx.x < nbw) {
        const long long i = ((long long)blockIdx.x * 16 + o) * 4;
        if (i < nw)
            for (int z = grp; z < nsplit; z += 16) {
                const float4 v = *reinterpret_cast<const float4*>(slab_w + (long long)z * sw + i);
                s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
            }
        part[threadIdx.x] = s4;
        __syncthreads();
        if (grp == 0 && i < nw) {
            float4 t = part[o];
            for (int k = 1; k < 16; ++k) { const float4 v = part[k * 16 + o]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
            *reinterpret_cast<float4*>(dw + i) = t;
        }
    } else {
        const int i = (int)(blockIdx.x - nbw) * 16 + o;
        float sacc = 0.f;
        if (i < nb)
            for (int z = grp; z < nsplit; z += 16) sacc += slab_b[(long long)z * sb + i];
        part[threadIdx.x].x = sacc;
        __syncthreads();
        if (grp == 0 && i < nb) {
            float t = part[o].x;
            for (int k = 1; k < 16; ++k) t += part[k * 16 + o].x;
            db[i] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// W[co][kh][kw][ci] (bf16) -> Wt[ci][KH-1-kh][KW-1-kw][co] for the data gradient
__global__ void k_weight_transpose(const bf16_raw* __restrict__ w, bf16_raw* __restrict__ wt, int Cout, int KH, int KW,
                                   int Cin, int Cout_pad) {
    // wt has row length KH*KW*Cout_pad (Cout padded to a multiple of 8 with zeros)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)Cin * KH * KW * Cout_pad;
    if (i >= total) return;
    const int co = (int)(i % Cout_pad);
    long long r = i / Cout_pad;
    const int kw = (int)(r % KW); r /= KW;
    const int kh = (int)(r % KH); r /= KH;
    const int ci = (int)r;
    bf16_raw v = 0;
    if (co < Cout) v = w[(((long long)co * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)) * Cin + ci];
    wt[i] = v;
}

// All transposed copies of a network in ONE launch: blockIdx.y = tensor, descriptor rows {src, dst, Cout, K, Cin, Cout_pad}
// (device array of int64), blockIdx.x = 32x32 (co, ci) tile x tap; the tile goes through LDS so that both the read (ci
// contiguous) and the write (co contiguous) are 64-byte runs.  33 launches of the per-tensor kernel cost 0.26 ms per step.
__global__ __launch_bounds__(256) void k_weight_transpose_batched(const long long* __restrict__ desc) {
    __shared__ bf16_raw tile[32][33];
    const long long* d = desc + (long long)blockIdx.y * 6;
    const bf16_raw* w = reinterpret_cast<const bf16_raw*>(d[0]);
    bf16_raw* wt = reinterpret_cast<bf16_raw*>(d[1]);
    // K field: kernel size in the low byte; bit 8 set = tap-major layout for the sparse head data gradient (sparse.hip):
    // wt[kh][kw][ci][co] = w[co][kh][kw][ci], not flipped
    const int Cout = (int)d[2], K = (int)d[3] & 0xff, tapmajor = ((int)d[3] >> 8) & 1, Cin = (int)d[4], Cout_pad = (int)d[5];
    const int tco = (Cout_pad + 31) >> 5, tci = (Cin + 31) >> 5;
    int t = blockIdx.x;
    if (t >= tco * tci * K * K) return;
    const int tap = t % (K * K); t /= K * K;
    const int ci0 = (t % tci) * 32, co0 = (t / tci) * 32;
    const int kh = tap / K, kw = tap - kh * K;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                 // 32 x 8
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = co0 + ty + 8 * j, ci = ci0 + tx;
        bf16_raw v = 0;
        const int skh = tapmajor ? kh : K - 1 - kh, skw = tapmajor ? kw : K - 1 - kw;
        if (co < Cout && ci < Cin) v = w[(((long long)co * K + skh) * K + skw) * Cin + ci];
        tile[ty + 8 * j][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ci = ci0 + ty + 8 * j, co = co0 + tx;
        if (ci >= Cin || co >= Cout_pad) continue;
        if (tapmajor) wt[(((long long)kh * K + kw) * Cin + ci) * Cout_pad + co] = tile[tx][ty + 8 * j];
        else wt[(((long long)ci * K + kh) * K + kw) * Cout_pad + co] = tile[tx][ty + 8 * j];
    }
}

// f32 -> bf16 cast (weights after an optimizer step)
__global__ void k_cast_bf16(const float* __restrict__ src, bf16_raw* __restrict__ dst, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = f2bf(src[i]);
}

// image f32 [B,H,W,3] in [0,1] -> bf16 [B,H,W,8], (x-0.5)*2 (models/ssd_model.py:214), channels 3..7 zero
__global__ void k_image_prep(const float* __restrict__ img, bf16_raw* __restrict__ out, long long npix, int normalize) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    float r = img[3 * i], gch = img[3 * i + 1], b = img[3 * i + 2];
    if (normalize) { r = (r - 0.5f) * 2.f; gch = (gch - 0.5f) * 2.f; b = (b - 0.5f) * 2.f; }
    *reinterpret_cast<uint4*>(out + 8 * i) =
        make_uint4(pack_bf16x2(r, gch), (unsigned)f2bf(b), 0u, 0u);
}

// 2x2 stride-2 max pooling, NHWC bf16, 8 channels per thread.  pad_b/pad_r = 1 for TF "SAME" on odd sizes.
__global__ void k_maxpool_fwd(const bf16_raw* __restrict__ x, bf16_raw* __restrict__ y, int B, int H, int W, int C,
                              int Ho, int Wo) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = C >> 3;
    const long long total = (long long)B * Ho * Wo * c8;
    if (i >= total) return;
    const int c = (int)(i % c8);
    long long r = i / c8;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float best[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) best[k] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int iy = 2 * oy + dy, ix = 2 * ox + dx;
            if (iy >= H || ix >= W) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(x + ((((long long)b * H + iy) * W + ix) * C + c * 8));
            const unsigned wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                best[2 * k] = fmaxf(best[2 * k], __uint_as_float(wds[k] << 16));
                best[2 * k + 1] = fmaxf(best[2 * k + 1], __uint_as_float(wds[k] & 0xffff0000u));
            }
        }
    unsigned o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (__float_as_uint(best[2 * k]) >> 16) | (__float_as_uint(best[2 * k + 1]) & 0xffff0000u);
    *reinterpret_cast<uint4*>(y + ((((long long)b * Ho + oy) * Wo + ox) * C + c * 8)) = make_uint4(o[0], o[1], o[2], o[3]);
}

// Backward of the pooling + the ReLU in front of it: dx = dy at the first maximum of each window (TF
// MaxPoolGrad), zero elsewhere and wherever x <= 0 (x is a post-ReLU activation).
__global__ void k_maxpool_bwd(const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ y, const bf16_raw* __restrict__ dy,
                              bf16_raw* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = C >> 3;
    const long long total = (long long)B * Ho * Wo * c8;
    if (i >= total) return;
    const int c = (int)(i % c8);
    long long r = i / c8;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const long long oidx = (((long long)b * Ho + oy) * Wo + ox) * C + c * 8;
    const uint4 yv = *reinterpret_cast<const uint4*>(y + oidx);
    const uint4 gv = *reinterpret_cast<const uint4*>(dy + oidx);
    const bf16_raw* yy = reinterpret_cast<const bf16_raw*>(&yv);
    const bf16_raw* gg = reinterpret_cast<const bf16_raw*>(&gv);
    bool done[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) done[k] = false;
#pragma unroll
    for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
        for (int dxx = 0; dxx < 2; ++dxx) {
            const int iy = 2 * oy + dyy, ix = 2 * ox + dxx;
            if (iy >= H || ix >= W) continue;
            const long long iidx = (((long long)b * H + iy) * W + ix) * C + c * 8;
            const uint4 xv = *reinterpret_cast<const uint4*>(x + iidx);
            const bf16_raw* xx = reinterpret_cast<const bf16_raw*>(&xv);
            bf16_raw o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool hit = !done[k] && xx[k] == yy[k];
                o[k] = (hit && bf2f(xx[k]) > 0.f) ? gg[k] : (bf16_raw)0;
                done[k] = done[k] || hit;
            }
            *reinterpret_cast<uint4*>(dx + iidx) = *reinterpret_cast<const uint4*>(o);
        }
}

// Pooling with a recorded winner: the forward pass also writes, per pooled element, a 4-bit code = position (2 dy + dx) of
// the first maximum of its window, or 4 if that maximum is <= 0 (post-ReLU input: no gradient flows).  The backward pass
// then needs only dy and the codes (1/4 byte per input element) instead of re-reading x and y: 0.97 GB instead of
// 1.84 GB for the first pool at batch 64.  Same routing rule as k_maxpool_bwd (TF MaxPoolGrad + ReLU mask).
__global__ void k_maxpool_fwd_argmax(const bf16_raw* __restrict__ x, bf16_raw* __restrict__ y, unsigned* __restrict__ code,
                                     int B, int H, int W, int C, int Ho, int Wo) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = C >> 3;
    if (i >= (long long)B * Ho * Wo * c8) return;
    const int c = (int)(i % c8);
    long long r = i / c8;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float best[8];
    unsigned pos[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; pos[k] = 4u; }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int iy = 2 * oy + dy, ix = 2 * ox + dx;
            if (iy >= H || ix >= W) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(x + ((((long long)b * H + iy) * W + ix) * C + c * 8));
            const unsigned wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float f = (k & 1) ? __uint_as_float(wds[k >> 1] & 0xffff0000u) : __uint_as_float(wds[k >> 1] << 16);
                if (f > best[k]) { best[k] = f; pos[k] = (unsigned)(2 * dy + dx); }   // strict: the first maximum wins
            }
        }
    unsigned o[4], cw = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (__float_as_uint(best[2 * k]) >> 16) | (__float_as_uint(best[2 * k + 1]) & 0xffff0000u);
#pragma unroll
    for (int k = 0; k < 8; ++k) cw |= (best[k] > 0.f ? pos[k] : 4u) << (4 * k);
    *reinterpret_cast<uint4*>(y + ((((long long)b * Ho + oy) * Wo + ox) * C + c * 8)) = make_uint4(o[0], o[1], o[2], o[3]);
    code[i] = cw;
}

__global__ void k_maxpool_bwd_argmax(const unsigned* __restrict__ code, const bf16_raw* __restrict__ dy, bf16_raw* __restrict__ dx,
                                     int B, int H, int W, int C, int Ho, int Wo) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = C >> 3;
    if (i >= (long long)B * Ho * Wo * c8) return;
    const int c = (int)(i % c8);
    long long r = i / c8;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const uint4 gv = *reinterpret_cast<const uint4*>(dy + ((((long long)b * Ho + oy) * Wo + ox) * C + c * 8));
    const unsigned g[4] = {gv.x, gv.y, gv.z, gv.w};
    const unsigned cw = code[i];
#pragma unroll
    for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
        for (int dxx = 0; dxx < 2; ++dxx) {
            const int iy = 2 * oy + dyy, ix = 2 * ox + dxx;
            if (iy >= H || ix >= W) continue;
            const unsigned p = (unsigned)(2 * dyy + dxx);
            unsigned o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned lo = ((cw >> (8 * k)) & 15u) == p ? 0x0000ffffu : 0u;
                const unsigned hi = ((cw >> (8 * k + 4)) & 15u) == p ? 0xffff0000u : 0u;
                o[k] = g[k] & (lo | hi);
            }
            *reinterpret_cast<uint4*>(dx + ((((long long)b * H + iy) * W + ix) * C + c * 8)) = make_uint4(o[0], o[1], o[2], o[3]);
        }
}

// dloc [B][A][4], dconf [B][A][classes] (bf16) -> one level's padded NHWC gradient [B][H*W][npad].  A pixel's row is the
// concatenation of its per_cell*4 loc values, its per_cell*classes conf values (both contiguous in the sources) and
// zero padding.  One thread per 16-byte chunk of the output (8 channels): the sources are only 2-byte aligned
// (classes = 81 is odd), so they are read element-wise (consecutive lanes -> consecutive addresses) and stored once.
__global__ void k_head_grad_pack(const bf16_raw* __restrict__ dloc, const bf16_raw* __restrict__ dconf,
                                 bf16_raw* __restrict__ out, int B, int hw, int per_cell, int classes, int npad,
                                 int anchors_total, int level_off) {
    const int cpr = npad >> 3;                                // chunks per row
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * hw * cpr) return;
    const int ch = (int)(i % cpr);
    const long long r = i / cpr;
    const int pix = (int)(r % hw);
    const int b = (int)(r / hw);
    const int n_loc = per_cell * 4, n_conf = per_cell * classes;
    const long long anchor0 = (long long)b * anchors_total + level_off + (long long)pix * per_cell;
    const bf16_raw* pl = dloc + anchor0 * 4;
    const bf16_raw* pc = dconf + anchor0 * classes - n_loc;
    if (!((per_cell | level_off | anchors_total) & 1)) {      // even anchor counts: every run 4-byte aligned, two channels per load
        unsigned w4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = ch * 8 + 2 * j;
            w4[j] = n < n_loc ? *reinterpret_cast<const unsigned*>(pl + n)
                              : (n < n_loc + n_conf ? *reinterpret_cast<const unsigned*>(pc + n) : 0u);
        }
        *reinterpret_cast<uint4*>(out + r * npad + ch * 8) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        return;
    }
    bf16_raw v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = ch * 8 + j;
        v[j] = n < n_loc ? pl[n] : (n < n_loc + n_conf ? pc[n] : (bf16_raw)0);
    }
    *reinterpret_cast<uint4*>(out + r * npad + ch * 8) =
        make_uint4((unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
                   (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16));
}

// Development overrides (ssd_dev_knob): A/B timing of kernel variants and forcing a dispatch path in tests, inside one
// process.  NOT configuration: the product never sets them, nothing is read from the environment, and results never
// depend on them (every variant computes the same convolution).  Values are relaxed atomics (-1 in the table = unset:
// the caller's default applies), so concurrent calls are safe.
struct Knob { const char* name; std::atomic<int> value; };
constexpr int KNOB_UNSET = INT_MIN;
Knob g_knobs[] = {{"SSD_ABLATE", {KNOB_UNSET}}, {"SSD_DGRAD_S2", {KNOB_UNSET}}, 
                  {"SSD_CONV_PATCH", {KNOB_UNSET}}, {"SSD_CONV_TILE", {KNOB_UNSET}}, {"SSD_SPLITK", {KNOB_UNSET}},
                  {"SSD_WGRAD_PATCH", {KNOB_UNSET}}, {"SSD_WGRAD_PATCH_SINGLE", {KNOB_UNSET}},
                  {"SSD_WGRAD_PATCH_SHAPE", {KNOB_UNSET}}, {"SSD_WGRAD_TILE", {KNOB_UNSET}},
                  {"SSD_CONV_PATCH_FLAT", {KNOB_UNSET}}, {"SSD_WGRAD_FIRST", {KNOB_UNSET}}, {"SSD_CONV_FIRST", {KNOB_UNSET}},
                  {"SSD_WGRAD_PATCH_XCD", {KNOB_UNSET}}, {"SSD_CONV_C64", {KNOB_UNSET}}, {"SSD_CONV_POOL_FUSE", {KNOB_UNSET}},
                  {"SSD_CONV_PATCH_ROWFLAT", {KNOB_UNSET}}, {"SSD_MATCH_FUSED", {KNOB_UNSET}}, {"SSD_CONV_P512", {KNOB_UNSET}},
                  {"SSD_C64B_WGS", {KNOB_UNSET}}, {"SSD_CONV_PW", {KNOB_UNSET}}, {"SSD_PW_WGS", {KNOB_UNSET}}, {"SSD_PW_DYNAMIC", {KNOB_UNSET}}, {"SSD_SP_ABLATE", {KNOB_UNSET}}, {"SSD_WGTILE_STAGES", {KNOB_UNSET}}, {"SSD_CHAIN_WAVES", {KNOB_UNSET}}, {"SSD_WGTILE_MIN_TILES", {KNOB_UNSET}}, {"SSD_CHAIN_TOUCH", {KNOB_UNSET}}, {"SSD_WGTILE_SLAB_X", {KNOB_UNSET}}, {"SSD_WGRAD_PATCH_WGS", {KNOB_UNSET}}};
Knob* find_knob(const char* name) {
    for (Knob& k : g_knobs) if (!strcmp(k.name, name)) return &k;
    return nullptr;
}
int knob(const char* name, int dflt) {
    Knob* k = find_knob(name);
    if (!k) return dflt;
    const int v = k->value.load(std::memory_order_relaxed);
    return v == KNOB_UNSET ? dflt : v;
}

// Which kernel a call resolves to (ssd_conv2d_*_plan): the dispatch code below runs as usual and, with `plan` set, records
// the id at the launch site and returns instead of launching -- the query cannot drift from the dispatch.
#define SSD_PLAN(ID_) do { if (plan) { *plan = (ID_); return SSD_OK; } } while (0)

template <int EPI>
bool staged_ok_host(const ConvGeom& g, const Epilogue& ep) {
    if (ep.slab) return false;
    if (EPI == EPI_HEAD) return true;
    return (g.N & 7) == 0 && (ep.ldo & 7) == 0;
}

template <int EPI>
int launch_igemm(const void* x, const void* w, const ConvGeom& g, const Epilogue& ep_in, hipStream_t s, void* ws = nullptr,
                 size_t ws_bytes = 0, bool* pooled = nullptr, int* plan = nullptr) {
    const bf16_raw* xp = static_cast<const bf16_raw*>(x);
    const bf16_raw* wp = static_cast<const bf16_raw*>(w);
    Epilogue ep = ep_in;
    ep.slab = nullptr;
    ep.ksplit = 1;
    if constexpr (EPI != EPI_HEAD) {
        if (knob("SSD_CONV_C64", 1) && g.KH == 3 && g.KW == 3 && g.mul == 1 && g.div == 1 && g.pad_t == 1 &&
            g.pad_l == 1 && g.C == 64 && g.N == 64 && g.ldw == 576 && g.H == g.Ho && g.W == g.Wo && g.H >= 16 && g.W >= 16 &&
            !ep.accumulate && !ep.up_out && (ep.ldo & 7) == 0 && (long long)g.B * g.H * g.W * 64 < (1ll << 31) - 16) {
            if (EPI == EPI_FWD && !ep.out && !(pooled && ep.pool_out)) return SSD_ERR_VALUE;
            // 8 x 16 blocks, two workgroups per CU
            const int tiles_x = (g.Wo + 15) / 16, tiles_y = (g.Ho + C64B_ROWS - 1) / C64B_ROWS;
            const int nblocks = g.B * tiles_x * tiles_y;
            auto kern = k_conv3x3_c64b<EPI, false>;
            SSD_PLAN(SSD_PLAN_C64B | ((pooled && ep.pool_out) ? SSD_PLAN_F_POOL_FUSED : 0));
            static OnceLds set; if (ensure_lds(set, reinterpret_cast<const void*>(kern), (int)(C64B_LDS)) != 0) return SSD_ERR_LAUNCH;
            const int maxwg = knob("SSD_C64B_WGS", 512);
            hipLaunchKernelGGL(kern, dim3((unsigned)(nblocks < maxwg ? nblocks : maxwg)), dim3(256), C64B_LDS, s, xp, wp, g, ep, tiles_x, tiles_y, C64W0{});
            if (pooled && ep.pool_out) *pooled = true;
            return ssd_launch_status();
        }
    }
    // LDS-patch kernel: 3x3 / stride 1 / pad 1 with N <= SSD_CONV_PATCH (default 256); beyond 128 channels only when
    // the 16x16 blocks waste little of the map (75x75 and larger: <= 14 %; 38x38 would waste 37 %)
    const int use_patch = knob("SSD_CONV_PATCH", 256);
    const bool patch_fits = g.N <= 128 || (long long)((g.Wo + 15) / 16) * ((g.Ho + 15) / 16) * 256 * 4 <= (long long)g.Wo * g.Ho * 5;
    // narrow maps (patch of 256 + 2 (W + 3) positions fits 32 KB): strip blocks, any channel count
    const int flat_knob = knob("SSD_CONV_PATCH_FLAT", 1);
    const bool use_flat = flat_knob && g.W <= 39 && g.W >= 16 && g.H >= 16 && (flat_knob >= 2 || !patch_fits || g.N > use_patch);
    if (g.KH == 3 && g.KW == 3 && g.mul == 1 && g.div == 1 && g.pad_t == 1 && g.pad_l == 1 &&
        g.C % 64 == 0 && g.H == g.Ho && g.W == g.Wo && ((g.N <= use_patch && patch_fits) || use_flat) && g.H >= 16 && g.W >= 16 &&
        (long long)g.B * g.H * g.W * g.C < (1ll << 31) - 16 && (long long)g.N * g.ldw < (1ll << 31) - 16) {   // 31-bit buffer offsets
        const int tiles_x = (g.Wo + 15) / 16, tiles_y = (g.Ho + 15) / 16;
        const unsigned gx = (unsigned)(tiles_x * tiles_y * g.B);
        {
            const bool flat = use_flat;
            // wide maps: one strip of rows over all images (k_conv3x3_patch32, "rowflat") when that needs fewer blocks and
            // the epilogue does not pool (2x2 windows would straddle blocks)
            const unsigned strip_rows = (unsigned)(((long long)g.B * (g.H + 1) + 15) / 16);
            const int rowflat = (!flat && !ep.pool_out && knob("SSD_CONV_PATCH_ROWFLAT", 1) &&
                                 strip_rows < (unsigned)(tiles_y * g.B)) ? 1 : 0;
            const unsigned gxx = flat ? (unsigned)(((long long)g.B * (g.H + 1) * (g.W + 2) + 255) / 256)
                                      : (rowflat ? strip_rows * (unsigned)tiles_x : gx);
#define SSD_LAUNCH_P32(BN_, FLAT_)                                                                                  \
            do {                                                                                                    \
                constexpr int lds_ = 2 * P32_PATCH + 2 * (BN_ == 64 ? 128 : BN_) * 64;   /* BN = 64 stages two taps per step */ \
                auto kern_ = k_conv3x3_patch32<BN_, EPI, FLAT_>;                                                    \
                SSD_PLAN((BN_ == 64 ? SSD_PLAN_P32_64 : SSD_PLAN_P32_128) | (FLAT_ ? SSD_PLAN_F_FLAT : 0) |            \
                         (rowflat ? SSD_PLAN_F_ROWFLAT : 0) | (can_pool ? SSD_PLAN_F_POOL_FUSED : 0));                \
                static OnceLds set_; if (ensure_lds(set_, reinterpret_cast<const void*>(kern_), (int)(lds_)) != 0) return SSD_ERR_LAUNCH; \
                const unsigned ntn_ = (unsigned)((g.N + BN_ - 1) / BN_);                                            \
                hipLaunchKernelGGL(kern_, dim3(8 * ntn_ * ((gxx + 7) / 8)), dim3(512), lds_, s, xp, wp, g, ep, tiles_x, tiles_y, (int)gxx, rowflat); \
            } while (0)
            const bool can_pool = pooled && ep.pool_out && !flat && (g.N & 7) == 0 && (ep.ldo & 7) == 0 && !(g.ablate & 8);
            if (EPI == EPI_FWD && !ep.out && !can_pool) return SSD_ERR_VALUE;
            if ((ep.up_out || ep.relu_bits || ep.mask_bits) && (!staged_ok_host<EPI>(g, ep) || (g.ablate & 8)))
                return SSD_ERR_UNSUPPORTED;                      // un-pooling and the sign bits live in the staged store
            // 512 px x 128 channels, one workgroup per CU: from 256 input channels on (eight 32-channel chunks amortise its longer
            // prologue / epilogue; measured per layer in DESIGN.md section 9).  SSD_CONV_P512: 0 never, 1 (default) that rule, 2 always
            const int p512 = knob("SSD_CONV_P512", 1);
            // (the heads never take it.  Head 0, 38x38 x 512 channels: its element-wise scatter epilogue has nothing to hide
            //  behind with one workgroup per CU, 385 vs 334 us.  Head 1, 19x19 x 1024 channels, is 208 vs 222 us ALONE -- but it
            //  runs beside the extras' chain of small launches, and a 512-pixel workgroup holds 152 of the CU's 160 KB of LDS:
            //  not one of those launches (64 KB each) found a CU until it had drained, 190 us for a 15-GFLOP layer and the
            //  loss 130 us later.  With the 78 KB workgroups of the patch kernel the two overlap: -0.07 ms per step.)
            if (p512 && g.C % 64 == 0 && g.N > 64 && (p512 >= 2 || (EPI != EPI_HEAD && g.C >= 256))) {
                const int ty32 = (g.Ho + 31) / 32;
                const unsigned strips32 = (unsigned)(((long long)g.B * (g.H + 1) + 31) / 32);
                const int rf = (!flat && !ep.pool_out && knob("SSD_CONV_PATCH_ROWFLAT", 1) && strips32 < (unsigned)(ty32 * g.B)) ? 1 : 0;
                const unsigned nb = flat ? (unsigned)(((long long)g.B * (g.H + 1) * (g.W + 2) + 511) / 512)
                                         : (rf ? strips32 * (unsigned)tiles_x : (unsigned)(tiles_x * ty32 * g.B));
                const unsigned ntn5 = (unsigned)((g.N + 127) / 128);
                SSD_PLAN(SSD_PLAN_P512 | (flat ? SSD_PLAN_F_FLAT : 0) | (rf ? SSD_PLAN_F_ROWFLAT : 0) | (can_pool ? SSD_PLAN_F_POOL_FUSED : 0));
                if (flat) {
                    auto kern5 = k_conv3x3_p512<EPI, true>;
                    static OnceLds set5; if (ensure_lds(set5, reinterpret_cast<const void*>(kern5), P5_LDS) != 0) return SSD_ERR_LAUNCH;
                    hipLaunchKernelGGL(kern5, dim3(8 * ntn5 * ((nb + 7) / 8)), dim3(512), P5_LDS, s, xp, wp, g, ep, tiles_x, ty32, (int)nb, rf);
                } else {
                    auto kern5 = k_conv3x3_p512<EPI, false>;
                    static OnceLds set5; if (ensure_lds(set5, reinterpret_cast<const void*>(kern5), P5_LDS) != 0) return SSD_ERR_LAUNCH;
                    hipLaunchKernelGGL(kern5, dim3(8 * ntn5 * ((nb + 7) / 8)), dim3(512), P5_LDS, s, xp, wp, g, ep, tiles_x, ty32, (int)nb, rf);
                }
                if (can_pool) *pooled = true;
                return ssd_launch_status();
            }
            if (g.N <= 64) { if (flat) SSD_LAUNCH_P32(64, true); else SSD_LAUNCH_P32(64, false); }
            else { if (flat) SSD_LAUNCH_P32(128, true); else SSD_LAUNCH_P32(128, false); }
#undef SSD_LAUNCH_P32
            if (can_pool) *pooled = true;
            return ssd_launch_status();
        }
    }
    if (EPI == EPI_FWD && !ep.out) return SSD_ERR_VALUE;         // pool-only needs a pooling kernel
    if (ep.up_out) return SSD_ERR_UNSUPPORTED;                   // only the LDS-patch kernels un-pool in their epilogue
    if constexpr (EPI != EPI_HEAD) {
        // 1x1 / stride 1 on a large map: the persistent GEMM (pwgemm.hip) -- the DMA stream runs on across tile boundaries
        // (SSD_CONV_PW: bit 0 forward, bit 1 data gradient)
        if ((knob("SSD_CONV_PW", 3) & (EPI == EPI_FWD ? 1 : 2)) && !(g.ablate & 8) && ssd_pw_gemm_serves(EPI, &g, &ep)) {
            SSD_PLAN(SSD_PLAN_PW);
            return ssd_pw_gemm_launch(EPI, x, w, &g, &ep, ws, ws_bytes, s);
        }
    }
    {
        // tile choice: the CU ingests ~28 B/clk from L2, so MACs per staged byte decide the ceiling: prefer the
        // largest tile that still gives every CU work (>= ~2 workgroups per CU), SSD_CONV_TILE overrides (testing)
        const int force = knob("SSD_CONV_TILE", 0);
        const long long wg_128 = (long long)((g.M + 127) / 128);
        const long long wg_256 = (long long)((g.M + 255) / 256);
        int bm = 128, bn = g.N <= 64 ? 64 : 128;
        const long long pad256 = (long long)((g.N + 255) / 256) * 256, pad128 = (long long)((g.N + 127) / 128) * 128;
        // (>= 352 rather than 384 workgroups: at 364 -- the 19x19 maps with 1024 channels -- the 8-phase kernel's deeper
        //  pipeline beats the 256x128 kernel's better fill by 12 %.  A third / fourth LDS stage in k_conv_igemm_dma itself was
        //  tried for the skinny layers and was neutral to 60 % slower: not the DMA latency bounds them)
        if (g.N > 128 && wg_256 * ((g.N + 255) / 256) >= 352 && pad256 * 4 <= pad128 * 5) { bm = 256; bn = 256; }
        else if (g.N > 64 && wg_256 * ((g.N + 127) / 128) >= 384) { bm = 256; bn = 128; }
        else if (g.N <= 64 && wg_256 >= 384) { bm = 256; bn = 64; }
        if (force == 1) { bm = 128; bn = g.N <= 64 ? 64 : 128; }
        if ((force == 2 || force == 4) && g.N > 128) { bm = 256; bn = 256; }   // (4: the LDS-DMA kernel instead of the 8-phase one)
        if (force == 3 && g.N > 64) { bm = 256; bn = 128; }
        (void)wg_128;
        // split-K for skinny problems (few tiles, long k loop): partial sums to the caller's workspace
        unsigned ksplit = 1;
        {
            const int sk_on = knob("SSD_SPLITK", 1);
            const long long tiles = (long long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn);
            const int nks_all = g.s2 ? 0 : (g.nchunks + 7) / 8;
            if (sk_on && ws && tiles < 160 && nks_all >= 8) {
                long long want = (256 + tiles - 1) / tiles;
                if (want > nks_all / 2) want = nks_all / 2;
                if (want > 32) want = 32;
                while (want > 1 && (size_t)want * g.M * g.N * sizeof(float) > ws_bytes) --want;
                if (want > 1) { ksplit = (unsigned)want; ep.slab = static_cast<float*>(ws); ep.ksplit = (int)want; }
            }
        }
        // the sign bits are written / read by the staged store only (not by the split-K finalize or the scattered epilogue)
        if ((ep.relu_bits || ep.mask_bits) && (ksplit > 1 || !staged_ok_host<EPI>(g, ep) || (g.ablate & 8))) return SSD_ERR_UNSUPPORTED;
#define SSD_LAUNCH_DMA(BM_, BN_)                                                                                   \
        do {                                                                                                       \
            constexpr int PT_ = (BM_ == 256 && BN_ == 256) ? SSD_PT256 : 4;                                         \
            constexpr int NT_ = (BM_ / (16 * PT_)) * (BN_ >= 128 ? BN_ / 64 : 2) * 64;                              \
            const size_t lds_ = 2 * (BM_ + BN_) * 128;                                                             \
            auto kern_ = k_conv_igemm_dma<BM_, BN_, EPI, PT_>;                                                     \
            SSD_PLAN((BM_ == 256 ? (BN_ == 256 ? SSD_PLAN_DMA_256_256 : (BN_ == 128 ? SSD_PLAN_DMA_256_128 : SSD_PLAN_DMA_256_64)) \
                                 : (BN_ == 64 ? SSD_PLAN_DMA_128_64 : SSD_PLAN_DMA_128_128)) |                       \
                     (ksplit > 1 ? SSD_PLAN_F_SPLITK : 0) | (g.s2 ? SSD_PLAN_F_S2 : 0));                              \
            if (lds_ > 65536) {                                                                                    \
                static OnceLds set_; if (ensure_lds(set_, reinterpret_cast<const void*>(kern_), (int)((int)lds_)) != 0) return SSD_ERR_LAUNCH; \
            }                                                                                                      \
            unsigned ntm_ = (unsigned)((g.M + BM_ - 1) / BM_);                                                      \
            const unsigned ntn_ = (unsigned)((g.N + BN_ - 1) / BN_);                                               \
            if (g.s2) { ntm_ = 0; for (int c_ = 0; c_ < 4; ++c_) ntm_ += (unsigned)((g.cls_n[c_] + BM_ - 1) / BM_); } \
            hipLaunchKernelGGL(kern_, dim3(8 * ntn_ * ((ntm_ + 7) / 8), ksplit), dim3(NT_), lds_, s, xp, wp, g, ep); \
        } while (0)
        if (bm == 256 && bn == 256 && force != 4 && !g.s2 && ksplit == 1 && g.cpt >= 8 &&
            (long long)g.B * g.H * g.W * g.C < (1ll << 31) - 16 && (long long)g.N * g.ldw < (1ll << 31) - 16) {
            const unsigned ntm = (unsigned)((g.M + 255) / 256), ntn = (unsigned)((g.N + 255) / 256);
#define SSD_LAUNCH_8PH(ABL_)                                                                                        \
            do {                                                                                                    \
                auto kern = k_conv_igemm_8ph<EPI, ABL_>;                                                            \
                SSD_PLAN(SSD_PLAN_8PH);                                                                             \
                static OnceLds set; if (ensure_lds(set, reinterpret_cast<const void*>(kern), (int)(131072)) != 0) return SSD_ERR_LAUNCH; \
                hipLaunchKernelGGL(kern, dim3(8 * ntn * ((ntm + 7) / 8)), dim3(512), 131072, s, xp, wp, g, ep);      \
            } while (0)
#ifdef SSD_DEV_ABLATE
            if constexpr (EPI == EPI_FWD) {
                switch (g.ablate) {
                    case 1: SSD_LAUNCH_8PH(1); break;
                    case 4: SSD_LAUNCH_8PH(4); break;
                    case 5: SSD_LAUNCH_8PH(5); break;
                    case 20: SSD_LAUNCH_8PH(20); break;
                    case 21: SSD_LAUNCH_8PH(21); break;
                    case 23: SSD_LAUNCH_8PH(23); break;
                    default: SSD_LAUNCH_8PH(0); break;
                }
            } else
#endif
            SSD_LAUNCH_8PH(0);
#undef SSD_LAUNCH_8PH
        }
        else if (bm == 256 && bn == 256) SSD_LAUNCH_DMA(256, 256);
        else if (bm == 256 && bn == 128) SSD_LAUNCH_DMA(256, 128);
        else if (bm == 256 && bn == 64) SSD_LAUNCH_DMA(256, 64);
        else if (bn == 64) SSD_LAUNCH_DMA(128, 64);
        else SSD_LAUNCH_DMA(128, 128);
#undef SSD_LAUNCH_DMA
        if (ksplit > 1) {
            if (hipGetLastError() != hipSuccess) return SSD_ERR_LAUNCH;
            const long long nthr = (long long)g.M * ((g.N + 3) / 4);
            hipLaunchKernelGGL(k_igemm_finalize<EPI>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, g, ep);
        }
        return ssd_launch_status();
    }
}

bool geom_ok(int B, int H, int W, int C, int Ho, int Wo, int N, int K) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || N <= 0 || K <= 0) return false;
    if (C % 8) return false;
    if ((long long)B * Ho * Wo >= (1ll << 31) || (long long)B * H * W >= (1ll << 31)) return false;
    if ((long long)B * H * W * C >= (1ll << 32) || (long long)B * Ho * Wo * N >= (1ll << 32)) return false;   // 32-bit element offsets
    return true;
}

}  // namespace

int ssd_knob(const char* name, int dflt) { return knob(name, dflt); }

// Optional second stream for the slab reductions (ssd_set_wgrad_reduce_stream): per calling thread.  The reduction is ordered
// behind the slab kernel by an event from a small ring (events are reused: a recorded-and-waited event can be re-recorded).
static thread_local hipStream_t t_reduce_stream = nullptr;
static thread_local hipEvent_t t_reduce_events[64];
static thread_local int t_reduce_event_next = 0, t_reduce_events_made = 0;

void ssd_launch_wgrad_reduce(hipStream_t s, const float* slab_w, long long sw, long long nw, float* dw, const float* slab_b,
                             long long sb, int nb, float* db, int ns) {
    if (t_reduce_stream && t_reduce_stream != s) {
        if (t_reduce_events_made < 64) {
            for (; t_reduce_events_made < 64; ++t_reduce_events_made)
                if (hipEventCreateWithFlags(&t_reduce_events[t_reduce_events_made], hipEventDisableTiming) != hipSuccess) break;
        }
        if (t_reduce_events_made == 64) {
            hipEvent_t ev = t_reduce_events[t_reduce_event_next];
            t_reduce_event_next = (t_reduce_event_next + 1) & 63;
            if (hipEventRecord(ev, s) == hipSuccess && hipStreamWaitEvent(t_reduce_stream, ev, 0) == hipSuccess) s = t_reduce_stream;
        }
    }
    if (ns >= 32) {
        const unsigned nbw = (unsigned)((nw / 4 + 15) / 16), nbb = db ? (unsigned)((nb + 15) / 16) : 0u;
        hipLaunchKernelGGL(k_wgrad_reduce_wide, dim3(nbw + nbb), dim3(256), 0, s, slab_w, sw, nw, dw, slab_b, sb, nb, db, ns, nbw);
    } else {
        const unsigned nbw = (unsigned)((nw / 4 + 255) / 256), nbb = db ? (unsigned)((nb + 255) / 256) : 0u;
        hipLaunchKernelGGL(k_wgrad_reduce2, dim3(nbw + nbb), dim3(256), 0, s, slab_w, sw, nw, dw, slab_b, sb, nb, db, ns, nbw);
    }
}

extern "C" {

int ssd_set_wgrad_reduce_stream(void* stream) {
    t_reduce_stream = (hipStream_t)stream;
    return SSD_OK;
}

int ssd_dev_knob(const char* name, int value) {
    if (!name) return SSD_ERR_VALUE;
    Knob* k = find_knob(name);
    if (!k) return SSD_ERR_VALUE;
    k->value.store(value, std::memory_order_relaxed);
    return SSD_OK;
}

static int conv2d_fwd_impl(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin, int Cout,
                           int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo, int relu, void* ws, size_t ws_bytes,
                           void* stream, int* plan, void* relu_bits = nullptr) {
    if (!x || !w || !y || !geom_ok(B, H, W, Cin, Ho, Wo, Cout, ksize) || stride <= 0) return SSD_ERR_VALUE;
    const ConvGeom g = make_geom(B, H, W, Cin, Ho, Wo, Cout, ksize, ksize, stride, 1, pad_t, pad_l);
    if (knob("SSD_CONV_FIRST", 1) && Cin == 8 && Cout == 64 && ksize == 3 && stride == 1 && pad_t == 1 && pad_l == 1 && H == Ho &&
        W == Wo && H >= 16 && W >= 16) {                     // the image layer
        SSD_PLAN(SSD_PLAN_CONV0_FWD);
        const int tx = (Wo + 15) / 16, ty = (Ho + 15) / 16;
        hipLaunchKernelGGL(k_conv0_fwd, dim3((unsigned)(B * tx * ty < 768 ? B * tx * ty : 768)), dim3(512), 0, (hipStream_t)stream,
                           static_cast<const bf16_raw*>(x), static_cast<const bf16_raw*>(w), bias, static_cast<bf16_raw*>(y), g, relu,
                           tx, ty, static_cast<unsigned char*>(relu_bits));
        return ssd_launch_status();
    }
    Epilogue ep = {};
    ep.bias = bias; ep.relu = relu; ep.out = static_cast<bf16_raw*>(y); ep.ldo = Cout;
    ep.relu_bits = static_cast<unsigned char*>(relu_bits);
    return launch_igemm<EPI_FWD>(x, w, g, ep, (hipStream_t)stream, ws, ws_bytes, nullptr, plan);
}

int ssd_conv2d_fwd(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin, int Cout,
                   int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo, int relu, void* ws, size_t ws_bytes,
                   void* stream) {
    return conv2d_fwd_impl(x, w, bias, y, B, H, W, Cin, Cout, ksize, stride, pad_t, pad_l, Ho, Wo, relu, ws, ws_bytes, stream,
                           nullptr);
}

static int conv2d_fwd_pool_impl(const void* x, const void* w, const float* bias, void* y, void* y_pool, void* pool_code, int B,
                                int H, int W, int Cin, int Cout, int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo,
                                int relu, int Hp, int Wp, void* ws, size_t ws_bytes, void* stream, int* plan) {
    if (!x || !w || !y_pool || !pool_code || !geom_ok(B, H, W, Cin, Ho, Wo, Cout, ksize) || stride <= 0 || Cout % 8)
        return SSD_ERR_VALUE;
    // y == NULL: the caller has no use for the full-resolution map (nothing but the pooling reads it): served only by the
    // kernels that pool in their epilogue, SSD_ERR_VALUE (before anything is launched) for any other layer shape
    if ((Hp != Ho / 2 && Hp != (Ho + 1) / 2) || (Wp != Wo / 2 && Wp != (Wo + 1) / 2) || Hp <= 0 || Wp <= 0) return SSD_ERR_VALUE;
    const ConvGeom g = make_geom(B, H, W, Cin, Ho, Wo, Cout, ksize, ksize, stride, 1, pad_t, pad_l);
    Epilogue ep = {};
    ep.bias = bias; ep.relu = relu; ep.out = static_cast<bf16_raw*>(y); ep.ldo = Cout;
    if (!y && !knob("SSD_CONV_POOL_FUSE", 1)) return SSD_ERR_VALUE;
    if (knob("SSD_CONV_POOL_FUSE", 1)) {
        ep.pool_out = static_cast<bf16_raw*>(y_pool); ep.pool_code = static_cast<unsigned*>(pool_code); ep.pool_h = Hp; ep.pool_w = Wp;
    }
    bool pooled = false;
    const int rc = launch_igemm<EPI_FWD>(x, w, g, ep, (hipStream_t)stream, ws, ws_bytes, &pooled, plan);
    if (rc != SSD_OK || pooled || plan) return rc;
    // this layer is not served by a 16x16-block kernel: pool in a second launch
    return ssd_maxpool2x2_fwd_argmax(y, y_pool, pool_code, B, Ho, Wo, Cout, Hp, Wp, stream);
}

int ssd_conv2d_fwd_pool(const void* x, const void* w, const float* bias, void* y, void* y_pool, void* pool_code, int B, int H,
                        int W, int Cin, int Cout, int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo, int relu, int Hp,
                        int Wp, void* ws, size_t ws_bytes, void* stream) {
    return conv2d_fwd_pool_impl(x, w, bias, y, y_pool, pool_code, B, H, W, Cin, Cout, ksize, stride, pad_t, pad_l, Ho, Wo, relu,
                                Hp, Wp, ws, ws_bytes, stream, nullptr);
}

static int conv2d_head_fwd_impl(const void* x, const void* w, const float* bias, void* loc, void* conf, int B, int H, int W,
                                int Cin, int per_cell, int classes, int anchors_total, int level_off, void* ws, size_t ws_bytes,
                                void* stream, int* plan) {
    const int N = per_cell * (4 + classes);
    if (!x || !w || !loc || !conf || !geom_ok(B, H, W, Cin, H, W, N, 3) || per_cell <= 0 || classes <= 0) return SSD_ERR_VALUE;
    const ConvGeom g = make_geom(B, H, W, Cin, H, W, N, 3, 3, 1, 1, 1, 1);   // 3x3 SAME stride 1 (models/ssd_model.py:155-162)
    Epilogue ep = {};
    ep.bias = bias; ep.loc = static_cast<bf16_raw*>(loc); ep.conf = static_cast<bf16_raw*>(conf);
    ep.n_loc = per_cell * 4; ep.n_conf = per_cell * classes; ep.anchors_total = anchors_total;
    ep.level_off = level_off; ep.per_cell = per_cell; ep.classes = classes;
    return launch_igemm<EPI_HEAD>(x, w, g, ep, (hipStream_t)stream, ws, ws_bytes, nullptr, plan);
}

int ssd_conv2d_head_fwd(const void* x, const void* w, const float* bias, void* loc, void* conf, int B, int H, int W,
                        int Cin, int per_cell, int classes, int anchors_total, int level_off, void* ws, size_t ws_bytes,
                        void* stream) {
    return conv2d_head_fwd_impl(x, w, bias, loc, conf, B, H, W, Cin, per_cell, classes, anchors_total, level_off, ws, ws_bytes,
                                stream, nullptr);
}

static int conv2d_bwd_data_impl(const void* dy, const void* w_t, const void* relu_src, void* dx, int B, int H, int W, int Cin,
                                int Cout_pad, int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo, int accumulate,
                                void* ws, size_t ws_bytes, void* stream, int* plan, const void* up_code = nullptr,
                                void* up_dx = nullptr, int up_h = 0, int up_w = 0, const void* mask_bits = nullptr) {
    // dy: [B,Ho,Wo,Cout_pad]; w_t: [Cin][k][k][Cout_pad] (ssd_weight_transpose); dx, relu_src: [B,H,W,Cin]
    if (!dy || !w_t || (!dx && !up_dx) || !geom_ok(B, Ho, Wo, Cout_pad, H, W, Cin, ksize) || stride <= 0) return SSD_ERR_VALUE;
    const ConvGeom g = make_geom(B, Ho, Wo, Cout_pad, H, W, Cin, ksize, ksize, 1, stride, ksize - 1 - pad_t,
                                 ksize - 1 - pad_l);
    Epilogue ep = {};
    ep.out = static_cast<bf16_raw*>(dx); ep.ldo = Cin; ep.mask_src = static_cast<const bf16_raw*>(relu_src);
    ep.accumulate = accumulate;
    ep.up_code = static_cast<const unsigned*>(up_code); ep.up_out = static_cast<bf16_raw*>(up_dx); ep.up_h = up_h; ep.up_w = up_w;
    ep.mask_bits = static_cast<const unsigned char*>(mask_bits);
    return launch_igemm<EPI_DGRAD>(dy, w_t, g, ep, (hipStream_t)stream, ws, ws_bytes, nullptr, plan);
}

int ssd_conv2d_fwd_relubits(const void* x, const void* w, const float* bias, void* y, void* relu_bits, int B, int H, int W, int Cin,
                            int Cout, int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo, void* ws, size_t ws_bytes,
                            void* stream) {
    // ssd_conv2d_fwd with ReLU that also writes the sign bits of its output: relu_bits [B*Ho*Wo][Cout/8] bytes
    if (!relu_bits || Cout % 8) return SSD_ERR_VALUE;
    return conv2d_fwd_impl(x, w, bias, y, B, H, W, Cin, Cout, ksize, stride, pad_t, pad_l, Ho, Wo, 1, ws, ws_bytes, stream, nullptr,
                           relu_bits);
}

int ssd_conv2d_bwd_data_bits(const void* dy, const void* w_t, const void* relu_bits, void* dx, int B, int H, int W, int Cin,
                             int Cout_pad, int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo, int accumulate, void* ws,
                             size_t ws_bytes, void* stream) {
    // ssd_conv2d_bwd_data with the ReLU mask given as sign bits ([B*H*W][Cin/8] bytes of ssd_conv2d_fwd_relubits)
    if (!relu_bits || Cin % 8) return SSD_ERR_VALUE;
    return conv2d_bwd_data_impl(dy, w_t, nullptr, dx, B, H, W, Cin, Cout_pad, ksize, stride, pad_t, pad_l, Ho, Wo, accumulate, ws,
                                ws_bytes, stream, nullptr, nullptr, nullptr, 0, 0, relu_bits);
}

int ssd_conv2d_bwd_data_unpool(const void* dy, const void* w_t, const void* relu_src, const void* pool_code, void* dx_pooled,
                               void* dx_full, int B, int H, int W, int Cin, int Cout_pad, int Hf, int Wf, void* ws, size_t ws_bytes,
                               void* stream) {
    // 3x3 / stride 1 / pad 1 data gradient w.r.t. a POOLED map [B,H,W,Cin], carried on through the 2x2 / stride-2 max pooling
    // that produced the map (pool_code [B,H,W,Cin/8] of ssd_maxpool2x2_fwd_argmax / ssd_conv2d_fwd_pool): dx_full [B,Hf,Wf,Cin]
    if (!pool_code || !dx_full || Cin % 8 || Hf <= 0 || Wf <= 0) return SSD_ERR_VALUE;
    if ((H != Hf / 2 && H != (Hf + 1) / 2) || (W != Wf / 2 && W != (Wf + 1) / 2)) return SSD_ERR_VALUE;
    if (2 * H < Hf || 2 * W < Wf) return SSD_ERR_UNSUPPORTED;     // VALID pooling of an odd size: the uncovered row / column is not written here
    if ((long long)B * Hf * Wf * Cin >= (1ll << 32)) return SSD_ERR_VALUE;
    return conv2d_bwd_data_impl(dy, w_t, relu_src, dx_pooled, B, H, W, Cin, Cout_pad, 3, 1, 1, 1, H, W, 0, ws, ws_bytes, stream, nullptr,
                                pool_code, dx_full, Hf, Wf);
}

int ssd_conv2d_bwd_data(const void* dy, const void* w_t, const void* relu_src, void* dx, int B, int H, int W, int Cin,
                        int Cout_pad, int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo, int accumulate,
                        void* ws, size_t ws_bytes, void* stream) {
    return conv2d_bwd_data_impl(dy, w_t, relu_src, dx, B, H, W, Cin, Cout_pad, ksize, stride, pad_t, pad_l, Ho, Wo, accumulate,
                                ws, ws_bytes, stream, nullptr);
}

static int wgrad_patch_min_hw() {           // SSD_WGRAD_PATCH = smallest feature-map side served by the patch kernel (0: off)
    return knob("SSD_WGRAD_PATCH", 16);
}

// dW / dbias = sum over splits of the slabs, fixed order
static void launch_wgrad_reduce(hipStream_t s, const float* slab_w, long long sw, long long nw, float* dw, const float* slab_b,
                                long long sb, int nb, float* db, int ns) {
    ssd_launch_wgrad_reduce(s, slab_w, sw, nw, dw, slab_b, sb, nb, db, ns);
}

// Data gradient of the second layer (64 -> 64, 3x3 / stride 1 / pad 1) fused with the weight gradient of the first
// (8 padded image channels -> 64): k_conv3x3_c64b<EPI_DGRAD, true>.  The gradient w.r.t. the first layer's output never
// reaches memory.  dy [B,H,W,64]; w_t [64][3][3][64] (ssd_weight_transpose of the second layer); relu_bits [B*H*W][8] (sign
// bits of the first layer's output, ssd_conv2d_fwd_relubits); image [B,H,W,8]; dw0 f32 [64][3][3][8] (pad channels: zeros),
// dbias0 f32 [64] or null.  One fp32 slab per workgroup, summed in a fixed order by the reduction kernel.
size_t ssd_conv2d_bwd_data_wgrad_first_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)512 * (64 * 72 + 64) * sizeof(float);
}

int ssd_conv2d_bwd_data_wgrad_first(const void* dy, const void* w_t, const void* relu_bits, const void* image, float* dw0,
                                    float* dbias0, int B, int H, int W, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !w_t || !relu_bits || !image || !dw0 || B <= 0 || H <= 0 || W <= 0) return SSD_ERR_VALUE;
    if (H < 16 || W < 16 || (long long)B * H * W * 64 >= (1ll << 31) - 16) return SSD_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < ssd_conv2d_bwd_data_wgrad_first_workspace_bytes(B, H, W)) return SSD_ERR_WORKSPACE;
    const ConvGeom g = make_geom(B, H, W, 64, H, W, 64, 3, 3, 1, 1, 1, 1);
    Epilogue ep = {};
    ep.ldo = 64;
    ep.mask_bits = static_cast<const unsigned char*>(relu_bits);
    const int tiles_x = (W + 15) / 16, tiles_y = (H + C64B_ROWS - 1) / C64B_ROWS;
    const int nblocks = B * tiles_x * tiles_y;
    const unsigned grid = (unsigned)(nblocks < 512 ? nblocks : 512);   // (one workgroup per CU, to run beside the next weight gradient: slower)
    float* slab_w = static_cast<float*>(ws);
    float* slab_b = slab_w + (size_t)512 * 64 * 72;
    hipStream_t s = (hipStream_t)stream;
    auto kern = k_conv3x3_c64b<EPI_DGRAD, true>;
    static OnceLds set; if (ensure_lds(set, reinterpret_cast<const void*>(kern), (int)(C64B_W0_LDS)) != 0) return SSD_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), C64B_W0_LDS, s, static_cast<const bf16_raw*>(dy), static_cast<const bf16_raw*>(w_t),
                       g, ep, tiles_x, tiles_y, C64W0{static_cast<const bf16_raw*>(image), slab_w, dbias0 ? slab_b : nullptr});
    if (hipGetLastError() != hipSuccess) return SSD_ERR_LAUNCH;
    launch_wgrad_reduce(s, slab_w, 64ll * 72, 64ll * 72, dw0, slab_b, 64ll, 64, dbias0, (int)grid);
    return ssd_launch_status();
}

// first layer (8 padded image channels): dedicated kernel, 512 pixel-block splits at most
static bool wgrad_first_layer(int H, int W, int Ho, int Wo, int Cin, int Cout, int ldy, int ksize, int stride, int pad_t, int pad_l) {
    return knob("SSD_WGRAD_FIRST", 1) && Cin == 8 && Cout <= 64 && ldy <= 64 && ksize == 3 && stride == 1 && pad_t == 1 && pad_l == 1 &&
           H == Ho && W == Wo && H >= 16 && W >= 16;
}

static bool wgrad_use_patch(int H, int W, int Ho, int Wo, int Cin, int ksize, int stride, int pad_t, int pad_l) {
    const int mn = wgrad_patch_min_hw();
    return mn > 0 && ksize == 3 && stride == 1 && pad_t == 1 && pad_l == 1 && H == Ho && W == Wo && Cin % 64 == 0 &&
           H >= mn && W >= mn;
}

// block shape of the patch kernel for a map: the candidate with the least padded work (0: 16x16, 1: 6x40, 2: 10x24)
static int wgrad_patch_shape(int Ho, int Wo, int* bh, int* bw) {
    static const int shapes[3][3] = {{16, 16, 16}, {6, 40, 15}, {10, 24, 15}};   // rows, columns, pair groups (of 16 slots)
    int best = 0;
    double best_cost = 0;
    for (int i = 0; i < 3; ++i) {
        // every block costs eight k-steps whatever its shape: fewest blocks wins
        const double cost = (double)((Ho + shapes[i][0] - 1) / shapes[i][0]) * ((Wo + shapes[i][1] - 1) / shapes[i][1]);
        if (i == 0 || cost < best_cost) { best = i; best_cost = cost; }
    }
    const int forced = knob("SSD_WGRAD_PATCH_SHAPE", -1);
    if (forced >= 0 && forced < 3) best = forced;
    *bh = shapes[best][0]; *bw = shapes[best][1];
    return best;
}

static void wgrad_patch_plan(int B, int Ho, int Wo, int Cin, int Cout, int* tiles_x, int* tiles_y, int* tps, int* ns) {
    int bh, bw;
    wgrad_patch_shape(Ho, Wo, &bh, &bw);
    *tiles_x = (Wo + bw - 1) / bw; *tiles_y = (Ho + bh - 1) / bh;
    const int ntiles = B * *tiles_x * *tiles_y;
    const int groups = (Cin / 64) * ((Cout + 63) / 64);
    const int wp_mult = knob("SSD_WGRAD_PATCH_SINGLE", 0) ? 2 : 1;
    int want = knob("SSD_WGRAD_PATCH_WGS", 256) * wp_mult / groups;   // one (two when single-buffered) workgroup per CU
    if (want < 1) want = 1;
    if (want > ntiles) want = ntiles;
    *tps = (ntiles + want - 1) / want;
    *ns = (ntiles + *tps - 1) / *tps;
}

static void wgrad_tiles(int Cout, long long ktot, int* bmo, int* bnc) {
    (void)Cout; (void)ktot;
    *bmo = 128; *bnc = 128;
}

static int wgrad_splits(long long M, int tiles) {
    long long want = 768 / tiles;                            // whole rounds: <= 3 workgroups per CU in total
    long long maxs = (M + 511) / 512;                        // at least 512 pixels per split
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    if (want > 512) want = 512;
    return (int)want;
}

// 256x256 GEMM weight-gradient kernel: used for wide layers the patch kernel does not serve
static bool wgrad_use_tile(long long M, int Cout, int ldy, long long ktot, long long x_elems) {
    // (SSD_WGTILE_MIN_TILES, development: layers with fewer 256 x 256 output tiles go to the 128 x 128 kernel -- a quarter of
    //  the slab bytes per split at the same workgroup count)
    const long long tiles = ((ktot + 255) / 256) * ((Cout + 255) / 256);
    return knob("SSD_WGRAD_TILE", 1) && Cout > 128 && ktot >= 256 && M >= 2048 && M * ldy < (1ll << 31) - 16 &&
           x_elems < (1ll << 31) - 16 && tiles >= knob("SSD_WGTILE_MIN_TILES", 0);
}

// pixel splits for that kernel (one workgroup per CU): estimated time = rounds x steps per split + slab traffic
static int wgrad_tile_splits(long long M, int tiles, long long slab_elems) {
    int best = 1;
    double best_cost = 0;
    const long long maxs = M / 256 > 0 ? M / 256 : 1;
    for (int ns = 1; ns <= 64 && ns <= maxs; ++ns) {
        const long long rounds = ((long long)tiles * ns + 255) / 256;
        const long long steps = ((M + ns - 1) / ns + 63) / 64;
        // (SSD_WGTILE_SLAB_X: weight of the slab term in tenths, development -- in the step the slab sums run beside another
        //  stream's kernels at a third of their speed alone, and CUs this kernel leaves idle are not wasted there)
        const double cost = (double)rounds * steps * 2.2 + (double)ns * slab_elems * 8.0 / 4.0e6 * (knob("SSD_WGTILE_SLAB_X", 10) / 10.0);   // microseconds
        if (ns == 1 || cost < best_cost) { best = ns; best_cost = cost; }
    }
    return best;
}

size_t ssd_conv2d_bwd_weight_workspace_bytes(int B, int Ho, int Wo, int Cin, int Cout, int ldy, int ksize) {
    if (B <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0 || ldy < Cout || ksize <= 0) return 0;
    const long long ktot = (long long)ksize * ksize * Cin;
    size_t patch_bytes = 0;
    if (wgrad_use_patch(Ho, Wo, Ho, Wo, Cin, ksize, 1, 1, 1)) {     // upper bound if the call turns out to be a patch case
        int tx, ty, tps, ns;
        wgrad_patch_plan(B, Ho, Wo, Cin, Cout, &tx, &ty, &tps, &ns);
        patch_bytes = (size_t)ns * ((size_t)ldy * ktot + ldy) * sizeof(float);
    }
    int bmo, bnc;
    wgrad_tiles(Cout, ktot, &bmo, &bnc);
    const int tiles = (int)(((ktot + bnc - 1) / bnc) * ((Cout + bmo - 1) / bmo));
    const int ns = wgrad_splits((long long)B * Ho * Wo, tiles);
    size_t gen = (size_t)ns * ((size_t)ldy * ktot + ldy) * sizeof(float);
    if (Cin == 8 && Cout <= 64 && ldy <= 64 && ksize == 3) {  // first-layer kernel: up to 512 splits
        const size_t first = (size_t)512 * ((size_t)ldy * ktot + ldy) * sizeof(float);
        if (first > gen) gen = first;
    }
    if (Cout > 128 && ktot >= 256) {                          // the 256x256 GEMM kernel may serve the call
        const int t2 = (int)(((ktot + 255) / 256) * ((Cout + 255) / 256));
        const int ns2 = wgrad_tile_splits((long long)B * Ho * Wo, t2, (long long)ldy * ktot);
        const size_t gen2 = (size_t)ns2 * ((size_t)ldy * ktot + ldy) * sizeof(float);
        if (gen2 > gen) gen = gen2;
    }
    return gen > patch_bytes ? gen : patch_bytes;
}

static int conv2d_bwd_weight_impl(const void* x, const void* dy, float* dw, float* dbias, int B, int H, int W, int Cin, int Cout,
                                  int ldy, int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo, void* ws, size_t ws_bytes,
                                  void* stream, int* plan) {
    // x: [B,H,W,Cin]; dy: [B,Ho,Wo,ldy] (first Cout channels used); dw: f32 [Cout][k][k][Cin]; dbias: f32 [Cout] or null
    if (!x || !dy || !dw || !geom_ok(B, H, W, Cin, Ho, Wo, Cout, ksize) || stride <= 0 || ldy < Cout || ldy % 8) return SSD_ERR_VALUE;
    if (!ws || ws_bytes < ssd_conv2d_bwd_weight_workspace_bytes(B, Ho, Wo, Cin, Cout, ldy, ksize)) return SSD_ERR_WORKSPACE;
    ConvGeom g = make_geom(B, H, W, Cin, Ho, Wo, ldy, ksize, ksize, stride, 1, pad_t, pad_l);
    const long long ktot = g.ldw;
    if (wgrad_first_layer(H, W, Ho, Wo, Cin, Cout, ldy, ksize, stride, pad_t, pad_l)) {
        const int tx = (Wo + 15) / 16, ty = (Ho + 15) / 16, ntiles = B * tx * ty;
        const int ns = ntiles < 512 ? ntiles : 512;
        const int tps = (ntiles + ns - 1) / ns;
        SSD_PLAN(SSD_PLAN_WG_FIRST | ((ntiles + tps - 1) / tps >= 32 ? SSD_PLAN_F_REDUCE_WIDE : 0));
        float* slab_w = static_cast<float*>(ws);
        float* slab_b = slab_w + (size_t)ns * ldy * ktot;
        hipStream_t s = (hipStream_t)stream;
        static OnceLds set; if (ensure_lds(set, reinterpret_cast<const void*>(k_conv0_wgrad), (int)(2 * W0_BUF)) != 0) return SSD_ERR_LAUNCH;
        hipLaunchKernelGGL(k_conv0_wgrad, dim3((unsigned)((ntiles + tps - 1) / tps)), dim3(512), 2 * W0_BUF, s,
                           static_cast<const bf16_raw*>(x), static_cast<const bf16_raw*>(dy), slab_w, dbias ? slab_b : nullptr, g,
                           tx, ty, tps, Cout);
        if (hipGetLastError() != hipSuccess) return SSD_ERR_LAUNCH;
        const int nsr = (ntiles + tps - 1) / tps;
        launch_wgrad_reduce(s, slab_w, (long long)ldy * ktot, (long long)Cout * ktot, dw, slab_b, (long long)ldy, Cout, dbias, nsr);
        return ssd_launch_status();
    }
    if (wgrad_use_patch(H, W, Ho, Wo, Cin, ksize, stride, pad_t, pad_l) &&
        (long long)B * H * W * Cin * 2 < (1ll << 32) - 16 && (long long)B * Ho * Wo * ldy * 2 < (1ll << 32) - 16) {   // 32-bit DMA offsets
        int tx, ty, tps, ns;
        wgrad_patch_plan(B, Ho, Wo, Cin, Cout, &tx, &ty, &tps, &ns);
        float* slab_w = static_cast<float*>(ws);
        float* slab_b = slab_w + (size_t)ns * ldy * ktot;
        hipStream_t s = (hipStream_t)stream;
        const int single = 0;                                   // (single-buffer mode was dropped with the loader-wave kernel)
        const int groups = (Cin / 64) * ((Cout + 63) / 64), nunits = groups * ns;
        // units per XCD group: the largest divisor of the channel-group count whose round-robin placement (group i on
        // XCD i % 8) keeps every XCD within ~7 % of its fair share of workgroups
        int xg = 1;
        for (int d = groups; d >= 1; --d) {
            if (groups % d) continue;
            const int ngr = (nunits + d - 1) / d;
            const int load = ((ngr + 7) / 8) * d, fair = (nunits + 7) / 8;
            if (load * 100 <= fair * 107) { xg = d; break; }
        }
        if (!knob("SSD_WGRAD_PATCH_XCD", 1)) xg = 1;
        const unsigned grid = (unsigned)(8 * xg * ((nunits + 8 * xg - 1) / (8 * xg)));
        int bh, bw;
        const int shape = wgrad_patch_shape(Ho, Wo, &bh, &bw);
        SSD_PLAN((shape == 1 ? SSD_PLAN_WG_PATCH_6x40 : (shape == 2 ? SSD_PLAN_WG_PATCH_10x24 : SSD_PLAN_WG_PATCH_16x16)) |
                 (ns >= 32 ? SSD_PLAN_F_REDUCE_WIDE : 0));
#define SSD_LAUNCH_WP(BH_, BW8_)                                                                                    \
        do {                                                                                                        \
            using G_ = WpGeom<BH_, BW8_>;                                                                           \
            auto kern_ = k_conv3x3_wgrad_patch<BH_, BW8_>;                                                          \
            static OnceLds set_; if (ensure_lds(set_, reinterpret_cast<const void*>(kern_), (int)(2 * G_::BUF)) != 0) return SSD_ERR_LAUNCH; \
            hipLaunchKernelGGL(kern_, dim3(grid), dim3(512), (size_t)(single ? 1 : 2) * G_::BUF, s,                  \
                               static_cast<const bf16_raw*>(x), static_cast<const bf16_raw*>(dy), slab_w,          \
                               dbias ? slab_b : nullptr, g, tx, ty, tps, ns, Cout, single, xg);                     \
        } while (0)
        if (shape == 1) SSD_LAUNCH_WP(6, 5);
        else if (shape == 2) SSD_LAUNCH_WP(10, 3);
        else SSD_LAUNCH_WP(16, 2);
#undef SSD_LAUNCH_WP
        if (hipGetLastError() != hipSuccess) return SSD_ERR_LAUNCH;
        launch_wgrad_reduce(s, slab_w, (long long)ldy * ktot, (long long)Cout * ktot, dw, slab_b, (long long)ldy, Cout, dbias, ns);
        return ssd_launch_status();
    }
    if (wgrad_use_tile(g.M, Cout, ldy, ktot, (long long)B * H * W * Cin)) {
        const int ctiles = (int)((ktot + 255) / 256), mtiles = (Cout + 255) / 256;
        const int ns = wgrad_tile_splits(g.M, ctiles * mtiles, (long long)ldy * ktot);
        int mps = (int)(((long long)g.M + ns - 1) / ns);
        mps = (mps + 63) / 64 * 64;
        SSD_PLAN(SSD_PLAN_WG_TILE | (ns >= 32 ? SSD_PLAN_F_REDUCE_WIDE : 0));
        float* slab_w = static_cast<float*>(ws);
        float* slab_b = slab_w + (size_t)ns * ldy * ktot;
        hipStream_t s = (hipStream_t)stream;
if (knob("SSD_WGTILE_STAGES", 4) == 4) {
            static OnceLds set; if (ensure_lds(set, reinterpret_cast<const void*>(k_conv_wgrad_tile<4>), (int)(4 * WT_TILE)) != 0) return SSD_ERR_LAUNCH;
            hipLaunchKernelGGL(k_conv_wgrad_tile<4>, dim3(ctiles * mtiles * ns), dim3(512), 4 * WT_TILE, s,
                           static_cast<const bf16_raw*>(x), static_cast<const bf16_raw*>(dy), slab_w, dbias ? slab_b : nullptr, g,
                           mps, ns, Cout);
        } else {
            static OnceLds set; if (ensure_lds(set, reinterpret_cast<const void*>(k_conv_wgrad_tile<2>), (int)(4 * WT_TILE)) != 0) return SSD_ERR_LAUNCH;
            hipLaunchKernelGGL(k_conv_wgrad_tile<2>, dim3(ctiles * mtiles * ns), dim3(512), 4 * WT_TILE, s,
                           static_cast<const bf16_raw*>(x), static_cast<const bf16_raw*>(dy), slab_w, dbias ? slab_b : nullptr, g,
                           mps, ns, Cout);
        }
        if (hipGetLastError() != hipSuccess) return SSD_ERR_LAUNCH;
        launch_wgrad_reduce(s, slab_w, (long long)ldy * ktot, (long long)Cout * ktot, dw, slab_b, (long long)ldy, Cout, dbias, ns);
        return ssd_launch_status();
    }
    int bmo, bnc;
    wgrad_tiles(Cout, ktot, &bmo, &bnc);
    const int ctiles = (int)((ktot + bnc - 1) / bnc), mtiles = (Cout + bmo - 1) / bmo;
    const int ns = wgrad_splits(g.M, ctiles * mtiles);
    int mps = (int)(((long long)g.M + ns - 1) / ns);
    mps = (mps + 63) / 64 * 64;
    SSD_PLAN(SSD_PLAN_WG_GENERIC | (ns >= 32 ? SSD_PLAN_F_REDUCE_WIDE : 0));
    float* slab_w = static_cast<float*>(ws);
    float* slab_b = slab_w + (size_t)ns * ldy * ktot;
    hipStream_t s = (hipStream_t)stream;
    const bf16_raw* xp = static_cast<const bf16_raw*>(x);
    const bf16_raw* dyp = static_cast<const bf16_raw*>(dy);
    float* sb = dbias ? slab_b : nullptr;
    {
        const size_t lds = 4 * 64 * WG_LD;
        static OnceLds attr_set; if (ensure_lds(attr_set, reinterpret_cast<const void*>(k_conv_wgrad), (int)((int)lds)) != 0) return SSD_ERR_LAUNCH;
        hipLaunchKernelGGL(k_conv_wgrad, dim3(ctiles, mtiles, ns), dim3(WG), lds, s, xp, dyp, slab_w, sb, g, mps);
    }
    if (hipGetLastError() != hipSuccess) return SSD_ERR_LAUNCH;
    launch_wgrad_reduce(s, slab_w, (long long)ldy * ktot, (long long)Cout * ktot, dw, slab_b, (long long)ldy, Cout, dbias, ns);
    return ssd_launch_status();
}

// Several SMALL layers' weight gradients in two launches (slab kernel + slab sums) instead of two per layer: the extras on the
// 10x10 ... 1x1 maps (reference models/ssd_model.py:124-150) are six launches of 2-70 workgroups each, ~25 us apiece on the side
// stream beside the other streams' kernels.  Served: the layers ssd_conv2d_bwd_weight itself runs on the generic kernel (not the
// first layer, not a patch / 256-wide tile case) with fewer than 32 splits; SSD_ERR_UNSUPPORTED otherwise, nothing launched.
// Every layer's blocks and sums are the ones its own call would have run: results are bit-identical to separate calls.
size_t ssd_conv2d_bwd_weight_batched_workspace_bytes(const ssd_wgrad_item* items, int count) {
    if (!items || count <= 0) return 0;
    size_t tot = 0;
    for (int i = 0; i < count; ++i)
        tot += ssd_align_up(ssd_conv2d_bwd_weight_workspace_bytes(items[i].B, items[i].Ho, items[i].Wo, items[i].Cin, items[i].Cout,
                                                                   items[i].ldy, items[i].ksize), 256);
    return tot;
}

int ssd_conv2d_bwd_weight_batched(const ssd_wgrad_item* items, int count, void* ws, size_t ws_bytes, void* stream) {
    if (!items || count <= 0) return SSD_ERR_VALUE;
    if (count > WGB_MAX) return SSD_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < ssd_conv2d_bwd_weight_batched_workspace_bytes(items, count)) return SSD_ERR_WORKSPACE;
    WgradBatchArgs wa;
    ReduceBatchArgs ra;
    wa.count = ra.count = count;
    char* p = static_cast<char*>(ws);
    int blk = 0, rblk = 0;
    for (int i = 0; i < count; ++i) {
        const ssd_wgrad_item& it = items[i];
        if (!it.x || !it.dy || !it.dw || !geom_ok(it.B, it.H, it.W, it.Cin, it.Ho, it.Wo, it.Cout, it.ksize) || it.stride <= 0 ||
            it.ldy < it.Cout || it.ldy % 8)
            return SSD_ERR_VALUE;
        const ConvGeom g = make_geom(it.B, it.H, it.W, it.Cin, it.Ho, it.Wo, it.ldy, it.ksize, it.ksize, it.stride, 1, it.pad_t, it.pad_l);
        const long long ktot = g.ldw;
        if (wgrad_first_layer(it.H, it.W, it.Ho, it.Wo, it.Cin, it.Cout, it.ldy, it.ksize, it.stride, it.pad_t, it.pad_l) ||
            wgrad_use_patch(it.H, it.W, it.Ho, it.Wo, it.Cin, it.ksize, it.stride, it.pad_t, it.pad_l) ||
            wgrad_use_tile(g.M, it.Cout, it.ldy, ktot, (long long)it.B * it.H * it.W * it.Cin))
            return SSD_ERR_UNSUPPORTED;
        int bmo, bnc;
        wgrad_tiles(it.Cout, ktot, &bmo, &bnc);
        const int ctiles = (int)((ktot + bnc - 1) / bnc), mtiles = (it.Cout + bmo - 1) / bmo;
        const int ns = wgrad_splits(g.M, ctiles * mtiles);
        if (ns >= 32) return SSD_ERR_UNSUPPORTED;              // (the wide reduction's case)
        int mps = (int)(((long long)g.M + ns - 1) / ns);
        mps = (mps + 63) / 64 * 64;
        float* slab_w = reinterpret_cast<float*>(p);
        float* slab_b = slab_w + (size_t)ns * it.ldy * ktot;
        p += ssd_align_up(ssd_conv2d_bwd_weight_workspace_bytes(it.B, it.Ho, it.Wo, it.Cin, it.Cout, it.ldy, it.ksize), 256);
        wa.it[i] = WgradBatchItem{static_cast<const bf16_raw*>(it.x), static_cast<const bf16_raw*>(it.dy), slab_w,
                                  it.dbias ? slab_b : nullptr, g, mps, ctiles, mtiles, blk};
        blk += ctiles * mtiles * ns;
        const long long nw = (long long)it.Cout * ktot;
        const unsigned nbw = (unsigned)((nw / 4 + 255) / 256), nbb = it.dbias ? (unsigned)((it.Cout + 255) / 256) : 0u;
        ra.it[i] = ReduceBatchItem{slab_w, slab_b, it.dw, it.dbias, (long long)it.ldy * ktot, nw, (long long)it.ldy, it.Cout, ns, rblk, nbw};
        rblk += (int)(nbw + nbb);
    }
    for (int i = count; i < WGB_MAX; ++i) { wa.it[i] = wa.it[0]; wa.it[i].blk0 = blk; ra.it[i] = ra.it[0]; ra.it[i].blk0 = rblk; }
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = 4 * 64 * WG_LD;
    static OnceLds attr_set; if (ensure_lds(attr_set, reinterpret_cast<const void*>(k_conv_wgrad_batched), (int)lds) != 0) return SSD_ERR_LAUNCH;
    hipLaunchKernelGGL(k_conv_wgrad_batched, dim3(blk), dim3(WG), lds, s, wa);
    if (hipGetLastError() != hipSuccess) return SSD_ERR_LAUNCH;
    hipLaunchKernelGGL(k_wgrad_reduce2_batched, dim3(rblk), dim3(256), 0, s, ra);
    return ssd_launch_status();
}

int ssd_conv2d_bwd_weight(const void* x, const void* dy, float* dw, float* dbias, int B, int H, int W, int Cin, int Cout,
                          int ldy, int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo, void* ws, size_t ws_bytes,
                          void* stream) {
    return conv2d_bwd_weight_impl(x, dy, dw, dbias, B, H, W, Cin, Cout, ldy, ksize, stride, pad_t, pad_l, Ho, Wo, ws, ws_bytes,
                                  stream, nullptr);
}

// ---- dispatch queries: the same code path with `plan` set (nothing is launched, no pointer is dereferenced) ----
static void* const PLAN_PTR = reinterpret_cast<void*>(static_cast<uintptr_t>(64));

int ssd_conv2d_fwd_plan(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo,
                        int pool, size_t ws_bytes) {
    int plan = 0, rc;
    void* ws = ws_bytes ? PLAN_PTR : nullptr;
    if (pool) {
        const int Hp = (Ho + 1) / 2, Wp = (Wo + 1) / 2;
        rc = conv2d_fwd_pool_impl(PLAN_PTR, PLAN_PTR, nullptr, pool == 2 ? nullptr : PLAN_PTR, PLAN_PTR, PLAN_PTR, B, H, W, Cin, Cout,
                                  ksize, stride, pad_t, pad_l, Ho, Wo, 1, Hp, Wp, ws, ws_bytes, nullptr, &plan);
    } else {
        rc = conv2d_fwd_impl(PLAN_PTR, PLAN_PTR, nullptr, PLAN_PTR, B, H, W, Cin, Cout, ksize, stride, pad_t, pad_l, Ho, Wo, 1, ws,
                             ws_bytes, nullptr, &plan);
    }
    return rc != SSD_OK ? rc : plan;
}

int ssd_conv2d_head_fwd_plan(int B, int H, int W, int Cin, int per_cell, int classes, size_t ws_bytes) {
    int plan = 0;
    const int rc = conv2d_head_fwd_impl(PLAN_PTR, PLAN_PTR, nullptr, PLAN_PTR, PLAN_PTR, B, H, W, Cin, per_cell, classes,
                                        H * W * per_cell, 0, ws_bytes ? PLAN_PTR : nullptr, ws_bytes, nullptr, &plan);
    return rc != SSD_OK ? rc : plan;
}

int ssd_conv2d_bwd_data_plan(int B, int H, int W, int Cin, int Cout_pad, int ksize, int stride, int pad_t, int pad_l, int Ho,
                             int Wo, int accumulate, size_t ws_bytes) {
    int plan = 0;
    const int rc = conv2d_bwd_data_impl(PLAN_PTR, PLAN_PTR, nullptr, PLAN_PTR, B, H, W, Cin, Cout_pad, ksize, stride, pad_t, pad_l,
                                        Ho, Wo, accumulate, ws_bytes ? PLAN_PTR : nullptr, ws_bytes, nullptr, &plan);
    return rc != SSD_OK ? rc : plan;
}

int ssd_conv2d_bwd_weight_plan(int B, int H, int W, int Cin, int Cout, int ldy, int ksize, int stride, int pad_t, int pad_l,
                               int Ho, int Wo) {
    int plan = 0;
    const size_t need = ssd_conv2d_bwd_weight_workspace_bytes(B, Ho, Wo, Cin, Cout, ldy, ksize);
    const int rc = conv2d_bwd_weight_impl(PLAN_PTR, PLAN_PTR, reinterpret_cast<float*>(PLAN_PTR), reinterpret_cast<float*>(PLAN_PTR),
                                          B, H, W, Cin, Cout, ldy, ksize, stride, pad_t, pad_l, Ho, Wo, PLAN_PTR, need, nullptr,
                                          &plan);
    return rc != SSD_OK ? rc : plan;
}

const char* ssd_conv_plan_name(int plan) {
    switch (plan & SSD_PLAN_KERNEL_MASK) {
        case SSD_PLAN_C64B: return "k_conv3x3_c64b";
        case SSD_PLAN_P32_64: return "k_conv3x3_patch32<64>";
        case SSD_PLAN_P32_128: return "k_conv3x3_patch32<128>";
        case SSD_PLAN_P512: return "k_conv3x3_p512";
        case SSD_PLAN_8PH: return "k_conv_igemm_8ph";
        case SSD_PLAN_PW: return "k_pw_gemm";
        case SSD_PLAN_DMA_256_256: return "k_conv_igemm_dma<256,256>";
        case SSD_PLAN_DMA_256_128: return "k_conv_igemm_dma<256,128>";
        case SSD_PLAN_DMA_256_64: return "k_conv_igemm_dma<256,64>";
        case SSD_PLAN_DMA_128_64: return "k_conv_igemm_dma<128,64>";
        case SSD_PLAN_DMA_128_128: return "k_conv_igemm_dma<128,128>";
        case SSD_PLAN_CONV0_FWD: return "k_conv0_fwd";
        case SSD_PLAN_WG_FIRST: return "k_conv0_wgrad";
        case SSD_PLAN_WG_PATCH_16x16: return "k_conv3x3_wgrad_patch<16,2>";
        case SSD_PLAN_WG_PATCH_6x40: return "k_conv3x3_wgrad_patch<6,5>";
        case SSD_PLAN_WG_PATCH_10x24: return "k_conv3x3_wgrad_patch<10,3>";
        case SSD_PLAN_WG_TILE: return "k_conv_wgrad_tile";
        case SSD_PLAN_WG_GENERIC: return "k_conv_wgrad";
        default: return "?";
    }
}

int ssd_weight_transpose(const void* w, void* w_t, int Cout, int ksize, int Cin, int Cout_pad, void* stream) {
    if (!w || !w_t || Cout <= 0 || ksize <= 0 || Cin <= 0 || Cout_pad < Cout || Cout_pad % 8) return SSD_ERR_VALUE;
    const long long total = (long long)Cin * ksize * ksize * Cout_pad;
    hipLaunchKernelGGL(k_weight_transpose, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const bf16_raw*>(w), static_cast<bf16_raw*>(w_t), Cout, ksize, ksize, Cin, Cout_pad);
    return ssd_launch_status();
}

int ssd_weight_transpose_batched(const long long* desc, int ntensors, int max_tiles, void* stream) {
    if (!desc || ntensors <= 0 || max_tiles <= 0) return SSD_ERR_VALUE;
    hipLaunchKernelGGL(k_weight_transpose_batched, dim3((unsigned)max_tiles, (unsigned)ntensors), dim3(256), 0, (hipStream_t)stream, desc);
    return ssd_launch_status();
}

int ssd_cast_bf16(const float* src, void* dst, long long n, void* stream) {
    if (n < 0 || (n > 0 && (!src || !dst))) return SSD_ERR_VALUE;
    if (n == 0) return SSD_OK;
    hipLaunchKernelGGL(k_cast_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       static_cast<bf16_raw*>(dst), n);
    return ssd_launch_status();
}

int ssd_image_prep(const float* img, void* out, int B, int H, int W, int normalize, void* stream) {
    if (!img || !out || B <= 0 || H <= 0 || W <= 0) return SSD_ERR_VALUE;
    const long long npix = (long long)B * H * W;
    hipLaunchKernelGGL(k_image_prep, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img,
                       static_cast<bf16_raw*>(out), npix, normalize);
    return ssd_launch_status();
}

int ssd_maxpool2x2_fwd(const void* x, void* y, int B, int H, int W, int C, int Ho, int Wo, void* stream) {
    if (!x || !y || B <= 0 || C <= 0 || C % 8) return SSD_ERR_VALUE;
    if ((Ho != H / 2 && Ho != (H + 1) / 2) || (Wo != W / 2 && Wo != (W + 1) / 2) || Ho <= 0 || Wo <= 0) return SSD_ERR_VALUE;
    const long long total = (long long)B * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(k_maxpool_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const bf16_raw*>(x), static_cast<bf16_raw*>(y), B, H, W, C, Ho, Wo);
    return ssd_launch_status();
}

int ssd_maxpool2x2_fwd_argmax(const void* x, void* y, void* code, int B, int H, int W, int C, int Ho, int Wo, void* stream) {
    if (!x || !y || !code || B <= 0 || C <= 0 || C % 8) return SSD_ERR_VALUE;
    if ((Ho != H / 2 && Ho != (H + 1) / 2) || (Wo != W / 2 && Wo != (W + 1) / 2) || Ho <= 0 || Wo <= 0) return SSD_ERR_VALUE;
    const long long total = (long long)B * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(k_maxpool_fwd_argmax, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const bf16_raw*>(x), static_cast<bf16_raw*>(y), static_cast<unsigned*>(code), B, H, W, C, Ho, Wo);
    return ssd_launch_status();
}

int ssd_maxpool2x2_bwd_argmax(const void* code, const void* dy, void* dx, int B, int H, int W, int C, int Ho, int Wo,
                              void* stream) {
    if (!code || !dy || !dx || B <= 0 || C <= 0 || C % 8 || Ho <= 0 || Wo <= 0) return SSD_ERR_VALUE;
    if (2 * Ho < H || 2 * Wo < W) {
        // VALID pooling of an odd size leaves the last row/column without gradient: clear it first
        if (hipMemsetAsync(dx, 0, (size_t)B * H * W * C * 2, (hipStream_t)stream) != hipSuccess) return SSD_ERR_LAUNCH;
    }
    const long long total = (long long)B * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(k_maxpool_bwd_argmax, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const unsigned*>(code), static_cast<const bf16_raw*>(dy), static_cast<bf16_raw*>(dx), B, H, W, C,
                       Ho, Wo);
    return ssd_launch_status();
}

int ssd_maxpool2x2_bwd(const void* x, const void* y, const void* dy, void* dx, int B, int H, int W, int C, int Ho, int Wo,
                       void* stream) {
    if (!x || !y || !dy || !dx || B <= 0 || C % 8) return SSD_ERR_VALUE;
    if (2 * Ho < H || 2 * Wo < W) {
        // VALID pooling of an odd size leaves the last row/column without gradient: clear it first
        if (hipMemsetAsync(dx, 0, (size_t)B * H * W * C * 2, (hipStream_t)stream) != hipSuccess) return SSD_ERR_LAUNCH;
    }
    const long long total = (long long)B * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(k_maxpool_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const bf16_raw*>(x), static_cast<const bf16_raw*>(y), static_cast<const bf16_raw*>(dy),
                       static_cast<bf16_raw*>(dx), B, H, W, C, Ho, Wo);
    return ssd_launch_status();
}

int ssd_head_grad_pack(const void* dloc, const void* dconf, void* out, int B, int hw, int per_cell, int classes, int npad,
                       int anchors_total, int level_off, void* stream) {
    if (!dloc || !dconf || !out || B <= 0 || hw <= 0 || npad < per_cell * (4 + classes) || npad % 8) return SSD_ERR_VALUE;
    const long long total = (long long)B * hw * (npad / 8);
    hipLaunchKernelGGL(k_head_grad_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const bf16_raw*>(dloc), static_cast<const bf16_raw*>(dconf), static_cast<bf16_raw*>(out),
                       B, hw, per_cell, classes, npad, anchors_total, level_off);
    return ssd_launch_status();
}

}  // extern "C"
