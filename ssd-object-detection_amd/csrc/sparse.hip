// Backward pass of the SSD heads from the loss's COMPACT gradient rows, for gfx950 (MI355X).
//
// Replaces, for the twelve head convolutions (models/ssd_model.py:153-162), the part of tape.gradient (:248) that the
// dense kernels of conv.hip computed from a 97 % zero operand.  The loss gives a gradient to the positives and the mined
// negatives only (:355-380: ~4P of B*A anchors), so a level's head output gradient dY [B,H,W,n*(4+C)] has a few per cent
// non-zero pixel rows; ssd_loss_fwd_bwd_heads hands over exactly those rows (ascending pixel order) with both index maps.
//
//   data gradient    dX[p] = sum over taps t of dY[p - (t - 1)] . W[:, t, :]      (3x3, stride 1, pad 1)
//       k_hz_gemm    Z[r][t][ci] = sum_co rows[r][co] * Wtap[t][ci][co]   -- one GEMM over the compact rows (fp32 out)
//       k_hz_col2im  dX[p][ci]   = sum_t Z[row_of_pixel[p - (t-1)]][t][ci], ReLU-masked, bf16 -- every output pixel, fixed
//                    tap order (deterministic); pixels no row reaches are written as zeros
//   weight gradient  dW[co][t][ci] = sum_r rows[r][co] * X[pixel_of_row[r] + (t - 1)][ci]
//       k_hw_gather  256 co x 256 (t, ci) tiles over the compact rows, X rows gathered through the index map by LDS-DMA,
//                    fixed pixel splits -> slabs, k_hw_reduce adds them in order
// All six levels of one network go through ONE launch of each kernel (per-level descriptors; the row counts are read
// on the device: nothing synchronises the host).  Work is proportional to the rows, 5 % of the dense form at SSD300's
// density; worst case (every pixel selected) it degenerates to the dense FLOPs plus the Z round trip.
#include <atomic>
#include <cstdint>
#include "common.h"
#include <hip/hip_bf16.h>
#include "conv_common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
constexpr unsigned OOB = 0xfffffff0u;

// ------------------------------------------------------------------------------------------------
// Z = rows x Wtap^T
struct ZLevel {
    const bf16_raw* rows;                    // [count][npad]
    const bf16_raw* wt;                      // [N][npad], N = 9*Cin rows ordered (tap, ci)
    float* z;                                // [count][N]
    int npad, N;
};
struct ZArgs {
    int levels;
    unsigned mask;                           // bit l: level l takes part in this launch
    ZLevel lv[SSD_MAX_LEVELS];
    const int* count;
};

constexpr int ZT = 128;                      // tile: 128 rows x 128 columns, k-step 64
constexpr int ZBUF = 2 * ZT * 128;           // one stage: A image + B image (128-byte rows)

__global__ __launch_bounds__(256) void k_hz_gemm(ZArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave & 1, wave_n = wave >> 1;
    int pre[SSD_MAX_LEVELS + 1];
    pre[0] = 0;
#pragma unroll
    for (int l = 0; l < SSD_MAX_LEVELS; ++l) {
        int t = 0;
        if (l < a.levels && ((a.mask >> l) & 1u)) t = ((a.count[l] + ZT - 1) / ZT) * (a.lv[l].N / ZT);
        pre[l + 1] = pre[l] + t;
    }
    const int total = pre[SSD_MAX_LEVELS];
    // DMA ownership (as k_conv_igemm_dma): instruction i fills tile rows 8i..8i+7; lane L -> row 8i + 2(L>>4) + ((L>>3)&1),
    // logical 16-byte chunk (L&7) ^ ((row>>1)&7).  Wave w issues instructions w, w+4, w+8, w+12 of both images.
    const int rl = 2 * (lane >> 4) + ((lane >> 3) & 1);
    const int chunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    const int frow = lane & 15, fk = lane >> 4;

    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        int l = 0;
#pragma unroll
        for (int k = 1; k < SSD_MAX_LEVELS; ++k) l += (k < a.levels && t >= pre[k]) ? 1 : 0;
        const ZLevel lv = a.lv[l];
        const int cnt = a.count[l];
        const int ntn = lv.N / ZT;
        const int local = t - pre[l];
        const int m0 = (local / ntn) * ZT, n0 = (local % ntn) * ZT;
        const int nchunks = lv.npad >> 3;
        const int nks = (nchunks + 7) >> 3;
        const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc((void*)lv.rows, 0, (unsigned)cnt * (unsigned)lv.npad * 2u, 0x00020000);
        const __amdgpu_buffer_rsrc_t bres = __builtin_amdgcn_make_buffer_rsrc((void*)lv.wt, 0, (unsigned)lv.N * (unsigned)lv.npad * 2u, 0x00020000);
        unsigned arow[4], brow[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = wave + 4 * j;
            arow[j] = (unsigned)(m0 + 8 * i + rl) * (unsigned)lv.npad * 2u;
            brow[j] = (unsigned)(n0 + 8 * i + rl) * (unsigned)lv.npad * 2u;
        }
        auto issue = [&](int ks, int buf) {
            const int q = ks * 8 + chunk;
            const bool kv = q < nchunks;
            char* base = smem + buf * ZBUF;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ares, (lds_void*)(base + (wave + 4 * j) * 1024), 16,
                                                         kv ? arow[j] + (unsigned)q * 16u : OOB, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(bres, (lds_void*)(base + ZT * 128 + (wave + 4 * j) * 1024), 16,
                                                         kv ? brow[j] + (unsigned)q * 16u : OOB, 0, 0, 0);
        };
        f32x4_t acc[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[c][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        __syncthreads();                                     // the previous tile's last fragment reads are done
        issue(0, 0);
        for (int ks = 0; ks < nks; ++ks) {
            const int cur = ks & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (ks + 1 < nks) issue(ks + 1, cur ^ 1);
            const char* sx = smem + cur * ZBUF;
            const char* sw = sx + ZT * 128;
#pragma unroll
            for (int ksub = 0; ksub < 2; ++ksub) {
                bf16x8_t fx[4], fw[4];
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    fx[p] = *reinterpret_cast<const bf16x8_t*>(sx + swz(wave_m * 64 + p * 16 + frow, ksub * 4 + fk));
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    fw[c] = *reinterpret_cast<const bf16x8_t*>(sw + swz(wave_n * 64 + c * 16 + frow, ksub * 4 + fk));
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[c], fx[p], acc[c][p], 0, 0, 0);
            }
        }
        // a lane holds columns n + 0..3 of row m: 16-byte stores, 64-byte runs per row and column tile
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int m = m0 + wave_m * 64 + p * 16 + (lane & 15);
            if (m >= cnt) continue;
            float* o = lv.z + (long long)m * lv.N + n0 + wave_n * 64 + (lane >> 4) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                *reinterpret_cast<float4*>(o + c * 16) = make_float4(acc[c][p][0], acc[c][p][1], acc[c][p][2], acc[c][p][3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dX from Z: one thread = one output pixel x 8 channels
struct CLevel {
    const float* z;
    const int* rop;
    const unsigned char* bits;               // ReLU sign bits [B*H*W][Cin/8] or null
    const bf16_raw* src;                     // else: the activation itself [B*H*W][Cin] or null (no mask)
    bf16_raw* dx;
    int H, W, Cin;
    int blk0;                                // first workgroup of the level
};
struct CArgs {
    int levels, B;
    int prezeroed;                           // the caller cleared the maps: pixels no row reaches are left alone
    CLevel lv[SSD_MAX_LEVELS];
};

// Four consecutive pixels per thread: a wave owns one pixel's 64 channel groups (Cin = 512), so per pixel it issues nine
// wave-uniform index loads, a byte of sign bits and one 1 KB store behind TWO dependent memory round trips -- at one pixel
// per thread the kernel ran at 1.2 TB/s of (mostly zero) stores, latency-bound.  All 36 indices and the four mask bytes
// are requested first; the rows that exist (5 % of the pixels have any) follow in tap order, pixel by pixel.
constexpr int C2I_PIX = 4;

__global__ __launch_bounds__(256) void k_hz_col2im(CArgs a) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < SSD_MAX_LEVELS; ++k) l += (k < a.levels && (int)blockIdx.x >= a.lv[k].blk0) ? 1 : 0;
    const CLevel lv = a.lv[l];
    const int cpp = lv.Cin >> 3;
    const long long idx = (long long)(blockIdx.x - lv.blk0) * 256 + threadIdx.x;
    const int hw = lv.H * lv.W, npix = a.B * hw;
    const int pg = (int)(idx / cpp);
    const int pixel0 = pg * C2I_PIX;
    if (pixel0 >= npix) return;
    const int cg = (int)(idx - (long long)pg * cpp);
    const int N = 9 * lv.Cin;
    int r[C2I_PIX][9];
    unsigned mb[C2I_PIX];
#pragma unroll
    for (int j = 0; j < C2I_PIX; ++j) {
        const int pixel = min(pixel0 + j, npix - 1);         // (a clamped duplicate is computed and not stored)
        const int b = pixel / hw, rem = pixel - b * hw;
        const int y = rem / lv.W, x = rem - y * lv.W;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int qy = y - (t / 3 - 1), qx = x - (t % 3 - 1);
            const bool in = (unsigned)qy < (unsigned)lv.H && (unsigned)qx < (unsigned)lv.W;
            r[j][t] = lv.rop[in ? b * hw + qy * lv.W + qx : pixel];
            if (!in) r[j][t] = -1;
        }
        mb[j] = 0xffu;
    }
    // A pixel no row reaches (95 % of them) is zeros whatever its mask: those stores go out FIRST, back to back, with nothing
    // to wait for.  Behind the conditional Z loads of the general path every pixel's store waited (s_waitcnt vmcnt(0) at the
    // join) for the previous pixel's STORE to complete -- four serialised write round trips per thread, 1.4 TB/s.
    bool any[C2I_PIX];
#pragma unroll
    for (int j = 0; j < C2I_PIX; ++j) {
        int m = r[j][0];
#pragma unroll
        for (int t = 1; t < 9; ++t) m = max(m, r[j][t]);
        any[j] = m >= 0;
    }
#pragma unroll
    for (int j = 0; j < C2I_PIX; ++j) {
        const int pixel = pixel0 + j;
        if (pixel < npix && !any[j] && !a.prezeroed)
            *reinterpret_cast<uint4*>(lv.dx + (long long)pixel * lv.Cin + cg * 8) = make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int j = 0; j < C2I_PIX; ++j) {
        const int pixel = pixel0 + j;
        if (pixel >= npix) break;
        if (!any[j]) continue;
        if (lv.bits) mb[j] = lv.bits[(long long)pixel * cpp + cg];
        float4 v0[9], v1[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            v0[t] = v1[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r[j][t] >= 0) {
                const float4* zp = reinterpret_cast<const float4*>(lv.z + (long long)r[j][t] * N + t * lv.Cin + cg * 8);
                v0[t] = zp[0]; v1[t] = zp[1];
            }
        }
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {                         // fixed tap order (adding the +0 of an absent row changes no bit)
            acc[0] += v0[t].x; acc[1] += v0[t].y; acc[2] += v0[t].z; acc[3] += v0[t].w;
            acc[4] += v1[t].x; acc[5] += v1[t].y; acc[6] += v1[t].z; acc[7] += v1[t].w;
        }
        uint4 v = make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]),
                             pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7]));
        const long long o = (long long)pixel * lv.Cin + cg * 8;
        if (lv.bits) {
            v = gate_bits8(v, mb[j]);
        } else if (lv.src) {
            v = gate_bits8(v, relu_bits8(*reinterpret_cast<const uint4*>(lv.src + o)));
        }
        *reinterpret_cast<uint4*>(lv.dx + o) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient over the compact rows (the tile scheme of k_conv_wgrad_tile, conv.hip): 256 co x 256 (tap, ci) columns
// per workgroup, 64 rows per step; the dY image is read straight from the compact rows, the X image is gathered: row m
// of a step comes from pixel pixel_of_row[m] shifted by the column's tap.  The index of a row is fetched one step ahead.
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
__device__ __forceinline__ s16x4_t lds_read_tr16_scoped(const char* __restrict__ p, const char* __restrict__ other) {
    (void)other;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
}

struct WLevel {
    const bf16_raw* x;                       // [B,H,W,Cin]
    const bf16_raw* rows;                    // [count][npad]
    const int* por;
    float* slab_w;                           // [nsplit][npad][ktot]
    float* slab_b;                           // [nsplit][npad]
    int H, W, Cin, npad, cout, nsplit;
    int blk0;                                // first workgroup of the level
};
struct WArgs {
    int levels, B;
    WLevel lv[SSD_MAX_LEVELS];
    const int* count;
};

constexpr int WT_TILE = 64 * 512;                          // one [64 rows][256 ch] image
// The host fixes the MAXIMUM number of pixel splits of a level (it does not know the row count); the kernels use only as
// many as give every active split at least HW_MIN_ROWS rows, so a level with few rows neither writes nor re-reads slabs
// of splits that would be empty.  Rows per active split (a multiple of 64) and their number, from the device-side count:
constexpr int HW_MIN_ROWS = 768;
__device__ __forceinline__ int hw_rows_per_split(int cnt, int nsplit_max) {
    const int per = max((cnt + nsplit_max - 1) / nsplit_max, HW_MIN_ROWS);
    return (per + 63) / 64 * 64;
}
__device__ __forceinline__ int hw_active_splits(int cnt, int nsplit_max) {
    const int per = hw_rows_per_split(cnt, nsplit_max);
    return max(1, (cnt + per - 1) / per);
}
__global__ __launch_bounds__(512) void k_hw_gather(WArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave & 1, wave_n = wave >> 1;           // 128 channels x 64 columns per wave
    int l = 0;
#pragma unroll
    for (int k = 1; k < SSD_MAX_LEVELS; ++k) l += (k < a.levels && (int)blockIdx.x >= a.lv[k].blk0) ? 1 : 0;
    const WLevel lv = a.lv[l];
    const int ktot = 9 * lv.Cin, cpt = lv.Cin >> 3, nchunks = 9 * cpt;
    const int ctiles = (ktot + 255) >> 8, mtiles = (lv.cout + 255) >> 8, tiles = ctiles * mtiles;
    const int local = blockIdx.x - lv.blk0;
    const int split = local / tiles, tile = local - split * tiles;
    const int bx = tile % ctiles, by = tile / ctiles;
    const int col0 = bx * 256, co0 = by * 256;
    const int cnt = a.count[l];
    if (split >= hw_active_splits(cnt, lv.nsplit)) return;     // (split 0 always runs: zero rows -> zero slab)
    const int m_per_split = hw_rows_per_split(cnt, lv.nsplit);
    const int m_begin = min(cnt, split * m_per_split);
    const int m_end = min(cnt, m_begin + m_per_split);
    const int hw = lv.H * lv.W;

    const __amdgpu_buffer_rsrc_t dyres = __builtin_amdgcn_make_buffer_rsrc((void*)lv.rows, 0, (unsigned)cnt * (unsigned)lv.npad * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)lv.x, 0, (unsigned)a.B * hw * lv.Cin * 2u, 0x00020000);
    const int drow = lane >> 5;
    int rowj[4];
    unsigned dycol[4], xcol[4];
    int xkh[4], xkw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 2 * (wave + 8 * j) + drow;
        rowj[j] = row;
        const int pc = lane & 31;
        const int lg = ((pc >> 1) & 8) | (((pc >> 1) ^ row) & 7);
        const int ch = (lg * 2 + (pc & 1)) * 8;
        dycol[j] = co0 + ch < lv.npad ? (unsigned)(co0 + ch) * 2u : OOB;
        const int q = (col0 + ch) >> 3;
        if (q < nchunks) {
            const int tap = q / cpt;
            xcol[j] = (unsigned)(q - tap * cpt) * 16u;
            xkh[j] = tap / 3;
            xkw[j] = tap - xkh[j] * 3;
        } else {
            xcol[j] = OOB; xkh[j] = 0; xkw[j] = 0;
        }
    }
    int pixn[4];                                             // flat pixel of this lane's rows of the step to be issued next
    auto fetch_pix = [&](int mstep) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mstep + rowj[j];
            pixn[j] = m < m_end ? lv.por[m] : -1;
        }
    };
    auto issue_dma = [&](int mstep, int buf) {
        char* base = smem + buf * (2 * WT_TILE);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mstep + rowj[j];
            const bool mok = m < m_end;
            const unsigned od = (unsigned)m * (unsigned)lv.npad * 2u + dycol[j];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dyres, (lds_void*)(base + (wave + 8 * j) * 1024), 16,
                                                     (mok && dycol[j] != OOB) ? od : OOB, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pix = pixn[j];
            bool ok = pix >= 0 && xcol[j] != OOB;
            const int pp = ok ? pix : 0;
            const int b = pp / hw, rem = pp - b * hw;
            const int oy = rem / lv.W, ox = rem - oy * lv.W;
            const int iy = oy - 1 + xkh[j], ix = ox - 1 + xkw[j];
            ok = ok && (unsigned)iy < (unsigned)lv.H && (unsigned)ix < (unsigned)lv.W;
            const unsigned off = (unsigned)((b * lv.H + iy) * lv.W + ix) * (unsigned)lv.Cin * 2u + xcol[j];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void*)(base + WT_TILE + (wave + 8 * j) * 1024), 16,
                                                     ok ? off : OOB, 0, 0, 0);
        }
    };

    f32x4_t acc[8][4];
    f32x4_t accb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        accb[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[q][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = bx == 0 && wave_n == 0;
    bf16x8_t ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;

    const int gq = lane >> 4, li = lane & 15;
    const int kk0 = (gq >> 1) * 16 + (gq & 1) * 4 + (li >> 2);
    const int key = kk0 & 7;
    int abase[8], bbase[4];
#pragma unroll
    for (int q = 0; q < 8; ++q) abase[q] = kk0 * 512 + ((wave_m * 8 + (q ^ key)) << 5) + (li & 3) * 8;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int gl = wave_n * 4 + c;
        bbase[c] = WT_TILE + kk0 * 512 + (((gl & 8) | ((gl & 7) ^ key)) << 5) + (li & 3) * 8;
    }
    auto rd = [&](int addr) { return lds_read_tr16_scoped(smem + addr, smem); };

    const int nsteps = (m_end - m_begin + 63) / 64;
    if (nsteps > 0) {
        fetch_pix(m_begin);
        issue_dma(m_begin, 0);
        fetch_pix(m_begin + 64);
    }
    auto run = [&](auto bias_tag) {
        constexpr bool BIAS = decltype(bias_tag)::value;
        for (int st = 0; st < nsteps; ++st) {
            const int cur = st & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (st + 1 < nsteps) {
                issue_dma(m_begin + (st + 1) * 64, cur ^ 1);
                fetch_pix(m_begin + (st + 2) * 64);
            }
            const int boff = cur * (2 * WT_TILE);
            int ab[8], bb[4];
#pragma unroll
            for (int q = 0; q < 8; ++q) ab[q] = abase[q] + boff;
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
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int half = 0; half < 2; ++half)
                        reinterpret_cast<s16x4_t*>(&fa[q])[half] = rd(ab[q] + ksub * 16384 + half * 4096);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[q][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[q], fb[c], acc[q][c], 0, 0, 0);
                    if constexpr (BIAS) accb[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[q], ones, accb[q], 0, 0, 0);
                }
            }
        }
    };
    if (do_bias) run(std::true_type{}); else run(std::false_type{});

    float* out = lv.slab_w + (long long)split * lv.npad * ktot;
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = col0 + wave_n * 64 + c * 16 + (lane & 15);
            if (col >= ktot) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + wave_m * 128 + q * 16 + (lane >> 4) * 4 + j;
                if (co < lv.npad) out[(long long)co * ktot + col] = acc[q][c][j];
            }
        }
    if (do_bias && (lane & 15) == 0) {
        float* ob = lv.slab_b + (long long)split * lv.npad;
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + wave_m * 128 + q * 16 + (lane >> 4) * 4 + j;
                if (co < lv.npad) ob[co] = accb[q][j];
            }
    }
}

// dW / dbias of every level = its slabs added in split order
struct RLevel {
    const float* slab_w;
    const float* slab_b;
    float* dw;
    float* db;
    long long sw, nw;                        // slab stride (npad*ktot), elements to reduce (cout*ktot)
    int sb, nb, nsplit;
    int blk0, nbw;                           // first workgroup of the level, workgroups of its weight part
};
struct RArgs {
    int levels;
    RLevel lv[SSD_MAX_LEVELS];
    const int* count;
};

__global__ __launch_bounds__(256) void k_hw_reduce(RArgs a) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < SSD_MAX_LEVELS; ++k) l += (k < a.levels && (int)blockIdx.x >= a.lv[k].blk0) ? 1 : 0;
    const RLevel lv = a.lv[l];
    const int blk = blockIdx.x - lv.blk0;
    const int nsplit = hw_active_splits(a.count[l], lv.nsplit);
    if (blk < lv.nbw) {
        const long long i = ((long long)blk * 256 + threadIdx.x) * 4;
        if (i >= lv.nw) return;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int z = 0; z < nsplit; ++z) {
            const float4 v = *reinterpret_cast<const float4*>(lv.slab_w + (long long)z * lv.sw + i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(lv.dw + i) = s;
    } else {
        const int i = (blk - lv.nbw) * 256 + threadIdx.x;
        if (i >= lv.nb || !lv.db) return;
        float s = 0.f;
        for (int z = 0; z < nsplit; ++z) s += lv.slab_b[(long long)z * lv.sb + i];
        lv.db[i] = s;
    }
}

OnceLds g_once_hz, g_once_hw;

int check_heads(const ssd_head_grads* hg, const ssd_head_layers* hl, int B) {
    if (!hg || !hl || B <= 0) return SSD_ERR_VALUE;
    if (hg->levels <= 0 || hg->levels > SSD_MAX_LEVELS || hl->levels != hg->levels || !hg->count) return SSD_ERR_VALUE;
    for (int l = 0; l < hg->levels; ++l) {
        if (hl->H[l] <= 0 || hl->W[l] <= 0 || hl->H[l] * hl->W[l] != hg->hw[l]) return SSD_ERR_VALUE;
        if (hl->cout[l] <= 0 || hl->cout[l] > hg->npad[l] || (hg->npad[l] & 7)) return SSD_ERR_VALUE;
        if (hl->Cin[l] <= 0 || hl->Cin[l] % 128) return SSD_ERR_UNSUPPORTED;
        if (!hg->rows[l] || !hg->row_of_pixel[l] || !hg->pixel_of_row[l]) return SSD_ERR_VALUE;
        // 32-bit buffer offsets
        if ((long long)B * hg->hw[l] * hl->Cin[l] * 2 >= (1ll << 32) || (long long)B * hg->hw[l] * hg->npad[l] * 2 >= (1ll << 32)) return SSD_ERR_UNSUPPORTED;
    }
    return SSD_OK;
}

int wsplits(int B, int hw, int tiles) {
    // fixed on the host (the row count is not known there): enough workgroups for ~2 rounds on 256 CUs if a tenth of the pixels
    // carry a gradient, never more than one split per 64 rows of that estimate
    const int est_steps = max(1, B * hw / 10 / 64);
    int ns = max(1, min(16, 512 / max(1, tiles)));
    return max(1, min(ns, est_steps));
}

}  // namespace

extern "C" {

size_t ssd_heads_bwd_data_sparse_workspace_bytes(int B, const ssd_head_layers* hl) {
    if (!hl || B <= 0) return 0;
    size_t tot = 0;
    for (int l = 0; l < hl->levels && l < SSD_MAX_LEVELS; ++l)
        tot += ssd_align_up((size_t)B * hl->H[l] * hl->W[l] * 9 * hl->Cin[l] * sizeof(float), 256);
    return tot;
}

int ssd_heads_bwd_data_sparse_levels(const ssd_head_grads* hg, const ssd_head_layers* hl, int B, unsigned level_mask, int prezeroed,
                                     void* ws, size_t ws_bytes, void* stream) {
    const int rc = check_heads(hg, hl, B);
    if (rc != SSD_OK) return rc;
    if (!ws || ws_bytes < ssd_heads_bwd_data_sparse_workspace_bytes(B, hl)) return SSD_ERR_WORKSPACE;
    level_mask &= (1u << hg->levels) - 1u;
    if (!level_mask) return SSD_ERR_VALUE;
    hipStream_t s = (hipStream_t)stream;
    ZArgs za;
    CArgs ca;
    za.levels = ca.levels = hg->levels;
    za.mask = level_mask;
    za.count = hg->count;
    ca.B = B;
    ca.prezeroed = prezeroed ? 1 : 0;
    char* p = static_cast<char*>(ws);
    int blk = 0, cap_tiles = 0;
    for (int l = 0; l < SSD_MAX_LEVELS; ++l) {
        if (l >= hg->levels) {
            za.lv[l] = ZLevel{nullptr, nullptr, nullptr, 8, ZT};
            ca.lv[l] = CLevel{nullptr, nullptr, nullptr, nullptr, nullptr, 1, 1, 8, blk};
            continue;
        }
        if (!hl->w_tap[l] || !hl->dx[l]) return SSD_ERR_VALUE;
        const int hw = hg->hw[l], N = 9 * hl->Cin[l];
        float* z = reinterpret_cast<float*>(p);                  // (every level keeps its own slice whatever the mask: two calls
        p += ssd_align_up((size_t)B * hw * N * sizeof(float), 256);   //  with disjoint masks may share `ws` on two streams)
        za.lv[l] = ZLevel{(const bf16_raw*)hg->rows[l], (const bf16_raw*)hl->w_tap[l], z, hg->npad[l], N};
        ca.lv[l] = CLevel{z, hg->row_of_pixel[l], (const unsigned char*)hl->relu_bits[l], (const bf16_raw*)hl->relu_src[l],
                          (bf16_raw*)hl->dx[l], hl->H[l], hl->W[l], hl->Cin[l], blk};
        if (!((level_mask >> l) & 1u)) continue;                 // no workgroups: the level lookup skips an empty range
        blk += (int)(((long long)((B * hw + C2I_PIX - 1) / C2I_PIX) * (hl->Cin[l] >> 3) + 255) / 256);
        cap_tiles += ((B * hw + ZT - 1) / ZT) * (N / ZT);
    }
    if (ensure_lds(g_once_hz, (const void*)k_hz_gemm, 2 * ZBUF) != 0) return SSD_ERR_LAUNCH;
    hipLaunchKernelGGL(k_hz_gemm, dim3(min(cap_tiles, 2048)), dim3(256), 2 * ZBUF, s, za);
    hipLaunchKernelGGL(k_hz_col2im, dim3(blk), dim3(256), 0, s, ca);
    return ssd_launch_status();
}

int ssd_heads_bwd_data_sparse(const ssd_head_grads* hg, const ssd_head_layers* hl, int B, void* ws, size_t ws_bytes,
                              void* stream) {
    return ssd_heads_bwd_data_sparse_levels(hg, hl, B, ~0u, 0, ws, ws_bytes, stream);
}

size_t ssd_heads_bwd_weight_sparse_workspace_bytes(int B, const ssd_head_grads* hg, const ssd_head_layers* hl) {
    if (!hl || !hg || B <= 0) return 0;
    size_t tot = 0;
    for (int l = 0; l < hl->levels && l < SSD_MAX_LEVELS; ++l) {
        const int ktot = 9 * hl->Cin[l];
        const int tiles = ((ktot + 255) >> 8) * ((hl->cout[l] + 255) >> 8);
        const int ns = wsplits(B, hg->hw[l], tiles);
        tot += ssd_align_up((size_t)ns * hg->npad[l] * ((size_t)ktot + 1) * sizeof(float), 256);
    }
    return tot;
}

int ssd_heads_bwd_weight_sparse(const ssd_head_grads* hg, const ssd_head_layers* hl, int B, void* ws, size_t ws_bytes,
                                void* stream) {
    const int rc = check_heads(hg, hl, B);
    if (rc != SSD_OK) return rc;
    if (!ws || ws_bytes < ssd_heads_bwd_weight_sparse_workspace_bytes(B, hg, hl)) return SSD_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    WArgs wa;
    RArgs ra;
    wa.levels = ra.levels = hg->levels;
    wa.B = B;
    wa.count = ra.count = hg->count;
    char* p = static_cast<char*>(ws);
    int blk = 0, rblk = 0;
    for (int l = 0; l < SSD_MAX_LEVELS; ++l) {
        if (l >= hg->levels) {
            wa.lv[l] = WLevel{nullptr, nullptr, nullptr, nullptr, nullptr, 1, 1, 8, 8, 1, 1, blk};
            ra.lv[l] = RLevel{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 1, rblk, 0};
            continue;
        }
        if (!hl->x[l] || !hl->dw[l]) return SSD_ERR_VALUE;
        const int ktot = 9 * hl->Cin[l], npad = hg->npad[l], cout = hl->cout[l];
        const int tiles = ((ktot + 255) >> 8) * ((cout + 255) >> 8);
        const int ns = wsplits(B, hg->hw[l], tiles);
        float* slab_w = reinterpret_cast<float*>(p);
        float* slab_b = slab_w + (size_t)ns * npad * ktot;
        p += ssd_align_up((size_t)ns * npad * ((size_t)ktot + 1) * sizeof(float), 256);
        wa.lv[l] = WLevel{(const bf16_raw*)hl->x[l], (const bf16_raw*)hg->rows[l], hg->pixel_of_row[l], slab_w, slab_b,
                          hl->H[l], hl->W[l], hl->Cin[l], npad, cout, ns, blk};
        blk += tiles * ns;
        const long long nw = (long long)cout * ktot;
        const int nbw = (int)((nw / 4 + 255) / 256), nbb = hl->dbias[l] ? (cout + 255) / 256 : 0;
        ra.lv[l] = RLevel{slab_w, slab_b, hl->dw[l], hl->dbias[l], (long long)npad * ktot, nw, npad, cout, ns, rblk, nbw};
        rblk += nbw + nbb;
    }
    if (ensure_lds(g_once_hw, (const void*)k_hw_gather, 4 * WT_TILE) != 0) return SSD_ERR_LAUNCH;
    hipLaunchKernelGGL(k_hw_gather, dim3(blk), dim3(512), 4 * WT_TILE, s, wa);
    hipLaunchKernelGGL(k_hw_reduce, dim3(rblk), dim3(256), 0, s, ra);
    return ssd_launch_status();
}

}  // extern "C"
