// Winograd F(2x2, 3x3) convolution for gfx950: the 3x3 / stride 1 / pad 1 layers of the VGG trunk (forward and data
// gradient; models/ssd_model.py:74-171 builds them, tape.gradient :248 differentiates them) at 2.25x fewer MFMA MACs
// than the direct kernels of conv.hip.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A     per 2x2 output tile, summed over input channels
//
//   * weights are transformed once per optimizer step (k_wino_weights): U[xi] = G g G^T in fp32, rounded to fp16 and
//     stored in MFMA A-fragment order, so the convolution kernel fetches a fragment as one contiguous 1 KB wave load;
//   * the convolution kernel works on the same 16x16-pixel blocks / 18x18 halo patches (LDS-DMA, 32-channel chunks,
//     96-byte rows) as k_conv3x3_patch32: a block is 8x8 tiles.  Per chunk every thread transforms one (tile, 4 channels)
//     item -- bf16 -> fp16 is exact in fp16's normal range, the B^T d B adds run as packed fp16 -- into V[16][64 tiles][32 ch]
//     in LDS; wave w then owns the two transform points xi = 2w, 2w+1 and accumulates M[xi] += U[xi] V[xi] for all
//     64 tiles x 64 output channels on 16x16x32 fp16 MFMAs (128 accumulator registers);
//   * operand type: fp16, not bf16.  With bf16-rounded V and U the error of a whole forward/backward pass sits on the 1e-2
//     acceptance bound (the transforms amplify the 8-bit mantissa's rounding); with fp16's 11 bits it is 6.6e-3, the same
//     class as the direct bf16 kernels.  The fp16 range is guarded by a power-of-two operand scale
//     (2^in_shift on conversion, undone on the fp32 accumulators), exact in both types;
//   * the inverse transform A^T M A needs all 16 points of a tile, which live in 8 different waves: the accumulators
//     cross through LDS 16 channels at a time, and the finished bf16 tile leaves through the shared staged store
//     (bias / ReLU / fused 2x2 pooling forward, accumulate / ReLU mask for the data gradient).
#include "conv_common.h"

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int WN_PITCH = 96;                               // bytes per patch pixel / per V row (64 B of data + 32 B pad)
constexpr int WN_PATCH = 32 * 1024;                        // 18 x 18 px x 96 B = 31104 B, rounded to 32 DMA instructions
constexpr int WN_PW = 18;
constexpr int WN_VPLANE = 64 * WN_PITCH;                   // V[xi]: 64 tiles x 96 B
constexpr int WN_OFF_V = 2 * WN_PATCH;
constexpr int WN_LDS = WN_OFF_V + 16 * WN_VPLANE;          // 163840: the whole LDS of a CU
constexpr int WN_MPITCH = 80;                              // output stage: M[xi][64 tiles][16 ch] fp32, 64 B + 16 B pad
constexpr int WN_MPLANE = 64 * WN_MPITCH;
constexpr int WN_OFF_IMG = 16 * WN_MPLANE;                 // [256 px][64 ch] bf16 tile image (staged_store layout) behind M
static_assert(WN_OFF_IMG + 256 * 128 <= WN_LDS, "output stage must fit");

// U[xi][n][c] = (G g G^T)[xi] of w[n][kh][kw][c], fp16, in fragment order:
// ((xi * N/16 + n/16) * C/32 + c/32) * 64 + lane) * 8 + (c & 7),  lane = (n & 15) + 16 * ((c & 31) >> 3)
__global__ void k_wino_weights(const bf16_raw* __restrict__ w, _Float16* __restrict__ u, int N, int C, float scale) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)N * C) return;
    const int n = (int)(idx / C), c = (int)(idx - (long long)n * C);
    float g[3][3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) g[kh][kw] = bf2f(w[((long long)(n * 3 + kh) * 3 + kw) * C + c]) * scale;
    float t[4][3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        t[0][kw] = g[0][kw];
        t[1][kw] = 0.5f * (g[0][kw] + g[1][kw] + g[2][kw]);
        t[2][kw] = 0.5f * (g[0][kw] - g[1][kw] + g[2][kw]);
        t[3][kw] = g[2][kw];
    }
    const int lane = (n & 15) + 16 * ((c & 31) >> 3);
    const long long base = ((long long)(n >> 4) * (C >> 5) + (c >> 5)) * 512 + lane * 8 + (c & 7);
    const long long plane = (long long)(N >> 4) * (C >> 5) * 512;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float r[4] = {t[a][0], 0.5f * (t[a][0] + t[a][1] + t[a][2]), 0.5f * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
#pragma unroll
        for (int b = 0; b < 4; ++b) u[(a * 4 + b) * plane + base] = (_Float16)r[b];
    }
}

struct WinoArgs {
    int tiles_x, tiles_y, nblocks, rowflat;
    int in_shift;                              // operands enter fp16 as x * 2^in_shift
    float out_scale;                           // 2^-(in_shift + weight shift), applied to the fp32 result
};

// V = B^T d B of one (tile, 2 channels) item: 16 patch pixels in, 16 transform points out (fp16).  The two pointers are
// __restrict__ on purpose: inlined, the LDS accesses then carry alias scopes, and the compiler's waitcnt pass only orders
// LDS reads behind an in-flight LDS-DMA when the read has NO scope information -- without them every first LDS read of
// an iteration waited for the NEXT chunk's patch DMA (vmcnt(0)), i.e. no prefetch at all.  The kernel orders DMA and
// reads itself (s_waitcnt vmcnt + barrier).
template <bool SCALE_IN>
__device__ __forceinline__ void wino_transform(const char* __restrict__ src, char* __restrict__ dst, int in_shift) {
    f16x2_t d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const unsigned raw = *reinterpret_cast<const unsigned*>(src + (a * WN_PW + bb) * WN_PITCH);
            unsigned lo = raw << 16, hi = raw & 0xffff0000u;
            if constexpr (SCALE_IN) {              // x * 2^in_shift on the exponent field; zero stays far below fp16's range
                lo += (unsigned)in_shift << 23;
                hi += (unsigned)in_shift << 23;
            }
            d[a][bb] = __builtin_bit_cast(f16x2_t, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(lo), __uint_as_float(hi)));
        }
    f16x2_t t[4][4];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
        t[0][bb] = d[0][bb] - d[2][bb];
        t[1][bb] = d[1][bb] + d[2][bb];
        t[2][bb] = d[2][bb] - d[1][bb];
        t[3][bb] = d[1][bb] - d[3][bb];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const f16x2_t v[4] = {t[a][0] - t[a][2], t[a][1] + t[a][2], t[a][2] - t[a][1], t[a][1] - t[a][3]};
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
            *reinterpret_cast<unsigned*>(dst + (a * 4 + bb) * WN_VPLANE) = __builtin_bit_cast(unsigned, v[bb]);
    }
}

// B fragments (V[xi] rows of 16 tiles) of one transform point, same reason for the __restrict__ pair
__device__ __forceinline__ void wino_load_v(const char* __restrict__ src, const char* __restrict__ other, f16x8_t (&fv)[2]) {
    (void)other;
#pragma unroll
    for (int p = 0; p < 2; ++p) fv[p] = *reinterpret_cast<const f16x8_t*>(src + p * 16 * WN_PITCH);
}

template <int EPI, bool SCALE_IN>
__global__ __launch_bounds__(512) void k_conv3x3_wino(const bf16_raw* __restrict__ x, const _Float16* __restrict__ u, ConvGeom g,
                                                      Epilogue ep, WinoArgs wa) {
    constexpr int BN = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // block order as in k_conv3x3_patch32: the channel tiles of one pixel block are consecutive on ONE XCD
    const int ntn = g.N / BN;
    const int kx = blockIdx.x >> 3;
    const int pblock = (kx / ntn) * 8 + (blockIdx.x & 7);
    if (pblock >= wa.nblocks) return;
    const int n0 = (kx % ntn) * BN;
    const int rowflat = wa.rowflat;
    int b = 0, y0 = 0, x0 = 0;
    {
        int t = pblock;
        const int tx = t % wa.tiles_x; t /= wa.tiles_x;
        x0 = tx * 16;
        if (rowflat) { y0 = t * 16; }
        else { const int ty = t % wa.tiles_y; b = t / wa.tiles_y; y0 = ty * 16; }
    }
    // block row r (-1 .. 16 with the halo) -> row index into [B * H] or -1 (padding); rowflat: the rows of all images
    // form one strip with ONE zero row between images (period H + 1, even for the odd map heights it is used on)
    auto image_row = [&](int r) {
        if (!rowflat) { const int y = y0 + r; return (unsigned)y < (unsigned)g.H ? b * g.H + y : -1; }
        const int R = y0 + r;
        if (R < 0) return -1;
        const int bb = fdiv(R, g.d_h1), yy = R - bb * (g.H + 1);
        return (bb < g.B && yy < g.H) ? bb * g.H + yy : -1;
    };

    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)g.B * g.H * g.W * g.C * 2u, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    // patch DMA: instruction i = wave + 8j (j < 4) fills slots 64i .. 64i+63; slot q -> pixel q / 6, 16-byte piece q % 6
    unsigned pvo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = (wave + 8 * j) * 64 + lane;
        const int pp = q / 6, sl = q - pp * 6;
        const int py = pp / WN_PW, px = pp - py * WN_PW;
        const int ix = x0 - 1 + px;
        const int ir = pp < WN_PW * WN_PW ? image_row(py - 1) : -1;
        const int pix = (ir >= 0 && (unsigned)ix < (unsigned)g.W) ? ir * g.W + ix : -1;
        pvo[j] = (sl < 4 && pix >= 0) ? ((unsigned)pix * (unsigned)g.C + (unsigned)(sl * 8)) * 2u : OOB;
    }
    auto dma_patch = [&](int chunk, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void*)(smem + buf * WN_PATCH + (wave + 8 * j) * 1024), 16, pvo[j],
                                                     chunk * 64, 0, 0);
    };
    const int nchunk = g.C >> 5;

    // transform item of this thread within a half block (tile rows 0-3 / 4-7): tile (ty = wave >> 1, tx = 4 (wave & 1) + (lane >> 4)),
    // channels 2 (lane & 15), +1 of the chunk.  V row of tile (ty, tx) = 8 ty + 2 (tx & 3) + (tx >> 2): a wave's four tiles
    // are adjacent in the patch (conflict-free reads) and two rows apart in V (conflict-free writes at the 96-byte pitch)
    const int t_src = ((2 * (wave >> 1)) * WN_PW + 2 * (4 * (wave & 1) + (lane >> 4))) * WN_PITCH + (lane & 15) * 4;
    const int t_dst = WN_OFF_V + ((wave >> 1) * 8 + 2 * (lane >> 4) + (wave & 1)) * WN_PITCH + (lane & 15) * 4;
    constexpr int T_HALF_SRC = 8 * WN_PW * WN_PITCH;       // second half: 4 tile rows = 8 pixel rows further down
    constexpr int T_HALF_DST = 32 * WN_PITCH;
    // MFMA operands of this wave: xi = 2 wave + i; B fragment p = tiles 16p .. 16p+15 (lane & 15), channels 8 (lane >> 4) ..
    const int v_src = WN_OFF_V + (2 * wave) * WN_VPLANE + (lane & 15) * WN_PITCH + (lane >> 4) * 16;
    // U fragments through a buffer descriptor: one per-lane offset register, the (xi, channel tile, chunk) part is scalar
    const unsigned uplane = (unsigned)(g.N >> 4) * (unsigned)nchunk * 1024u;     // bytes per transform point
    const __amdgpu_buffer_rsrc_t ures = __builtin_amdgcn_make_buffer_rsrc((void*)u, 0, 16u * uplane, 0x00020000);
    const unsigned ubase = (unsigned)(2 * wave) * uplane + (unsigned)(n0 >> 4) * (unsigned)nchunk * 1024u;
    const unsigned ulane = (unsigned)lane * 16u;

    f32x4_t acc[2][4][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[i][c][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // Software pipeline over half blocks (A = tile rows 0-3, B = 4-7), two phases per chunk, one barrier each:
    //   phase 1 of chunk c:  MFMAs of B(c-1)  ||  transform A(c)      (+ issue patch DMA and U loads of chunk c+1)
    //   phase 2 of chunk c:  MFMAs of A(c)    ||  transform B(c)
    // so every phase has matrix work and VALU / LDS work to overlap, within a wave and between the two waves of a SIMD.
    // U fragments live in two register sets (chunk parity); the patch of chunk c+1 lands in the other LDS buffer.
    auto load_u = [&](u32x4_t (&f)[2][4], int chunk) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                f[i][c] = __builtin_amdgcn_raw_buffer_load_b128(ures, ulane, ubase + (unsigned)i * uplane + (unsigned)(c * nchunk + chunk) * 1024u, 0);
    };
    const int abl = g.ablate;                                // dev only (SSD_ABLATE): 1 no transform, 2 no MFMA, 4 no epilogue, 8 no DMA
    auto mfma_half = [&](int half, u32x4_t (&f)[2][4]) {
        if (abl & 2) return;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f16x8_t fv[2];
            wino_load_v(smem + v_src + i * WN_VPLANE + half * T_HALF_DST, smem, fv);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    f32x4_t& a4 = half ? acc[i][c][2 + p] : acc[i][c][p];
                    a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, f[i][c]), fv[p], a4, 0, 0, 0);
                }
        }
    };
    auto transform_half = [&](int half, int chunk) {
        if (abl & 1) return;
        wino_transform<SCALE_IN>(smem + (chunk & 1) * WN_PATCH + t_src + half * T_HALF_SRC, smem + t_dst + half * T_HALF_DST, wa.in_shift);
    };
    // one chunk: fcur = U(chunk) (landed), fnext receives U(chunk + 1)
    auto step = [&](int chunk, u32x4_t (&fcur)[2][4], u32x4_t (&fprev)[2][4]) {
        // ---- phase 1: V_B still holds chunk - 1 (its MFMAs run now), patch(chunk) has landed
        {   // patch / U of chunk + 1 (the last iteration re-fetches its own: no divergent register merge).  The other patch
            // buffer is free: transform B(chunk - 1) finished before the last barrier.  fprev = U(chunk - 1) is needed by the
            // B MFMAs of this phase, so the U loads of chunk + 1 (into fprev's registers) are issued after them
            const int nx = chunk + 1 < nchunk ? chunk + 1 : chunk;
            if (!(abl & 8)) dma_patch(nx, (chunk + 1) & 1);
            if (chunk > 0) mfma_half(1, fprev);
            transform_half(0, chunk);
            load_u(fprev, nx);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- phase 2
        mfma_half(0, fcur);
        transform_half(1, chunk);
        // the next phase reads patch(chunk + 1).  vmcnt(0), not a count that would leave the U loads in flight: the compiler is
        // free to move those loads across the DMA issue and across this statement, so their position in the queue is unknown
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    u32x4_t fa[2][4], fb[2][4];
    dma_patch(0, 0);
    load_u(fa, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int chunk = 0; chunk < nchunk; chunk += 2) {        // nchunk is even (C % 64 == 0)
        step(chunk, fa, fb);
        step(chunk + 1, fb, fa);
    }
    mfma_half(1, fb);                                        // B of the last chunk (odd index: U in fb)

    if (abl & 4) return;
    // inverse transform, 16 output channels (one channel tile c) per pass
    const int o_tile = tid >> 3, o_cp = tid & 7;            // item: V row (tile), channel pair
    const int o_ty = o_tile >> 3, o_tx = ((o_tile & 7) >> 1) + 4 * (o_tile & 1);
    char* img = smem + WN_OFF_IMG;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        __builtin_amdgcn_s_barrier();                        // V (first pass) / the previous pass's M planes are no longer read
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 4; ++p)
                *reinterpret_cast<f32x4_t*>(smem + (2 * wave + i) * WN_MPLANE + (p * 16 + (lane & 15)) * WN_MPITCH + (lane >> 4) * 16) =
                    acc[i][c][p];
        __syncthreads();
        float2 m[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
                m[a][bb] = *reinterpret_cast<const float2*>(smem + (a * 4 + bb) * WN_MPLANE + o_tile * WN_MPITCH + o_cp * 8);
        const int col = c * 16 + o_cp * 2;
        float b2[2] = {0.f, 0.f};
        if constexpr (EPI == EPI_FWD) {
            if (ep.bias) { b2[0] = ep.bias[n0 + col]; b2[1] = ep.bias[n0 + col + 1]; }
        }
        float y[2][2][2];                                    // [dy][dx][channel]
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float s[4], dd[4];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const float m0 = h ? m[0][bb].y : m[0][bb].x, m1 = h ? m[1][bb].y : m[1][bb].x;
                const float m2 = h ? m[2][bb].y : m[2][bb].x, m3 = h ? m[3][bb].y : m[3][bb].x;
                s[bb] = m0 + m1 + m2;
                dd[bb] = m1 - m2 - m3;
            }
            y[0][0][h] = (s[0] + s[1] + s[2]) * wa.out_scale + b2[h];
            y[0][1][h] = (s[1] - s[2] - s[3]) * wa.out_scale + b2[h];
            y[1][0][h] = (dd[0] + dd[1] + dd[2]) * wa.out_scale + b2[h];
            y[1][1][h] = (dd[1] - dd[2] - dd[3]) * wa.out_scale + b2[h];
        }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float v0 = y[dy][dx][0], v1 = y[dy][dx][1];
                if constexpr (EPI == EPI_FWD) {
                    if (ep.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                }
                const int row = (2 * o_ty + dy) * 16 + 2 * o_tx + dx;
                *reinterpret_cast<unsigned*>(img + row * (BN * 2) + ((((col >> 3) ^ row) & 7) << 4) + (col & 7) * 2) =
                    (unsigned)f2bf(v0) | ((unsigned)f2bf(v1) << 16);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    auto row_to_m = [&](int row) {
        const int ir = image_row(row >> 4), xx = x0 + (row & 15);
        return (ir >= 0 && xx < g.Wo) ? ir * g.Wo + xx : -1;
    };
    auto pool_index = [&](int py, int px) -> long long {
        if (rowflat) return -1;
        const int gy = (y0 >> 1) + py, gx = (x0 >> 1) + px;
        return (gy < ep.pool_h && gx < ep.pool_w) ? ((long long)b * ep.pool_h + gy) * ep.pool_w + gx : -1;
    };
    staged_store<EPI, 256, BN, 512>(img, g, ep, n0, tid, row_to_m, pool_index);
}

bool wino_shape_ok(int B, int H, int W, int C, int N) {
    if (B <= 0 || H < 16 || W < 16 || C <= 0 || N <= 0) return false;
    if (C % 64 || N % 64) return false;
    if ((long long)B * H * W * C >= (1ll << 31) - 16 || (long long)B * H * W * N >= (1ll << 31) - 16) return false;
    return true;
}

template <int EPI>
int launch_wino(const void* x, const void* u, const ConvGeom& g, const Epilogue& ep, int in_shift, int w_shift, hipStream_t s) {
    WinoArgs wa;
    wa.tiles_x = (g.Wo + 15) / 16;
    wa.tiles_y = (g.Ho + 15) / 16;
    const unsigned strip_rows = (unsigned)(((long long)g.B * (g.H + 1) + 15) / 16);
    // one strip of rows over all images when that needs fewer blocks; needs an even period (tiles are 2 rows) and no pooling
    wa.rowflat = (!ep.pool_out && ((g.H + 1) & 1) == 0 && ssd_knob("SSD_CONV_PATCH_ROWFLAT", 1) &&
                  strip_rows < (unsigned)(wa.tiles_y * g.B)) ? 1 : 0;
    wa.nblocks = wa.rowflat ? (int)strip_rows * wa.tiles_x : wa.tiles_x * wa.tiles_y * g.B;
    wa.in_shift = in_shift;
    wa.out_scale = ldexpf(1.f, -(in_shift + w_shift));
    const unsigned ntn = (unsigned)(g.N / 64);
    const dim3 grid(8 * ntn * (((unsigned)wa.nblocks + 7) / 8));
    if (in_shift) {
        auto kern = k_conv3x3_wino<EPI, true>;
        static OnceLds set; if (ensure_lds(set, reinterpret_cast<const void*>(kern), WN_LDS) != 0) return SSD_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, grid, dim3(512), WN_LDS, s, static_cast<const bf16_raw*>(x), static_cast<const _Float16*>(u), g, ep, wa);
    } else {
        auto kern = k_conv3x3_wino<EPI, false>;
        static OnceLds set; if (ensure_lds(set, reinterpret_cast<const void*>(kern), WN_LDS) != 0) return SSD_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, grid, dim3(512), WN_LDS, s, static_cast<const bf16_raw*>(x), static_cast<const _Float16*>(u), g, ep, wa);
    }
    return ssd_launch_status();
}

}  // namespace

extern "C" {

size_t ssd_wino_weights_bytes(int Cout, int Cin) { return (size_t)16 * Cout * Cin * 2; }

int ssd_wino_weights(const void* w, void* u, int Cout, int Cin, int w_shift, void* stream) {
    // w: [Cout][3][3][Cin] bf16 (forward weights, or ssd_weight_transpose's output for the data gradient with the roles of
    // Cout / Cin swapped); u: ssd_wino_weights_bytes(Cout, Cin) bytes
    if (!w || !u || Cout <= 0 || Cin <= 0 || Cout % 16 || Cin % 64 || w_shift < -24 || w_shift > 24) return SSD_ERR_VALUE;
    const long long n = (long long)Cout * Cin;
    hipLaunchKernelGGL(k_wino_weights, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const bf16_raw*>(w), static_cast<_Float16*>(u), Cout, Cin, ldexpf(1.f, w_shift));
    return ssd_launch_status();
}

int ssd_conv3x3_wino_supported(int B, int H, int W, int Cin, int Cout) { return wino_shape_ok(B, H, W, Cin, Cout) ? 1 : 0; }

int ssd_conv3x3_wino_fwd(const void* x, const void* u, const float* bias, void* y, void* y_pool, void* pool_code, int B, int H, int W,
                         int Cin, int Cout, int relu, int Hp, int Wp, int in_shift, int w_shift, void* stream) {
    if (!x || !u || (!y && !y_pool) || !wino_shape_ok(B, H, W, Cin, Cout) || in_shift < 0 || in_shift > 24) return SSD_ERR_VALUE;
    if (y_pool && (!pool_code || (Hp != H / 2 && Hp != (H + 1) / 2) || (Wp != W / 2 && Wp != (W + 1) / 2))) return SSD_ERR_VALUE;
    const ConvGeom g = make_geom(B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1);
    Epilogue ep = {};
    ep.bias = bias; ep.relu = relu; ep.out = static_cast<bf16_raw*>(y); ep.ldo = Cout;
    if (y_pool) { ep.pool_out = static_cast<bf16_raw*>(y_pool); ep.pool_code = static_cast<unsigned*>(pool_code); ep.pool_h = Hp; ep.pool_w = Wp; }
    return launch_wino<EPI_FWD>(x, u, g, ep, in_shift, w_shift, (hipStream_t)stream);
}

int ssd_conv3x3_wino_bwd_data(const void* dy, const void* u_t, const void* relu_src, void* dx, int B, int H, int W, int Cin,
                              int Cout, int accumulate, int in_shift, int w_shift, void* stream) {
    // dy: [B,H,W,Cout]; u_t: transform of the flipped, transposed weights [Cin][3][3][Cout]; dx, relu_src: [B,H,W,Cin]
    if (!dy || !u_t || !dx || !wino_shape_ok(B, H, W, Cout, Cin) || in_shift < 0 || in_shift > 24) return SSD_ERR_VALUE;
    const ConvGeom g = make_geom(B, H, W, Cout, H, W, Cin, 3, 3, 1, 1, 1, 1);
    Epilogue ep = {};
    ep.out = static_cast<bf16_raw*>(dx); ep.ldo = Cin; ep.mask_src = static_cast<const bf16_raw*>(relu_src); ep.accumulate = accumulate;
    return launch_wino<EPI_DGRAD>(dy, u_t, g, ep, in_shift, w_shift, (hipStream_t)stream);
}

}  // extern "C"
