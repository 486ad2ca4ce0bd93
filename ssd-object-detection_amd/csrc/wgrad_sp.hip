// Weight gradient of a 3x3 / stride 1 / pad 1 convolution whose OUTPUT IS MAX-POOLED (2x2 / stride 2), from the gradient of the
// pooled map and the pooling's winner codes -- the un-pooled gradient as the structured-sparse operand of
// v_smfmac_f32_16x16x64_bf16 (gfx950).
//
// Replaces, for block1_conv2 / block2_conv2 / block3_conv3 (models/ssd_model.py:77-84: Keras VGG16 convolutions in front of a
// MaxPooling2D), the part of tape.gradient (:248) that is   dW = correlate(x, unpool(dP))   where unpool() routes every pooled
// gradient to its window's winner (Keras MaxPool gradient) and leaves the other three positions ZERO.  The dense weight-gradient
// kernel (k_conv3x3_wgrad_patch) multiplies those zeros: 9.95 of its 16.59 GMAC per image.  Here the four pixels of a pooling
// window are four consecutive k of the MFMA, so dY^T is a 1:4 (hence 2:4) sparse A operand: the pooled gradient is the
// compressed value, the winner code is the index -- half the matrix instructions (measured: smfmac 16x16x64 runs at exactly
// 2.0x the dense 16x16x32 rate, LDS-fed too, tools_dev/ubench_smfmac.hip), a quarter of the dY bytes from HBM.
//
// Operand layout of v_smfmac_f32_16x16x64_bf16, probed on the device (tools_dev/smfmac_layout_check.hip; lane = 16 gq + li):
//   B (dense)   lane (gq, li): column li, its 16 values j = 0..15 are k-slots (gq, j)
//   A (sparse)  lane (G, li): row li, 8 values = 4 groups of two kept out of four; group g covers the k-slots
//               (gq = 2 (G & 1) + (g >> 1),  j = 8 (G >> 1) + 4 (g & 1) + 0..3);  index VGPR bits [4g+1:4g] / [4g+3:4g+2] = position of
//               the first / second kept value (low 16 bits with abid = 0)
//   C           lane (gq, li), element e: row 4 gq + e, column li  (as the dense 16x16 MFMAs)
// Here: k-step s of a 16x16-pixel block = window rows 2s, 2s+1 x 8 window columns; B's j = 4q + p is pixel p = 2 dy + dx of window
// (row 2s + (q >> 1), column 4 (q & 1) + PERM[gq]), fetched by ONE transposing read per q whose four lane quads address the four
// pixels of the window; PERM = {0, 2, 1, 3} puts the two windows of a half-wave two columns apart, which keeps the eight pixel
// rows of a read on eight different bank groups under every tap shift (the patch image and its key are k_conv3x3_wgrad_patch's).
//
// A workgroup = 64 output channels x 64 input channels x nine taps over a range of 16x16 blocks.  Per block: pooled-gradient tile [64 windows][64 co] + codes [64][8 words] + 18x18 halo
// patch of x by LDS-DMA (51 KB instead of 73 KB); a short producer pass turns (gradient, code) into the 16 A fragments of the
// block in MFMA operand order (values + index word per lane), which every wave then reads with plain 16 + 4 byte loads.
#include <atomic>
#include <cstdint>
#include <type_traits>
#include "common.h"
#include <hip/hip_bf16.h>
#include "conv_common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((__vector_size__(16 * sizeof(__bf16)))) __bf16 bf16x16_t;

constexpr int SP_PW = 18, SP_PPIX = 18 * 18, SP_PINSTR = (SP_PPIX * 8 + 63) / 64, SP_PBYTES = SP_PINSTR * 1024;   // 41 KB
constexpr int SP_OFF_DP = SP_PBYTES;              // pooled-gradient tile [64 windows][64 co] bf16
constexpr int SP_OFF_CODE = SP_OFF_DP + 8192;     // winner codes [64 windows][8 words]
constexpr int SP_OFF_AV = SP_OFF_CODE + 2048;     // A values  [16 fragments][64 lanes][16 B]
constexpr int SP_OFF_AI = SP_OFF_AV + 16384;      // A indices [16 fragments][64 lanes][4 B]
constexpr int SP_LDS = SP_OFF_AI + 4096;          // 71 KB: two workgroups per CU

__device__ __forceinline__ int sp_key(int px) { return (px >> 1) & 3; }          // = wp_key of k_conv3x3_wgrad_patch
__device__ __forceinline__ int sp_perm(int gq) { return ((gq & 1) << 1) | (gq >> 1); }   // {0, 2, 1, 3}

__device__ __forceinline__ uint4 sp_ld16(const char* __restrict__ p, const char* __restrict__ other) { (void)other; return *reinterpret_cast<const uint4*>(p); }
// (the transposing read through a __restrict__ parameter: inlined, it carries an alias scope.  Without one the compiler orders every
//  LDS read behind ALL LDS-DMA in flight -- s_waitcnt vmcnt(0) in front of the first read after a wave's requests)
typedef __attribute__((address_space(3))) s16x4_t sp_lds_s16x4;
__device__ __forceinline__ s16x4_t sp_tr(sp_lds_s16x4* __restrict__ p, const char* __restrict__ other) {
    (void)other;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
}
__device__ __forceinline__ int sp_ld4(const char* __restrict__ p, const char* __restrict__ other) { (void)other; return *reinterpret_cast<const int*>(p); }

// Four waves per workgroup, TWO workgroups per CU (71 KB of LDS, <= 256 registers): a block's phases -- requests, landing,
// producer, matrix work -- are strictly serial inside a workgroup (everything is single-buffered), and the OTHER workgroup of the
// CU fills them: measured on the first form of this kernel (eight waves, one workgroup per CU, next block's DMA in flight during
// the matrix work) the phases simply added up -- 1.2 us of fixed cost + 0.65 DMA + 1.1 smfmac + 0.6 fragment reads + 0.6-1.0
// producer per block, 3.3-3.9 us against 2.3 us of matrix work in the dense kernel's block -- because all eight waves sat in
// the same phase at the same time.  Wave w owns input-channel tile w, all nine taps, all four output-channel tiles: 36
// accumulator tiles, 144 registers.
__global__ __launch_bounds__(256, 2) void k_conv3x3_wgrad_unpool(const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ dp,
                                                                 const unsigned* __restrict__ code, float* __restrict__ slab_w,
                                                                 float* __restrict__ slab_b, ConvGeom g, int Hp, int Wp, int tiles_x,
                                                                 int tiles_y, int tiles_per_split, int nsplit, int xg, int abl) {
    // g: source = x (B,H,W,C); N = output channels (a multiple of 64)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int ct = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave = input-channel tile
    const int nchunk = g.C >> 6, cotiles = g.N >> 6;
    const int nunits = nchunk * cotiles * nsplit;
    const int jb = blockIdx.x >> 3;
    int id = ((jb / xg) * 8 + (blockIdx.x & 7)) * xg + jb % xg;
    if (id >= nunits) return;
    const int chunk = id % nchunk; id /= nchunk;
    const int cot = id % cotiles;
    const int split = id / cotiles;
    const int co0 = cot * 64, ci0 = chunk * 64;
    const int ntiles = g.B * tiles_x * tiles_y;
    const int t_begin = split * tiles_per_split, t_end = min(ntiles, t_begin + tiles_per_split);

    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)g.B * g.H * g.W * g.C * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t dres = __builtin_amdgcn_make_buffer_rsrc((void*)dp, 0, (unsigned)g.B * Hp * Wp * g.N * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t cres = __builtin_amdgcn_make_buffer_rsrc((void*)code, 0, (unsigned)g.B * Hp * Wp * (g.N >> 3) * 4u, 0x00020000);
    constexpr unsigned SP_OOB = 0xfffffff0u;
    // the x patch, the pooled-gradient tile and the codes of block t: instruction i of each goes to wave i % 4
    auto issue_dma = [&](int t) {
        int r = t;
        const int tx = r % tiles_x; r /= tiles_x;
        const int ty = r % tiles_y;
        const int b = r / tiles_y;
        const int y0 = ty * 16, x0 = tx * 16;
        // (the per-lane parts of the offsets are recomputed per block: hoisted out of the block loop they cost 30+ registers)
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = ct + 4 * j;                       // 8 windows x 128 B per instruction
            const int wi = 8 * i + (ln >> 3), sl = ln & 7;
            const int wy = ty * 8 + (wi >> 3), wx = tx * 8 + (wi & 7);
            const bool ok = wy < Hp && wx < Wp;
            const unsigned off = ((unsigned)((b * Hp + wy) * Wp + wx) * (unsigned)g.N + (unsigned)(co0 + sl * 8)) * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dres, (lds_void*)(smem + SP_OFF_DP + i * 1024), 16, ok ? off : SP_OOB, 0, 0, 0);
        }
        if (ct < 2) {                                       // 32 windows x 32 B per instruction
            const int wi = 32 * ct + (ln >> 1), half = ln & 1;
            const int wy = ty * 8 + (wi >> 3), wx = tx * 8 + (wi & 7);
            const bool ok = wy < Hp && wx < Wp;
            const unsigned off = ((unsigned)((b * Hp + wy) * Wp + wx) * (unsigned)(g.N >> 3) + (unsigned)((co0 >> 3) + half * 4)) * 4u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(cres, (lds_void*)(smem + SP_OFF_CODE + ct * 1024), 16, ok ? off : SP_OOB, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < (SP_PINSTR + 3) / 4; ++j) {
            const int i = ct + 4 * j;
            if (i < SP_PINSTR) {
                const int pp = 8 * i + (ln >> 3), sl = ln & 7;
                const int c16 = (((sl >> 1) ^ sp_key(pp)) << 1) | (sl & 1);
                const int py = pp / SP_PW, px = pp - py * SP_PW;
                const int iy = y0 - 1 + py, ix = x0 - 1 + px;
                const bool ok = pp < SP_PPIX && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
                const unsigned off = ((unsigned)((b * g.H + iy) * g.W + ix) * (unsigned)g.C + (unsigned)(ci0 + c16 * 8)) * 2u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void*)(smem + i * 1024), 16, ok ? off : SP_OOB, 0, 0, 0);
            }
        }
    };

    f32x4_t acc[4][9];                                      // [co tile][tap]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) acc[a][t9] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // bias gradient = sum of the routed pooled gradients: the producer sees every (window, channel) of a block exactly once, so
    // each lane keeps the sum of its items (channel 16 a + li, a = 0..3) -- no matrix instruction against a block of ones
    const bool do_bias = slab_b != nullptr && chunk == 0;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};

    const int gq = lane >> 4, li = lane & 15;
    // patch rows of this lane: pixel ((li >> 2) >> 1, (li >> 2) & 1) of window column PERM[gq]; eight bases (pixel offset mod 8)
    int gb[8];
    {
        const int p0 = ((li >> 3) & 1) * SP_PW + 2 * sp_perm(gq) + ((li >> 2) & 1);
#pragma unroll
        for (int r = 0; r < 8; ++r) gb[r] = (p0 + r) * 128 + ((ct ^ sp_key(p0 + r)) << 5) + (li & 3) * 8;
    }

    {
        for (int t = t_begin; t < t_end; ++t) {
            __syncthreads();                                 // every wave is done with block t-1: the buffers are free
            if (!(abl & 1) || t == t_begin) issue_dma(t);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                 // tile + codes + patch of block t have landed
            // ---- producer: the 16 A fragments of the block (k-step s, co tile a), four per wave, in smfmac operand order
            if (!(abl & 8)) {
                // wave ct makes the four fragments of k-step s = ct (co tiles a = 0..3), two at a time: all LDS loads of a pair
                // first, then the arithmetic (written item by item the compiler emitted dependent load -> wait -> use round trips)
                // (lane-dependent addresses recomputed per block: kept live across the matrix loop they spill)
                int lq = lane;
                asm volatile("" : "+v"(lq));
                const int gq = lq >> 4, li = lq & 15;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    unsigned pv[2][4], pc[2][4];
                    const int wy = 2 * ct + (gq >> 1);
#pragma unroll
                    for (int ff = 0; ff < 2; ++ff) {
                        const int col = (h2 * 2 + ff) * 16 + li;     // output channel within the 64
#pragma unroll
                        for (int gi = 0; gi < 4; ++gi) {
                            const int wx = 4 * (gi & 1) + sp_perm(2 * (gq & 1) + (gi >> 1));
                            const int wi = wy * 8 + wx;
                            pv[ff][gi] = *reinterpret_cast<const unsigned short*>(smem + SP_OFF_DP + wi * 128 + col * 2);
                            pc[ff][gi] = *reinterpret_cast<const unsigned*>(smem + SP_OFF_CODE + wi * 32 + (col >> 3) * 4);
                        }
                    }
#pragma unroll
                    for (int ff = 0; ff < 2; ++ff) {
                        const int a = h2 * 2 + ff, f = ct * 4 + a;
                        const int col = a * 16 + li;
                        unsigned vals[4];
                        int idx = 0;
#pragma unroll
                        for (int gi = 0; gi < 4; ++gi) {
                            const unsigned jc = (pc[ff][gi] >> (4 * (col & 7))) & 15u;   // winner position 2 dy + dx, 4 = none (no gradient)
                            const unsigned live = jc < 4u ? pv[ff][gi] : 0u;
                            vals[gi] = live << ((jc & 1u) << 4);                          // the pair (0,1) or (2,3) that holds the winner
                            idx |= ((jc & 2u) ? 0xE : 0x4) << (4 * gi);
                            bsum[a] += __uint_as_float(live << 16);
                        }
                        *reinterpret_cast<uint4*>(smem + SP_OFF_AV + (f * 64 + lq) * 16) = make_uint4(vals[0], vals[1], vals[2], vals[3]);
                        *reinterpret_cast<int*>(smem + SP_OFF_AI + (f * 64 + lq) * 4) = idx;
                    }
                }
            }
            __syncthreads();                                 // fragments visible
            // One unit u = (k-step s, tap): four transposing reads bring the tap's B fragment (64 k x 16 ci), four smfmac multiply
            // it with the step's four A fragments; the next unit's B fragment is requested before this unit's matrix instructions.
            uint4 fa[4];
            int fi[4];
            bf16x16_t fb[2];
            auto load_a = [&](int s) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    fa[a] = sp_ld16(smem + SP_OFF_AV + ((s * 4 + a) * 64 + lane) * 16, smem);
                    fi[a] = sp_ld4(smem + SP_OFF_AI + ((s * 4 + a) * 64 + lane) * 4, smem);
                }
            };
            auto load_b = [&](int u, int set) {
                const int s = u / 9, tap = u % 9;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // window row 2s + (q >> 1), window columns 4 (q & 1) + ...: the pixel offset of the read (a constant once unrolled)
                    const int ctap = (2 * (2 * s + (q >> 1)) + tap / 3) * SP_PW + 8 * (q & 1) + (tap % 3);
                    reinterpret_cast<s16x4_t*>(&fb[set])[q] = sp_tr((sp_lds_s16x4*)(smem + gb[ctap & 7] + (ctap >> 3) * 1024), smem);
                }
            };
            load_a(0);
            load_b(0, 0);
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int u = 0; u < 36; ++u) {
                const int s = u / 9, t9 = u % 9;
                if (u + 1 < 36 && !(abl & 4)) load_b(u + 1, (u + 1) & 1);
                if (t9 == 0 && u > 0) load_a(s);
                if (abl & 2) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        struct W8 { unsigned w[8]; };
                        const W8 bw = __builtin_bit_cast(W8, fb[u & 1]);
                        asm volatile("" :: "v"(fa[a].x), "v"(fa[a].w), "v"(bw.w[0]), "v"(bw.w[2]), "v"(bw.w[4]), "v"(bw.w[7]), "v"(fi[a]));
                    }
                } else
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    acc[a][t9] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(__builtin_bit_cast(bf16x8_t, fa[a]), fb[u & 1], acc[a][t9], fi[a], 0, 0);
            }
            __builtin_amdgcn_s_setprio(3);
        }
    }
    // slab[split][co][tap][ci]  (dW layout [Cout][kh][kw][Cin])
    const int ktot = g.ldw;
    float* out = slab_w + (long long)split * g.N * ktot;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
            const int colw = t9 * g.C + ci0 + ct * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + a * 16 + (lane >> 4) * 4 + j;
                out[(long long)co * ktot + colw] = acc[a][t9][j];
            }
        }
    if (do_bias) {                                           // 16 partial sums per channel (4 waves x 4 lane groups), fixed order
        __syncthreads();
        float* sb = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int a = 0; a < 4; ++a) sb[(ct * 4 + gq) * 64 + a * 16 + li] = bsum[a];
        __syncthreads();
        if (tid < 64) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) v += sb[k * 64 + tid];
            slab_b[(long long)split * g.N + co0 + tid] = v;
        }
    }
}

struct SpPlan { int tx, ty, tps, ns, xg; unsigned grid; };

bool sp_plan(int B, int H, int W, int Cin, int Cout, int Hp, int Wp, SpPlan* p) {
    if (B <= 0 || H < 16 || W < 16 || Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 64) return false;
    if ((Hp != H / 2 && Hp != (H + 1) / 2) || (Wp != W / 2 && Wp != (W + 1) / 2)) return false;
    if ((long long)B * H * W * Cin * 2 >= (1ll << 32) - 16 || (long long)B * Hp * Wp * Cout * 2 >= (1ll << 32) - 16) return false;   // 32-bit DMA offsets
    p->tx = (W + 15) / 16; p->ty = (H + 15) / 16;
    const int ntiles = B * p->tx * p->ty;
    const int groups = (Cin / 64) * (Cout / 64);
    int want = 512 / groups;                                  // two workgroups per CU
    if (want < 1) want = 1;
    if (want > ntiles) want = ntiles;
    p->tps = (ntiles + want - 1) / want;
    p->ns = (ntiles + p->tps - 1) / p->tps;
    const int nunits = groups * p->ns;
    int xg = 1;                                               // channel groups that walk the same blocks share an XCD (as the dense kernel)
    for (int d = groups; d >= 1; --d) {
        if (groups % d) continue;
        const int ngr = (nunits + d - 1) / d;
        const int load = ((ngr + 7) / 8) * d, fair = (nunits + 7) / 8;
        if (load * 100 <= fair * 107) { xg = d; break; }
    }
    p->xg = xg;
    p->grid = (unsigned)(8 * xg * ((nunits + 8 * xg - 1) / (8 * xg)));
    return true;
}

OnceLds g_sp_once;

}  // namespace

extern "C" {

size_t ssd_conv2d_bwd_weight_unpooled_workspace_bytes(int B, int H, int W, int Cin, int Cout, int Hp, int Wp) {
    SpPlan p;
    if (!sp_plan(B, H, W, Cin, Cout, Hp, Wp, &p)) return 0;
    return (size_t)p.ns * ((size_t)Cout * 9 * Cin + Cout) * sizeof(float);
}

int ssd_conv2d_bwd_weight_unpooled(const void* x, const void* dpool, const void* pool_code, float* dw, float* dbias, int B, int H,
                                   int W, int Cin, int Cout, int Hp, int Wp, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !dpool || !pool_code || !dw) return SSD_ERR_VALUE;
    SpPlan p;
    if (!sp_plan(B, H, W, Cin, Cout, Hp, Wp, &p)) return SSD_ERR_UNSUPPORTED;
    const size_t need = (size_t)p.ns * ((size_t)Cout * 9 * Cin + Cout) * sizeof(float);
    if (!ws || ws_bytes < need) return SSD_ERR_WORKSPACE;
    const ConvGeom g = make_geom(B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1);
    const long long ktot = g.ldw;
    float* slab_w = static_cast<float*>(ws);
    float* slab_b = slab_w + (size_t)p.ns * Cout * ktot;
    hipStream_t s = (hipStream_t)stream;
    if (ensure_lds(g_sp_once, reinterpret_cast<const void*>(k_conv3x3_wgrad_unpool), SP_LDS) != 0) return SSD_ERR_LAUNCH;
    hipLaunchKernelGGL(k_conv3x3_wgrad_unpool, dim3(p.grid), dim3(256), SP_LDS, s, static_cast<const bf16_raw*>(x),
                       static_cast<const bf16_raw*>(dpool), static_cast<const unsigned*>(pool_code), slab_w, dbias ? slab_b : nullptr, g,
                       Hp, Wp, p.tx, p.ty, p.tps, p.ns, p.xg, ssd_knob("SSD_SP_ABLATE", 0));
    if (hipGetLastError() != hipSuccess) return SSD_ERR_LAUNCH;
    ssd_launch_wgrad_reduce(s, slab_w, (long long)Cout * ktot, (long long)Cout * ktot, dw, slab_b, (long long)Cout, Cout, dbias, p.ns);
    return ssd_launch_status();
}

}  // extern "C"
