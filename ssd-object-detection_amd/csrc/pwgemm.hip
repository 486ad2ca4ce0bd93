// Pointwise (1x1, stride 1) convolution as a PERSISTENT bf16 GEMM for gfx950 (MI355X): forward and data gradient of the
// reference's 1x1 layers (models/ssd_model.py:94-97 conv 512->512 at 38x38, :107-111 1024->1024 and :113-116 1024->256 at 19x19;
// Keras Conv2D(kernel_size=1) + ReLU and its tape.gradient w.r.t. the input).
//
//   out[m][n] = sum_k A[m][k] * Wm[n][k]      A = activations [M = B*H*W][K = Cin]  (forward, Wm = filters [Cout][Cin])
//                                             A = dY          [M][K = Cout_pad]     (data gradient, Wm = transposed filters [Cin][Cout_pad])
//
// Why a second GEMM kernel next to k_conv_igemm_8ph (conv.hip): these layers have K = 256 ... 1024, i.e. 4 ... 16 k-tiles per
// 256 x 256 output tile, and move 100-190 MB each -- at the chip's MFMA rate they are HBM-bound (256 FLOP per byte against a
// machine balance of ~310).  One workgroup per CU and one tile per workgroup leaves the pipeline fill (first tile's DMA
// latency) and the drain (a 128 KB store tail) of every tile exposed: k_conv_igemm_8ph runs them at 0.20-0.37 of the MFMA peak
// with the matrix pipes idle 2/3 of the time.  Here a workgroup walks ITS tiles in one launch and the LDS-DMA stream never
// stops at a tile boundary: while the waves convert and store tile i, the first two k-tiles of tile i+1 are already in flight
// (the ring holds 128 KB); the store stage is wave-private (each wave turns its own 128 x 64 sub-tile through 2 KB of LDS, no
// workgroup barrier), and 1x1 addressing is base + constant (no im2col arithmetic: 3-5 vector instructions per MFMA in the
// generic kernel).
//
// Main loop = the 8-phase schedule of k_conv_igemm_8ph (two groups of four waves half a phase apart, a k-tile staged as four
// 16 KB half-tiles W0 X0 W1 X1, one counted vmcnt(6) per k-tile, raw s_barriers), same LDS image, same fragment order: the two
// kernels accumulate in the same order and agree bit for bit.
#include <atomic>
#include <cstdint>
#include "common.h"
#include <hip/hip_bf16.h>
#include "conv_common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;

constexpr int PW_HALF = 128 * 128;           // one half-tile image: 128 rows x 64 k bf16
constexpr int PW_OFF_W0 = 0, PW_OFF_X0 = PW_HALF, PW_OFF_W1 = 2 * PW_HALF, PW_OFF_X1 = 3 * PW_HALF, PW_BUF = 4 * PW_HALF;
constexpr int PW_RING = 2 * PW_BUF;          // 128 KB: two k-tiles
constexpr int PW_STAGE = 2048;               // per wave: one 16 px x 64 ch bf16 slab of the store stage (LDS executes a wave's
                                             // accesses in order: the next slab's writes cannot overtake this one's reads)
constexpr int PW_AUX = PW_RING + 8 * PW_STAGE;   // forward: bias as float [N <= 2048]; data gradient: the tile's ReLU sign bytes [256][32]
constexpr int PW_TICKET = PW_AUX + 8192;     // one word: the claimed ticket, wave 0 -> everybody
constexpr int PW_LDS = PW_TICKET + 16;       // 152 KB
constexpr unsigned PW_OOB = 0x80000000u;     // stream base of "no tile": every lane's offset is beyond the buffer -> zeros

__device__ __forceinline__ unsigned lds_ld4_scoped(const char* __restrict__ p, const char* __restrict__ other) {
    (void)other;
    return *reinterpret_cast<const unsigned*>(p);
}
__device__ __forceinline__ void lds_st4_scoped(char* __restrict__ p, const char* __restrict__ other, unsigned v) {
    (void)other;
    *reinterpret_cast<unsigned*>(p) = v;
}
__device__ __forceinline__ unsigned lds_ld1_scoped(const char* __restrict__ p, const char* __restrict__ other) {
    (void)other;
    return *reinterpret_cast<const unsigned char*>(p);
}

template <int HX, int HW>
__device__ __forceinline__ void pw_mma_quadrant(f32x4_t (&acc)[4][8], const bf16x8_t (&fx)[4][2], const bf16x8_t (&fw)[2][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int p = 0; p < 4; ++p)
                acc[HW * 2 + c][HX * 4 + p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[c][ks], fx[p][ks], acc[HW * 2 + c][HX * 4 + p], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int p = 0; p < 4; ++p) asm volatile("" : "+v"(acc[HW * 2 + c][HX * 4 + p]));
}

// Tile order.  Workgroup b belongs to class c = b % 8 (blocks are dealt round-robin over the 8 XCDs, so a class shares an XCD and
// its L2: a matter of speed only).  Class c owns the pixel tiles mt = 8q + c; its tiles are numbered by a ticket
// k = q * ntn + nt (the ntn channel tiles of one pixel tile are consecutive tickets: they run at about the same time on one XCD,
// which fetches the activation tile into its L2 once).  Tickets 0 .. G/8 - 1 of a class are the first tiles of its G/8
// workgroups; the following ones are CLAIMED from a per-class counter (one returning atomic per tile), so that a workgroup
// that reaches its CU late -- the kernel shares the chip with another stream's launches -- simply takes fewer tiles; with
// counters == nullptr every workgroup has a fixed share (ticket += G/8).  See the launch function for which is the default.
struct PwCursor {
    int k;                                   // ticket in the class, >= kmax: exhausted
    int kt;                                  // k-tile inside the tile
    int m0, n0;
};

struct PwSched {
    int ntm, ntn, G8, KT;                    // pixel / channel tiles, workgroups per class, k-tiles per tile
    FastDiv d_ntn;
    unsigned* counters;                      // [8] zeroed before the launch, or nullptr
};

__device__ __forceinline__ void pw_place(PwCursor& c, const PwSched& s, int cls) {
    const int q = fdiv(c.k, s.d_ntn);
    c.m0 = (q * 8 + cls) * 256;
    c.n0 = (c.k - q * s.ntn) * 256;
}

// ACC (data gradient only): out += result (two gradients meet at a feature map)
template <int EPI, bool ACC>
__global__ __launch_bounds__(512) void k_pw_gemm(const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ w, ConvGeom g,
                                                 Epilogue ep, PwSched sc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;    // wave tile: pixels wr*128.., channels wc*64..; group = wr
    const int K = g.C;
    const unsigned rowb = (unsigned)K * 2u;     // bytes per operand row

    // DMA ownership inside a half-tile (16 one-KiB pieces = 8 rows each): piece i = (wave&1) + 2*((wave>>1) + 4j), lane L ->
    // row 8i + 2*(L>>4) + ((L>>3)&1), k-chunk (L&7) ^ (4*(wave&1) + (L>>4))   [the image k_conv_igemm_8ph reads]
    const int rl = 2 * (lane >> 4) + ((lane >> 3) & 1);
    const int slot = (lane & 7) ^ (4 * (wave & 1) + (lane >> 4));
    unsigned xl[2][2], wl[2][2];                // per-lane byte offsets relative to the tile's first row / k-tile
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 8 * ((wave & 1) + 2 * ((wave >> 1) + 4 * j)) + rl;
            xl[h][j] = (unsigned)((r >> 6) * 128 + h * 64 + (r & 63)) * rowb + (unsigned)slot * 16u;
            wl[h][j] = (unsigned)((r >> 5) * 64 + h * 32 + (r & 31)) * rowb + (unsigned)slot * 16u;
        }
    // rows beyond M (last pixel tile) / N (ragged channel tile) are beyond the descriptor's range: the hardware writes zeros
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)g.M * rowb, 0x00020000);
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (unsigned)g.N * rowb, 0x00020000);

    const int cls = (int)blockIdx.x & 7;
    const int kmax = cls < sc.ntm ? ((sc.ntm - cls + 7) >> 3) * sc.ntn : 0;     // tickets of this class
    PwCursor cx, cw, cc;                        // activation stream, weight stream, compute
    cx.k = (int)blockIdx.x >> 3; cx.kt = 0;
    pw_place(cx, sc, cls);
    cw = cx; cc = cx;
    if (cc.k >= kmax) return;
    // ticket of the tile after the one the streams are in: known long before a stream crosses into it (fixed share: at once;
    // claimed: requested at the head of the current tile, see claim / publish / fetch below)
    const bool dyn = sc.counters != nullptr;
    int k_next = dyn ? kmax : cc.k + sc.G8;
    unsigned tk = 0;                            // wave 0, lane 0: the claimed counter value, in flight until a counted wait has passed
    auto claim = [&]() {
        if (dyn && wave == 0 && lane == 0) {
            const unsigned one = 1u;
            const unsigned* cp = sc.counters + cls;
            // (inline asm: an ordinary returning atomic makes the compiler wait vmcnt(0) -- the whole DMA ring -- before its use)
            asm volatile("global_atomic_add %0, %1, %2, off sc0 sc1" : "=v"(tk) : "v"(cp), "v"(one) : "memory");
        }
    };
    auto publish = [&]() {
        if (dyn && wave == 0 && lane == 0) lds_st4_scoped(smem + PW_TICKET, smem, tk);
    };
    auto fetch = [&]() {
        if (dyn) k_next = sc.G8 + (int)__builtin_amdgcn_readfirstlane(lds_ld4_scoped(smem + PW_TICKET, smem));
    };
    auto advance = [&](PwCursor& c) {           // one k-tile further along the stream
        if (++c.kt == sc.KT) {
            c.kt = 0;
            c.k = k_next;
            pw_place(c, sc, cls);
        }
    };
    auto xbase = [&]() { return cx.k < kmax ? (unsigned)cx.m0 * rowb + (unsigned)cx.kt * 128u : PW_OOB; };
    auto wbase = [&]() { return cw.k < kmax ? (unsigned)cw.n0 * rowb + (unsigned)cw.kt * 128u : PW_OOB; };
    auto issue_x = [&](int off, int h) {
        const unsigned b = xbase();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = (wave & 1) + 2 * ((wave >> 1) + 4 * j);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void*)(smem + off + i * 1024), 16, xl[h][j] + b, 0, 0, 0);
        }
    };
    auto issue_w = [&](int off, int h) {
        const unsigned b = wbase();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = (wave & 1) + 2 * ((wave >> 1) + 4 * j);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (lds_void*)(smem + off + i * 1024), 16, wl[h][j] + b, 0, 0, 0);
        }
    };

    f32x4_t acc[4][8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < 8; ++p) acc[c][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // Nothing but LDS-DMA may load from memory once the stream runs: the compiler waits for EVERY outstanding vector-memory
    // operation (vmcnt(0)) in front of the first use of an ordinary load's result, i.e. for the next tile's k-tiles in the
    // middle of a store stage.  The bias therefore goes to LDS now (forward); the ReLU sign bytes of a tile travel as one more
    // LDS-DMA piece per wave at the head of the tile (data gradient).
    if constexpr (EPI == EPI_FWD) {
        float* sb = reinterpret_cast<float*>(smem + PW_AUX);
        for (int i = tid; i < 2048; i += 512) sb[i] = (ep.bias && i < g.N) ? ep.bias[i] : 0.f;
    }
    const int nb8 = g.N >> 3;
    const __amdgpu_buffer_rsrc_t mres = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == EPI_DGRAD ? ep.mask_bits : nullptr), 0,
                                                                          EPI == EPI_DGRAD && ep.mask_bits ? (unsigned)g.M * (unsigned)nb8 : 0u, 0x00020000);
    // sign bytes of tile (m0, n0): row r of the tile = 32 bytes (256 channels); wave w brings rows 32w .. 32w+31, lane L the
    // 16-byte half (L & 1) of row 32w + (L >> 1); rows beyond M and halves beyond N are out of range (zeros: masked anyway)
    auto issue_mask = [&]() {
        if constexpr (EPI == EPI_DGRAD) {
            if (ep.mask_bits) {
                const int row = wave * 32 + (lane >> 1), half = lane & 1;
                const bool in = (cc.n0 >> 3) + half * 16 < nb8;
                const unsigned off = (unsigned)(cc.m0 + row) * (unsigned)nb8 + (unsigned)(cc.n0 >> 3) + (unsigned)half * 16u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(mres, (lds_void*)(smem + PW_AUX + wave * 1024), 16, in ? off : PW_OOB, 0, 0, 0);
            }
        }
    };

    // prologue: k-tile 0 completely, W0 X0 W1 of k-tile 1 (X1 of it goes out in the first phase)
    issue_w(PW_OFF_W0, 0); issue_x(PW_OFF_X0, 0); issue_w(PW_OFF_W1, 1); issue_x(PW_OFF_X1, 1);
    advance(cw);
    advance(cx);
    issue_w(PW_BUF + PW_OFF_W0, 0); issue_x(PW_BUF + PW_OFF_X0, 0); issue_w(PW_BUF + PW_OFF_W1, 1);
    advance(cw);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // group 1 runs half a phase behind group 0

    const int frow = lane & 15, fk = lane >> 4;
    const int xf0 = swz(wr * 64 + frow, fk), xf1 = swz(wr * 64 + frow, 4 + fk);     // + p * 2048
    const int wf0 = swz(wc * 32 + frow, fk), wf1 = swz(wc * 32 + frow, 4 + fk);     // + c * 2048
    auto ldf = [&](int off) { return *reinterpret_cast<const bf16x8_t*>(smem + off); };
    bf16x8_t fx[4][2] = {}, fw0[2][2] = {}, fw1[2][2] = {};

#define PW_PHASE_MMA(HX_, HW_, FW_)                                                                                 \
    __builtin_amdgcn_s_barrier();                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    __builtin_amdgcn_s_setprio(1);                                                                                  \
    pw_mma_quadrant<HX_, HW_>(acc, fx, FW_);                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    __builtin_amdgcn_s_setprio(0);                                                                                  \
    __builtin_amdgcn_s_barrier();                                                                                   \
    __builtin_amdgcn_sched_barrier(0);

    int gk = 0;                                 // k-tiles consumed so far (ring slot = gk & 1)
    while (cc.k < kmax) {
        for (int t = 0; t < sc.KT; ++t, ++gk) {
            const int cb = (gk & 1) * PW_BUF, nb = PW_BUF - cb;
            // ---- phase 0: quadrant (X0, W0); stage X1 of k-tile gk+1
#pragma unroll
            for (int c = 0; c < 2; ++c) { fw0[c][0] = ldf(cb + PW_OFF_W0 + wf0 + c * 2048); fw0[c][1] = ldf(cb + PW_OFF_W0 + wf1 + c * 2048); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 4; ++p) { fx[p][0] = ldf(cb + PW_OFF_X0 + xf0 + p * 2048); fx[p][1] = ldf(cb + PW_OFF_X0 + xf1 + p * 2048); }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");      // the W0 reads are done: W0 may be re-staged next phase
            if (t == 1) publish();                // the ticket claimed in k-tile 0: retired by that k-tile's counted wait
            issue_x(nb + PW_OFF_X1, 1);
            advance(cx);
            PW_PHASE_MMA(0, 0, fw0)
            // ---- phase 1: quadrant (X0, W1); stage W0 of k-tile gk+2
#pragma unroll
            for (int c = 0; c < 2; ++c) { fw1[c][0] = ldf(cb + PW_OFF_W1 + wf0 + c * 2048); fw1[c][1] = ldf(cb + PW_OFF_W1 + wf1 + c * 2048); }
            // (the tile's sign bytes: not before this phase -- past phase 0's barriers every wave of either group has left the
            //  previous tile's store stage, which reads the region; issued ahead of W0, so the counted wait of phase 3 retires it)
            if (t == 0) { claim(); issue_mask(); }
            issue_w(cb + PW_OFF_W0, 0);
            PW_PHASE_MMA(0, 1, fw1)
            // ---- phase 2: quadrant (X1, W1); stage X0 of k-tile gk+2
#pragma unroll
            for (int p = 0; p < 4; ++p) { fx[p][0] = ldf(cb + PW_OFF_X1 + xf0 + p * 2048); fx[p][1] = ldf(cb + PW_OFF_X1 + xf1 + p * 2048); }
            issue_x(cb + PW_OFF_X0, 0);
            // (two barriers behind wave 0's publish for either group; the streams cross into the next tile in k-tile KT - 2 >= 2)
            if (t == 1) fetch();
            PW_PHASE_MMA(1, 1, fw1)
            // ---- phase 3: quadrant (X1, W0); stage W1 of k-tile gk+2; all of k-tile gk+1 has landed once only the last three
            // half-tiles (6 DMA instructions of this wave) are still in flight
            issue_w(cb + PW_OFF_W1, 1);
            advance(cw);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            PW_PHASE_MMA(1, 0, fw0)
        }
        // ---- store stage of this tile, wave-private (no workgroup barrier; the DMA of the next tile's first k-tiles, issued
        // above, lands meanwhile).  A wave turns its 128 px x 64 ch sub-tile through LDS 16 pixels at a time: the accumulator
        // layout (a lane holds 4 channels of 16 different pixels) becomes whole 16-byte chunks of 128-byte row segments.
        {
            char* st = smem + PW_RING + wave * PW_STAGE;
            const int r16 = lane & 15, q4 = lane >> 4;
            const int ncol0 = cc.n0 + wc * 64;
            float b4[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                b4[c][0] = b4[c][1] = b4[c][2] = b4[c][3] = 0.f;
                if constexpr (EPI == EPI_FWD) {
                    const int nn = ncol0 + c * 16 + q4 * 4;         // < 2048 (host check); zeros beyond N
                    const uint4 bv = lds_ld16_scoped(smem + PW_AUX + nn * 4, smem);
                    b4[c][0] = __uint_as_float(bv.x); b4[c][1] = __uint_as_float(bv.y); b4[c][2] = __uint_as_float(bv.z); b4[c][3] = __uint_as_float(bv.w);
                }
            }
            const int rr = lane >> 3, ch = lane & 7;          // read side: rows rr and rr + 8, chunk ch
            const int n = ncol0 + ch * 8;
            const bool nok = n < g.N;
            const int mrow = cc.m0 + wr * 128 + rr;           // + p * 16 + h * 8
            // accumulate: all sixteen chunks this lane will add onto are requested up front, unconditionally (a chunk outside
            // the map re-reads element 0 and is dropped): ONE memory round trip per tile instead of one per 16-pixel slab
            uint4 old[ACC ? 8 : 1][2];
            if constexpr (ACC) {
#pragma unroll
                for (int p = 0; p < 8; ++p)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int m = mrow + p * 16 + h * 8;
                        old[p][h] = *reinterpret_cast<const uint4*>(ep.out + ((m < g.M && nok) ? (long long)m * ep.ldo + n : 0));
                    }
            }
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                unsigned mb[2] = {0xffu, 0xffu};
                if constexpr (EPI == EPI_DGRAD) {
                    if (ep.mask_bits) {                           // from the tile's LDS copy (issue_mask)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            mb[h] = lds_ld1_scoped(smem + PW_AUX + (wr * 128 + p * 16 + h * 8 + rr) * 32 + wc * 8 + ch, smem);
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v[4] = {acc[c][p][0] + b4[c][0], acc[c][p][1] + b4[c][1], acc[c][p][2] + b4[c][2], acc[c][p][3] + b4[c][3]};
                    if constexpr (EPI == EPI_FWD) {
                        if (ep.relu) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                        }
                    }
                    acc[c][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    const int chunk = c * 2 + (q4 >> 1);
                    lds_st8_scoped(st + r16 * 128 + (((chunk ^ r16) & 7) << 4) + (q4 & 1) * 8, smem,
                                   make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])));
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = h * 8 + rr;
                    uint4 v = lds_ld16_scoped(st + row * 128 + (((ch ^ row) & 7) << 4), smem);
                    const int m = mrow + p * 16 + h * 8;
                    // (every loaded value is consumed on every path -- only the stores are predicated: a load whose use is
                    //  skipped stays "pending" for the compiler, which then waits vmcnt(0) in front of the main loop's fragment
                    //  reads that reuse its registers, i.e. for the whole DMA ring, every k-tile)
                    if constexpr (ACC) {
                        auto add2 = [](unsigned a, unsigned b) {
                            const float lo = __uint_as_float(a << 16) + __uint_as_float(b << 16);
                            const float hi = __uint_as_float(a & 0xffff0000u) + __uint_as_float(b & 0xffff0000u);
                            return pack_bf16x2(lo, hi);
                        };
                        v.x = add2(v.x, old[p][h].x); v.y = add2(v.y, old[p][h].y); v.z = add2(v.z, old[p][h].z); v.w = add2(v.w, old[p][h].w);
                    }
                    if constexpr (EPI == EPI_DGRAD) {
                        if (ep.mask_bits) v = gate_bits8(v, mb[h]);
                    }
                    if (m < g.M && nok) {
                        if constexpr (EPI == EPI_FWD) {
                            if (ep.relu_bits) ep.relu_bits[(long long)m * nb8 + (n >> 3)] = (unsigned char)relu_bits8(v);
                        }
                        *reinterpret_cast<uint4*>(ep.out + (long long)m * ep.ldo + n) = v;
                    }
                }
            }
        }
        // (k_next still names the tile the streams crossed into: the next claim is fetched in k-tile 1 of that tile)
        cc.k = k_next;
        pw_place(cc, sc, cls);
        if (!dyn) k_next = cc.k + sc.G8;
    }
#undef PW_PHASE_MMA
    if (wr == 0) __builtin_amdgcn_s_barrier();          // balance the stagger
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the (all-zero) stages issued past the last tile
}

OnceLds g_pw_once[3];

}  // namespace

// Called from conv.hip's dispatch (launch_igemm); geometry / epilogue are the structs of conv_common.h (same header, both
// translation units).  Returns SSD_ERR_UNSUPPORTED (nothing launched) for shapes this kernel does not serve.
bool ssd_pw_gemm_serves(int epi, const void* geom, const void* epilogue) {
    const ConvGeom& g = *static_cast<const ConvGeom*>(geom);
    const Epilogue& ep = *static_cast<const Epilogue*>(epilogue);
    if (epi != EPI_FWD && epi != EPI_DGRAD) return false;
    if (g.KH != 1 || g.KW != 1 || g.mul != 1 || g.div != 1 || g.pad_t != 0 || g.pad_l != 0 || g.H != g.Ho || g.W != g.Wo) return false;
    if (g.C % 64 || g.C < 256 || (g.N & 7) || (ep.ldo & 7) || g.ldw != g.C || !ep.out) return false;
    if (ep.slab || ep.pool_out || ep.up_out || ep.mask_src) return false;     // (the ReLU mask as sign bytes only)
    if (epi == EPI_FWD && ep.accumulate) return false;
    if ((long long)(g.M + 256) * g.C * 2 >= (1ll << 31) || (long long)(g.N + 256) * g.C * 2 >= (1ll << 31)) return false;   // 31-bit stream offsets
    if (g.N > 2048) return false;                 // the bias / sign-byte region of the LDS image
    if ((long long)((g.N + 255) / 256) * 256 * 5 > (long long)g.N * 8) return false;   // 256-wide channel tiles: at most 3/8 padding
    if (g.M < 8192) return false;                 // few tiles: nothing to walk, the split-K path fills the chip better
    // fewer 256 x 256 tiles than ~3/4 of the CUs: the 128 x 128 tiles of the generic kernel fill the chip better
    // (conv15 forward, 19x19 1024 -> 256 at batch 64: 91 tiles, 28.4 us here against 26.5 us)
    if ((long long)((g.M + 255) / 256) * ((g.N + 255) / 256) < 192) return false;
    return true;
}

int ssd_pw_gemm_launch(int epi, const void* x, const void* w, const void* geom, const void* epilogue, void* ws, size_t ws_bytes,
                       void* stream) {
    if (!ssd_pw_gemm_serves(epi, geom, epilogue)) return SSD_ERR_UNSUPPORTED;
    const ConvGeom& g = *static_cast<const ConvGeom*>(geom);
    const Epilogue& ep = *static_cast<const Epilogue*>(epilogue);
    PwSched sc;
    sc.ntm = (g.M + 255) / 256;
    sc.ntn = (g.N + 255) / 256;
    sc.KT = g.C / 64;
    sc.d_ntn = make_fastdiv(sc.ntn);
    const int tiles = sc.ntm * sc.ntn;
    int G = ssd_knob("SSD_PW_WGS", 256);
    if (G > tiles) G = tiles;
    G = (G + 7) / 8 * 8;
    sc.G8 = G / 8;
    // claimed tiles need eight zeroed counters: the head of the caller's split-K workspace (stream-ordered with everything
    // else that uses it), cleared by a memset node in front of the launch; without a workspace every workgroup takes a fixed share
    // Default: fixed shares.  Measured on the batch-64 train step (same box, interleaved, tools_dev/ab_train_step.py and
    // tools_dev/ab_knobs.sh): fixed shares 9.43-9.51 ms, claimed tiles + the memset node 9.50-9.51 ms, the one-tile-per-workgroup
    // kernels this replaces 9.45-9.56 ms -- the memset is a fill KERNEL that needs a CU slot (5 us alone, 66 us once behind
    // another stream's resident workgroups), which eats what claiming gains.  SSD_PW_DYNAMIC=1 selects claiming.
    sc.counters = (ws && ws_bytes >= 64 && ssd_knob("SSD_PW_DYNAMIC", 0)) ? static_cast<unsigned*>(ws) : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (sc.counters && hipMemsetAsync(sc.counters, 0, 32, s) != hipSuccess) return SSD_ERR_LAUNCH;
#define PW_LAUNCH(KERN_, SLOT_)                                                                                      \
    do {                                                                                                            \
        auto kern = KERN_;                                                                                          \
        if (ensure_lds(g_pw_once[SLOT_], reinterpret_cast<const void*>(kern), PW_LDS) != 0) return SSD_ERR_LAUNCH;   \
        hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(512), PW_LDS, s, static_cast<const bf16_raw*>(x),           \
                           static_cast<const bf16_raw*>(w), g, ep, sc);                                             \
    } while (0)
    if (epi == EPI_FWD) PW_LAUNCH((k_pw_gemm<EPI_FWD, false>), 0);
    else if (ep.accumulate) PW_LAUNCH((k_pw_gemm<EPI_DGRAD, true>), 1);
    else PW_LAUNCH((k_pw_gemm<EPI_DGRAD, false>), 2);
#undef PW_LAUNCH
    return ssd_launch_status();
}
