// A chain of small convolutions in ONE launch, one workgroup per image, activations in LDS -- for gfx950 (MI355X).
//
// The reference's "extras" behind the 19x19 map (models/ssd_model.py:124-150: conv 1x1 512->128, 3x3/2 128->256, 1x1
// 256->128, 3x3 valid 128->256, 1x1 256->128, 3x3 valid 128->256 on 10x10 ... 1x1 maps) are six layers of <= 100 pixels per
// image: as six launches (+ split-K finalizes) each fills a few CUs and pays its own launch, pipeline fill and drain; the six
// data gradients are the head of the backward pass's critical path (nothing large can start before the chain reaches the 19x19
// map): 226 us of a nearly idle GPU at batch 64 (profiles/r03_step_timeline.txt).  Per image the whole chain is 36 MFLOP and
// its activations fit in LDS (the largest map is 10x10x512 bf16 = 100 KB), so a workgroup walks all layers of ITS image:
//   * the layer's input lives in LDS ([pixel][channel], rows padded by 16 bytes against bank conflicts), its output goes to
//     LDS (the next layer's input) AND to HBM (activations: the backward pass; gradients: the weight-gradient kernels);
//   * the filters stream from L2 / HBM straight into registers as MFMA A operands, CH_D fragments (1 KB each) in flight per
//     wave; every wave owns n-tiles, so a filter element is read once per workgroup.  They are read from a FRAGMENT-PACKED copy
//     (k_chain_pack: [n/16][k/32][64 lanes][8], one contiguous KB per operand) of the k-contiguous filters [n][tap][k] -- the
//     forward layout [Cout][kh][kw][Cin] / the data gradient's transposed copy [Cin][kh][kw][Cout_pad]: read in place, the 16
//     rows of a fragment are 64-byte pieces of 16 lines and the CU's address path delivered 16 B/clk (120 / 150 us per chain
//     instead of 50 / 79);
//   * one code path for forward and data gradient, the implicit-GEMM kernels' geometry (conv.hip): source pixel of tap t of
//     output o = (o * mul + t - pad) / div, taken when divisible and inside the map;
//   * epilogues as conv_common.h's epi_store: forward bias + ReLU (+ sign bits), data gradient accumulate-then-mask.
// fp32 accumulation over k = (tap, channel) ascending in ONE pass (no split-K): results are deterministic; they differ from the
// per-layer kernels' (split-K partial sums) in the last fp32 bits only -- tests/test_chain_gpu.py pins both against the fp32
// oracle with the per-layer kernels' bounds.
#include <atomic>
#include <cstdint>
#include <type_traits>
#include "common.h"
#include <hip/hip_bf16.h>
#include "conv_common.h"

namespace {

constexpr int CH_D = 16;                     // filter fragments in flight per wave (four groups of four)
constexpr int CH_MAX_MT = 7;                 // <= 112 output pixels per image and layer
constexpr int CH_LDS_MAX = 160 * 1024;

struct ChainLayer {
    const bf16_raw* w;                       // packed by ssd_chain_pack_weights: [N/16][KH*KW*Kc/32][64 lanes][8]
    const float* bias;                       // [N] or null
    bf16_raw* out;                           // [B][Ho*Wo][N]
    const unsigned char* mask_bits;          // [B][Ho*Wo][N/8] or null    (data gradient: ReLU sign bits of the layer's input activation)
    const bf16_raw* mask_src;                // [B][Ho*Wo][N] or null      (... or the activation itself)
    unsigned char* relu_bits;                // [B][Ho*Wo][N/8] or null    (forward: written)
    int Hi, Wi, Kc, Ho, Wo, N, KH, KW, mul, dshift, pad_t, pad_l, relu, accumulate;
    int in_off, out_off;                     // LDS byte offsets of the input / output image (out_off < 0: the output is not kept)
};
struct ChainArgs {
    int nlayers;
    int zero_off, zero_bytes;                // LDS: a row of zeros as long as the widest input row
    int touch;                               // SSD_CHAIN_TOUCH: every workgroup first reads its eighth of the packed filters
    const bf16_raw* in0;                     // [B][Hi*Wi][Kc] of layer 0
    ChainLayer L[SSD_CHAIN_MAX_LAYERS];
};

__device__ __forceinline__ bf16x8_t ld_frag(const bf16_raw* p) { return *reinterpret_cast<const bf16x8_t*>(p); }

template <int MT, int CH_THREADS>
__device__ __forceinline__ void chain_layer(const ChainLayer& L, char* smem, int b, int zero_off) {
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, gq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int taps = L.KH * L.KW, kpt = L.Kc >> 5, S = taps * kpt;
    const int npix = L.Ho * L.Wo;
    const int in_stride = L.Kc * 2 + 16, out_stride = L.N * 2 + 16;
    const int div_mask = (1 << L.dshift) - 1;
    int by[MT], bx[MT];
    {
        const int Wo = L.Wo, mul = L.mul, pad_t = L.pad_t, pad_l = L.pad_l;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int p = t * 16 + li;
            const int oy = p / Wo, ox = p - oy * Wo;
            by[t] = p < npix ? oy * mul - pad_t : -(1 << 20);      // (a row beyond the image never finds a source pixel)
            bx[t] = ox * mul - pad_l;
        }
    }
    const int ntiles = L.N >> 4;
    // byte offset of the source pixel's LDS row for the current tap; a tap without a source pixel (padding, a stride-2 gap,
    // a row beyond the image) reads the zero row instead: no select behind the LDS read
    int poff[MT];
    // (the layer record lives in kernel-argument memory: every field the loops use is copied to a scalar ONCE -- read through
    //  the reference, each use was an s_load + s_waitcnt of its own, 2-3 us per tap)
    const int Hi = L.Hi, Wi = L.Wi, dshift = L.dshift, in_off = L.in_off, KW = L.KW;
    auto set_tap = [&](int kh, int kw) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int iy = by[t] + kh, ix = bx[t] + kw;
            const int sy = iy >> dshift, sx = ix >> dshift;
            const bool ok = ((iy | ix) >= 0) & (((iy | ix) & div_mask) == 0) & (sy < Hi) & (sx < Wi);
            poff[t] = ok ? in_off + (sy * Wi + sx) * in_stride : zero_off;
        }
    };
    // k runs in groups of four 32-channel steps (Kc % 128 == 0: a group never straddles a tap); the filter ring holds CH_D / 4
    // groups, group g lives in ring quarter g % (CH_D / 4): static register indices with a loop body of CH_D steps.  The body has
    // NO conditional loads (the compiler's vmcnt bookkeeping gives up at a branch and waits for every load in flight -- measured:
    // one full memory latency per group, 141 us for the six layers): the group count is padded to a multiple of four, a padding
    // group multiplies (valid, clamped) filter words with the zero row.
    const int G = S >> 2, gpt = kpt >> 2, Gp = (G + CH_D / 4 - 1) & ~(CH_D / 4 - 1);
    for (int nt = wave; nt < ntiles; nt += CH_THREADS / 64) {
        const bf16_raw* wrow = L.w + ((long long)nt * S * 64 + lane) * 8;      // fragment (nt, s): 1 KB contiguous, lane-major
        bf16x8_t ring[CH_D];
#pragma unroll
        for (int d = 0; d < CH_D; ++d) ring[d] = ld_frag(wrow + min(d, S - 1) * 512);
        f32x4_t acc[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        int kh = 0, kw = 0, gk = 0;                              // tap and group-in-tap of group g (wave-uniform)
        set_tap(0, 0);
        auto group = [&](int g, auto r0_tag) {
            constexpr int R0 = decltype(r0_tag)::value;
            const int koff = gk * 256 + gq * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x8_t f[MT];
#pragma unroll
                for (int t = 0; t < MT; ++t) f[t] = ld_frag(reinterpret_cast<const bf16_raw*>(smem + poff[t] + koff + j * 64));
                __builtin_amdgcn_sched_barrier(0);               // (the reads of a step go out together, ahead of its MFMAs)
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[R0 + j], f[t], acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)                          // refill the quarter: CH_D - 4 steps ahead of its use
                ring[R0 + j] = ld_frag(wrow + min((g + CH_D / 4) * 4 + j, S - 1) * 512);
            if (g + 1 >= G) {                                    // behind the last group: zeros
#pragma unroll
                for (int t = 0; t < MT; ++t) poff[t] = zero_off;
                gk = 0;
            } else if (++gk == gpt) {
                gk = 0;
                if (++kw == KW) { kw = 0; ++kh; }
                set_tap(kh, kw);
            }
        };
        for (int g0 = 0; g0 < Gp; g0 += CH_D / 4) {
            group(g0, std::integral_constant<int, 0>{});
            group(g0 + 1, std::integral_constant<int, 4>{});
            group(g0 + 2, std::integral_constant<int, 8>{});
            group(g0 + 3, std::integral_constant<int, 12>{});
        }
        // lane: channels n..n+3 of pixel 16 t + li
        const int n = nt * 16 + gq * 4;
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (L.bias) {
            const float4 bv = *reinterpret_cast<const float4*>(L.bias + n);
            b4[0] = bv.x; b4[1] = bv.y; b4[2] = bv.z; b4[3] = bv.w;
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int p = t * 16 + li;
            const bool live = p < npix;
            const long long row = (long long)b * npix + (live ? p : 0);
            float v[4] = {acc[t][0] + b4[0], acc[t][1] + b4[1], acc[t][2] + b4[2], acc[t][3] + b4[3]};
            if (L.relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            bf16_raw* o = L.out + row * L.N + n;
            if (L.accumulate && live) {                          // out += result (a head's gradient is already there)
                const uint2 old = *reinterpret_cast<const uint2*>(o);
                v[0] += __uint_as_float(old.x << 16); v[1] += __uint_as_float(old.x & 0xffff0000u);
                v[2] += __uint_as_float(old.y << 16); v[3] += __uint_as_float(old.y & 0xffff0000u);
            }
            if (L.mask_bits) {
                const unsigned mb = live ? L.mask_bits[row * (L.N >> 3) + (n >> 3)] : 0u;
                const unsigned m4 = mb >> ((gq & 1) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) if (!((m4 >> j) & 1u)) v[j] = 0.f;
            } else if (L.mask_src && live) {
                const uint2 mk = *reinterpret_cast<const uint2*>(L.mask_src + row * L.N + n);
                if (!(__uint_as_float(mk.x << 16) > 0.f)) v[0] = 0.f;
                if (!(__uint_as_float(mk.x & 0xffff0000u) > 0.f)) v[1] = 0.f;
                if (!(__uint_as_float(mk.y << 16) > 0.f)) v[2] = 0.f;
                if (!(__uint_as_float(mk.y & 0xffff0000u) > 0.f)) v[3] = 0.f;
            }
            const uint2 pk = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            if (live) {
                *reinterpret_cast<uint2*>(o) = pk;
                if (L.out_off >= 0) *reinterpret_cast<uint2*>(smem + L.out_off + p * out_stride + n * 2) = pk;
            }
            if (L.relu_bits) {                                   // bit k of a byte = channel 8c + k of the pixel is > 0 (on the stored bf16)
                auto pos = [](unsigned h) { return (h & 0x8000u) == 0u && (h & 0x7fffu) != 0u ? 1u : 0u; };
                unsigned m4 = pos(pk.x & 0xffffu) | (pos(pk.x >> 16) << 1) | (pos(pk.y & 0xffffu) << 2) | (pos(pk.y >> 16) << 3);
                const unsigned other = (unsigned)__shfl_xor((int)m4, 16);
                if (live && (gq & 1) == 0) L.relu_bits[row * (L.N >> 3) + (n >> 3)] = (unsigned char)(m4 | (other << 4));
            }
        }
    }
}

// CH_THREADS = 512: 8 waves, the whole register file of a CU (2 waves x ~240 VGPRs per SIMD) -- fastest alone; 256: 4 waves, each
// owning twice the n-tiles: half a CU, so a workgroup can start beside other streams' workgroups (SSD_CHAIN_WAVES).
template <int CH_THREADS>
__global__ __launch_bounds__(CH_THREADS) void k_conv_chain(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    if (a.touch) {
        // cold filters: the eight workgroups of an XCD (ids b, b + 8, ...) walk the same lines in lockstep and would stream them
        // at one miss latency per 128 KB in flight; each first touches ITS eighth of every layer's packed filters (~35 16-byte
        // loads per thread, all in flight together), after which the XCD's L2 holds the whole set
        const int nslice = max(1, min(8, (int)gridDim.x >> 3)), slice = ((int)blockIdx.x >> 3) % nslice;
        unsigned acc = 0;
        for (int l = 0; l < a.nlayers; ++l) {
            const ChainLayer& L = a.L[l];
            const int pieces = L.N * L.KH * L.KW * (L.Kc >> 3);
            const int per = (pieces + nslice - 1) / nslice, lo = slice * per, hi = min(pieces, lo + per);
            const uint4* p = reinterpret_cast<const uint4*>(L.w);
            for (int i = lo + threadIdx.x; i < hi; i += 4 * CH_THREADS) {
                uint4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = p[min(i + u * CH_THREADS, hi - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc ^= v[u].x;
            }
        }
        if (acc == 0x9e3779b9u) smem[a.zero_off] = 1;          // (keeps the loads alive; the zero row is rewritten right below)
        __syncthreads();
    }
    for (int i = threadIdx.x * 16; i < a.zero_bytes; i += CH_THREADS * 16)
        *reinterpret_cast<uint4*>(smem + a.zero_off + i) = make_uint4(0u, 0u, 0u, 0u);
    {   // image of layer 0 -> LDS, rows padded
        const ChainLayer& L0 = a.L[0];
        const int cpr = L0.Kc >> 3, stride = L0.Kc * 2 + 16;         // 16-byte chunks per pixel row
        const int total = L0.Hi * L0.Wi * cpr;
        const uint4* src = reinterpret_cast<const uint4*>(a.in0 + (long long)b * L0.Hi * L0.Wi * L0.Kc);
        for (int i0 = threadIdx.x; i0 < total; i0 += 4 * CH_THREADS) {   // four loads in flight per thread
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * CH_THREADS;
                v[u] = i < total ? src[i] : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * CH_THREADS;
                if (i < total) {
                    const int r = i / cpr, c = i - r * cpr;
                    *reinterpret_cast<uint4*>(smem + L0.in_off + r * stride + c * 16) = v[u];
                }
            }
        }
    }
    for (int l = 0; l < a.nlayers; ++l) {
        __syncthreads();                                         // the layer's input is complete; the buffer it overwrites is free
        const ChainLayer& L = a.L[l];
        const int mt = (L.Ho * L.Wo + 15) >> 4;
        if (mt <= 1) chain_layer<1, CH_THREADS>(L, smem, b, a.zero_off);
        else if (mt <= 2) chain_layer<2, CH_THREADS>(L, smem, b, a.zero_off);
        else if (mt <= 4) chain_layer<4, CH_THREADS>(L, smem, b, a.zero_off);
        else chain_layer<CH_MAX_MT, CH_THREADS>(L, smem, b, a.zero_off);
    }
}

// Filters [N][K] (k contiguous: forward [Cout][kh*kw*Cin], data gradient [Cin][kh*kw*Cout_pad]) -> the MFMA A fragments
// k_conv_chain loads: [N/16][K/32][64 lanes][8], fragment (nt, s), lane (li, gq) = w[16 nt + li][32 s + 8 gq .. + 7].  A wave's
// load of a fragment is then ONE contiguous KB: read in place, the 16 rows of a fragment are 64-byte pieces of 16 lines and the
// CU's texture-address path delivered 16 B/clk (measured: ~540 cycles per k-step whatever the LDS / MFMA work was).
struct PackItem { const bf16_raw* src; bf16_raw* dst; int N, K; };
struct PackArgs { PackItem it[SSD_CHAIN_PACK_MAX]; };
__global__ __launch_bounds__(256) void k_chain_pack(PackArgs a) {
    const PackItem it = a.it[blockIdx.y];
    const int S = it.K >> 5;
    const long long total = (long long)(it.N >> 4) * S * 64;          // 16-byte pieces
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        const long long f = i >> 6;
        const int s = (int)(f % S), nt = (int)(f / S);
        const uint4 v = *reinterpret_cast<const uint4*>(it.src + (long long)(nt * 16 + (lane & 15)) * it.K + s * 32 + (lane >> 4) * 8);
        *reinterpret_cast<uint4*>(it.dst + i * 8) = v;
    }
}

// Warm every XCD's L2 (and the memory-side cache) with the packed filters shortly before a chain launch: the filters were last
// touched by the optimizer step's pack launch ~9 ms and ~20 GB of traffic ago, and the chain's workgroups -- eight per XCD, in
// lockstep on the same lines -- then stream them at one miss latency per 128 KB in flight (147 / 173 us in the step against 50 /
// 79 us with a warm L2).  Workgroup b reads slice b / 8 of every tensor; workgroups b, b + 8, ... share an XCD.
__global__ __launch_bounds__(256) void k_chain_prefetch(PackArgs a, int count, unsigned* sink) {
    const int slice = blockIdx.x >> 3, nslice = gridDim.x >> 3;
    unsigned acc = 0;
    for (int t = 0; t < count; ++t) {
        const PackItem it = a.it[t];
        const long long pieces = (long long)it.N * it.K / 8;
        const long long per = (pieces + nslice - 1) / nslice, lo = slice * per, hi = lo + per < pieces ? lo + per : pieces;
        const uint4* p = reinterpret_cast<const uint4*>(it.dst);
        for (long long i = lo + threadIdx.x; i < hi; i += 256) {
            const uint4 v = p[i];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x9e3779b9u && sink) *sink = acc;             // (keeps the loads alive; never true in practice, harmless if it is)
}

std::atomic<int> g_lds_set{0}, g_lds_set4{0};

}  // namespace

extern "C" {

int ssd_chain_pack_weights(const ssd_chain_pack* items, int count, void* stream) {
    if (!items || count <= 0 || count > SSD_CHAIN_PACK_MAX) return SSD_ERR_VALUE;
    PackArgs a;
    long long most = 0;
    for (int i = 0; i < count; ++i) {
        if (!items[i].src || !items[i].dst || items[i].N <= 0 || items[i].K <= 0 || (items[i].N & 15) || (items[i].K & 31)) return SSD_ERR_VALUE;
        a.it[i] = PackItem{static_cast<const bf16_raw*>(items[i].src), static_cast<bf16_raw*>(items[i].dst), items[i].N, items[i].K};
        const long long pieces = (long long)items[i].N * items[i].K / 8;
        most = pieces > most ? pieces : most;
    }
    for (int i = count; i < SSD_CHAIN_PACK_MAX; ++i) a.it[i] = a.it[0];
    const unsigned gx = (unsigned)((most + 255) / 256 < 256 ? (most + 255) / 256 : 256);
    hipLaunchKernelGGL(k_chain_pack, dim3(gx, count), dim3(256), 0, (hipStream_t)stream, a);
    return ssd_launch_status();
}

int ssd_chain_prefetch(const ssd_chain_pack* items, int count, void* stream) {
    if (!items || count <= 0 || count > SSD_CHAIN_PACK_MAX) return SSD_ERR_VALUE;
    PackArgs a;
    for (int i = 0; i < count; ++i) {
        if (!items[i].dst || items[i].N <= 0 || items[i].K <= 0) return SSD_ERR_VALUE;
        a.it[i] = PackItem{nullptr, static_cast<bf16_raw*>(items[i].dst), items[i].N, items[i].K};
    }
    for (int i = count; i < SSD_CHAIN_PACK_MAX; ++i) a.it[i] = a.it[0];
    hipLaunchKernelGGL(k_chain_prefetch, dim3(64), dim3(256), 0, (hipStream_t)stream, a, count, static_cast<unsigned*>(nullptr));
    return ssd_launch_status();
}

int ssd_conv_chain(const void* in0, const ssd_chain_layer* layers, int nlayers, int B, void* stream) {
    if (!in0 || !layers || nlayers <= 0 || nlayers > SSD_CHAIN_MAX_LAYERS || B <= 0) return SSD_ERR_VALUE;
    ChainArgs a;
    a.nlayers = nlayers;
    a.in0 = static_cast<const bf16_raw*>(in0);
    size_t region[2] = {0, 0};                                   // image l lives in region l & 1 (image 0 = the chain's input)
    auto image_bytes = [](int pixels, int ch) { return (size_t)pixels * ((size_t)ch * 2 + 16); };
    for (int l = 0; l < nlayers; ++l) {
        const ssd_chain_layer& s = layers[l];
        if (!s.w || !s.out || s.Hi <= 0 || s.Wi <= 0 || s.Ho <= 0 || s.Wo <= 0 || s.Kc <= 0 || s.N <= 0 || s.ksize <= 0 || s.mul <= 0 ||
            s.pad_t < 0 || s.pad_l < 0)
            return SSD_ERR_VALUE;
        if (l > 0 && (s.Hi != layers[l - 1].Ho || s.Wi != layers[l - 1].Wo || s.Kc != layers[l - 1].N)) return SSD_ERR_VALUE;
        if ((s.Kc & 127) || (s.N & 15) || s.Ho * s.Wo > 16 * CH_MAX_MT || (s.div != 1 && s.div != 2) || s.ksize > 7) return SSD_ERR_UNSUPPORTED;
        if ((long long)B * s.Ho * s.Wo * s.N >= (1ll << 31) || (long long)s.N * s.ksize * s.ksize * s.Kc >= (1ll << 31)) return SSD_ERR_UNSUPPORTED;
        ChainLayer& d = a.L[l];
        d.w = static_cast<const bf16_raw*>(s.w); d.bias = s.bias; d.out = static_cast<bf16_raw*>(s.out);
        d.mask_bits = static_cast<const unsigned char*>(s.mask_bits); d.mask_src = static_cast<const bf16_raw*>(s.mask_src);
        d.relu_bits = static_cast<unsigned char*>(s.relu_bits);
        d.Hi = s.Hi; d.Wi = s.Wi; d.Kc = s.Kc; d.Ho = s.Ho; d.Wo = s.Wo; d.N = s.N; d.KH = d.KW = s.ksize; d.mul = s.mul;
        d.dshift = s.div == 2 ? 1 : 0; d.pad_t = s.pad_t; d.pad_l = s.pad_l; d.relu = s.relu; d.accumulate = s.accumulate;
        const size_t in_b = image_bytes(s.Hi * s.Wi, s.Kc);
        if (in_b > region[l & 1]) region[l & 1] = in_b;
        if (l + 1 < nlayers) {
            const size_t out_b = image_bytes(s.Ho * s.Wo, s.N);
            if (out_b > region[(l + 1) & 1]) region[(l + 1) & 1] = out_b;
        }
    }
    region[0] = ssd_align_up(region[0], 16);
    region[1] = ssd_align_up(region[1], 16);
    size_t zero_bytes = 0;
    for (int l = 0; l < nlayers; ++l) zero_bytes = zero_bytes > (size_t)layers[l].Kc * 2 ? zero_bytes : (size_t)layers[l].Kc * 2;
    a.zero_off = (int)(region[0] + region[1]);
    a.zero_bytes = (int)zero_bytes;
    a.touch = ssd_knob("SSD_CHAIN_TOUCH", 0);
    const size_t lds = region[0] + region[1] + zero_bytes;
    if (lds > (size_t)CH_LDS_MAX) return SSD_ERR_UNSUPPORTED;
    for (int l = 0; l < nlayers; ++l) {
        a.L[l].in_off = (l & 1) ? (int)region[0] : 0;
        a.L[l].out_off = l + 1 < nlayers ? (((l + 1) & 1) ? (int)region[0] : 0) : -1;
    }
    for (int l = nlayers; l < SSD_CHAIN_MAX_LAYERS; ++l) a.L[l] = a.L[0];
    const bool four = ssd_knob("SSD_CHAIN_WAVES", 8) == 4;
    const void* fn = four ? reinterpret_cast<const void*>(k_conv_chain<256>) : reinterpret_cast<const void*>(k_conv_chain<512>);
    std::atomic<int>& once = four ? g_lds_set4 : g_lds_set;
    if (lds > 65536 && !once.load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS_MAX) != hipSuccess) return SSD_ERR_LAUNCH;
        once.store(1, std::memory_order_release);
    }
    if (four) hipLaunchKernelGGL(k_conv_chain<256>, dim3(B), dim3(256), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_conv_chain<512>, dim3(B), dim3(512), lds, (hipStream_t)stream, a);
    return ssd_launch_status();
}

}  // extern "C"
