// SSD training loss, forward + backward, for gfx950 (MI355X).
//
// Replaces SSDObjectDetectionModel._ssd_loss (models/ssd_model.py:341-396) and its autodiff:
//   L_pos = sum(CE(conf, gt_cls) * pos) / P                                  (:354-358)
//   ce_bg = CE(conf, C-1) * (1 - pos);  tau = (3P)-th largest of ce_bg over the whole
//           (micro)batch; neg = ce_bg >= tau;  L_neg = sum(ce_bg * neg) / sum(neg)   (:362-380)
//   L_loc = sum(sum_4 |pred_box - gt_box| * pos) / P                          (:383-386)
// with CE = logsumexp(z) - z[label] (tf.nn.sparse_softmax_cross_entropy_with_logits).
// Gradients (what tf.GradientTape returns for the sum of the three terms; the comparison that
// builds `neg` is not differentiable):
//   dconf = pos * (softmax - onehot(gt_cls)) / P + neg * (softmax - onehot(C-1)) / N
//   dloc  = pos * sign(pred_box - gt_box) / P
//
// HBM plan: `conf` (B*A*C logits, the only large tensor) is read ONCE, coalesced through LDS
// (k_loss_rows); the threshold tau is found exactly by a 3-level radix select over the B*A f32 keys
// (k_loss_hist: 2 tiny passes); `dconf` is written once, coalesced (k_loss_grad) -- only the
// ~4P selected rows re-read their logits.  All reductions are deterministic (integer atomics and
// fixed-order partial sums only).
#include "common.h"
#include <hip/hip_bf16.h>
#include "rowblock.h"

namespace {

constexpr int WG = 256;
constexpr int ROWS = 128;                    // anchor rows per workgroup (2 threads per row)
constexpr int HB1 = 2048, HB2 = 2048, HB3 = 1024;   // radix digits: 11 + 11 + 10 bits
// Level-1 keys of a real batch sit in one or two bins, and every workgroup's flush would hit the
// same few addresses (same-address global atomics serialise at ~12 ns each): the level-1 histogram
// and the positive counter are replicated NREP times (workgroup b uses replica b % NREP) and the
// streaming kernels run as persistent grids so that each workgroup flushes once.
constexpr int NREP = 4;                      // (16 until round 4: k_loss_hist<2> spent most of its 15 us summing 16 x 8 KB per workgroup; four replicas
                                             //  put <= 192 same-address flushes on a bin over the ~30 us of k_loss_rows)
constexpr int H1STRIDE = HB1 + 16;           // [HB1] bins, then [HB1] = P, padded
constexpr int MAX_PERSIST = 768;             // 3 workgroups per CU

struct LossWs {                              // layout of the caller's workspace
    float* ce_bg;                            // [n] masked background CE (0 at positives)
    float* lse;                              // [n] logsumexp per anchor
    int* hist1;                              // [NREP][H1STRIDE] | hist2 [HB2] | hist3 [HB3] | counters[8]  (zeroed per call)
    int* hist1s;                             // [H1STRIDE] replicas collapsed (by k_loss_hist<2>)
    int* hist2;
    int* hist3;
    int* counters;                           // [4..7] = select result {tau_bits, n_neg lo, n_neg hi, ok}; [3] = P; [2] = a logit / offset was not finite
    double* part_pos;                        // [nblk] per-block sum of positive CE
    double* part_l1;                         // [nblk] per-block sum of |pred-gt| over positives
    double* part_neg;                        // [nblk] per-block sum of selected background CE
    size_t zero_bytes;                       // bytes to clear starting at hist1
    size_t hist_words;                       // 32-bit words from hist1 up to (not including) counters: what the rows form re-zeroes itself
};

__host__ __device__ inline size_t al256(size_t x) { return (x + 255) / 256 * 256; }

inline size_t loss_ws_layout(size_t n, char* base, LossWs* w) {
    const size_t nblk = (n + ROWS - 1) / ROWS;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += al256(bytes); return p; };
    char* p_ce = take(n * sizeof(float));
    char* p_lse = take(n * sizeof(float));
    const size_t zero_start = off;
    char* p_h1 = take((size_t)NREP * H1STRIDE * sizeof(int));
    char* p_h1s = take((size_t)H1STRIDE * sizeof(int));
    char* p_h2 = take(HB2 * sizeof(int));
    char* p_h3 = take(HB3 * sizeof(int));
    char* p_cnt = take(8 * sizeof(int));
    const size_t zero_end = off;
    char* p_pp = take(nblk * sizeof(double));
    char* p_pl = take(nblk * sizeof(double));
    char* p_pn = take(nblk * sizeof(double));
    if (w) {
        w->ce_bg = (float*)p_ce; w->lse = (float*)p_lse;
        w->hist1 = (int*)p_h1; w->hist1s = (int*)p_h1s; w->hist2 = (int*)p_h2; w->hist3 = (int*)p_h3; w->counters = (int*)p_cnt;
        w->part_pos = (double*)p_pp; w->part_l1 = (double*)p_pl; w->part_neg = (double*)p_pn;
        w->zero_bytes = zero_end - zero_start;
        w->hist_words = (size_t)(p_cnt - p_h1) / 4;
    }
    return off;
}

__device__ __forceinline__ double block_sum(double v, double* s_red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int lo = __shfl_xor(__double2loint(v), off);
        const int hi = __shfl_xor(__double2hiint(v), off);
        v += __hiloint2double(hi, lo);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// ------------------------------------------------------------------------------------------------
// Pass 1: one read of conf.  Per anchor: logsumexp, both CEs; per block: partial sums, P; level-1
// histogram of the masked background CE keys.
// CC: the logit count when it is known at compile time (81), else 0.  Fixed count: both per-row loops unroll (the shorter half
// row pads with max(m, -inf) / s + 0, which change no bit), the LDS reads of a half row are issued together, and the next
// block's logits are requested before the current block is reduced (as k_score_decode).
template <typename T, int CC>
__global__ __launch_bounds__(WG) void k_loss_rows(const T* __restrict__ conf, const T* __restrict__ loc,
                                                  const int* __restrict__ cls, const float* __restrict__ gloc,
                                                  const uint8_t* __restrict__ mask, size_t n, int C_rt, LossWs w) {
    extern __shared__ __attribute__((aligned(16))) float s_z[];   // [ROWS*C]
    __shared__ int s_hist[HB1];
    __shared__ double s_red[4];
    const int C = CC ? CC : C_rt;
    const size_t nblk = (n + ROWS - 1) / ROWS;
    for (int i = threadIdx.x; i < HB1; i += WG) s_hist[i] = 0;
    int my_pos = 0;
    bool not_finite = false;                  // NaN / Inf in a logit row or in a positive's offsets (status 3)
    const int r = threadIdx.x >> 1, half = threadIdx.x & 1;
    const int k0 = half ? (C + 1) / 2 : 0, k1 = half ? C : (C + 1) / 2;
    constexpr int FIXED = (CC + 1) / 2;                       // trip count of the longer half row
    constexpr int NV = CC ? (ROWS * CC * (int)sizeof(T) / 16 + WG - 1) / WG : 1;
    uint4 raw[NV];
    if constexpr (CC != 0) {
        if (blockIdx.x < nblk) {
            const size_t row0 = (size_t)blockIdx.x * ROWS;
            stage_load<T, NV>(conf + row0 * C, (size_t)min((size_t)ROWS, n - row0) * C, raw);
        }
    }
    for (size_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const size_t row0 = blk * ROWS;
        const int nrow = (int)min((size_t)ROWS, n - row0);
        __syncthreads();                                   // previous block's LDS reads are done
        if constexpr (CC != 0) stage_store<T, NV>(conf + row0 * C, (size_t)nrow * C, raw, s_z);
        else stage_block<T>(conf + row0 * C, (size_t)nrow * C, s_z);
        __syncthreads();
        if constexpr (CC != 0) {
            const size_t nxt = blk + gridDim.x;
            if (nxt < nblk) {
                const size_t r1 = nxt * ROWS;
                stage_load<T, NV>(conf + r1 * C, (size_t)min((size_t)ROWS, n - r1) * C, raw);
            }
        }
        double acc_pos = 0.0, acc_l1 = 0.0;
        if (r < nrow) {
            const float* z = s_z + r * C;
            float m = -INFINITY;
            float s = 0.f;
            if constexpr (CC != 0) {
                float zz[FIXED];
#pragma unroll
                for (int j = 0; j < FIXED; ++j) zz[j] = k0 + j < k1 ? z[k0 + j] : -INFINITY;
#pragma unroll
                for (int j = 0; j < FIXED; ++j) m = fmaxf(m, zz[j]);
                m = fmaxf(m, __shfl_xor(m, 1));
#pragma unroll
                for (int j = 0; j < FIXED; ++j) s += k0 + j < k1 ? __expf(zz[j] - m) : 0.f;
            } else {
                for (int k = k0; k < k1; ++k) m = fmaxf(m, z[k]);
                m = fmaxf(m, __shfl_xor(m, 1));
                for (int k = k0; k < k1; ++k) s += __expf(z[k] - m);
            }
            s += __shfl_xor(s, 1);
            if (half == 0) {
                const size_t g = row0 + r;
                const float logs = __logf(s);
                const float lse = m + logs;
                not_finite |= !(fabsf(lse) < INFINITY);
                const bool is_pos = mask[g] != 0;
                const float ce_bg = is_pos ? 0.f : (m - z[C - 1]) + logs;      // >= 0 by construction
                w.ce_bg[g] = ce_bg;
                w.lse[g] = lse;
                atomicAdd(&s_hist[__float_as_uint(ce_bg) >> 21], 1);
                if (is_pos) {
                    ++my_pos;
                    acc_pos = (double)((m - z[cls[g]]) + logs);
                    const float4 gl = reinterpret_cast<const float4*>(gloc)[g];
                    const T* pl = loc + 4 * g;
                    acc_l1 = (double)(fabsf(to_f32<T>(pl[0]) - gl.x) + fabsf(to_f32<T>(pl[1]) - gl.y) +
                                      fabsf(to_f32<T>(pl[2]) - gl.z) + fabsf(to_f32<T>(pl[3]) - gl.w));
                    not_finite |= !(acc_l1 < (double)INFINITY);
                }
            }
        }
        const double bp = block_sum(acc_pos, s_red);
        const double bl = block_sum(acc_l1, s_red);
        if (threadIdx.x == 0) { w.part_pos[blk] = bp; w.part_l1[blk] = bl; }
    }
    __shared__ int s_np;
    if (threadIdx.x == 0) s_np = 0;
    __syncthreads();
    if (my_pos) atomicAdd(&s_np, my_pos);
    __syncthreads();
    int* rep = w.hist1 + (size_t)(blockIdx.x % NREP) * H1STRIDE;
    for (int i = threadIdx.x; i < HB1; i += WG)
        if (s_hist[i]) atomicAdd(&rep[i], s_hist[i]);
    if (threadIdx.x == 0 && s_np) atomicAdd(&rep[HB1], s_np);
    if (__any(not_finite) && (threadIdx.x & 63) == 0) atomicOr(&w.counters[2], 1);
}

// Find, in a histogram of `nb` bins scanned from the top, the bin holding the k-th largest key.
// Returns the bin (-1 if there are fewer than k keys); *k_in_bin = rank of the target inside that
// bin (1-based), *above = number of keys in higher bins.  Executed redundantly by every
// workgroup (nb <= 8*WG): each thread owns `per` consecutive bins, a workgroup prefix sum finds
// the owner of rank k, who then walks its own bins.
__device__ __forceinline__ int find_bin(const int* __restrict__ hist, int nb, long long k, long long* k_in_bin,
                                        long long* above, int* s_scan) {
    __shared__ int s_bin;
    __shared__ long long s_kin, s_above;
    const int per = (nb + WG - 1) / WG;
    const int hi = nb - 1 - (int)threadIdx.x * per;          // my highest bin
    int h[8];
    int mine = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int b = hi - j;
        h[j] = (j < per && b >= 0) ? hist[b] : 0;
        mine += h[j];
    }
    // inclusive prefix over threads (thread 0 owns the top bins)
    int incl = mine;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    __syncthreads();                                         // previous users of s_scan / s_bin are done
    if (lane == 63) s_scan[wave] = incl;
    if (threadIdx.x == 0) { s_bin = -1; s_kin = 0; s_above = 0; }
    __syncthreads();
    long long base = 0;
    for (int w = 0; w < wave; ++w) base += s_scan[w];
    const long long before = base + incl - mine;             // keys in bins above mine
    if (k > 0 && before < k && k <= before + mine) {         // exactly one thread
        long long run = before;
        int b = hi;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (run + h[j] >= k) { b = hi - j; break; }
            run += h[j];
        }
        s_bin = b; s_kin = k - run; s_above = run;
    }
    __syncthreads();
    *k_in_bin = s_kin;
    *above = s_above;
    return s_bin;
}

__device__ __forceinline__ int load_num_pos(const LossWs& w) { return w.hist1s[HB1]; }

// Radix select, levels 2 and 3: histogram the next digit of the keys that share the prefix so far.
// LEVEL 2 also collapses the NREP replicas of the level-1 histogram (and of P): every workgroup sums them into its own
// LDS copy (132 KB from L2 each: cheaper than the launch a separate collapse kernel costs), workgroup 0 stores the sums for
// the kernels behind (hist1s).
template <int LEVEL>
__global__ __launch_bounds__(WG) void k_loss_hist(size_t n, LossWs w) {
    __shared__ int s_scan[WG];
    __shared__ int s_hist[HB2];
    __shared__ int s_h1[LEVEL == 2 ? H1STRIDE : 1];
    const int* h1 = w.hist1s;
    if constexpr (LEVEL == 2) {
        for (int b = threadIdx.x; b <= HB1; b += WG) {
            int v[NREP];
#pragma unroll
            for (int rp = 0; rp < NREP; ++rp) v[rp] = w.hist1[(size_t)rp * H1STRIDE + b];
            int sum = 0;
#pragma unroll
            for (int rp = 0; rp < NREP; ++rp) sum += v[rp];
            s_h1[b] = sum;
            if (blockIdx.x == 0) w.hist1s[b] = sum;
        }
        __syncthreads();
        h1 = s_h1;
    }
    const long long k = 3ll * h1[HB1];
    long long kin, above;
    const int b1 = find_bin(h1, HB1, k, &kin, &above, s_scan);
    unsigned prefix = (unsigned)b1, shift = 21;
    int nb = HB2;
    if (LEVEL == 3) {
        long long kin2, above2;
        const int b2 = find_bin(w.hist2, HB2, kin, &kin2, &above2, s_scan);
        prefix = ((unsigned)b1 << 11) | (unsigned)b2;
        shift = 10;
        nb = HB3;
    }
    if (b1 < 0) return;
    for (int i = threadIdx.x; i < nb; i += WG) s_hist[i] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
        const unsigned key = __float_as_uint(w.ce_bg[i]);
        if ((key >> shift) == prefix) atomicAdd(&s_hist[LEVEL == 2 ? (key >> 10) & 2047u : key & 1023u], 1);
    }
    __syncthreads();
    int* out = LEVEL == 2 ? w.hist2 : w.hist3;
    for (int i = threadIdx.x; i < nb; i += WG)
        if (s_hist[i]) atomicAdd(&out[i], s_hist[i]);
}

struct Select {                              // result of the radix select
    unsigned tau_bits;
    long long n_neg;                         // number of keys >= tau (ties included, models/ssd_model.py:372)
    int ok;
};

__device__ __forceinline__ Select finish_select(const LossWs& w, int* s_scan) {
    Select s;
    const long long k = 3ll * load_num_pos(w);
    long long kin1, ab1, kin2, ab2, kin3, ab3;
    const int b1 = find_bin(w.hist1s, HB1, k, &kin1, &ab1, s_scan);
    if (b1 < 0 || k <= 0) { s.tau_bits = 0; s.n_neg = 0; s.ok = 0; return s; }
    const int b2 = find_bin(w.hist2, HB2, kin1, &kin2, &ab2, s_scan);
    const int b3 = find_bin(w.hist3, HB3, kin2, &kin3, &ab3, s_scan);
    s.tau_bits = ((unsigned)b1 << 21) | ((unsigned)b2 << 10) | (unsigned)b3;
    s.n_neg = ab1 + ab2 + ab3 + w.hist3[b3];
    s.ok = 1;
    return s;
}

__global__ __launch_bounds__(WG) void k_loss_select(LossWs w) {
    __shared__ int s_scan[WG];
    const Select sel = finish_select(w, s_scan);
    if (threadIdx.x == 0) {
        w.counters[4] = (int)sel.tau_bits;
        w.counters[5] = (int)(sel.n_neg & 0xffffffffll);
        w.counters[6] = (int)(sel.n_neg >> 32);
        w.counters[7] = sel.ok;
        w.counters[3] = load_num_pos(w);
    }
}

__device__ __forceinline__ Select load_select(const LossWs& w) {
    Select s;
    s.tau_bits = (unsigned)w.counters[4];
    s.n_neg = (long long)(unsigned)w.counters[5] | ((long long)w.counters[6] << 32);
    s.ok = w.counters[7];
    return s;
}

// Pass 2: gradients, written once and coalesced.  Only selected rows re-read their logits.
template <typename T>
__global__ __launch_bounds__(WG) void k_loss_grad(const T* __restrict__ conf, const T* __restrict__ loc,
                                                  const int* __restrict__ cls, const float* __restrict__ gloc,
                                                  const uint8_t* __restrict__ mask, size_t n, int C, float grad_scale,
                                                  T* __restrict__ dconf, T* __restrict__ dloc, LossWs w) {
    extern __shared__ __attribute__((aligned(16))) float s_g[];   // [ROWS*C] gradient block
    __shared__ double s_red[4];
    const Select sel = load_select(w);
    const size_t row0 = (size_t)blockIdx.x * ROWS;
    const int nrow = (int)min((size_t)ROWS, n - row0);
    const float tau = __uint_as_float(sel.tau_bits);
    const float P = (float)w.counters[3];
    const float inv_p = sel.ok ? grad_scale / P : 0.f;
    const float inv_n = sel.ok && sel.n_neg > 0 ? grad_scale / (float)sel.n_neg : 0.f;

    for (int i = threadIdx.x; i < (ROWS * C + 3) / 4; i += WG)
        reinterpret_cast<float4*>(s_g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int r = threadIdx.x >> 1, half = threadIdx.x & 1;
    double acc_neg = 0.0;
    if (r < nrow) {
        const size_t g = row0 + r;
        const bool pos = mask[g] != 0;
        const float ce = w.ce_bg[g];
        const bool neg = !pos && sel.ok && ce >= tau;
        if (pos || neg) {
            const float lse = w.lse[g];
            const float sc = pos ? inv_p : inv_n;
            const int label = pos ? cls[g] : C - 1;
            const T* z = conf + g * C;
            float* o = s_g + r * C;
            const int k0 = half ? (C + 1) / 2 : 0, k1 = half ? C : (C + 1) / 2;
            for (int k = k0; k < k1; ++k) o[k] = (__expf(to_f32<T>(z[k]) - lse) - (k == label ? 1.f : 0.f)) * sc;
            if (neg && half == 0) acc_neg = (double)ce;
        }
        if (half == 0) {
            float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pos) {
                const float4 gl = reinterpret_cast<const float4*>(gloc)[g];
                const T* pl = loc + 4 * g;
                const float e0 = to_f32<T>(pl[0]) - gl.x, e1 = to_f32<T>(pl[1]) - gl.y;
                const float e2 = to_f32<T>(pl[2]) - gl.z, e3 = to_f32<T>(pl[3]) - gl.w;
                d.x = e0 > 0.f ? inv_p : (e0 < 0.f ? -inv_p : 0.f);
                d.y = e1 > 0.f ? inv_p : (e1 < 0.f ? -inv_p : 0.f);
                d.z = e2 > 0.f ? inv_p : (e2 < 0.f ? -inv_p : 0.f);
                d.w = e3 > 0.f ? inv_p : (e3 < 0.f ? -inv_p : 0.f);
            }
            T* o = dloc + 4 * g;
            o[0] = from_f32<T>(d.x); o[1] = from_f32<T>(d.y); o[2] = from_f32<T>(d.z); o[3] = from_f32<T>(d.w);
        }
    }
    const double bn = block_sum(acc_neg, s_red);       // also orders the LDS writes before the stores
    if (threadIdx.x == 0) w.part_neg[blockIdx.x] = bn;
    // coalesced store of the gradient block
    T* out = dconf + row0 * C;
    const size_t count = (size_t)nrow * C;
    if constexpr (sizeof(T) == 4) {
        const size_t nvec = count / 4;
        for (size_t i = threadIdx.x; i < nvec; i += WG)
            reinterpret_cast<float4*>(out)[i] = *reinterpret_cast<const float4*>(s_g + 4 * i);
        for (size_t i = nvec * 4 + threadIdx.x; i < count; i += WG) out[i] = s_g[i];
    } else {
        const size_t nvec = count / 8;
        for (size_t i = threadIdx.x; i < nvec; i += WG) {
            const float* f = s_g + 8 * i;
            unsigned wds[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const __hip_bfloat16 a = __float2bfloat16(f[2 * k]), b = __float2bfloat16(f[2 * k + 1]);
                wds[k] = (unsigned)(*reinterpret_cast<const unsigned short*>(&a)) |
                         ((unsigned)(*reinterpret_cast<const unsigned short*>(&b)) << 16);
            }
            reinterpret_cast<uint4*>(out)[i] = make_uint4(wds[0], wds[1], wds[2], wds[3]);
        }
        for (size_t i = nvec * 8 + threadIdx.x; i < count; i += WG) out[i] = from_f32<T>(s_g[i]);
    }
}

// Final: fixed-order reduction of the per-block partial sums -> the reference's three loss scalars.
// out[0..7] = loc, cls-pos, cls-neg, total, P, N, tau, status (0 ok; 1 = P==0 / k>n (TF top_k would
// raise); 2 = tau==0, i.e. the reference's assert at models/ssd_model.py:375 would fire; 3 = a logit row or a positive's
// offsets were not finite -- the reference would log NaN losses from there on; takes precedence: a diverged run must not
// look like an empty batch)
__global__ __launch_bounds__(WG) void k_loss_final(size_t nblk, LossWs w, float* __restrict__ out) {
    __shared__ double s_red[4];
    const Select sel = load_select(w);
    double a = 0.0, b = 0.0, c = 0.0;
    for (size_t i = threadIdx.x; i < nblk; i += WG) { a += w.part_pos[i]; b += w.part_l1[i]; c += w.part_neg[i]; }
    a = block_sum(a, s_red);
    b = block_sum(b, s_red);
    c = block_sum(c, s_red);
    if (threadIdx.x == 0) {
        const double P = (double)w.counters[3];
        const float l_pos = sel.ok ? (float)(a / P) : 0.f;
        const float l_loc = sel.ok ? (float)(b / P) : 0.f;
        const float l_neg = sel.ok && sel.n_neg > 0 ? (float)(c / (double)sel.n_neg) : 0.f;
        out[0] = l_loc; out[1] = l_pos; out[2] = l_neg; out[3] = l_loc + l_pos + l_neg;
        out[4] = (float)P; out[5] = (float)sel.n_neg; out[6] = __uint_as_float(sel.tau_bits);
        const bool bad = w.counters[2] != 0 || !(fabs(a) < (double)INFINITY) || !(fabs(b) < (double)INFINITY) || !(fabs(c) < (double)INFINITY);
        out[7] = bad ? 3.f : (!sel.ok ? 1.f : (sel.tau_bits == 0 ? 2.f : 0.f));
    }
}

// ------------------------------------------------------------------------------------------------
// Gradient as compact per-level pixel rows (ssd_loss_fwd_bwd_heads).  An anchor carries a gradient iff it is a
// positive or a mined negative (`selected`): ~4P of B*A rows.  Three launches behind the radix select:
//   k_hg_count   (image, level): pixels with at least one selected anchor
//   k_hg_assign  (image, level): rows in ascending pixel order (offset = counts of the images before: fixed, no
//                atomics), both index maps, the rows zero-filled
//   k_loss_grad_rows: the gradient of every selected anchor into its segment of its pixel's row
struct HeadGradsDev {
    int levels, A, B;
    int hw[SSD_MAX_LEVELS], n[SSD_MAX_LEVELS], npad[SSD_MAX_LEVELS], off[SSD_MAX_LEVELS + 1];
    __hip_bfloat16* rows[SSD_MAX_LEVELS];
    int* rop[SSD_MAX_LEVELS];
    int* por[SSD_MAX_LEVELS];
    int* count;                              // [SSD_MAX_LEVELS]
    int* img_count;                          // [levels][B] (workspace)
};

__device__ __forceinline__ bool hg_selected(const uint8_t* __restrict__ mask, const float* __restrict__ ce_bg, size_t g,
                                            const Select& sel, float tau) {
    return sel.ok && (mask[g] != 0 || ce_bg[g] >= tau);
}

__device__ __forceinline__ bool hg_pixel_flag(const HeadGradsDev& h, const uint8_t* __restrict__ mask, const LossWs& w,
                                              int l, int b, int pix, const Select& sel, float tau) {
    const size_t g0 = (size_t)b * h.A + h.off[l] + (size_t)pix * h.n[l];
    bool f = false;
    for (int a = 0; a < h.n[l]; ++a) f |= hg_selected(mask, w.ce_bg, g0 + a, sel, tau);
    return f;
}

__device__ __forceinline__ int block_sum_int(int v, int* s4) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return s4[0] + s4[1] + s4[2] + s4[3];
}

// (also the last step of the radix select: every workgroup evaluates it from the three histograms -- the separate
// one-workgroup launch cost more than 384 redundant evaluations -- and workgroup (0, 0) stores it for the kernels behind)
__global__ __launch_bounds__(WG) void k_hg_count(const uint8_t* __restrict__ mask, LossWs w, HeadGradsDev h, double* __restrict__ part_neg_il,
                                                 size_t nblk) {
    __shared__ int s4[4];
    __shared__ int s_scan[WG];
    __shared__ double s_red[4];
    const int b = blockIdx.x, l = blockIdx.y;
    const Select sel = finish_select(w, s_scan);
    if (b == 0 && l == 0 && threadIdx.x == 0) {
        w.counters[4] = (int)sel.tau_bits;
        w.counters[5] = (int)(sel.n_neg & 0xffffffffll);
        w.counters[6] = (int)(sel.n_neg >> 32);
        w.counters[7] = sel.ok;
        w.counters[3] = load_num_pos(w);
    }
    const float tau = __uint_as_float(sel.tau_bits);
    int c = 0;
    double neg = 0.0;                          // sum of the mined negatives' background CE of this (image, level): the loss's third
    for (int pix = threadIdx.x; pix < h.hw[l]; pix += WG) {     // scalar no longer waits for the gradient pass
        const size_t g0 = (size_t)b * h.A + h.off[l] + (size_t)pix * h.n[l];
        bool f = false;
        for (int a = 0; a < h.n[l]; ++a) {
            const bool pos = mask[g0 + a] != 0;
            const float ce = w.ce_bg[g0 + a];
            const bool ng = !pos && sel.ok && ce >= tau;
            f |= sel.ok && (pos || ng);
            if (ng) neg += (double)ce;
        }
        c += f ? 1 : 0;
    }
    const int tot = block_sum_int(c, s4);
    const double bn = block_sum(neg, s_red);
    // ... and a slice of k_loss_rows' per-block sums (positive CE, L1): the 4 366 partial sums of a batch-64 call reach the
    // workgroup that writes the scalars as levels x B values each (one workgroup summing them all took 6 us)
    const int idx = l * h.B + b, nwg = h.levels * h.B;
    const size_t per = (nblk + nwg - 1) / nwg, i0 = (size_t)idx * per;
    double pp = 0.0, pl = 0.0;
    for (size_t i = i0 + threadIdx.x; i < min(nblk, i0 + per); i += WG) { pp += w.part_pos[i]; pl += w.part_l1[i]; }
    pp = block_sum(pp, s_red);
    pl = block_sum(pl, s_red);
    if (threadIdx.x == 0) {
        h.img_count[idx] = tot;
        part_neg_il[idx] = bn; part_neg_il[nwg + idx] = pp; part_neg_il[2 * nwg + idx] = pl;
    }
}

__global__ __launch_bounds__(WG) void k_hg_assign(const uint8_t* __restrict__ mask, LossWs w, HeadGradsDev h, size_t nblk,
                                                  const double* __restrict__ part_neg_il, float* __restrict__ out) {
    __shared__ int s4[4];
    __shared__ int s_wave[4];
    __shared__ double s_red[4];
    const int b = blockIdx.x, l = blockIdx.y;
    const Select sel = load_select(w);
    const float tau = __uint_as_float(sel.tau_bits);
    int before = 0;
    for (int i = threadIdx.x; i < b; i += WG) before += h.img_count[l * h.B + i];
    int base = block_sum_int(before, s4);                     // rows of this level in the images before this one
    const int hw = h.hw[l], npad = h.npad[l];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int p0 = 0; p0 < hw; p0 += WG) {
        const int pix = p0 + threadIdx.x;
        const bool f = pix < hw && hg_pixel_flag(h, mask, w, l, b, pix, sel, tau);
        const unsigned long long bal = __ballot(f);
        __syncthreads();                                     // previous chunk's s_wave reads are done
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int prefix = __popcll(bal & ((1ull << lane) - 1ull));
        for (int k = 0; k < wave; ++k) prefix += s_wave[k];
        const int chunk_total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        if (pix < hw) {
            const int flat = b * hw + pix;
            const int row = f ? base + prefix : -1;
            h.rop[l][flat] = row;
            if (f) {
                h.por[l][row] = flat;
                uint4* dst = reinterpret_cast<uint4*>(h.rows[l] + (size_t)row * npad);
                for (int k = 0; k < npad / 8; ++k) dst[k] = make_uint4(0, 0, 0, 0);
            }
        }
        base += chunk_total;
    }
    if (b == h.B - 1 && threadIdx.x == 0) h.count[l] = base;
    if (b == 0 && l == 0) {
        // the loss scalars (k_loss_final's arithmetic, same fixed orders): everything they need exists once k_hg_count has run,
        // so this workgroup writes them here instead of a one-workgroup launch at the end of the chain
        double a = 0.0, bb = 0.0, c = 0.0;
        const int nwg = h.levels * h.B;
        for (int i = threadIdx.x; i < nwg; i += WG) { c += part_neg_il[i]; a += part_neg_il[nwg + i]; bb += part_neg_il[2 * nwg + i]; }
        a = block_sum(a, s_red);
        bb = block_sum(bb, s_red);
        c = block_sum(c, s_red);
        if (threadIdx.x == 0) {
            const double P = (double)w.counters[3];
            const float l_pos = sel.ok ? (float)(a / P) : 0.f;
            const float l_loc = sel.ok ? (float)(bb / P) : 0.f;
            const float l_neg = sel.ok && sel.n_neg > 0 ? (float)(c / (double)sel.n_neg) : 0.f;
            out[0] = l_loc; out[1] = l_pos; out[2] = l_neg; out[3] = l_loc + l_pos + l_neg;
            out[4] = (float)P; out[5] = (float)sel.n_neg; out[6] = __uint_as_float(sel.tau_bits);
            const bool bad = w.counters[2] != 0 || !(fabs(a) < (double)INFINITY) || !(fabs(bb) < (double)INFINITY) || !(fabs(c) < (double)INFINITY);
            out[7] = bad ? 3.f : (!sel.ok ? 1.f : (sel.tau_bits == 0 ? 2.f : 0.f));
            w.counters[2] = 0;                 // (the non-finite flag is an atomicOr target: clean for the next call)
        }
    }
}

// Gradient of the selected anchors only, written into the compact rows (the arithmetic of k_loss_grad).
__global__ __launch_bounds__(WG) void k_loss_grad_rows(const __hip_bfloat16* __restrict__ conf, const __hip_bfloat16* __restrict__ loc,
                                                       const int* __restrict__ cls, const float* __restrict__ gloc,
                                                       const uint8_t* __restrict__ mask, size_t n, int C, float grad_scale,
                                                       LossWs w, HeadGradsDev h) {
    typedef __hip_bfloat16 T;
    const Select sel = load_select(w);
    const size_t row0 = (size_t)blockIdx.x * ROWS;
    const int nrow = (int)min((size_t)ROWS, n - row0);
    const float tau = __uint_as_float(sel.tau_bits);
    const float P = (float)w.counters[3];
    const float inv_p = sel.ok ? grad_scale / P : 0.f;
    const float inv_n = sel.ok && sel.n_neg > 0 ? grad_scale / (float)sel.n_neg : 0.f;
    const int r = threadIdx.x >> 1, half = threadIdx.x & 1;
    // the histograms are read for the last time by k_hg_count / k_hg_assign: this launch leaves them zeroed for the next call
    // (ws_clean), which saves the hipMemsetAsync node -- a fill KERNEL -- in front of every call
    for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < w.hist_words; i += (size_t)gridDim.x * WG) w.hist1[i] = 0;
    if (r < nrow) {
        const size_t g = row0 + r;
        const bool pos = mask[g] != 0;
        const float ce = w.ce_bg[g];
        const bool neg = !pos && sel.ok && ce >= tau;
        if (sel.ok && (pos || neg)) {
            const int b = (int)(g / (size_t)h.A);
            const int a = (int)(g - (size_t)b * h.A);
            int l = 0;
            for (int k = 1; k < h.levels; ++k) l += a >= h.off[k] ? 1 : 0;
            const int idx = a - h.off[l];
            const int pix = idx / h.n[l], slot = idx - pix * h.n[l];
            const int row = h.rop[l][b * h.hw[l] + pix];
            if (row >= 0) {
                T* d = h.rows[l] + (size_t)row * h.npad[l];
                const float lse = w.lse[g];
                const float sc = pos ? inv_p : inv_n;
                const int label = pos ? cls[g] : C - 1;
                const T* z = conf + g * C;
                T* o = d + h.n[l] * 4 + slot * C;
                const int k0 = half ? (C + 1) / 2 : 0, k1 = half ? C : (C + 1) / 2;
                for (int k = k0; k < k1; ++k) o[k] = from_f32<T>((__expf(to_f32<T>(z[k]) - lse) - (k == label ? 1.f : 0.f)) * sc);
                if (pos && half == 0) {
                    const float4 gl = reinterpret_cast<const float4*>(gloc)[g];
                    const T* pl = loc + 4 * g;
                    const float e0 = to_f32<T>(pl[0]) - gl.x, e1 = to_f32<T>(pl[1]) - gl.y;
                    const float e2 = to_f32<T>(pl[2]) - gl.z, e3 = to_f32<T>(pl[3]) - gl.w;
                    T* ol = d + slot * 4;
                    ol[0] = from_f32<T>(e0 > 0.f ? inv_p : (e0 < 0.f ? -inv_p : 0.f));
                    ol[1] = from_f32<T>(e1 > 0.f ? inv_p : (e1 < 0.f ? -inv_p : 0.f));
                    ol[2] = from_f32<T>(e2 > 0.f ? inv_p : (e2 < 0.f ? -inv_p : 0.f));
                    ol[3] = from_f32<T>(e3 > 0.f ? inv_p : (e3 < 0.f ? -inv_p : 0.f));
                }
            }
        }
    }
}

// the launches both forms share: conf read once, exact radix select of tau
template <typename T>
void launch_loss_select(const void* conf, const void* loc, const int32_t* cls, const float* gloc, const uint8_t* mask,
                        size_t n, int C, LossWs w, hipStream_t s, size_t lds, bool select_launch = true) {
    const size_t nblk = (n + ROWS - 1) / ROWS;
    const unsigned pgrid = (unsigned)min((size_t)MAX_PERSIST, nblk);
    if (C == 81)
        hipLaunchKernelGGL((k_loss_rows<T, 81>), dim3(pgrid), dim3(WG), lds, s, (const T*)conf, (const T*)loc, cls, gloc, mask, n, C, w);
    else
        hipLaunchKernelGGL((k_loss_rows<T, 0>), dim3(pgrid), dim3(WG), lds, s, (const T*)conf, (const T*)loc, cls, gloc, mask, n, C, w);
    const unsigned hgrid = (unsigned)min((size_t)256, (n + WG - 1) / WG);
    hipLaunchKernelGGL(k_loss_hist<2>, dim3(hgrid), dim3(WG), 0, s, n, w);
    hipLaunchKernelGGL(k_loss_hist<3>, dim3(hgrid), dim3(WG), 0, s, n, w);
    if (select_launch) hipLaunchKernelGGL(k_loss_select, dim3(1), dim3(WG), 0, s, w);
}

template <typename T>
int launch_loss(const void* conf, const void* loc, const int32_t* cls, const float* gloc, const uint8_t* mask,
                size_t n, int C, float* out, void* dconf, void* dloc, float grad_scale, LossWs w, hipStream_t s) {
    const size_t nblk = (n + ROWS - 1) / ROWS;
    const size_t lds = ((size_t)ROWS * C * sizeof(float) + 15) / 16 * 16;
    if (hipMemsetAsync(w.hist1, 0, w.zero_bytes, s) != hipSuccess) return SSD_ERR_LAUNCH;
    launch_loss_select<T>(conf, loc, cls, gloc, mask, n, C, w, s, lds);
    hipLaunchKernelGGL(k_loss_grad<T>, dim3((unsigned)nblk), dim3(WG), lds, s, (const T*)conf, (const T*)loc, cls, gloc,
                       mask, n, C, grad_scale, (T*)dconf, (T*)dloc, w);
    hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(WG), 0, s, nblk, w, out);
    return ssd_launch_status();
}

}  // namespace

extern "C" {

size_t ssd_loss_workspace_bytes(int B, int A, int C) {
    if (B <= 0 || A <= 0 || C <= 0) return 0;
    return loss_ws_layout((size_t)B * A, nullptr, nullptr);
}

size_t ssd_loss_heads_workspace_bytes(int B, int A, int C) {
    if (B <= 0 || A <= 0 || C <= 0) return 0;
    return loss_ws_layout((size_t)B * A, nullptr, nullptr) + al256((size_t)SSD_MAX_LEVELS * B * sizeof(int)) +
           al256((size_t)3 * SSD_MAX_LEVELS * B * sizeof(double));
}

int ssd_loss_fwd_bwd_heads(const void* conf, const void* loc, int dtype, const int32_t* gt_cls, const float* gt_loc,
                           const uint8_t* gt_mask, int B, int A, int C, float grad_scale, float* out8,
                           const ssd_head_grads* hg, void* ws, size_t ws_bytes, int ws_clean, void* stream) {
    if (B <= 0 || A <= 0 || C < 2 || !hg) return SSD_ERR_VALUE;
    if (!conf || !loc || !gt_cls || !gt_loc || !gt_mask || !out8 || !hg->count) return SSD_ERR_VALUE;
    if (dtype != SSD_BF16) return SSD_ERR_UNSUPPORTED;
    if ((size_t)ROWS * C * sizeof(float) > 144 * 1024) return SSD_ERR_UNSUPPORTED;
    if (hg->levels <= 0 || hg->levels > SSD_MAX_LEVELS) return SSD_ERR_VALUE;
    HeadGradsDev h;
    h.levels = hg->levels; h.A = A; h.B = B;
    h.off[0] = 0;
    for (int l = 0; l < hg->levels; ++l) {
        if (hg->hw[l] <= 0 || hg->per_cell[l] <= 0 || hg->npad[l] < hg->per_cell[l] * (4 + C) || (hg->npad[l] & 7)) return SSD_ERR_VALUE;
        if (!hg->rows[l] || !hg->row_of_pixel[l] || !hg->pixel_of_row[l]) return SSD_ERR_VALUE;
        if ((long long)B * hg->hw[l] >= (1ll << 31)) return SSD_ERR_VALUE;
        h.hw[l] = hg->hw[l]; h.n[l] = hg->per_cell[l]; h.npad[l] = hg->npad[l];
        h.off[l + 1] = h.off[l] + hg->hw[l] * hg->per_cell[l];
        h.rows[l] = (__hip_bfloat16*)hg->rows[l]; h.rop[l] = hg->row_of_pixel[l]; h.por[l] = hg->pixel_of_row[l];
    }
    for (int l = hg->levels; l < SSD_MAX_LEVELS; ++l) { h.hw[l] = h.n[l] = h.npad[l] = 0; h.off[l + 1] = h.off[l]; h.rows[l] = nullptr; h.rop[l] = h.por[l] = nullptr; }
    if (h.off[hg->levels] != A) return SSD_ERR_ASSERT;            // models/ssd_model.py:350-351
    h.count = hg->count;
    const size_t n = (size_t)B * A;
    const size_t base = loss_ws_layout(n, nullptr, nullptr);
    const size_t cnt_bytes = al256((size_t)SSD_MAX_LEVELS * B * sizeof(int));
    if (!ws || ws_bytes < base + cnt_bytes + al256((size_t)3 * SSD_MAX_LEVELS * B * sizeof(double))) return SSD_ERR_WORKSPACE;
    LossWs w;
    loss_ws_layout(n, static_cast<char*>(ws), &w);
    h.img_count = reinterpret_cast<int*>(static_cast<char*>(ws) + base);
    double* part_neg_il = reinterpret_cast<double*>(static_cast<char*>(ws) + base + cnt_bytes);
    hipStream_t s = (hipStream_t)stream;
    typedef __hip_bfloat16 T;
    const size_t nblk = (n + ROWS - 1) / ROWS;
    const size_t lds = ((size_t)ROWS * C * sizeof(float) + 15) / 16 * 16;
    // ws_clean: the caller states that the histogram words of THIS workspace layout are zero -- true after a completed call of
    // this function with the same B, A (its last launch re-zeroes them) -- and the memset node is skipped
    if (!ws_clean && hipMemsetAsync(w.hist1, 0, w.zero_bytes, s) != hipSuccess) return SSD_ERR_LAUNCH;
    launch_loss_select<T>(conf, loc, gt_cls, gt_loc, gt_mask, n, C, w, s, lds, false);
    hipLaunchKernelGGL(k_hg_count, dim3(B, hg->levels), dim3(WG), 0, s, gt_mask, w, h, part_neg_il, nblk);
    hipLaunchKernelGGL(k_hg_assign, dim3(B, hg->levels), dim3(WG), 0, s, gt_mask, w, h, nblk, (const double*)part_neg_il, out8);
    hipLaunchKernelGGL(k_loss_grad_rows, dim3((unsigned)nblk), dim3(WG), 0, s, (const T*)conf, (const T*)loc, gt_cls, gt_loc,
                       gt_mask, n, C, grad_scale, w, h);
    return ssd_launch_status();
}

int ssd_loss_fwd_bwd(const void* conf, const void* loc, int dtype, const int32_t* gt_cls, const float* gt_loc,
                     const uint8_t* gt_mask, int B, int A, int C, float grad_scale, float* out8, void* dconf,
                     void* dloc, void* ws, size_t ws_bytes, void* stream) {
    if (B <= 0 || A <= 0 || C < 2) return SSD_ERR_VALUE;
    if (!conf || !loc || !gt_cls || !gt_loc || !gt_mask || !out8 || !dconf || !dloc) return SSD_ERR_VALUE;
    if (dtype != SSD_F32 && dtype != SSD_BF16) return SSD_ERR_VALUE;
    if ((size_t)ROWS * C * sizeof(float) > 144 * 1024) return SSD_ERR_UNSUPPORTED;
    const size_t n = (size_t)B * A;
    if (!ws || ws_bytes < loss_ws_layout(n, nullptr, nullptr)) return SSD_ERR_WORKSPACE;
    LossWs w;
    loss_ws_layout(n, static_cast<char*>(ws), &w);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SSD_F32)
        return launch_loss<float>(conf, loc, gt_cls, gt_loc, gt_mask, n, C, out8, dconf, dloc, grad_scale, w, s);
    return launch_loss<__hip_bfloat16>(conf, loc, gt_cls, gt_loc, gt_mask, n, C, out8, dconf, dloc, grad_scale, w, s);
}

}  // extern "C"
