// Inference post-processing for gfx950 (MI355X): scoring + box decoding + per-image NMS.
//
//   k_score_decode  replaces the scoring of SSDObjectDetectionModel.visualize (models/ssd_model.py:
//                   479-488, mask=None branch: softmax, best non-background probability, candidate =
//                   score > thresh and not p_bg > thresh, class = argmax) and the decode at :466-467
//                   (cxcy = (t_xy*d_wh + d_xy)*300, wh = exp(t_wh)*d_wh*300; f64 then stored as f32).
//                   One coalesced read of conf through LDS; boxes are decoded for candidates only.
//   k_nms           build-defined per-image, per-class greedy hard NMS (SURVEY.md A9'; the reference has
//                   no suppression at all).  Candidates are ordered by (score desc, anchor asc); the
//                   first max_cand take part.  IoU is the reference's scalar formula (utils/bbox.py:6-25)
//                   in float32.  One workgroup per image: ordered compaction (ballot prefix), optional
//                   exact top-max_cand cut (radix select on the score bits), bitonic sort of 64-bit keys
//                   (class | ~score | anchor) in LDS, then one wave per class segment runs the greedy
//                   pass with 64-wide suppression.
// Compile with -ffp-contract=off: keep masks must be bit-exact against the oracle.
#include "common.h"
#include "rowblock.h"
#include <limits.h>

namespace {

constexpr int WG = 256;
constexpr int ROWS = 128;
constexpr int CAP = 1024;                    // candidates per image that can take part in NMS

// CC: the class count when it is known at compile time (81: the reference's 80 classes + background), else 0.  With a fixed
// trip count both per-row loops unroll, the 40 LDS reads of a half row are issued together and the exponentials pipeline;
// the order of the additions (and so every bit of the result) is that of the rolled loop.
template <typename T, int CC>
__global__ __launch_bounds__(WG) void k_score_decode(const T* __restrict__ conf, const T* __restrict__ loc,
                                                     const double* __restrict__ priors, size_t n, int A, int C_rt,
                                                     float thresh, double in_size, float* __restrict__ score,
                                                     int* __restrict__ cls, float4* __restrict__ box,
                                                     uint8_t* __restrict__ cand) {
    extern __shared__ __attribute__((aligned(16))) float s_z[];
    const int C = CC ? CC : C_rt;
    const size_t nblk = (n + ROWS - 1) / ROWS;
    const int r = threadIdx.x >> 1, half = threadIdx.x & 1;
    const int nfg = C - 1;
    const int k0 = half ? (nfg + 1) / 2 : 0, k1 = half ? nfg : (nfg + 1) / 2;
    constexpr int FIXED = CC && ((CC - 1) % 2 == 0) ? (CC - 1) / 2 : 0;      // trip count of both halves when it is the same
    // with a fixed class count the next block's logits are requested before the current block is scored (registers), and
    // written to LDS after the barrier that ends it: the loads are in flight during the exponentials
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
        __syncthreads();
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
        if (r < nrow) {
            const float* z = s_z + r * C;
            float m = -INFINITY;                 // best foreground logit of this half, first index wins
            int mi = INT_MAX;
            if constexpr (FIXED > 0) {
#pragma unroll
                for (int j = 0; j < FIXED; ++j) {
                    const int k = k0 + j;
                    if (z[k] > m) { m = z[k]; mi = k; }
                }
            } else {
                for (int k = k0; k < k1; ++k)
                    if (z[k] > m) { m = z[k]; mi = k; }
            }
            const float om = __shfl_xor(m, 1);
            const int oi = __shfl_xor(mi, 1);
            if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
            const float zb = z[nfg];
            const float top = fmaxf(m, zb);
            float s = 0.f;
            if constexpr (FIXED > 0) {
#pragma unroll
                for (int j = 0; j < FIXED; ++j) s += __expf(z[k0 + j] - top);
            } else {
                for (int k = k0; k < k1; ++k) s += __expf(z[k] - top);
            }
            s += __shfl_xor(s, 1);
            if (half == 0) {
                const size_t g = row0 + r;
                const float eb = __expf(zb - top);
                s += eb;
                const float sc = __expf(m - top) / s;
                const float pb = eb / s;
                const bool is_cand = sc > thresh && !(pb > thresh);
                score[g] = sc;
                cls[g] = mi;
                cand[g] = is_cand ? 1 : 0;
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (is_cand) {
                    const int a = (int)(g % (size_t)A);
                    const double2 lo = *reinterpret_cast<const double2*>(priors + 4 * (size_t)a);
                    const double2 hi = *reinterpret_cast<const double2*>(priors + 4 * (size_t)a + 2);
                    const T* t = loc + 4 * g;
                    o.x = (float)(((double)to_f32<T>(t[0]) * hi.x + lo.x) * in_size);
                    o.y = (float)(((double)to_f32<T>(t[1]) * hi.y + lo.y) * in_size);
                    o.z = (float)(exp((double)to_f32<T>(t[2])) * hi.x * in_size);
                    o.w = (float)(exp((double)to_f32<T>(t[3])) * hi.y * in_size);
                }
                box[g] = o;
            }
        }
    }
}

// IoU of utils/bbox.py:6-25 in float32 (sides clamped at 0, +1e-10 in the union).
__device__ __forceinline__ float iou_f32(float4 a, float4 b) {
    const float a1 = a.z * a.w, a2 = b.z * b.w;
    const float lo_x = fmaxf(a.x - a.z / 2.0f, b.x - b.z / 2.0f);
    const float lo_y = fmaxf(a.y - a.w / 2.0f, b.y - b.w / 2.0f);
    const float hi_x = fminf(a.x + a.z / 2.0f, b.x + b.z / 2.0f);
    const float hi_y = fminf(a.y + a.w / 2.0f, b.y + b.w / 2.0f);
    const float inter = fmaxf(0.0f, hi_x - lo_x) * fmaxf(0.0f, hi_y - lo_y);
    return inter / (a1 + a2 - inter + 1e-10f);
}

// exclusive prefix of a 0/1 flag over the workgroup (index order), and the total
__device__ __forceinline__ int wg_prefix(bool flag, int* s_wave, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_wave[w];
    total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    return base + before;
}

__global__ __launch_bounds__(WG) void k_nms(const float* __restrict__ score, const int* __restrict__ cls,
                                            const float4* __restrict__ box, const uint8_t* __restrict__ cand, int A,
                                            float iou_thresh, int max_cand, uint8_t* __restrict__ keep,
                                            int* __restrict__ keep_count, int ablate) {
    __shared__ unsigned long long s_key[CAP];
    __shared__ float4 s_box[CAP];
    __shared__ unsigned char s_alive[CAP];
    __shared__ int s_seg[CAP + 1];
    __shared__ int s_hist[256];
    __shared__ int s_wave[4];
    __shared__ int s_misc[4];

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t off = (size_t)b * A;
    const float* sc = score + off;
    const uint8_t* cd = cand + off;

    // clear the output row, count the candidates and -- optimistically -- compact them in the same sweep: when all of them
    // take part (total <= max_cand, the usual case) the sort below orders them completely (the anchor index is part of the
    // key), so their order in s_key is irrelevant: one LDS counter, no barrier between the anchor strides, every load of
    // the sweep in flight at once.  (The ordered compaction further down -- needed only for the exact top-max_cand cut --
    // costs two workgroup prefix sums and a global-memory round trip per stride of 256 anchors, 34 strides per image.)
    if (tid < 4) s_misc[tid] = 0;
    __syncthreads();
    auto take_cand = [&](int a) {
        const int slot = atomicAdd(&s_misc[0], 1);
        if (slot < CAP)
            s_key[slot] = ((unsigned long long)(unsigned)cls[off + a] << 48) |
                          ((unsigned long long)(0xffffffffu - __float_as_uint(sc[a])) << 16) | (unsigned long long)(unsigned)a;
    };
    if ((A & 3) == 0) {                      // rows of cand / keep are 4-byte aligned: four anchors per load / store
        const unsigned* cd4 = reinterpret_cast<const unsigned*>(cd);
        unsigned* kp4 = reinterpret_cast<unsigned*>(keep + off);
        for (int q = tid; q < (A >> 2); q += WG) {
            kp4[q] = 0u;
            const unsigned c4 = cd4[q];
            if (c4) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if ((c4 >> (8 * e)) & 0xffu) take_cand(4 * q + e);
            }
        }
    } else {
        for (int a = tid; a < A; a += WG) {
            keep[off + a] = 0;
            if (cd[a]) take_cand(a);
        }
    }
    __syncthreads();
    const int total = s_misc[0];
    if (total == 0) {
        if (keep_count && tid == 0) keep_count[b] = 0;
        return;
    }

    // exact cut to the max_cand best (score desc, anchor asc): radix select on the score bits
    unsigned cut_bits = 0;                   // take score > cut, and the first `need_eq` with score == cut
    int need_eq = INT_MAX;
    const int take = min(total, max_cand);
    if (total > max_cand) {
        unsigned prefix = 0;
        int k = max_cand;                    // rank (1-based, from the top) still to locate
        for (int shift = 24; shift >= 0; shift -= 8) {
            s_hist[tid] = 0;
            __syncthreads();
            for (int a = tid; a < A; a += WG)
                if (cd[a]) {
                    const unsigned key = __float_as_uint(sc[a]);
                    if (shift == 24 || (key >> (shift + 8)) == prefix) atomicAdd(&s_hist[(key >> shift) & 255u], 1);
                }
            __syncthreads();
            if (tid == 0) {
                int run = 0, d = 255;
                for (; d > 0; --d) {
                    if (run + s_hist[d] >= k) break;
                    run += s_hist[d];
                }
                s_misc[1] = d;
                s_misc[2] = k - run;
            }
            __syncthreads();
            prefix = (prefix << 8) | (unsigned)s_misc[1];
            k = s_misc[2];
            __syncthreads();
        }
        cut_bits = prefix;
        need_eq = k;                         // how many of the keys equal to the cut are still wanted
    }

    if (total > max_cand) {
    // ordered compaction (anchor order) of the participating candidates
    int filled = 0, eq_seen = 0;
    for (int a0 = 0; a0 < A; a0 += WG) {
        const int a = a0 + tid;
        bool is_c = a < A && cd[a];
        unsigned key = is_c ? __float_as_uint(sc[a]) : 0u;
        bool is_eq = false;
        if (total > max_cand && is_c) {
            is_eq = key == cut_bits;
            is_c = key > cut_bits || is_eq;
        }
        int tot_eq = 0;
        const int eq_rank = wg_prefix(is_eq, s_wave, tot_eq);
        if (is_eq && eq_seen + eq_rank >= need_eq) is_c = false;
        eq_seen += tot_eq;
        int tot = 0;
        const int pos = wg_prefix(is_c, s_wave, tot);
        if (is_c) {
            const int slot = filled + pos;
            s_key[slot] = ((unsigned long long)(unsigned)cls[off + a] << 48) |
                          ((unsigned long long)(0xffffffffu - key) << 16) | (unsigned long long)(unsigned)a;
        }
        filled += tot;
    }
    }
    if (ablate & 1) return;
    // sort ascending: class asc, score desc, anchor asc.  Up to 512 keys by rank (keys are unique: every thread counts the keys
    // below its own with broadcast LDS reads and writes its key to that slot -- two barriers instead of the 45 of a 512-key
    // bitonic network); more keys: bitonic, padded to a power of two
    if (take <= 512) {
        __syncthreads();
        unsigned long long mine[2];
        int rank[2] = {0, 0};
#pragma unroll
        for (int e = 0; e < 2; ++e) mine[e] = tid + e * WG < take ? s_key[tid + e * WG] : ~0ull;
        for (int j = 0; j < take; ++j) {
            const unsigned long long kj = s_key[j];
#pragma unroll
            for (int e = 0; e < 2; ++e) rank[e] += kj < mine[e] ? 1 : 0;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; ++e)
            if (tid + e * WG < take) s_key[rank[e]] = mine[e];
        __syncthreads();
    } else {
    int npow = 1;
    while (npow < take) npow <<= 1;
    for (int i = take + tid; i < npow; i += WG) s_key[i] = ~0ull;
    __syncthreads();
    for (int k = 2; k <= npow; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npow; i += WG) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned long long x = s_key[i], y = s_key[p];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { s_key[i] = y; s_key[p] = x; }
                }
            }
            __syncthreads();
        }
    }
    if (ablate & 2) return;
    // gather boxes, mark class-segment starts
    for (int i = tid; i < take; i += WG) {
        const int a = (int)(s_key[i] & 0xffffull);
        s_box[i] = box[off + a];
        s_alive[i] = 1;
    }
    if (tid == 0) s_misc[3] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < take; i0 += WG) {
        const int i = i0 + tid;
        const bool start = i < take && (i == 0 || (s_key[i] >> 48) != (s_key[i - 1] >> 48));
        int tot = 0;
        const int pos = wg_prefix(start, s_wave, tot);
        if (start) s_seg[s_misc[3] + pos] = i;
        __syncthreads();
        if (tid == 0) s_misc[3] += tot;
        __syncthreads();
    }
    const int nseg = s_misc[3];
    if (tid == 0) s_seg[nseg] = take;
    __syncthreads();

    if (ablate & 4) return;
    // greedy pass: one wave per class segment
    volatile unsigned char* alive = s_alive;
    int kept = 0;
    for (int sg = wave; sg < nseg; sg += 4) {
        const int lo = s_seg[sg], hi = s_seg[sg + 1];
        if (hi - lo <= 64) {
            // a segment that fits a wave (nearly all do) runs in registers: lane l holds box lo + l, the alive set is a
            // 64-bit wave-uniform mask, box i reaches the lanes by a broadcast -- no LDS round trip per kept box (the LDS
            // form below spent ~1 us per box on its dependent reads: 35 of this kernel's 64 us)
            const int nb = hi - lo;
            const float4 bj = lane < nb ? s_box[lo + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned long long am = nb == 64 ? ~0ull : ((1ull << nb) - 1ull);
            for (int i = 0; i < nb; ++i) {
                if (!((am >> i) & 1ull)) continue;           // wave-uniform
                const float4 bi = make_float4(__shfl(bj.x, i), __shfl(bj.y, i), __shfl(bj.z, i), __shfl(bj.w, i));
                const bool sup = lane > i && lane < nb && iou_f32(bi, bj) > iou_thresh;
                am &= ~__ballot(sup);
            }
            if (lane < nb && ((am >> lane) & 1ull)) keep[off + (int)(s_key[lo + lane] & 0xffffull)] = 1;
            if (lane == 0) kept += __popcll(am);
            continue;
        }
        for (int i = lo; i < hi; ++i) {
            if (!alive[i]) continue;             // wave-uniform
            if (lane == 0) {
                keep[off + (int)(s_key[i] & 0xffffull)] = 1;
                ++kept;
            }
            const float4 bi = s_box[i];
            for (int j = i + 1 + lane; j < hi; j += 64)
                if (alive[j] && iou_f32(bi, s_box[j]) > iou_thresh) alive[j] = 0;
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (keep_count) {
        __syncthreads();
        if (tid == 0) s_misc[0] = 0;
        __syncthreads();
        if (kept) atomicAdd(&s_misc[0], kept);
        __syncthreads();
        if (tid == 0) keep_count[b] = s_misc[0];
    }
}

}  // namespace

// timing-only ablations of k_nms (they skip stages, i.e. change results): development builds only
static int nms_ablate() {
#ifdef SSD_DEV_ABLATE
    return ssd_knob("SSD_ABLATE", 0);
#else
    return 0;
#endif
}

extern "C" {

int ssd_score_decode(const void* conf, const void* loc, int dtype, const double* priors, int B, int A, int C,
                     float score_thresh, double in_size, float* score, int32_t* cls, float* box, uint8_t* cand,
                     void* stream) {
    if (B <= 0 || A <= 0 || C < 2) return SSD_ERR_VALUE;
    if (!conf || !loc || !priors || !score || !cls || !box || !cand) return SSD_ERR_VALUE;
    if (dtype != SSD_F32 && dtype != SSD_BF16) return SSD_ERR_VALUE;
    const size_t lds = ((size_t)ROWS * C * sizeof(float) + 15) / 16 * 16;
    if (lds > 150 * 1024) return SSD_ERR_UNSUPPORTED;
    const size_t n = (size_t)B * A;
    const size_t nblk = (n + ROWS - 1) / ROWS;
    const unsigned grid = (unsigned)(nblk < 768 ? nblk : 768);
    hipStream_t s = (hipStream_t)stream;
#define SSD_LAUNCH_SCORE(T_, CC_)                                                                                     \
    hipLaunchKernelGGL((k_score_decode<T_, CC_>), dim3(grid), dim3(WG), lds, s, (const T_*)conf, (const T_*)loc, priors, n, A, C, \
                       score_thresh, in_size, score, cls, reinterpret_cast<float4*>(box), cand)
    if (dtype == SSD_F32) { if (C == 81) SSD_LAUNCH_SCORE(float, 81); else SSD_LAUNCH_SCORE(float, 0); }
    else { if (C == 81) SSD_LAUNCH_SCORE(__hip_bfloat16, 81); else SSD_LAUNCH_SCORE(__hip_bfloat16, 0); }
#undef SSD_LAUNCH_SCORE
    return ssd_launch_status();
}

int ssd_nms_max_candidates(void) { return CAP; }

int ssd_nms(const float* score, const int32_t* cls, const float* box, const uint8_t* cand, int B, int A,
            float iou_thresh, int max_cand, uint8_t* keep, int32_t* keep_count, void* stream) {
    if (B <= 0 || A <= 0 || A > 65536 || max_cand <= 0 || max_cand > CAP) return SSD_ERR_VALUE;
    if (!score || !cls || !box || !cand || !keep) return SSD_ERR_VALUE;
    hipLaunchKernelGGL(k_nms, dim3(B), dim3(WG), 0, (hipStream_t)stream, score, cls, reinterpret_cast<const float4*>(box),
                       cand, A, iou_thresh, max_cand, keep, keep_count, nms_ablate());
    return ssd_launch_status();
}

}  // extern "C"
