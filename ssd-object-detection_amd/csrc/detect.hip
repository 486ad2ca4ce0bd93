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
template <int NW>
__device__ __forceinline__ int wg_prefix(bool flag, int* s_wave, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_wave[w];
    total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) total += s_wave[w];
    return base + before;
}

// One workgroup of 16 waves per image: the greedy pass runs one wave per class segment, ~80 short segments per image, and with
// four waves it was 21 of the kernel's 39 us (stage times from a -DSSD_DEV_ABLATE build, tools_dev/time_detect.py).
constexpr int NMS_WG = 1024;
__global__ __launch_bounds__(NMS_WG) void k_nms(const float* __restrict__ score, const int* __restrict__ cls,
                                            const float4* __restrict__ box, const uint8_t* __restrict__ cand, int A,
                                            float iou_thresh, int max_cand, uint8_t* __restrict__ keep,
                                            int* __restrict__ keep_count, int ablate) {
    __shared__ unsigned long long s_key[CAP];
    __shared__ float4 s_box[CAP];
    __shared__ unsigned char s_alive[CAP];
    __shared__ int s_seg[CAP + 1];
    __shared__ int s_hist[256];
    __shared__ int s_wave[NMS_WG / 64];
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
    // Two memory round trips for the whole sweep: (1) every thread's candidate flags, all requests in flight together (the
    // per-stride form -- load a flag word, branch, load class and score of its candidates, next stride -- ran ~17 dependent
    // round trips: most of this kernel's time); the candidates' anchor indices go to an LDS list; (2) further down every thread
    // fetches class, score AND box of "its" list entries at once (the boxes used to be a third trip behind the sort).
    if (tid < 4) s_misc[tid] = 0;
    __syncthreads();
    int* s_idx = s_seg;                      // [CAP] candidate anchors, unordered (s_seg is not in use yet)
    auto note_cand = [&](int a) {
        const int slot = atomicAdd(&s_misc[0], 1);
        if (slot < CAP) s_idx[slot] = a;
    };
    if ((A & 3) == 0) {                      // rows of cand / keep are 4-byte aligned: four anchors per load / store
        const unsigned* cd4 = reinterpret_cast<const unsigned*>(cd);
        unsigned* kp4 = reinterpret_cast<unsigned*>(keep + off);
        const int nq = A >> 2;
        constexpr int UN = 3;                // 3 x 1024 x 4 = 12288 >= 8732 anchors in one pass
        for (int q0 = 0; q0 < nq; q0 += UN * NMS_WG) {
            unsigned c4v[UN];
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                const int q = q0 + j * NMS_WG + tid;
                const unsigned v = cd4[q < nq ? q : nq - 1];
                c4v[j] = q < nq ? v : 0u;
            }
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                const int q = q0 + j * NMS_WG + tid;
                if (q < nq) kp4[q] = 0u;
            }
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                if (!c4v[j]) continue;
                const int q = q0 + j * NMS_WG + tid;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if ((c4v[j] >> (8 * e)) & 0xffu) note_cand(4 * q + e);
            }
        }
    } else {
        for (int a = tid; a < A; a += NMS_WG) {
            keep[off + a] = 0;
            if (cd[a]) note_cand(a);
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
            if (tid < 256) s_hist[tid] = 0;
            __syncthreads();
            for (int a = tid; a < A; a += NMS_WG)
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

    // keys (+ boxes) of the listed candidates: one round trip, every load of a thread in flight together
    constexpr int PER = CAP / NMS_WG;
    float4 mybox[PER];
    {
        // (unconditional loads at clamped indices: a conditionally filled register array would live in scratch memory)
        int an[PER];
        unsigned cl[PER], sb[PER];
        const int listed = min(total, CAP);
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int i = tid + e * NMS_WG;
            const int a = s_idx[i < listed ? i : 0];
            an[e] = min(max(a, 0), A - 1);
        }
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            cl[e] = (unsigned)cls[off + an[e]];
            sb[e] = __float_as_uint(sc[an[e]]);
            mybox[e] = box[off + an[e]];
        }
        __syncthreads();                     // s_idx (= s_seg) is dead from here on
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int i = tid + e * NMS_WG;
            if (total <= max_cand && i < total)
                s_key[i] = ((unsigned long long)cl[e] << 48) | ((unsigned long long)(0xffffffffu - sb[e]) << 16) |
                           (unsigned long long)(unsigned)an[e];
        }
    }
    if (total > max_cand) {
    // ordered compaction (anchor order) of the participating candidates
    int filled = 0, eq_seen = 0;
    for (int a0 = 0; a0 < A; a0 += NMS_WG) {
        const int a = a0 + tid;
        bool is_c = a < A && cd[a];
        unsigned key = is_c ? __float_as_uint(sc[a]) : 0u;
        bool is_eq = false;
        if (total > max_cand && is_c) {
            is_eq = key == cut_bits;
            is_c = key > cut_bits || is_eq;
        }
        int tot_eq = 0;
        const int eq_rank = wg_prefix<NMS_WG / 64>(is_eq, s_wave, tot_eq);
        if (is_eq && eq_seen + eq_rank >= need_eq) is_c = false;
        eq_seen += tot_eq;
        int tot = 0;
        const int pos = wg_prefix<NMS_WG / 64>(is_c, s_wave, tot);
        if (is_c) {
            const int slot = filled + pos;
            s_key[slot] = ((unsigned long long)(unsigned)cls[off + a] << 48) |
                          ((unsigned long long)(0xffffffffu - key) << 16) | (unsigned long long)(unsigned)a;
        }
        filled += tot;
    }
    }
    if (ablate & 1) return;
    // sort ascending: class asc, score desc, anchor asc -- by rank (keys are unique: every thread counts the keys below its own
    // with broadcast LDS reads and writes its key to that slot: two barriers instead of the 45+ of a bitonic network)
    static_assert(CAP <= NMS_WG, "one key per thread");
    {
        __syncthreads();
        const unsigned long long mine = tid < take ? s_key[tid] : ~0ull;
        int rank = 0;
        for (int j = 0; j < take; ++j) rank += s_key[j] < mine ? 1 : 0;
        __syncthreads();
        if (tid < take) {
            s_key[rank] = mine;
            // the box travels with its key: no gather behind the sort (component-wise: see stage_store)
            if (total <= max_cand) s_box[rank] = make_float4(mybox[0].x, mybox[0].y, mybox[0].z, mybox[0].w);
        }
        __syncthreads();
    }
    if (ablate & 2) return;
    // gather boxes (unless they came with the keys), mark class-segment starts
    const bool boxes_placed = total <= max_cand;
    for (int i = tid; i < take; i += NMS_WG) {
        if (!boxes_placed) s_box[i] = box[off + (int)(s_key[i] & 0xffffull)];
        s_alive[i] = 1;
    }
    if (tid == 0) s_misc[3] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < take; i0 += NMS_WG) {
        const int i = i0 + tid;
        const bool start = i < take && (i == 0 || (s_key[i] >> 48) != (s_key[i - 1] >> 48));
        int tot = 0;
        const int pos = wg_prefix<NMS_WG / 64>(start, s_wave, tot);
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
    for (int sg = wave; sg < nseg; sg += NMS_WG / 64) {
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
    hipLaunchKernelGGL(k_nms, dim3(B), dim3(NMS_WG), 0, (hipStream_t)stream, score, cls, reinterpret_cast<const float4*>(box),
                       cand, A, iou_thresh, max_cand, keep, keep_count, nms_ablate());
    return ssd_launch_status();
}

}  // extern "C"
