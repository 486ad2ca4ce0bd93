// Batched anchor matching + box encoding for gfx950 (MI355X).
//
// Replaces utils/bbox.py:44-101 of the reference (match_bbox + apply_anchor_box, driven per image
// from models/ssd_model.py:212-213) for a whole batch in two launches:
//
//   k_match_pairs   grid (chunks, B): one thread per prior column, all of the image's gt rows.
//                   Streams priors in (f64, L2-resident), writes cls/loc/mask out (the HBM
//                   traffic), computes the mixed f32/f64 IoU arithmetic of iou_n
//                   (utils/bbox.py:28-41) bit-exactly but WITHOUT a division for almost every
//                   pair: a pair is only divided out when inter >= bound*union, where bound is
//                   a proven lower bound of what could still matter (column: current column
//                   best, starting at thresh; row: an exactly evaluated seed of the row
//                   maximum).  Emits phase-2 assignments (column max > thresh -> lowest row
//                   attaining it) and, per (row, chunk), the best column candidate.
//   k_match_phase1  grid (B): reduces the row candidates, runs the sequential phase 1
//                   (utils/bbox.py:62-68).  If all row-best columns are distinct the n_t rounds
//                   collapse to "every row takes its best column"; otherwise the literal
//                   round-by-round elimination is run with exact re-scans.  Patches the <= n_t
//                   phase-1 columns in the outputs.
//
// Compile with -ffp-contract=off: results must match numpy's unfused IEEE arithmetic bit for bit.
#include "common.h"
#include <limits.h>

namespace {

constexpr int WG = 256;
constexpr int NWAVE = WG / SSD_WAVE;
#define SSD_MARGIN (1.0 - 0x1p-50)           // filter slack: see DESIGN.md "division-free pruning"

struct Corner {                              // (x_lo, y_lo, x_hi, y_hi, area) of one box, as f64
    double lx, ly, hx, hy, a;
};

// gt side of iou_n: corners and area evaluated in float32, then widened (numpy promotion).
__device__ __forceinline__ Corner gt_corner(float4 g) {
    const float hw = g.z / 2.0f, hh = g.w / 2.0f;
    Corner c;
    c.lx = (double)(g.x - hw);
    c.ly = (double)(g.y - hh);
    c.hx = (double)(g.x + hw);
    c.hy = (double)(g.y + hh);
    c.a = (double)(g.z * g.w);
    return c;
}

// prior side of iou_n: float64 throughout.
__device__ __forceinline__ Corner prior_corner(double cx, double cy, double w, double h) {
    Corner c;
    c.lx = cx - w / 2.0;
    c.ly = cy - h / 2.0;
    c.hx = cx + w / 2.0;
    c.hy = cy + h / 2.0;
    c.a = w * h;
    return c;
}

// intersection (sides clamped at 1e-10) and union (+1e-10) exactly as utils/bbox.py:34-41.
__device__ __forceinline__ void inter_union(const Corner& g, const Corner& p, double& inter, double& uni) {
    const double x_lo = fmax(g.lx, p.lx);
    const double y_lo = fmax(g.ly, p.ly);
    const double x_hi = fmin(g.hx, p.hx);
    const double y_hi = fmin(g.hy, p.hy);
    inter = fmax(1e-10, x_hi - x_lo) * fmax(1e-10, y_hi - y_lo);
    uni = g.a + p.a - inter + 1e-10;
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// (q, c) ordering used everywhere for "first maximum in row-major order": larger q wins, ties go
// to the smaller index.
__device__ __forceinline__ bool better(double q, int c, double bq, int bc) {
    return q > bq || (q == bq && c < bc);
}

// apply_anchor_box for one matched row (utils/bbox.py:98-99), cast to f32 (ssd_model.py:222).
__device__ __forceinline__ float4 encode_row(float4 g, double pcx, double pcy, double pw, double ph) {
    float4 o;
    o.x = (float)(((double)g.x - pcx) / pw);
    o.y = (float)(((double)g.y - pcy) / ph);
    o.z = (float)log((double)fmaxf(g.z, 1e-5f) / fmax(pw, 1e-5));
    o.w = (float)log((double)fmaxf(g.w, 1e-5f) / fmax(ph, 1e-5));
    return o;
}

struct GridHint {
    int levels;
    int gh[SSD_MAX_LEVELS], gw[SSD_MAX_LEVELS], k[SSD_MAX_LEVELS];
    int col_off[SSD_MAX_LEVELS + 1];        // first column of each level
    int cand_off[SSD_MAX_LEVELS + 1];       // prefix sum of k
};

// One record per gt row, produced by k_match_rows and read through the scalar cache by
// k_match_pairs: float32-evaluated corners/area widened to f64, and the row pruning bound.
struct __attribute__((aligned(16))) RowRec {
    double lx, ly, hx, hy, a, lbm;
};

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
    const int lo = __shfl_xor(__double2loint(v), mask);
    const int hi = __shfl_xor(__double2hiint(v), mask);
    return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------------------------------------
// K0: per gt row (32 lanes each): corner record + an exact seed of the row maximum taken from the
// priors of the cell under the gt centre on every level (geometry hint).  The seed L is the IoU
// of a real column of this row, hence L <= row maximum; lbm = L*(1-2^-50) is the pruning bound.
constexpr int LIST_CAP = 48;                 // row candidates kept per gt row
struct __attribute__((aligned(16))) Cand { double q; int c; int pad; };

// float32 copy of a row record for the pre-filter: corners and area (exact: they ARE float32 values) and the row bound.
struct __attribute__((aligned(16))) RowF32 { float lx, ly, hx, hy, a, lb, pad1, pad2; };

// f32 pre-filter.  With g the gt row (its corners and area ARE float32 values, utils/bbox.py evaluates them in float32) and p
// the prior (float64), the exact test is  inter >= bound * uni,  inter = max(1e-10, w) * max(1e-10, h),
// uni = ga + pa - inter + 1e-10.  Rounding p's corners to float32 moves each by <= 2^-24 * 1.6 < 1e-7, so the float32
// side w32 = fl(min(ghx, phx32) - max(glx, plx32)) is within 2.5e-7 of w: W = max(w32, 0) + 4e-7 >= max(1e-10, w), likewise
// H, and iub = fl(W * H) * (1 + 2^-23) >= inter.  With pa32 = fl32(pa): ulb = fl(fl(ga + pa32) - iub) - 5e-7 <= uni (two
// roundings of values <= 3 and the rounding of pa: < 4e-7 in all).  b32 = fl32(bound) is within 2^-24 relative.  Hence
//     iub * 1.00001f < b32 * ulb   ==>   inter < bound * uni
// (the 1e-5 relative slack covers the remaining float32 roundings, ~3e-7 relative, with room to spare; ulb <= 0 never
// rejects because iub > 0).  A rejected pair is one the exact filter rejects too; everything else goes through the exact
// arithmetic unchanged.
__device__ __forceinline__ bool prefilter_rejects(const RowF32& g, float plx, float ply, float phx, float phy, float pa,
                                                  float b32) {
    const float w = fminf(g.hx, phx) - fmaxf(g.lx, plx);
    const float h = fminf(g.hy, phy) - fmaxf(g.ly, ply);
    const float iub = (fmaxf(w, 0.f) + 4e-7f) * (fmaxf(h, 0.f) + 4e-7f);
    const float ulb = ((g.a + pa) - iub) - 5e-7f;
    return iub * 1.00001f < b32 * ulb;
}

// Workspace of the three-launch path: one RowSlot per gt row of the batch (pure scratch: every word is written before
// it is read in the same call).
struct __attribute__((aligned(16))) RowSlot {
    int cnt;                                 // entries appended to `list` (> LIST_CAP: overflowed / no list)
    int pad[3];
    Cand list[LIST_CAP];                     // scratch: unordered row candidates
    RowRec rec;                              // the row record k_match_rows leaves for k_match_pairs,
    RowF32 recf;                             //   and its float32 copy
    double rq;                               // scratch (three-launch path, n_t > P1_LDS_ROWS): phase-1 row state
    int rc, rs;
};
static_assert(sizeof(RowSlot) == 16 + LIST_CAP * 16 + 48 + 32 + 16, "RowSlot layout");

__global__ __launch_bounds__(WG) void k_match_rows(const float4* __restrict__ gt_box, int total_gt,
                                                   const double* __restrict__ priors, int A, GridHint hint,
                                                   RowSlot* __restrict__ slots) {
    const int gid = blockIdx.x * WG + threadIdx.x;
    const int row = gid >> 5, sub = gid & 31;
    const bool live = row < total_gt;
    const float4 g = gt_box[live ? row : 0];
    const Corner gc = gt_corner(g);
    double best = 0.0;
    if (hint.levels > 0) {
        const int ncand = hint.cand_off[hint.levels];
        for (int k = sub; k < ncand; k += 32) {
            int l = 0;
            while (l + 1 < hint.levels && k >= hint.cand_off[l + 1]) ++l;
            int x = (int)floorf(g.x * (float)hint.gw[l]);
            int y = (int)floorf(g.y * (float)hint.gh[l]);
            x = min(max(x, 0), hint.gw[l] - 1);
            y = min(max(y, 0), hint.gh[l] - 1);
            int c = hint.col_off[l] + (y * hint.gw[l] + x) * hint.k[l] + (k - hint.cand_off[l]);
            c = min(max(c, 0), A - 1);
            const double2 lo = *reinterpret_cast<const double2*>(priors + 4 * (size_t)c);
            const double2 hi = *reinterpret_cast<const double2*>(priors + 4 * (size_t)c + 2);
            double inter, uni;
            inter_union(gc, prior_corner(lo.x, lo.y, hi.x, hi.y), inter, uni);
            const double q = inter / uni;
            if (q > best) best = q;
        }
    }
#pragma unroll
    for (int m = 16; m > 0; m >>= 1) {
        const double o = shfl_xor_f64(best, m);
        if (o > best) best = o;
    }
    if (live && sub == 0) {
        RowRec r;
        r.lx = gc.lx; r.ly = gc.ly; r.hx = gc.hx; r.hy = gc.hy; r.a = gc.a;
        // Row candidates are collected down to 0.8 x the seed (typically 5-15 columns, never more than ~50): the list then usually holds the runners-up that a row
        // needs when it loses its best column in phase 1.  No seed (no geometry hint): no list, the row is re-scanned.
        r.lbm = best * 0.8 * SSD_MARGIN;
        slots[row].rec = r;
        RowF32 f;
        f.lx = (float)gc.lx; f.ly = (float)gc.ly; f.hx = (float)gc.hx; f.hy = (float)gc.hy; f.a = (float)gc.a;   // exact
        f.lb = (float)r.lbm; f.pad1 = f.pad2 = 0.f;
        slots[row].recf = f;
        slots[row].cnt = best > 0.0 ? 0 : LIST_CAP + 1;
    }
}

// ------------------------------------------------------------------------------------------------
// K1: the streaming kernel.  A workgroup = WG prior columns of one image, numbered so that eight consecutive workgroups
// belong to eight different images (a CU's resident workgroups mix heavy and light images); thread = one prior column,
// looping over the image's gt rows, four at a time, whose records arrive through the scalar cache (uniform address).  Per
// pair the float32 pre-filter first (prefilter_rejects); only a wave with a surviving lane runs the float64 arithmetic: no
// division unless inter >= bound*union (bound = min(column bound, row bound)).  Exactly evaluated pairs that reach the row
// bound are appended to the row's candidate list (a handful per row).
constexpr int PAIRS_GROUP = 8;
__global__ __launch_bounds__(WG) void k_match_pairs(
    const float4* __restrict__ gt_box, const float* __restrict__ gt_cls, const int* __restrict__ gt_off,
    RowSlot* __restrict__ slots, const double* __restrict__ priors, const float4* __restrict__ enc_zero,
    int A, int B, int nchunk, double thresh, int* __restrict__ out_cls, float4* __restrict__ out_loc,
    uint8_t* __restrict__ out_mask, int* __restrict__ out_owner) {
    const int per_group = PAIRS_GROUP * nchunk;
    const int grp = blockIdx.x / per_group, within = blockIdx.x - grp * per_group;
    const int b = grp * PAIRS_GROUP + (within & (PAIRS_GROUP - 1)), chunk = within / PAIRS_GROUP;
    if (b >= B) return;
    const int tid = threadIdx.x;
    const int g0 = gt_off[b];
    const int nt = gt_off[b + 1] - g0;
    const int c = chunk * WG + tid;
    const bool valid = c < A;
    const int cc = valid ? c : A - 1;

    const double2 plo = *reinterpret_cast<const double2*>(priors + 4 * (size_t)cc);
    const double2 phi = *reinterpret_cast<const double2*>(priors + 4 * (size_t)cc + 2);
    const float4 ez = enc_zero[cc];
    Corner pc = prior_corner(plo.x, plo.y, phi.x, phi.y);
    const float plx = (float)pc.lx, ply = (float)pc.ly, phx = (float)pc.hx, phy = (float)pc.hy, pa = (float)pc.a;
    double cbq = thresh;                       // phase 2 needs max > thresh (utils/bbox.py:73)
    double cbm = thresh * SSD_MARGIN;
    float cb32 = (float)cbm;
    int cbr = -1;

    RowF32 nxt[4];                             // uniform -> scalar loads, four in flight, one batch ahead
#pragma unroll
    for (int i = 0; i < 4; ++i) nxt[i] = slots[g0 + min(i, max(nt - 1, 0))].recf;
    for (int r0 = 0; r0 < nt; r0 += 4) {
        RowF32 f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[i] = nxt[i]; nxt[i] = slots[g0 + min(r0 + 4 + i, nt - 1)].recf; }
        unsigned maybe = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (valid && r0 + i < nt && !prefilter_rejects(f[i], plx, ply, phx, phy, pa, fminf(cb32, f[i].lb))) maybe |= 1u << i;
        if (__ballot(maybe != 0)) {            // rare: some lane's pair survives the pre-filter
            for (int i = 0; i < 4 && r0 + i < nt; ++i) {       // rows in order: the column bound only grows
                const int r = r0 + i;
                const RowRec g = slots[g0 + r].rec;
                Corner gc;
                gc.lx = g.lx; gc.ly = g.ly; gc.hx = g.hx; gc.hy = g.hy; gc.a = g.a;
                double inter, uni;
                inter_union(gc, pc, inter, uni);
                if (((maybe >> i) & 1u) && inter >= fmin(cbm, g.lbm) * uni) {      // the exact filter, then the division
                    const double q = inter / uni;
                    if (q > cbq) { cbq = q; cbr = r; cbm = q * SSD_MARGIN; cb32 = (float)cbm; }
                    if (g.lbm > 0.0 && q >= g.lbm) {
                        const int pos = atomicAdd(&slots[g0 + r].cnt, 1);
                        if (pos < LIST_CAP) {
                            Cand e; e.q = q; e.c = c; e.pad = 0;
                            slots[g0 + r].list[pos] = e;
                        }
                    }
                }
            }
        }
    }

    // phase-2 outputs for this thread's column (phase-1 columns are patched by k_match_phase1)
    if (valid) {
        const size_t o = (size_t)b * A + c;
        if (out_owner) out_owner[o] = cbr;
        if (cbr >= 0) {
            const float4 g = gt_box[g0 + cbr];
            out_cls[o] = (int)gt_cls[g0 + cbr];
            out_mask[o] = 1;
            out_loc[o] = encode_row(g, plo.x, plo.y, phi.x, phi.y);
        } else {
            out_cls[o] = 0;
            out_mask[o] = 0;
            out_loc[o] = ez;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Workgroup-wide argmax of (q, idx) with the `better` ordering. Result broadcast to all threads.
__device__ __forceinline__ void wg_argmax(double& q, int& c, double* s_q, int* s_c) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double oq = shfl_xor_f64(q, off);
        const int oc = __shfl_xor(c, off);
        if (better(oq, oc, q, c)) { q = oq; c = oc; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { s_q[wave] = q; s_c[wave] = c; }
    __syncthreads();
    q = s_q[0]; c = s_c[0];
#pragma unroll
    for (int w = 1; w < NWAVE; ++w)
        if (better(s_q[w], s_c[w], q, c)) { q = s_q[w]; c = s_c[w]; }
}

// Exact maximum of one gt row over all columns whose bit in `taken` is clear; first maximum wins.
__device__ __forceinline__ void row_scan(const Corner& g, const double* __restrict__ priors, int A,
                                         const unsigned* taken, double& q_out, int& c_out,
                                         double* s_q, int* s_c) {
    double bq = -1.0, bqm = -1.0;
    int bc = INT_MAX;
    for (int c = threadIdx.x; c < A; c += WG) {
        if (taken[c >> 5] & (1u << (c & 31))) continue;
        const double2 lo = *reinterpret_cast<const double2*>(priors + 4 * (size_t)c);
        const double2 hi = *reinterpret_cast<const double2*>(priors + 4 * (size_t)c + 2);
        double inter, uni;
        inter_union(g, prior_corner(lo.x, lo.y, hi.x, hi.y), inter, uni);
        if (inter >= bqm * uni) {
            const double q = inter / uni;
            if (q > bq) { bq = q; bc = c; bqm = q * SSD_MARGIN; }
        }
    }
    wg_argmax(bq, bc, s_q, s_c);
    q_out = bq;
    c_out = bc;
}

constexpr int P1_LDS_ROWS = 512;               // row state lives in LDS up to this many gt rows (three-launch path)

template <int ROWS>
struct Phase1Lds {                             // static LDS of the phase-1 pass
    double rq[ROWS];
    int rc[ROWS];
    int rs[ROWS];
    int res[ROWS];                             // rows to re-scan in this round
    double q[NWAVE];
    int c[NWAVE];
    int flag, nres;
};

// Sequential phase 1 (utils/bbox.py:62-68) of image b + patch of the phase-1 columns, by one workgroup.
// Literal semantics: n_t rounds, each taking the largest remaining IoU (ties: lowest row, then
// lowest column) and eliminating its row and column.  Executed as: while some still-free rows
// share their best column, let the best-priority such row (the pivot) and every free row ahead of
// it take their columns at once (none of those can be disturbed: nobody else wants their
// columns), re-scan the rows that wanted the pivot's column over the untaken columns, repeat;
// when no two free rows share a best column every row takes its own.
template <int ROWS>
__device__ __forceinline__ void phase1_image(
    int b, const float4* __restrict__ gt_box, const float* __restrict__ gt_cls, const int* __restrict__ gt_off,
    const double* __restrict__ priors, int A, RowSlot* slots, int* __restrict__ out_cls,
    float4* __restrict__ out_loc, uint8_t* __restrict__ out_mask, int* __restrict__ out_owner, unsigned* s_bits,
    Phase1Lds<ROWS>& S) {
    auto list_len = [&](int row) -> int { return slots[row].cnt; };
    auto list_at = [&](int row, int k) -> Cand { return slots[row].list[k]; };
    double* s_q = S.q;
    int* s_c = S.c;
    const int tid = threadIdx.x;
    const int g0 = gt_off[b];
    const int nt = gt_off[b + 1] - g0;
    if (nt == 0) return;
    const int nwords = (A + 31) >> 5;
    unsigned* taken = s_bits;
    unsigned* seen = s_bits + nwords;
    unsigned* dup = s_bits + 2 * nwords;
    const bool in_lds = nt <= ROWS;            // row state: LDS, or the rows' slots (many rows)
    auto RQ = [&](int r) -> double& { return in_lds ? S.rq[r] : slots[g0 + r].rq; };
    auto RC = [&](int r) -> int& { return in_lds ? S.rc[r] : slots[g0 + r].rc; };
    auto RS = [&](int r) -> int& { return in_lds ? S.rs[r] : slots[g0 + r].rs; };      // 1 = free, 0 = done, 2 = re-scan

    for (int i = tid; i < 3 * nwords; i += WG) s_bits[i] = 0u;
    if (tid == 0) { S.flag = 0; S.nres = 0; }
    __syncthreads();

    // best candidate of every row (lists are unordered: larger q wins, ties to the lower column); 8 lanes per row so
    // that a row's entries are fetched in one round trip
    for (int r0 = 0; r0 < nt; r0 += WG / 8) {
        const int r = r0 + (tid >> 3), sub = tid & 7;
        double bq = 0.0;
        int bc = INT_MAX;
        if (r < nt) {
            const int n = list_len(g0 + r);
            if (n <= LIST_CAP) {
                for (int k = sub; k < n; k += 8) {
                    const Cand e = list_at(g0 + r, k);
                    if (better(e.q, e.c, bq, bc)) { bq = e.q; bc = e.c; }
                }
            }
        }
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) {
            const double oq = shfl_xor_f64(bq, off);
            const int oc = __shfl_xor(bc, off);
            if (better(oq, oc, bq, bc)) { bq = oq; bc = oc; }
        }
        if (r < nt && sub == 0) {
            RQ(r) = bq;
            RC(r) = bc;
            RS(r) = 1;
            if (bc == INT_MAX) atomicOr(&S.flag, 2);
        }
    }
    __syncthreads();
    if (S.flag & 2) {                          // a row without candidate (never for valid boxes): exact scan
        for (int r = 0; r < nt; ++r) {
            if (RC(r) != INT_MAX) continue;
            double q; int c;
            row_scan(gt_corner(gt_box[g0 + r]), priors, A, taken, q, c, s_q, s_c);
            __syncthreads();
            if (tid == 0) { RQ(r) = q; RC(r) = c; }
            __syncthreads();
        }
    }

    for (int iter = 0; iter <= nt; ++iter) {
        // which free rows share their best column with another free row?
        for (int r = tid; r < nt; r += WG)
            if (RS(r)) {
                const int c = RC(r);
                const unsigned bit = 1u << (c & 31);
                if (atomicOr(&seen[c >> 5], bit) & bit) atomicOr(&dup[c >> 5], bit);
            }
        __syncthreads();
        double pq = -1.0;
        int pr = INT_MAX;
        for (int r = tid; r < nt; r += WG)
            if (RS(r)) {
                const int c = RC(r);
                if ((dup[c >> 5] >> (c & 31)) & 1u)
                    if (better(RQ(r), r, pq, pr)) { pq = RQ(r); pr = r; }
            }
        wg_argmax(pq, pr, s_q, s_c);
        if (pr == INT_MAX) break;              // no sharing left: every free row keeps its column
        const int cstar = RC(pr);
        __syncthreads();
        // pivot and every free row ahead of it take their columns; rows that wanted cstar re-scan
        for (int r = tid; r < nt; r += WG)
            if (RS(r)) {
                const int c = RC(r);
                if (r == pr || better(RQ(r), r, pq, pr)) {
                    RS(r) = 0;
                    atomicOr(&taken[c >> 5], 1u << (c & 31));
                } else if (c == cstar) {
                    RS(r) = 2;                 // needs a re-scan
                    if (in_lds) S.res[atomicAdd(&S.nres, 1)] = r;
                }
            }
        for (int i = tid; i < 2 * nwords; i += WG) seen[i] = 0u;   // seen and dup are adjacent
        __syncthreads();
        // the rows to re-scan: from the list built above (any order: re-scans do not change `taken`), or -- row state in
        // global memory -- by walking all rows
        const int nres = in_lds ? S.nres : nt;
        for (int k = 0; k < nres; ++k) {       // uniform; usually one row
            const int r = in_lds ? S.res[k] : k;
            if (RS(r) != 2) continue;
            // the row's list holds every column with IoU >= 0.8 x its seed: if any of them is still free, the best
            // free one is the row's new maximum (everything outside the list is smaller); else re-scan exactly
            double q = -1.0; int c = INT_MAX;
            const int n = list_len(g0 + r);
            if (n <= LIST_CAP && tid < n) {
                const Cand e = list_at(g0 + r, tid);
                if (!((taken[e.c >> 5] >> (e.c & 31)) & 1u)) { q = e.q; c = e.c; }
            }
            wg_argmax(q, c, s_q, s_c);
            if (c == INT_MAX) row_scan(gt_corner(gt_box[g0 + r]), priors, A, taken, q, c, s_q, s_c);
            __syncthreads();
            if (tid == 0) { RQ(r) = q; RC(r) = c; RS(r) = 1; }
            __syncthreads();
        }
        if (tid == 0) S.nres = 0;
        __syncthreads();
    }
    __syncthreads();
    // patch the phase-1 columns (utils/bbox.py:84-90 scatter + apply_anchor_box)
    for (int r = tid; r < nt; r += WG) {
        const int c = RC(r);
        const size_t o = (size_t)b * A + c;
        const float4 g = gt_box[g0 + r];
        const double2 lo = *reinterpret_cast<const double2*>(priors + 4 * (size_t)c);
        const double2 hi = *reinterpret_cast<const double2*>(priors + 4 * (size_t)c + 2);
        out_cls[o] = (int)gt_cls[g0 + r];
        out_mask[o] = 1;
        out_loc[o] = encode_row(g, lo.x, lo.y, hi.x, hi.y);
        if (out_owner) out_owner[o] = r;
    }
}

// K2: the phase-1 pass as a launch of its own (one workgroup per image): the three-launch path.
__global__ __launch_bounds__(WG) void k_match_phase1(
    const float4* __restrict__ gt_box, const float* __restrict__ gt_cls, const int* __restrict__ gt_off,
    const double* __restrict__ priors, int A, RowSlot* slots, int* __restrict__ out_cls,
    float4* __restrict__ out_loc, uint8_t* __restrict__ out_mask, int* __restrict__ out_owner) {
    extern __shared__ unsigned s_bits[];       // 3 A-bit maps: taken | seen | dup
    __shared__ Phase1Lds<P1_LDS_ROWS> L;
    phase1_image<P1_LDS_ROWS>(blockIdx.x, gt_box, gt_cls, gt_off, priors, A, slots, out_cls, out_loc, out_mask, out_owner,
                              s_bits, L);
}

// ------------------------------------------------------------------------------------------------
// The whole of ssd_match_encode in ONE launch, with no communication between workgroups (images with at most LOC_MAX_ROWS
// boxes and a VERIFIED geometry, ssd_prior_grid_verify).  A workgroup = WG prior columns of one image; every workgroup of
// an image works out that image's phase 1 for itself -- cheaply, because on a regular grid the columns that can matter to a
// gt row are few and can be enumerated from the geometry -- and then writes the FINAL value of each of its columns once,
// with plain stores.  Workgroups are numbered so that eight consecutive ones belong to eight different images (a CU's
// resident workgroups mix heavy and light images).
//
//   1. rows     gt boxes -> LDS (raw box + class, f64 record, f32 record for the pre-filter); the (w, h) of every anchor
//               type from the first cell of its level.
//   2. seed     L_r = best exact IoU of row r with the priors of the cell under its centre on every level (computed from
//               the model: centre (x + .5) / grid, size from the table -- the values the verified prior array holds);
//               L'_r = L_r (1 - 1e-9) <= row maximum.
//   3. window   with b = 0.8 L' (the chain bound): a column with IoU >= b needs  inter >= b uni >= T := b max(aP, aG)(1 - 1e-6)  while
//               inter <= (ow + 1e-10)(min(hP, hG) + 1e-10) with ow the overlap of the x extents: so ow >= m_w := T /
//               (min(hP, hG) + 2e-10) - 2e-10, i.e. the prior's centre lies in [g.lx + m_w - wP/2, g.hx - m_w + wP/2]; likewise
//               in y.  Per (row, anchor type) that is a rectangle of cells (typically 0-9 cells, ~25 per row in all),
//               queued in LDS one cell per entry.
//   4. evaluate every queued cell EXACTLY, one per thread and pass (the prior comes from the model: the same bits as the array; iou_n's arithmetic, division only past the
//               division-free filter); pairs with q >= 0.8 L' go to the row's candidate chain in LDS.  The chain holds
//               every column of the row with IoU >= 0.8 L' -- what the lists of the three-launch path hold.
//   5. phase 1  the literal order by pivot batching, as phase1_image, on the LDS chains; a row whose chain is exhausted
//               (or overflowed, or that has no seed) is re-scanned exactly over all columns.
//   6. stream   one thread = one column x all rows: f32 pre-filter, exact filter, division (k_match_pairs' loop); a
//               phase-1 column takes its row, any other column the phase-2 rule; one plain store per output.
// Nothing here depends on lists in memory, tickets, or another workgroup: no workspace, no state.
constexpr int LOC_MAX_ROWS = 64;
constexpr int LOC_MAX_TYPES = 48;              // anchor types = sum of per_cell over the levels (SSD300: 30)
constexpr int LOC_UNITS = 2048;              // window cells per round of LOC_ROUND rows (overflow: the row is scanned exactly)
constexpr int LOC_ROUND = 32;
constexpr int LOC_MAX_CELLS = 256;             // sum of the grid widths (heights) over the levels (SSD300: 76)
constexpr int LOC_CANDS = 512;
constexpr int LOC_GROUP = 8;

struct __attribute__((aligned(16))) LocCand { double q; int c; int next; };

struct LocalLds {
    RowRec rows[LOC_MAX_ROWS];                 // .lbm = 0.8 L' (1 - 2^-50): the chain bound; 0 = no seed
    RowF32 rowf[LOC_MAX_ROWS];
    float4 gt[LOC_MAX_ROWS];
    int cls[LOC_MAX_ROWS];
    double tw[LOC_MAX_TYPES], th[LOC_MAX_TYPES];           // anchor type t: size,
    int tgw[LOC_MAX_TYPES], tgh[LOC_MAX_TYPES];            //   grid of its level,
    int tcol[LOC_MAX_TYPES], tk[LOC_MAX_TYPES];            //   column of the type in cell (0, 0), priors per cell,
    int tcx[LOC_MAX_TYPES], tcy[LOC_MAX_TYPES];            //   where its level's cell centres start in ccx / ccy
    double ccx[LOC_MAX_CELLS], ccy[LOC_MAX_CELLS];         // cell centres (x + .5) / gw, (y + .5) / gh of every level: what k_priors stored
    unsigned units[LOC_UNITS];
    LocCand cand[LOC_CANDS];
    int head[LOC_MAX_ROWS];                    // chain head (-1: empty)
    int nolist[LOC_MAX_ROWS];                  // the chain is incomplete (overflow) or there is no seed
    double rq[LOC_MAX_ROWS];
    int rc[LOC_MAX_ROWS], rs[LOC_MAX_ROWS], res[LOC_MAX_ROWS];
    double q[NWAVE];
    int c[NWAVE];
    int nunits, ncand, nres;
};

template <int CPW>                             // columns per thread (a workgroup owns WG * CPW consecutive-by-chunk columns)
__global__ __launch_bounds__(WG) void k_match_local(
    const float4* __restrict__ gt_box, const float* __restrict__ gt_cls, const int* __restrict__ gt_off,
    const double* __restrict__ priors, const float4* __restrict__ enc_zero, int A, int B, GridHint hint, double thresh,
    int* __restrict__ out_cls, float4* __restrict__ out_loc, uint8_t* __restrict__ out_mask, int* __restrict__ out_owner,
    int nchunk, int ablate) {
    extern __shared__ unsigned s_bits[];       // 3 A-bit maps: taken | seen | dup
    __shared__ LocalLds S;
    const int tid = threadIdx.x;
    const int per_group = LOC_GROUP * nchunk;
    const int grp = blockIdx.x / per_group, within = blockIdx.x - grp * per_group;
    const int b = grp * LOC_GROUP + (within & (LOC_GROUP - 1)), chunk = within / LOC_GROUP;
    if (b >= B) return;
    // this thread's columns: nothing they need depends on the gt data -- in flight under steps 1-5
    double2 plo_[CPW], phi_[CPW];
    float4 ez_[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        const int cj = min((chunk * CPW + j) * WG + tid, A - 1);
        plo_[j] = *reinterpret_cast<const double2*>(priors + 4 * (size_t)cj);
        phi_[j] = *reinterpret_cast<const double2*>(priors + 4 * (size_t)cj + 2);
        ez_[j] = enc_zero[cj];
    }

    const int g0 = gt_off[b];
    const int nt = gt_off[b + 1] - g0;
    const int ntypes = hint.cand_off[hint.levels];
    const int nwords = (A + 31) >> 5;
    unsigned* taken = s_bits;
    unsigned* seen = s_bits + nwords;
    unsigned* dup = s_bits + 2 * nwords;

    // ---- 1. rows and anchor types ----
    for (int i = tid; i < 3 * nwords; i += WG) s_bits[i] = 0u;
    if (tid < nt) {
        const float4 g = gt_box[g0 + tid];
        const Corner gc = gt_corner(g);
        S.gt[tid] = g;
        S.cls[tid] = (int)gt_cls[g0 + tid];
        RowRec rr;
        rr.lx = gc.lx; rr.ly = gc.ly; rr.hx = gc.hx; rr.hy = gc.hy; rr.a = gc.a; rr.lbm = 0.0;
        S.rows[tid] = rr;
        RowF32 f;
        f.lx = (float)gc.lx; f.ly = (float)gc.ly; f.hx = (float)gc.hx; f.hy = (float)gc.hy; f.a = (float)gc.a;   // exact
        f.lb = 0.f; f.pad1 = f.pad2 = 0.f;
        S.rowf[tid] = f;
        S.head[tid] = -1;
        S.nolist[tid] = 0;
        S.rs[tid] = 1;
    }
    if (tid < ntypes) {
        // the level of type `tid` with compile-time indices into the kernel argument (a per-lane index would turn every
        // access into a memory load: that alone cost 25 us)
        int gw = 1, gh = 1, col = 0, k = 1, ox = 0, oy = 0, sx = 0, sy = 0;
#pragma unroll
        for (int l = 0; l < SSD_MAX_LEVELS; ++l) {
            if (l < hint.levels && tid >= hint.cand_off[l]) {
                gw = hint.gw[l]; gh = hint.gh[l]; k = hint.k[l]; col = hint.col_off[l] + tid - hint.cand_off[l];
                ox = sx; oy = sy;
            }
            if (l < hint.levels) { sx += hint.gw[l]; sy += hint.gh[l]; }
        }
        const double2 wh = *reinterpret_cast<const double2*>(priors + 4 * (size_t)col + 2);
        S.tw[tid] = wh.x;
        S.th[tid] = wh.y;
        S.tgw[tid] = gw; S.tgh[tid] = gh; S.tcol[tid] = col; S.tk[tid] = k; S.tcx[tid] = ox; S.tcy[tid] = oy;
    }
    {   // cell centres of every level, with k_priors' arithmetic: (x + .5) / gw -- the values the verified prior array holds
        int gwx = 1, x = 0, ghy = 1, y = 0, sx = 0, sy = 0;
        bool inx = false, iny = false;
#pragma unroll
        for (int l = 0; l < SSD_MAX_LEVELS; ++l)
            if (l < hint.levels) {
                if (tid >= sx && tid < sx + hint.gw[l]) { gwx = hint.gw[l]; x = tid - sx; inx = true; }
                if (tid >= sy && tid < sy + hint.gh[l]) { ghy = hint.gh[l]; y = tid - sy; iny = true; }
                sx += hint.gw[l]; sy += hint.gh[l];
            }
        if (inx) S.ccx[tid] = ((double)x + 0.5) / (double)gwx;
        if (iny) S.ccy[tid] = ((double)y + 0.5) / (double)ghy;
    }
    if (tid == 0) { S.nunits = 0; S.ncand = 0; S.nres = 0; }
    __syncthreads();

    // ---- 2. seeds: 8 lanes per row over the anchor types of the cell under the gt centre ----
    for (int r0 = 0; r0 < nt; r0 += WG / 8) {
        const int r = r0 + (tid >> 3), sub = tid & 7;
        const bool live = r < nt;
        double best = 0.0;
        if (live) {
            const float4 g = S.gt[r];
            const RowRec rr = S.rows[r];
            Corner gc;
            gc.lx = rr.lx; gc.ly = rr.ly; gc.hx = rr.hx; gc.hy = rr.hy; gc.a = rr.a;
            for (int t = sub; t < ntypes; t += 8) {
                const int tgw = S.tgw[t], tgh = S.tgh[t];
                int x = (int)floorf(g.x * (float)tgw);
                int y = (int)floorf(g.y * (float)tgh);
                x = min(max(x, 0), tgw - 1);
                y = min(max(y, 0), tgh - 1);
                const double cx = S.ccx[S.tcx[t] + x], cy = S.ccy[S.tcy[t] + y];
                double inter, uni;
                inter_union(gc, prior_corner(cx, cy, S.tw[t], S.th[t]), inter, uni);
                const double q = inter / uni;
                if (q > best) best = q;
            }
        }
#pragma unroll
        for (int m = 4; m > 0; m >>= 1) {
            const double o = shfl_xor_f64(best, m);
            if (o > best) best = o;
        }
        if (live && sub == 0) {
            S.rows[r].lbm = best * (1.0 - 1e-9) * 0.8 * SSD_MARGIN;
            if (!(best > 0.0)) S.nolist[r] = 1;
        }
    }
    __syncthreads();

    if (ablate == 1) return;
    // ---- 3 + 4. windows -> cells (work units in LDS) -> exact evaluation -> candidate chains, LOC_ROUND rows at a time ----
    for (int rbase = 0; rbase < nt; rbase += LOC_ROUND) {
        const int nrows = min(LOC_ROUND, nt - rbase);
        for (int item = tid; item < nrows * ntypes; item += WG) {
            const int r = rbase + item / ntypes, t = item % ntypes;
            const RowRec g = S.rows[r];
            if (!(g.lbm > 0.0)) continue;          // no seed: the row is scanned exactly in step 5
            const int tgw = S.tgw[t], tgh = S.tgh[t];
            const double wP = S.tw[t], hP = S.th[t];
            const double T = g.lbm * fmax(wP * hP, g.a) * (1.0 - 1e-6);   // the chain bound 0.8 L': the window of what step 4 keeps
            if (fmin(wP * hP, g.a) + 4e-10 < T) continue;                  // inter <= min(aP, aG) + 3e-10: this type cannot reach the bound
            const double m_w = T / (fmin(hP, fmax(g.hy - g.ly, 0.0)) + 2e-10) - 2e-10;
            const double m_h = T / (fmin(wP, fmax(g.hx - g.lx, 0.0)) + 2e-10) - 2e-10;
            const double gw = (double)tgw, gh = (double)tgh;
            // cell x has its centre at (x + .5) / gw: x in [xlo, xhi]; 1e-6 of a cell against the roundings of this block
            const double xlo = (g.lx + m_w - 0.5 * wP) * gw - 0.5 - 1e-6, xhi = (g.hx - m_w + 0.5 * wP) * gw - 0.5 + 1e-6;
            const double ylo = (g.ly + m_h - 0.5 * hP) * gh - 0.5 - 1e-6, yhi = (g.hy - m_h + 0.5 * hP) * gh - 0.5 + 1e-6;
            if (!(xlo <= xhi) || !(ylo <= yhi)) continue;                 // empty (also catches NaN)
            const int x0 = max((int)fmax(ceil(xlo), -1.0), 0), x1 = min((int)fmin(floor(xhi), 1e6), tgw - 1);
            const int y0 = max((int)fmax(ceil(ylo), -1.0), 0), y1 = min((int)fmin(floor(yhi), 1e6), tgh - 1);
            const int ncell = (x1 - x0 + 1) * (y1 - y0 + 1);
            if (x1 < x0 || y1 < y0) continue;
            int idx = atomicAdd(&S.nunits, ncell);     // one reservation per window (same-address LDS atomics serialise)
            if (idx + ncell > LOC_UNITS) {             // does not fit: hand the range back (it is never read: a successful
                atomicSub(&S.nunits, ncell);           // reservation cannot lie above a pending failed one), scan the row exactly
                S.nolist[r] = 1;
                continue;
            }
            for (int y = y0; y <= y1; ++y)
                for (int x = x0; x <= x1; ++x)
                    S.units[idx++] = (unsigned)r | ((unsigned)t << 6) | ((unsigned)y << 12) | ((unsigned)x << 19);
        }
        __syncthreads();
        const int nu = min(S.nunits, LOC_UNITS);
        for (int u = tid; u < nu; u += WG) {       // one cell per thread and pass: the gathers of a pass are all in flight
            const unsigned w = S.units[u];
            const int r = w & 63, t = (w >> 6) & 63, y = (w >> 12) & 127, x = (w >> 19) & 127;
            const int col = S.tcol[t] + (y * S.tgw[t] + x) * S.tk[t];
            const RowRec g = S.rows[r];
            Corner gc;
            gc.lx = g.lx; gc.ly = g.ly; gc.hx = g.hx; gc.hy = g.hy; gc.a = g.a;
            double inter, uni;
            // the prior of this cell from the model -- (cx, cy, w, h) are the very values the verified array holds, so no
            // load (and no memory round trip) is needed to evaluate it exactly
            inter_union(gc, prior_corner(S.ccx[S.tcx[t] + x], S.ccy[S.tcy[t] + y], S.tw[t], S.th[t]), inter, uni);
            if (inter >= g.lbm * uni) {
                const double q = inter / uni;
                if (q >= g.lbm) {
                    const int idx = atomicAdd(&S.ncand, 1);
                    if (idx < LOC_CANDS) {
                        S.cand[idx].q = q;
                        S.cand[idx].c = col;
                        S.cand[idx].next = atomicExch(&S.head[r], idx);
                    } else {
                        S.nolist[r] = 1;
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0) S.nunits = 0;
        __syncthreads();
    }

    if (ablate == 2) return;
    // ---- 5. phase 1 (utils/bbox.py:62-68) on the chains: the algorithm of phase1_image ----
    auto chain_best = [&](int r, bool only_free, double& q_out, int& c_out) {   // best (free) entry of row r's chain
        double bq = -1.0;
        int bc = INT_MAX;
        if (!S.nolist[r])
            for (int k = S.head[r]; k >= 0; k = S.cand[k].next) {
                const double q = S.cand[k].q;
                const int col = S.cand[k].c;
                if (only_free && ((taken[col >> 5] >> (col & 31)) & 1u)) continue;
                if (better(q, col, bq, bc)) { bq = q; bc = col; }
            }
        q_out = bq; c_out = bc;
    };
    if (tid < nt) {
        double q; int col;
        chain_best(tid, false, q, col);
        S.rq[tid] = q;
        S.rc[tid] = col;
        if (col == INT_MAX) S.res[atomicAdd(&S.nres, 1)] = tid;
    }
    __syncthreads();
    {
        const int nres = S.nres;               // rows without a usable chain: exact scan (never for valid boxes with a seed)
        for (int k = 0; k < nres; ++k) {
            const int r = S.res[k];
            double q; int col;
            row_scan(gt_corner(S.gt[r]), priors, A, taken, q, col, S.q, S.c);
            __syncthreads();
            if (tid == 0) { S.rq[r] = q; S.rc[r] = col; }
            __syncthreads();
        }
        if (tid == 0) S.nres = 0;
        __syncthreads();
    }
    for (int iter = 0; iter <= nt; ++iter) {
        if (tid < nt && S.rs[tid]) {           // which free rows share their best column with another free row?
            const int col = S.rc[tid];
            const unsigned bit = 1u << (col & 31);
            if (atomicOr(&seen[col >> 5], bit) & bit) atomicOr(&dup[col >> 5], bit);
        }
        __syncthreads();
        double pq = -1.0;
        int pr = INT_MAX;
        if (tid < nt && S.rs[tid]) {
            const int col = S.rc[tid];
            if ((dup[col >> 5] >> (col & 31)) & 1u) { pq = S.rq[tid]; pr = tid; }
        }
        wg_argmax(pq, pr, S.q, S.c);
        if (pr == INT_MAX) break;              // no sharing left: every free row keeps its column
        const int cstar = S.rc[pr];
        __syncthreads();
        if (tid < nt && S.rs[tid]) {           // pivot and every free row ahead of it take their columns
            const int col = S.rc[tid];
            if (tid == pr || better(S.rq[tid], tid, pq, pr)) {
                S.rs[tid] = 0;
                atomicOr(&taken[col >> 5], 1u << (col & 31));
            } else if (col == cstar) {
                S.rs[tid] = 2;                 // wanted the pivot's column: next-best free column
                S.res[atomicAdd(&S.nres, 1)] = tid;
            }
        }
        for (int i = tid; i < 2 * nwords; i += WG) seen[i] = 0u;   // seen and dup are adjacent
        __syncthreads();
        const int nres = S.nres;
        for (int k = 0; k < nres; ++k) {       // uniform; usually one row
            const int r = S.res[k];
            // the chain holds every column with IoU >= 0.8 L': if one of them is still free, the best free one is the
            // row's new maximum (everything outside the chain is smaller); else re-scan exactly
            double q; int col;
            chain_best(r, true, q, col);       // (every thread walks the same few entries)
            if (col == INT_MAX) row_scan(gt_corner(S.gt[r]), priors, A, taken, q, col, S.q, S.c);
            __syncthreads();
            if (tid == 0) { S.rq[r] = q; S.rc[r] = col; S.rs[r] = 1; }
            __syncthreads();
        }
        if (tid == 0) S.nres = 0;
        __syncthreads();
    }
    __syncthreads();
    // phase-1 columns -> bitmap (`taken` is reused: every row's final column)
    for (int i = tid; i < nwords; i += WG) taken[i] = 0u;
    __syncthreads();
    if (tid < nt) atomicOr(&taken[S.rc[tid] >> 5], 1u << (S.rc[tid] & 31));
    __syncthreads();

    if (ablate == 3) return;
    // ---- 6. stream: this thread's columns against all rows, four rows per step (their LDS records and pre-filters in
    //         flight together: one row per step left the loop waiting ~400 cycles on each LDS read) ----
    Corner pc[CPW];
    float plx[CPW], ply[CPW], phx[CPW], phy[CPW], pa[CPW], cb32[CPW];
    double cbq[CPW], cbm[CPW];
    int cbr[CPW], own[CPW];
    bool valid[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        const int c = (chunk * CPW + j) * WG + tid;
        valid[j] = c < A;
        pc[j] = prior_corner(plo_[j].x, plo_[j].y, phi_[j].x, phi_[j].y);
        plx[j] = (float)pc[j].lx; ply[j] = (float)pc[j].ly; phx[j] = (float)pc[j].hx; phy[j] = (float)pc[j].hy; pa[j] = (float)pc[j].a;
        cbq[j] = thresh;                           // phase 2 needs max > thresh (utils/bbox.py:73)
        cbm[j] = thresh * SSD_MARGIN;
        cb32[j] = (float)cbm[j];
        cbr[j] = -1;
        own[j] = -1;                               // the row that took this column in phase 1
        if (valid[j] && ((taken[c >> 5] >> (c & 31)) & 1u))
            for (int r = 0; r < nt; ++r)
                if (S.rc[r] == c) own[j] = r;
    }
    const int nt_s = ablate == 4 ? 0 : nt;
    for (int r0 = 0; r0 < nt_s; r0 += 4) {
        RowF32 f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = S.rowf[min(r0 + i, nt - 1)];     // one address for the wave: LDS broadcast
        unsigned maybe = 0;                        // bit 4 j + i: column j against row r0 + i survives the pre-filter
#pragma unroll
        for (int j = 0; j < CPW; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (valid[j] && r0 + i < nt && !prefilter_rejects(f[i], plx[j], ply[j], phx[j], phy[j], pa[j], cb32[j]))
                    maybe |= 1u << (4 * j + i);
        if (__ballot(maybe != 0)) {                // rare: some lane's pair survives
            for (int i = 0; i < 4 && r0 + i < nt; ++i) {           // rows in order: the bound of a column only grows
                const RowRec g = S.rows[r0 + i];
                Corner gc;
                gc.lx = g.lx; gc.ly = g.ly; gc.hx = g.hx; gc.hy = g.hy; gc.a = g.a;
#pragma unroll
                for (int j = 0; j < CPW; ++j) {
                    if (!((maybe >> (4 * j + i)) & 1u)) continue;
                    double inter, uni;
                    inter_union(gc, pc[j], inter, uni);
                    if (inter >= cbm[j] * uni) {   // the exact filter of k_match_pairs, then the division
                        const double q = inter / uni;
                        if (q > cbq[j]) { cbq[j] = q; cbr[j] = r0 + i; cbm[j] = q * SSD_MARGIN; cb32[j] = (float)cbm[j]; }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        if (!valid[j]) continue;
        const int c = (chunk * CPW + j) * WG + tid;
        const int row = own[j] >= 0 ? own[j] : cbr[j];      // phase 1 first (utils/bbox.py:84-90: its scatter comes last)
        const size_t o = (size_t)b * A + c;
        if (out_owner) out_owner[o] = row;
        if (row >= 0) {
            out_cls[o] = S.cls[row];
            out_mask[o] = 1;
            out_loc[o] = encode_row(S.gt[row], plo_[j].x, plo_[j].y, phi_[j].x, phi_[j].y);
        } else {
            out_cls[o] = 0;
            out_mask[o] = 0;
            out_loc[o] = ez_[j];
        }
    }
}

// Does `hint` describe `priors`?  Every prior must sit at the cell centre k_priors computes -- bit for bit -- and share its
// (w, h) with the same anchor type of the first cell of its level; mismatches are counted.
__global__ void k_grid_verify(const double* __restrict__ priors, int A, GridHint hint, int* __restrict__ mismatches) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A) return;
    int l = 0;
    while (l + 1 < hint.levels && i >= hint.col_off[l + 1]) ++l;
    const int rel = i - hint.col_off[l];
    const int cell = rel / hint.k[l], j = rel - cell * hint.k[l];
    const int y = cell / hint.gw[l], x = cell - y * hint.gw[l];
    const double cx = ((double)x + 0.5) / (double)hint.gw[l];
    const double cy = ((double)y + 0.5) / (double)hint.gh[l];
    const double* p = priors + 4 * (size_t)i;
    const double* p0 = priors + 4 * (size_t)(hint.col_off[l] + j);
    // (areas >= 1e-3: the window bound of k_match_local takes uni >= max(aP, aG)(1 - 1e-6), which needs areas >> 1e-10)
    const bool ok = p[0] == cx && p[1] == cy && p[2] == p0[2] && p[3] == p0[3] && p[2] > 0.0 && p[3] > 0.0 && p[2] * p[3] >= 1e-3;
    if (!ok) atomicAdd(mismatches, 1);
}

// ------------------------------------------------------------------------------------------------
__global__ void k_iou_n(const float4* __restrict__ b1, const double* __restrict__ b2, int n,
                        double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double2 lo = *reinterpret_cast<const double2*>(b2 + 4 * (size_t)i);
    const double2 hi = *reinterpret_cast<const double2*>(b2 + 4 * (size_t)i + 2);
    double inter, uni;
    inter_union(gt_corner(b1[i]), prior_corner(lo.x, lo.y, hi.x, hi.y), inter, uni);
    out[i] = inter / uni;
}

// apply_anchor_box (utils/bbox.py:94-101) on n paired rows, float64 result as numpy produces.
__global__ void k_apply_anchor_box(const float4* __restrict__ box, const double* __restrict__ priors, int n,
                                   double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 g = box[i];
    const double2 lo = *reinterpret_cast<const double2*>(priors + 4 * (size_t)i);
    const double2 hi = *reinterpret_cast<const double2*>(priors + 4 * (size_t)i + 2);
    double* o = out + 4 * (size_t)i;
    o[0] = ((double)g.x - lo.x) / hi.x;
    o[1] = ((double)g.y - lo.y) / hi.y;
    o[2] = log((double)fmaxf(g.z, 1e-5f) / fmax(hi.x, 1e-5));
    o[3] = log((double)fmaxf(g.w, 1e-5f) / fmax(hi.y, 1e-5));
}

__global__ void k_encode_zero(const double* __restrict__ priors, int A, float4* __restrict__ enc_zero) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= A) return;
    const double2 lo = *reinterpret_cast<const double2*>(priors + 4 * (size_t)c);
    const double2 hi = *reinterpret_cast<const double2*>(priors + 4 * (size_t)c + 2);
    enc_zero[c] = encode_row(make_float4(0.f, 0.f, 0.f, 0.f), lo.x, lo.y, hi.x, hi.y);
}

struct PriorLevels {
    int levels;
    int gh[SSD_MAX_LEVELS], gw[SSD_MAX_LEVELS], k[SSD_MAX_LEVELS], col_off[SSD_MAX_LEVELS + 1];
    double s_k[SSD_MAX_LEVELS], s_prime[SSD_MAX_LEVELS];
    int ratio_off[SSD_MAX_LEVELS + 1];
    double ratio_sqrt[4 * SSD_MAX_LEVELS];   // host-computed sqrt(ratio) (correctly rounded, as math.sqrt)
};

// models/ssd_model.py:173-194.  s_k and s' are computed on the host with the same IEEE
// operations (division, multiply, correctly rounded sqrt); per-prior terms here.
__global__ void k_priors(PriorLevels L, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.col_off[L.levels]) return;
    int l = 0;
    while (l + 1 < L.levels && i >= L.col_off[l + 1]) ++l;
    const int rel = i - L.col_off[l];
    const int cell = rel / L.k[l], j = rel - cell * L.k[l];
    const int y = cell / L.gw[l], x = cell - y * L.gw[l];
    const double cx = ((double)x + 0.5) / (double)L.gw[l];
    const double cy = ((double)y + 0.5) / (double)L.gh[l];
    double w, h;
    if (j == 0) { w = h = L.s_k[l]; }
    else if (j == 1) { w = h = L.s_prime[l]; }
    else {
        const int ri = (j - 2) >> 1;
        const double q = L.ratio_sqrt[L.ratio_off[l] + ri];
        if (((j - 2) & 1) == 0) { w = L.s_k[l] * q; h = L.s_k[l] / q; }
        else { w = L.s_k[l] / q; h = L.s_k[l] * q; }
    }
    double* o = out + 4 * (size_t)i;
    o[0] = cx; o[1] = cy; o[2] = w; o[3] = h;
}

}  // namespace

// ================================================================================================
extern "C" {

int ssd_hip_abi_version(void) { return SSD_ABI_VERSION; }

const char* ssd_status_string(int status) {
    switch (status) {
        case SSD_OK: return "ok";
        case SSD_ERR_ASSERT: return "assertion of the reference violated";
        case SSD_ERR_VALUE: return "invalid argument";
        case SSD_ERR_WORKSPACE: return "workspace too small";
        case SSD_ERR_LAUNCH: return "HIP launch failed";
        case SSD_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown status";
    }
}

int ssd_priors_count(const int* grid_hw, int levels, const int* ratio_off) {
    if (!grid_hw || !ratio_off || levels <= 0 || levels > SSD_MAX_LEVELS) return SSD_ERR_VALUE;
    long long n = 0;
    for (int l = 0; l < levels; ++l)
        n += (long long)grid_hw[2 * l] * grid_hw[2 * l + 1] * (2 + 2 * (ratio_off[l + 1] - ratio_off[l]));
    return n > INT_MAX ? SSD_ERR_VALUE : (int)n;
}

int ssd_priors(const int* grid_hw, int levels, const double* s_ref, const int* ratios, const int* ratio_off,
               double in_size, double* out, void* stream) {
    const int A = ssd_priors_count(grid_hw, levels, ratio_off);
    if (A <= 0 || !s_ref || !ratios || !out || !(in_size > 0.0)) return SSD_ERR_VALUE;
    if (ratio_off[levels] > 4 * SSD_MAX_LEVELS) return SSD_ERR_UNSUPPORTED;
    PriorLevels L;
    L.levels = levels;
    L.col_off[0] = 0;
    for (int l = 0; l < levels; ++l) {
        L.gh[l] = grid_hw[2 * l];
        L.gw[l] = grid_hw[2 * l + 1];
        L.k[l] = 2 + 2 * (ratio_off[l + 1] - ratio_off[l]);
        L.col_off[l + 1] = L.col_off[l] + L.gh[l] * L.gw[l] * L.k[l];
        const double s_k = s_ref[l] / in_size;                       // :184
        L.s_k[l] = s_k;
        L.s_prime[l] = __builtin_sqrt(s_k * (s_ref[l + 1] / in_size));   // :187
        L.ratio_off[l] = ratio_off[l];
    }
    L.ratio_off[levels] = ratio_off[levels];
    for (int i = 0; i < ratio_off[levels]; ++i) L.ratio_sqrt[i] = __builtin_sqrt((double)ratios[i]);   // :191
    hipLaunchKernelGGL(k_priors, dim3((A + 255) / 256), dim3(256), 0, (hipStream_t)stream, L, out);
    return ssd_launch_status();
}

int ssd_encode_zero(const double* priors, int A, float* enc_zero, void* stream) {
    if (!priors || !enc_zero || A <= 0) return SSD_ERR_VALUE;
    hipLaunchKernelGGL(k_encode_zero, dim3((A + 255) / 256), dim3(256), 0, (hipStream_t)stream, priors, A,
                       reinterpret_cast<float4*>(enc_zero));
    return ssd_launch_status();
}

int ssd_apply_anchor_box(const float* box, const double* priors, int n, double* out, void* stream) {
    if (n < 0 || (n > 0 && (!box || !priors || !out))) return SSD_ERR_VALUE;
    if (n == 0) return SSD_OK;
    hipLaunchKernelGGL(k_apply_anchor_box, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(box), priors, n, out);
    return ssd_launch_status();
}

int ssd_iou_n(const float* b1, const double* b2, int n, double* out, void* stream) {
    if (n < 0 || (n > 0 && (!b1 || !b2 || !out))) return SSD_ERR_VALUE;
    if (n == 0) return SSD_OK;
    hipLaunchKernelGGL(k_iou_n, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(b1), b2, n, out);
    return ssd_launch_status();
}

static inline int match_nchunk(int A) { return (A + WG - 1) / WG; }

static bool make_hint(const ssd_prior_grid* grid, GridHint* hint) {
    hint->levels = 0;
    if (!grid || grid->levels <= 0 || grid->levels > SSD_MAX_LEVELS) return false;
    hint->levels = grid->levels;
    hint->col_off[0] = hint->cand_off[0] = 0;
    for (int l = 0; l < grid->levels; ++l) {
        hint->gh[l] = grid->grid_h[l] > 0 ? grid->grid_h[l] : 1;
        hint->gw[l] = grid->grid_w[l] > 0 ? grid->grid_w[l] : 1;
        hint->k[l] = grid->per_cell[l] > 0 ? grid->per_cell[l] : 1;
        hint->col_off[l + 1] = hint->col_off[l] + hint->gh[l] * hint->gw[l] * hint->k[l];
        hint->cand_off[l + 1] = hint->cand_off[l] + hint->k[l];
    }
    return true;
}

int ssd_prior_grid_verify(const double* priors, int A, ssd_prior_grid* grid, int32_t* scratch, void* stream) {
    if (!priors || !grid || !scratch || A <= 0) return SSD_ERR_VALUE;
    grid->verified = 0;
    GridHint hint;
    if (!make_hint(grid, &hint) || hint.col_off[hint.levels] != A) return SSD_OK;      // cannot describe these priors
    for (int l = 0; l < hint.levels; ++l)
        if (grid->grid_h[l] <= 0 || grid->grid_w[l] <= 0 || grid->per_cell[l] <= 0) return SSD_OK;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(scratch, 0, sizeof(int32_t), s) != hipSuccess) return SSD_ERR_LAUNCH;
    hipLaunchKernelGGL(k_grid_verify, dim3((A + 255) / 256), dim3(256), 0, s, priors, A, hint, scratch);
    int32_t bad = -1;
    if (hipMemcpyAsync(&bad, scratch, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess) return SSD_ERR_LAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return SSD_ERR_LAUNCH;
    grid->verified = bad == 0 ? SSD_GRID_VERIFIED : 0;
    return SSD_OK;
}

size_t ssd_match_encode_workspace_bytes(int B, int A, int total_gt) {
    (void)B;
    if (A <= 0 || total_gt < 0) return 0;
    return (size_t)(total_gt > 0 ? total_gt : 1) * sizeof(RowSlot);
}

int ssd_match_encode(const float* gt_box, const float* gt_cls, const int32_t* gt_off, int B, int total_gt,
                     int max_nt, const double* priors, const float* enc_zero, int A, const ssd_prior_grid* grid,
                     double thresh, int32_t* out_cls, float* out_loc, uint8_t* out_mask, int32_t* out_owner,
                     void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || A <= 0 || total_gt < 0 || max_nt < 0) return SSD_ERR_VALUE;
    if (max_nt > A) return SSD_ERR_ASSERT;                     // utils/bbox.py:50
    if (!(thresh > 0.0)) return SSD_ERR_ASSERT;                // utils/bbox.py:51
    if (B == 0) return SSD_OK;
    if (!gt_off || !priors || !enc_zero || !out_cls || !out_loc || !out_mask) return SSD_ERR_VALUE;
    if (total_gt > 0 && (!gt_box || !gt_cls)) return SSD_ERR_VALUE;
    const size_t lds_bitmaps = 3 * (size_t)((A + 31) / 32) * sizeof(unsigned);
    if (lds_bitmaps > 100 * 1024) return SSD_ERR_UNSUPPORTED;

    GridHint hint;
    const bool have_hint = make_hint(grid, &hint);
    const int nchunk = match_nchunk(A);
    hipStream_t s = (hipStream_t)stream;

    // One launch, no workspace (k_match_local): needs a geometry that was checked against the prior array
    // (ssd_prior_grid_verify) and images whose rows fit the kernel's LDS tables.  Bit-identical to the three-launch path
    // (tests/test_match_gpu.py) but, as measured on MI355X, not faster on COCO-shaped batches (50 vs 36 us at batch 64: the
    // heaviest image's phase 1 and streaming run on too few workgroups) -- so it is opt-in (development override
    // SSD_MATCH_FUSED != 0) until its phases are tuned; DESIGN.md section 5 has the numbers.
    bool local = have_hint && grid->verified == SSD_GRID_VERIFIED && hint.col_off[hint.levels] == A && max_nt <= LOC_MAX_ROWS &&
                 hint.cand_off[hint.levels] <= LOC_MAX_TYPES && lds_bitmaps <= 40 * 1024 && ssd_knob("SSD_MATCH_FUSED", 0) != 0;
    int sum_w = 0, sum_h = 0;
    for (int l = 0; local && l < hint.levels; ++l) { local = hint.gh[l] <= 128 && hint.gw[l] <= 128; sum_w += hint.gw[l]; sum_h += hint.gh[l]; }
    local = local && sum_w <= LOC_MAX_CELLS && sum_h <= LOC_MAX_CELLS;
    if (local) {
        const int ngroups = (B + LOC_GROUP - 1) / LOC_GROUP;
        const int knob = ssd_knob("SSD_MATCH_FUSED", 0);
        const int cpw = (knob & 2) ? 4 : ((knob & 4) ? 1 : 2);        // development: bits 1 / 2 select 4 / 1 columns per thread (default 2)
        const int nch = (A + WG * cpw - 1) / (WG * cpw);
#define SSD_LAUNCH_LOCAL(CPW_)                                                                                         \
        hipLaunchKernelGGL(k_match_local<CPW_>, dim3((unsigned)(ngroups * LOC_GROUP * nch)), dim3(WG), lds_bitmaps, s,   \
                           reinterpret_cast<const float4*>(gt_box), gt_cls, gt_off, priors, reinterpret_cast<const float4*>(enc_zero), \
                           A, B, hint, thresh, out_cls, reinterpret_cast<float4*>(out_loc), out_mask, out_owner, nch, knob >> 4)
        if (cpw == 1) SSD_LAUNCH_LOCAL(1); else if (cpw == 2) SSD_LAUNCH_LOCAL(2); else SSD_LAUNCH_LOCAL(4);
#undef SSD_LAUNCH_LOCAL
        return ssd_launch_status();
    }

    if (ws_bytes < ssd_match_encode_workspace_bytes(B, A, total_gt) || !ws) return SSD_ERR_WORKSPACE;
    RowSlot* slots = static_cast<RowSlot*>(ws);
    if (total_gt > 0) {
        hipLaunchKernelGGL(k_match_rows, dim3((total_gt * 32 + WG - 1) / WG), dim3(WG), 0, s,
                           reinterpret_cast<const float4*>(gt_box), total_gt, priors, A, hint, slots);
        if (hipGetLastError() != hipSuccess) return SSD_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(k_match_pairs, dim3((unsigned)((B + PAIRS_GROUP - 1) / PAIRS_GROUP * PAIRS_GROUP * nchunk)), dim3(WG), 0, s,
                       reinterpret_cast<const float4*>(gt_box), gt_cls, gt_off, slots, priors,
                       reinterpret_cast<const float4*>(enc_zero), A, B, nchunk, thresh, out_cls,
                       reinterpret_cast<float4*>(out_loc), out_mask, out_owner);
    if (hipGetLastError() != hipSuccess) return SSD_ERR_LAUNCH;
    if (total_gt > 0) {
        hipLaunchKernelGGL(k_match_phase1, dim3(B), dim3(WG), lds_bitmaps, s,
                           reinterpret_cast<const float4*>(gt_box), gt_cls, gt_off, priors, A, slots, out_cls,
                           reinterpret_cast<float4*>(out_loc), out_mask, out_owner);
    }
    return ssd_launch_status();
}

}  // extern "C"
