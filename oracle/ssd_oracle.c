/* CPU oracle, C restatement -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Literal restatement of the reference's target assignment (paths relative to /root/reference):
 *   oracle_iou_n          utils/bbox.py:28-41   (float32 box_1, float64 box_2, as match_bbox calls it)
 *   oracle_match_literal  utils/bbox.py:44-91   (materialised n_t x A matrix, one full argmax per pick)
 *   oracle_encode         utils/bbox.py:94-101  + float32 cast (models/ssd_model.py:222)
 * and of the build-defined NMS (SURVEY.md A9'; same definition as oracle/ssd_oracle.py:nms).
 * Pinned against the golden vectors captured from the reference (tests/test_oracle_c.py).
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: numpy's arithmetic is unfused IEEE). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static double iou_pair(const float* g, const double* p) {
    const float hw = g[2] / 2.0f, hh = g[3] / 2.0f;
    const double glx = (double)(g[0] - hw), ghx = (double)(g[0] + hw);
    const double gly = (double)(g[1] - hh), ghy = (double)(g[1] + hh);
    const double ga = (double)(g[2] * g[3]);
    const double plx = p[0] - p[2] / 2.0, phx = p[0] + p[2] / 2.0;
    const double ply = p[1] - p[3] / 2.0, phy = p[1] + p[3] / 2.0;
    const double pa = p[2] * p[3];
    const double x_lo = glx > plx ? glx : plx, y_lo = gly > ply ? gly : ply;
    const double x_hi = ghx < phx ? ghx : phx, y_hi = ghy < phy ? ghy : phy;
    const double dx = x_hi - x_lo, dy = y_hi - y_lo;
    const double inter = (dx > 1e-10 ? dx : 1e-10) * (dy > 1e-10 ? dy : 1e-10);
    return inter / (ga + pa - inter + 1e-10);
}

void oracle_iou_n(const float* b1, const double* b2, int n, double* out) {
    for (int i = 0; i < n; ++i) out[i] = iou_pair(b1 + 4 * i, b2 + 4 * i);
}

static long argmax(const double* m, long n) {       /* first maximum, as np.argmax */
    long best = 0;
    for (long i = 1; i < n; ++i)
        if (m[i] > m[best]) best = i;
    return best;
}

/* returns the number of recorded pairs, or -1 on the reference's asserts (utils/bbox.py:50-51) */
int oracle_match_literal(const float* gt_cls, const float* gt_box, int n_t, const double* priors, int A, double thresh,
                         int32_t* out_cls, float* out_box, uint8_t* out_mask, int32_t* out_owner) {
    if (n_t > A || !(thresh > 0.0)) return -1;
    memset(out_cls, 0, sizeof(int32_t) * (size_t)A);
    memset(out_box, 0, sizeof(float) * 4 * (size_t)A);
    memset(out_mask, 0, (size_t)A);
    if (out_owner) for (int c = 0; c < A; ++c) out_owner[c] = -1;
    if (n_t == 0) return 0;
    const long n = (long)n_t * A;
    double* live = (double*)malloc(sizeof(double) * n);
    double* work = (double*)malloc(sizeof(double) * n);
    for (int r = 0; r < n_t; ++r)
        for (int c = 0; c < A; ++c) live[(long)r * A + c] = iou_pair(gt_box + 4 * r, priors + 4 * c);
    memcpy(work, live, sizeof(double) * n);
    int picks = 0;
    for (int it = 0; it < n_t; ++it) {                /* phase 1, :62-68 */
        const long idx = argmax(work, n);
        const int r = (int)(idx / A), c = (int)(idx % A);
        for (int k = 0; k < A; ++k) work[(long)r * A + k] = 0.0;
        for (int k = 0; k < n_t; ++k) { work[(long)k * A + c] = 0.0; live[(long)k * A + c] = 0.0; }
        out_mask[c] = 1; out_cls[c] = (int32_t)gt_cls[r];
        memcpy(out_box + 4 * c, gt_box + 4 * r, 4 * sizeof(float));
        if (out_owner) out_owner[c] = r;
        ++picks;
    }
    for (;;) {                                        /* phase 2, :71-79 */
        const long idx = argmax(live, n);
        if (live[idx] <= thresh) break;
        const int r = (int)(idx / A), c = (int)(idx % A);
        for (int k = 0; k < n_t; ++k) live[(long)k * A + c] = 0.0;
        out_mask[c] = 1; out_cls[c] = (int32_t)gt_cls[r];
        memcpy(out_box + 4 * c, gt_box + 4 * r, 4 * sizeof(float));
        if (out_owner) out_owner[c] = r;
        ++picks;
    }
    free(live); free(work);
    return picks;
}

void oracle_encode(const float* box, const double* priors, int n, float* out) {
    for (int i = 0; i < n; ++i) {
        const float* g = box + 4 * i;
        const double* p = priors + 4 * i;
        const float gw = g[2] > 1e-5f ? g[2] : 1e-5f, gh = g[3] > 1e-5f ? g[3] : 1e-5f;
        const double pw = p[2] > 1e-5 ? p[2] : 1e-5, ph = p[3] > 1e-5 ? p[3] : 1e-5;
        out[4 * i + 0] = (float)(((double)g[0] - p[0]) / p[2]);
        out[4 * i + 1] = (float)(((double)g[1] - p[1]) / p[3]);
        out[4 * i + 2] = (float)log((double)gw / pw);
        out[4 * i + 3] = (float)log((double)gh / ph);
    }
}

static float iou_f32(const float* a, const float* b) {
    const float a1 = a[2] * a[3], a2 = b[2] * b[3];
    const float lx = fmaxf(a[0] - a[2] / 2.0f, b[0] - b[2] / 2.0f), ly = fmaxf(a[1] - a[3] / 2.0f, b[1] - b[3] / 2.0f);
    const float hx = fminf(a[0] + a[2] / 2.0f, b[0] + b[2] / 2.0f), hy = fminf(a[1] + a[3] / 2.0f, b[1] + b[3] / 2.0f);
    const float inter = fmaxf(0.0f, hx - lx) * fmaxf(0.0f, hy - ly);
    return inter / (a1 + a2 - inter + 1e-10f);
}

typedef struct { float s; int i; } cand_t;
static int cand_cmp(const void* x, const void* y) {
    const cand_t* a = (const cand_t*)x; const cand_t* b = (const cand_t*)y;
    if (a->s != b->s) return a->s > b->s ? -1 : 1;
    return a->i < b->i ? -1 : (a->i > b->i);
}

/* per-image per-class greedy NMS; returns the number kept */
int oracle_nms(const float* score, const int32_t* cls, const float* box, const uint8_t* cand, int A, float iou_thresh,
               int max_cand, uint8_t* keep) {
    cand_t* list = (cand_t*)malloc(sizeof(cand_t) * (size_t)A);
    int n = 0;
    for (int a = 0; a < A; ++a) { keep[a] = 0; if (cand[a]) { list[n].s = score[a]; list[n].i = a; ++n; } }
    qsort(list, (size_t)n, sizeof(cand_t), cand_cmp);
    if (n > max_cand) n = max_cand;
    int* kept = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int nk = 0;
    for (int j = 0; j < n; ++j) {
        const int i = list[j].i;
        int ok = 1;
        for (int k = 0; k < nk && ok; ++k)
            if (cls[kept[k]] == cls[i] && iou_f32(box + 4 * i, box + 4 * kept[k]) > iou_thresh) ok = 0;
        if (ok) { kept[nk++] = i; keep[i] = 1; }
    }
    free(kept); free(list);
    return nk;
}
