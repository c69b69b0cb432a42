/*
 * orb_ref.c -- CPU restatement of the reference ORB extractor / matcher cores.
 * TEST INFRASTRUCTURE ONLY (parity oracle + timed CPU baseline); see orb_ref.h.
 * PARITY UNPINNED against the OpenCV-backed original (no upstream fixtures).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  All
 * float steps are separate IEEE-754 single operations, as in the reference's
 * -O0, non-FMA build (CMakeLists.txt:8-9).
 */
#include "orb_ref.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* OpenCV scalar conversions (opencv2/core/fast_math.hpp semantics)           */
/* ------------------------------------------------------------------------- */
int orbref_round_f(float v) { return (int)lrintf(v); }  /* cvtss2si: ties-to-even */
int orbref_round_d(double v) { return (int)lrint(v); }
int orbref_floor_f(float v) { int i = (int)v; return i - (i > v); }
int orbref_ceil_f(float v) { int i = (int)v; return i + (i < v); }

static const int8_t g_pattern[1024] = {
#include "orb_pattern.inc"
};
const int8_t *orbref_pattern(void) { return g_pattern; }

/* ------------------------------------------------------------------------- */
/* Construction: modules/ORB/ORBExtractor.cpp:424-475                          */
/* ------------------------------------------------------------------------- */
static void compute_quotas(orbref_cfg *cfg, int n_features)
{
    /* modules/ORB/ORBExtractor.cpp:443-452 (and :483-492 for the re-quota ctor).
     * pow(float,int) resolves to the double overload, so the quotient is
     * evaluated in double and rounded once to float. */
    cfg->n_features = n_features;
    float inv2 = 1.0f / (cfg->scale_factor * cfg->scale_factor);
    float num = (float)n_features * (1 - inv2);
    float nd = (float)((double)num / (1.0 - pow((double)inv2, (double)cfg->n_levels)));
    int sum = 0;
    for (int l = 0; l < cfg->n_levels - 1; ++l) {
        cfg->n_features_per_level[l] = orbref_round_f(nd);
        sum += cfg->n_features_per_level[l];
        nd *= inv2;
    }
    int last = n_features - sum;
    cfg->n_features_per_level[cfg->n_levels - 1] = last > 1 ? last : 1;
}

void orbref_set_blur_taps(orbref_cfg *cfg, int variant)
{
    /* 8.8 fixed-point 7-tap sigma=2 kernel.  Variant 0 is
     * the error-diffused set that sums to 256, variant 1 the plainly rounded
     * set (sum 257).  Which one OpenCV 4.2.0 ships is not verifiable here. */
    static const int t0[7] = {18, 34, 48, 56, 48, 34, 18};
    static const int t1[7] = {18, 34, 49, 55, 49, 34, 18};
    memcpy(cfg->blur_taps, variant == 1 ? t1 : t0, sizeof t0);
}

void orbref_cfg_init(orbref_cfg *cfg, int n_features, float scale_factor, int n_levels,
                     int ini_th_fast, int min_th_fast)
{
    memset(cfg, 0, sizeof *cfg);
    if (n_levels > ORBREF_MAX_LEVELS) n_levels = ORBREF_MAX_LEVELS;
    cfg->n_levels = n_levels;
    cfg->ini_th_fast = ini_th_fast;
    cfg->min_th_fast = min_th_fast;
    cfg->scale_factor = scale_factor;
    cfg->log_scale_factor = logf(scale_factor); /* :428, std::log(float) */
    /* :432-438 */
    cfg->scale_factors[0] = cfg->inv_scale_factors[0] = 1.f;
    cfg->square_sigmas[0] = cfg->inv_square_sigmas[0] = 1.f;
    for (int i = 1; i < n_levels; ++i) {
        cfg->scale_factors[i] = cfg->scale_factors[i - 1] * scale_factor;
        cfg->inv_scale_factors[i] = 1.f / cfg->scale_factors[i];
        cfg->square_sigmas[i] = cfg->scale_factors[i] * cfg->scale_factors[i];
        cfg->inv_square_sigmas[i] = 1.f / cfg->square_sigmas[i];
    }
    compute_quotas(cfg, n_features);

    /* :460-474 -- row half-widths of the radius-15 circular patch */
    const int HP = ORBREF_HALF_PATCH;
    int v, v0;
    int vmax = orbref_floor_f((float)HP * sqrtf(2.f) / 2 + 1);
    int vmin = orbref_ceil_f((float)HP * sqrtf(2.f) / 2);
    const double hp2 = HP * HP;
    for (v = 0; v <= vmax; ++v) cfg->u_max[v] = orbref_round_d(sqrt(hp2 - v * v));
    for (v = HP, v0 = 0; v >= vmin; --v) {
        while (cfg->u_max[v0] == cfg->u_max[v0 + 1]) ++v0;
        cfg->u_max[v] = v0;
        ++v0;
    }
    orbref_set_blur_taps(cfg, 0);
}

void orbref_cfg_requota(orbref_cfg *cfg, int n_features) { compute_quotas(cfg, n_features); }

void orbref_level_size(const orbref_cfg *cfg, int w0, int h0, int level, int *w, int *h)
{
    /* modules/ORB/ORBExtractor.cpp:563-564 -- always from the level-0 size */
    if (level == 0) { *w = w0; *h = h0; return; }
    float s = cfg->inv_scale_factors[level];
    *w = orbref_round_f((float)w0 * s);
    *h = orbref_round_f((float)h0 * s);
}

/* ------------------------------------------------------------------------- */
/* cv::resize(INTER_LINEAR), 8UC1 -- modules/ORB/ORBExtractor.cpp:565           */
/* OpenCV 4.2 imgproc/resize.cpp: HResizeLinear<uchar,int,short,2048> +        */
/* VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>       */
/* ------------------------------------------------------------------------- */
static short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

static void linear_coeffs(int dn, int sn, int clamp_ofs, int *ofs, short *coef)
{
    double inv_scale = (double)dn / sn;
    double scale = 1. / inv_scale;
    for (int d = 0; d < dn; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = orbref_floor_f(f);
        f -= (float)s;
        if (clamp_ofs) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= sn - 1) { f = 0; s = sn - 1; }
        }
        ofs[d] = s;
        coef[2 * d] = sat_short(orbref_round_f((1.f - f) * 2048.f));
        coef[2 * d + 1] = sat_short(orbref_round_f(f * 2048.f));
    }
}

void orbref_resize_linear(const uint8_t *src, int sw, int sh, int sstride,
                          uint8_t *dst, int dw, int dh, int dstride)
{
    int *xofs = (int *)malloc(sizeof(int) * dw);
    int *yofs = (int *)malloc(sizeof(int) * dh);
    short *alpha = (short *)malloc(sizeof(short) * 2 * dw);
    short *beta = (short *)malloc(sizeof(short) * 2 * dh);
    int *row0 = (int *)malloc(sizeof(int) * dw);
    int *row1 = (int *)malloc(sizeof(int) * dw);
    linear_coeffs(dw, sw, 1, xofs, alpha);
    linear_coeffs(dh, sh, 0, yofs, beta); /* rows are clipped, beta is not zeroed */
    for (int dy = 0; dy < dh; ++dy) {
        int sy0 = yofs[dy], sy1 = yofs[dy] + 1;
        if (sy0 < 0) sy0 = 0; if (sy0 > sh - 1) sy0 = sh - 1;
        if (sy1 < 0) sy1 = 0; if (sy1 > sh - 1) sy1 = sh - 1;
        const uint8_t *S0 = src + (size_t)sy0 * sstride, *S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; ++dx) {
            int sx = xofs[dx];
            int a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
            if (sx >= sw - 1) { /* dx >= xmax: S[sx]*ONE */
                row0[dx] = S0[sx] * 2048;
                row1[dx] = S1[sx] * 2048;
            } else {
                row0[dx] = S0[sx] * a0 + S0[sx + 1] * a1;
                row1[dx] = S1[sx] * a0 + S1[sx + 1] * a1;
            }
        }
        int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; ++dx)
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(yofs); free(alpha); free(beta); free(row0); free(row1);
}

void orbref_pyramid(const orbref_cfg *cfg, const uint8_t *img, int w, int h, int stride,
                    uint8_t **levels)
{
    /* modules/ORB/ORBExtractor.cpp:559-570: level k resampled from level k-1,
     * sized from the level-0 dimensions, no border padding */
    int pw = w, ph = h;
    for (int l = 0; l < cfg->n_levels; ++l) {
        if (l == 0) {
            for (int y = 0; y < h; ++y) memcpy(levels[0] + (size_t)y * w, img + (size_t)y * stride, w);
        } else {
            int lw, lh;
            orbref_level_size(cfg, w, h, l, &lw, &lh);
            orbref_resize_linear(levels[l - 1], pw, ph, pw, levels[l], lw, lh, lw);
            pw = lw; ph = lh;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* cv::FAST TYPE_9_16 with NMS -- modules/ORB/ORBExtractor.cpp:601-607          */
/* OpenCV 4.2 features2d/fast.cpp FAST_t<16> + fast_score.cpp cornerScore<16>  */
/* ------------------------------------------------------------------------- */
static const int RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

static int fast_is_corner(const uint8_t *p, int stride, int threshold)
{
    /* "more than 8 contiguous" of the 25-long unrolled ring (fast.cpp, K=8, N=25) */
    int v = p[0];
    int vt_lo = v - threshold, vt_hi = v + threshold;
    /* quick reject, same idea as fast.cpp's tab[] ORs: an arc of 9 out of 16
     * contains one pixel of every opposite pair (pure optimisation) */
    {
        int a = p[RING_DY[0] * stride], b = p[RING_DY[8] * stride];
        int c = p[RING_DX[4]], e = p[RING_DX[12]];
        int dark = (a < vt_lo || b < vt_lo) && (c < vt_lo || e < vt_lo);
        int bright = (a > vt_hi || b > vt_hi) && (c > vt_hi || e > vt_hi);
        if (!dark && !bright) return 0;
    }
    int cd = 0, cb = 0;
    for (int k = 0; k < 25; ++k) {
        int x = p[RING_DX[k & 15] + RING_DY[k & 15] * stride];
        if (x < vt_lo) { if (++cd > 8) return 1; } else cd = 0;
        if (x > vt_hi) { if (++cb > 8) return 1; } else cb = 0;
    }
    return 0;
}

static int fast_corner_score(const uint8_t *p, int stride, int threshold)
{
    /* cornerScore<16> (fast_score.cpp); the early-outs do not change the value */
    short d[25];
    int v = p[0];
    for (int k = 0; k < 25; ++k) d[k] = (short)(v - p[RING_DX[k & 15] + RING_DY[k & 15] * stride]);
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        a = a < d[k + 3] ? a : d[k + 3];
        if (a <= a0) continue;
        for (int j = 4; j <= 8; ++j) a = a < d[k + j] ? a : d[k + j];
        int t = a < d[k] ? a : d[k];
        if (t > a0) a0 = t;
        t = a < d[k + 9] ? a : d[k + 9];
        if (t > a0) a0 = t;
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int j = 3; j <= 5; ++j) b = b > d[k + j] ? b : d[k + j];
        if (b >= b0) continue;
        for (int j = 6; j <= 8; ++j) b = b > d[k + j] ? b : d[k + j];
        int t = b > d[k] ? b : d[k];
        if (t < b0) b0 = t;
        t = b > d[k + 9] ? b : d[k + 9];
        if (t < b0) b0 = t;
    }
    return -b0 - 1;
}

int orbref_fast_strength(const uint8_t *img, int stride, int x, int y)
{
    /* threshold-free closed form used by the HIP path; tests check
     * it against fast_is_corner/fast_corner_score for every threshold. */
    const uint8_t *p = img + (size_t)y * stride + x;
    int v = p[0], d[16];
    for (int k = 0; k < 16; ++k) d[k] = v - p[RING_DX[k] + RING_DY[k] * stride];
    int best = INT_MIN;
    for (int s = 0; s < 16; ++s) {
        int mn = INT_MAX, mx = INT_MIN;
        for (int j = 0; j < 9; ++j) {
            int e = d[(s + j) & 15];
            if (e < mn) mn = e;
            if (e > mx) mx = e;
        }
        if (mn > best) best = mn;
        if (-mx > best) best = -mx;
    }
    return best - 1;
}

int orbref_fast_box(const uint8_t *img, int stride, int x0, int y0, int x1, int y1,
                    int threshold, orbref_cand *out, int cap)
{
    /* FAST_t evaluates rows/cols [3, dim-3) of the sub-image handed to it; the
     * caller passes the cell plus a 3-px apron, so the evaluated box is the cell.
     * Scores outside the box are 0 for the 3x3 strict NMS. */
    int bw = x1 - x0, bh = y1 - y0;
    if (bw <= 0 || bh <= 0) return 0;
    if (threshold < 0) threshold = 0;
    if (threshold > 255) threshold = 255;
    int sw = bw + 2;
    uint8_t *score = (uint8_t *)calloc((size_t)sw * (bh + 2), 1);
    uint8_t *corner = (uint8_t *)calloc((size_t)bw * bh, 1);
    for (int y = 0; y < bh; ++y)
        for (int x = 0; x < bw; ++x) {
            const uint8_t *p = img + (size_t)(y0 + y) * stride + (x0 + x);
            if (fast_is_corner(p, stride, threshold)) {
                corner[y * bw + x] = 1;
                score[(y + 1) * sw + (x + 1)] = (uint8_t)fast_corner_score(p, stride, threshold);
            }
        }
    int n = 0;
    for (int y = 0; y < bh; ++y)
        for (int x = 0; x < bw; ++x) {
            if (!corner[y * bw + x]) continue;
            const uint8_t *s = score + (y + 1) * sw + (x + 1);
            int sc = s[0];
            if (sc > s[-1] && sc > s[1] && sc > s[-sw - 1] && sc > s[-sw] && sc > s[-sw + 1] &&
                sc > s[sw - 1] && sc > s[sw] && sc > s[sw + 1]) {
                if (n < cap) { out[n].x = (float)(x0 + x); out[n].y = (float)(y0 + y); out[n].response = (float)sc; }
                ++n;
            }
        }
    free(corner);
    free(score);
    return n;
}

int orbref_level_candidates(const orbref_cfg *cfg, const uint8_t *img, int w, int h, int stride,
                            orbref_cand *out, int cap)
{
    /* modules/ORB/ORBExtractor.cpp:578-617 */
    const int W = ORBREF_CELL;
    const int minBX = ORBREF_EDGE, minBY = ORBREF_EDGE;
    const int maxBX = w - ORBREF_EDGE, maxBY = h - ORBREF_EDGE;
    const int width = maxBX - minBX, height = maxBY - minBY;
    if (width <= 0 || height <= 0) return 0;
    const int nCols = width % W == 0 ? width / W : width / W + 1;
    const int nRows = height % W == 0 ? height / W : height / W + 1;
    int n = 0;
    orbref_cand cell[ORBREF_CELL * ORBREF_CELL];
    for (int i = 0; i < nRows; ++i) {
        const int iniY = minBY + i * W;
        int maxY = iniY + W < maxBY ? iniY + W : maxBY;
        for (int j = 0; j < nCols; ++j) {
            const int iniX = minBX + j * W;
            int maxX = iniX + W < maxBX ? iniX + W : maxBX;
            int m = orbref_fast_box(img, stride, iniX, iniY, maxX, maxY, cfg->ini_th_fast, cell, W * W);
            if (m == 0) m = orbref_fast_box(img, stride, iniX, iniY, maxX, maxY, cfg->min_th_fast, cell, W * W);
            for (int k = 0; k < m; ++k) {
                if (n < cap) {
                    /* :611-612 -- shift to border-relative coordinates */
                    out[n].x = cell[k].x - (float)minBX;
                    out[n].y = cell[k].y - (float)minBY;
                    out[n].response = cell[k].response;
                }
                ++n;
            }
        }
    }
    return n;
}

/* ------------------------------------------------------------------------- */
/* DistributeOctree -- modules/ORB/ORBExtractor.cpp:640-830, DivideNode :367-413 */
/* std::list emulated with index links; node "address" tie-break canonicalised */
/* to creation sequence.                                       */
/* ------------------------------------------------------------------------- */
typedef struct qnode {
    int ulx, uly, brx, bry; /* UL and BR; UR.x == BR.x, BL.y == BR.y for every node */
    int *pts;               /* indices into the candidate array, in insertion order */
    int npts;
    int no_more;
    int prev, next;
    int alive;
} qnode;

typedef struct qlist {
    qnode *nodes;
    int n_alloc, cap;
    int head, tail, size;
} qlist;

static int ql_new(qlist *L)
{
    if (L->n_alloc == L->cap) {
        L->cap = L->cap ? L->cap * 2 : 64;
        L->nodes = (qnode *)realloc(L->nodes, sizeof(qnode) * L->cap);
    }
    qnode *q = &L->nodes[L->n_alloc];
    memset(q, 0, sizeof *q);
    q->prev = q->next = -1;
    return L->n_alloc++;
}
static void ql_push_back(qlist *L, int id)
{
    qnode *q = &L->nodes[id];
    q->alive = 1; q->prev = L->tail; q->next = -1;
    if (L->tail >= 0) L->nodes[L->tail].next = id; else L->head = id;
    L->tail = id; L->size++;
}
static void ql_push_front(qlist *L, int id)
{
    qnode *q = &L->nodes[id];
    q->alive = 1; q->next = L->head; q->prev = -1;
    if (L->head >= 0) L->nodes[L->head].prev = id; else L->tail = id;
    L->head = id; L->size++;
}
static int ql_erase(qlist *L, int id) /* returns the following node */
{
    qnode *q = &L->nodes[id];
    int nx = q->next;
    if (q->prev >= 0) L->nodes[q->prev].next = q->next; else L->head = q->next;
    if (q->next >= 0) L->nodes[q->next].prev = q->prev; else L->tail = q->prev;
    q->alive = 0; L->size--;
    free(q->pts); q->pts = NULL;
    return nx;
}

/* DivideNode: creates four detached children ids[0..3] (n1..n4) */
static void divide_node(qlist *L, int id, const orbref_cand *c, int ids[4])
{
    for (int k = 0; k < 4; ++k) ids[k] = ql_new(L);
    qnode *p = &L->nodes[id]; /* after ql_new: realloc may have moved the array */
    const int halfX = (p->brx - p->ulx) / 2;
    const int halfY = (p->bry - p->uly) / 2;
    const int midx = p->ulx + halfX, midy = p->uly + halfY;
    qnode *n1 = &L->nodes[ids[0]], *n2 = &L->nodes[ids[1]], *n3 = &L->nodes[ids[2]], *n4 = &L->nodes[ids[3]];
    n1->ulx = p->ulx; n1->uly = p->uly; n1->brx = midx;   n1->bry = midy;
    n2->ulx = midx;   n2->uly = p->uly; n2->brx = p->brx; n2->bry = midy;
    n3->ulx = p->ulx; n3->uly = midy;   n3->brx = midx;   n3->bry = p->bry;
    n4->ulx = midx;   n4->uly = midy;   n4->brx = p->brx; n4->bry = p->bry;
    for (int k = 0; k < 4; ++k) L->nodes[ids[k]].pts = (int *)malloc(sizeof(int) * (p->npts ? p->npts : 1));
    for (int i = 0; i < p->npts; ++i) {
        int ci = p->pts[i];
        qnode *dst;
        if (orbref_floor_f(c[ci].x) < midx) dst = orbref_floor_f(c[ci].y) < midy ? n1 : n3;
        else dst = orbref_floor_f(c[ci].y) < midy ? n2 : n4;
        dst->pts[dst->npts++] = ci;
    }
    for (int k = 0; k < 4; ++k) if (L->nodes[ids[k]].npts == 1) L->nodes[ids[k]].no_more = 1;
}

typedef struct szptr { int size; int id; } szptr;
static int szptr_cmp(const void *a, const void *b)
{
    const szptr *x = (const szptr *)a, *y = (const szptr *)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    return x->id < y->id ? -1 : x->id > y->id; /* creation order stands in for the heap address */
}

int orbref_distribute_octree(const orbref_cand *c, int n, int minX, int maxX, int minY,
                             int maxY, int n_features, orbref_cand *out, int cap)
{
    qlist L; memset(&L, 0, sizeof L); L.head = L.tail = -1;
    const int nIni = orbref_ceil_f((float)(maxX - minX) / (float)(maxY - minY));
    const int hX = orbref_ceil_f((float)(maxX - minX) / (float)nIni);
    int *ini = (int *)malloc(sizeof(int) * (nIni > 0 ? nIni : 1));
    for (int i = 0; i < nIni; ++i) {
        int id = ql_new(&L);
        qnode *q = &L.nodes[id];
        q->ulx = hX * i; q->uly = 0;
        q->brx = (i == nIni - 1) ? maxX : hX * (i + 1); /* :665,:667 absolute maxX for the last node */
        q->bry = maxY - minY;
        q->pts = (int *)malloc(sizeof(int) * (n ? n : 1));
        ql_push_back(&L, id);
        ini[i] = id;
    }
    for (int i = 0; i < n; ++i) {
        qnode *q = &L.nodes[ini[orbref_floor_f(c[i].x) / hX]];
        q->pts[q->npts++] = i;
    }
    for (int it = L.head; it >= 0;) {
        qnode *q = &L.nodes[it];
        if (q->npts == 1) { q->no_more = 1; it = q->next; }
        else if (q->npts == 0) it = ql_erase(&L, it);
        else it = q->next;
    }

    int finish = 0;
    szptr *vec = NULL, *prevvec = NULL;
    int nvec = 0, capvec = 0, capprev = 0;
#define VEC_PUSH(sz, idd) do { if (nvec == capvec) { capvec = capvec ? capvec * 2 : 256; \
        vec = (szptr *)realloc(vec, sizeof(szptr) * capvec); } vec[nvec].size = (sz); vec[nvec].id = (idd); ++nvec; } while (0)
#define ADD_CHILDREN(ids) do { for (int k_ = 0; k_ < 4; ++k_) { int cid_ = (ids)[k_]; \
        if (L.nodes[cid_].npts > 0) { ql_push_front(&L, cid_); \
            if (L.nodes[cid_].npts > 1) { nToExpand++; VEC_PUSH(L.nodes[cid_].npts, cid_); } } \
        else { free(L.nodes[cid_].pts); L.nodes[cid_].pts = NULL; } } } while (0)

    while (!finish) {
        int preSize = L.size;
        int nToExpand = 0;
        nvec = 0;
        for (int it = L.head; it >= 0;) {
            if (L.nodes[it].no_more) { it = L.nodes[it].next; continue; }
            int ids[4];
            divide_node(&L, it, c, ids);
            ADD_CHILDREN(ids);
            it = ql_erase(&L, it);
        }
        if (L.size > n_features || L.size == preSize) {
            finish = 1;
        } else if (L.size + nToExpand * 3 > n_features) {
            while (!finish) {
                preSize = L.size;
                if (nvec > capprev) { capprev = nvec; prevvec = (szptr *)realloc(prevvec, sizeof(szptr) * capprev); }
                if (nvec) memcpy(prevvec, vec, sizeof(szptr) * nvec);
                int nprev = nvec;
                nvec = 0;
                qsort(prevvec, nprev, sizeof(szptr), szptr_cmp);
                for (int j = 0; j < nprev; ++j) {
                    int ids[4];
                    divide_node(&L, prevvec[j].id, c, ids);
                    ADD_CHILDREN(ids);
                    ql_erase(&L, prevvec[j].id);
                    if (L.size >= n_features) break;
                }
                if (L.size >= n_features || L.size == preSize) finish = 1;
            }
        }
    }
#undef VEC_PUSH
#undef ADD_CHILDREN

    /* :812-827 keep the strongest point per node, first wins on ties */
    int m = 0;
    for (int it = L.head; it >= 0; it = L.nodes[it].next) {
        qnode *q = &L.nodes[it];
        int best = q->pts[0];
        float bestR = c[best].response;
        for (int k = 1; k < q->npts; ++k)
            if (c[q->pts[k]].response > bestR) { best = q->pts[k]; bestR = c[best].response; }
        if (m < cap) out[m] = c[best];
        ++m;
    }
    for (int i = 0; i < L.n_alloc; ++i) free(L.nodes[i].pts);
    free(L.nodes); free(vec); free(prevvec); free(ini);
    return m;
}

/* ------------------------------------------------------------------------- */
/* Orientation -- modules/ORB/ORBExtractor.cpp:18-42 + cv::fastAtan2 (B.4)      */
/* ------------------------------------------------------------------------- */
float orbref_fast_atan2(float y, float x)
{
    /* OpenCV 4.2 core/mathfuncs_core.simd.hpp atan_f32 (scalar), no FMA */
    static const float rad2deg = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * rad2deg;
    const float p3 = -0.3258083974640975f * rad2deg;
    const float p5 = 0.1555786518463281f * rad2deg;
    const float p7 = -0.04432655554792128f * rad2deg;
    const float eps = (float)2.2204460492503131e-16; /* (float)DBL_EPSILON */
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

float orbref_ic_angle(const orbref_cfg *cfg, const uint8_t *img, int stride, int x, int y)
{
    int m_01 = 0, m_10 = 0;
    const uint8_t *center = img + (size_t)y * stride + x;
    for (int u = -ORBREF_HALF_PATCH; u <= ORBREF_HALF_PATCH; ++u) m_10 += u * center[u];
    for (int v = 1; v <= ORBREF_HALF_PATCH; ++v) {
        int v_sum = 0;
        int d = cfg->u_max[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orbref_fast_atan2((float)m_01, (float)m_10);
}

/* ------------------------------------------------------------------------- */
/* cv::GaussianBlur 7x7 sigma 2 REFLECT_101, 8U fixed point -- :527-528 (B.3)   */
/* ------------------------------------------------------------------------- */
static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

void orbref_gaussian_blur7(const orbref_cfg *cfg, const uint8_t *src, int w, int h, int sstride,
                           uint8_t *dst, int dstride)
{
    const int *k = cfg->blur_taps;
    uint16_t *hbuf = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)w * h);
    for (int y = 0; y < h; ++y) {
        const uint8_t *S = src + (size_t)y * sstride;
        for (int x = 0; x < w; ++x) {
            uint32_t acc = 0;
            for (int i = 0; i < 7; ++i) acc += (uint32_t)k[i] * S[reflect101(x + i - 3, w)];
            hbuf[(size_t)y * w + x] = (uint16_t)(acc > 65535u ? 65535u : acc); /* ufixedpoint16 saturates */
        }
    }
    for (int y = 0; y < h; ++y) {
        uint8_t *D = dst + (size_t)y * dstride;
        for (int x = 0; x < w; ++x) {
            uint32_t acc = 0;
            for (int i = 0; i < 7; ++i) acc += (uint32_t)k[i] * hbuf[(size_t)reflect101(y + i - 3, h) * w + x];
            uint32_t r = (acc + (1u << 15)) >> 16;
            D[x] = (uint8_t)(r > 255u ? 255u : r);
        }
    }
    free(hbuf);
}

/* ------------------------------------------------------------------------- */
/* rBRIEF -- modules/ORB/ORBExtractor.cpp:50-97                                 */
/* ------------------------------------------------------------------------- */
/* cos(angle), sin(angle) on a float (ORBExtractor.cpp:54; `using namespace std` at :11 makes these
 * std::cos(float) / std::sin(float) = glibc cosf / sinf).  glibc is a third-party dependency outside
 * /root/reference; this restates the published algorithm of glibc >= 2.28
 * (sysdeps/ieee754/flt-32/s_sincosf.h, s_sinf.c, s_cosf.c, s_sincosf_data.c), pinned by the
 * container's glibc 2.35: orbref_sincosf_check_libm() below compares it with the host libm and
 * tests/test_oracle_kat.py runs that over every float in [0, 2*pi] (0 mismatches for sinf, cosf and
 * sincosf; on that range the x86-64 FMA and SSE2 ifunc variants give the same floats, so the plain
 * multiply-add form is restated).  Only the branches reachable for |y| < 120 are restated; the
 * descriptor angle is in [0, 360) degrees. */
static float sincosf_poly(double x, double x2, int negated, int odd)
{
    /* __sincosf_table[0]; table[1] (used when n & 2) is the same with c0..c4 negated */
    const double sg = negated ? -1.0 : 1.0;
    const double c0 = sg * 0x1p0, c1 = sg * -0x1.ffffffd0c621cp-2, c2 = sg * 0x1.55553e1068f19p-5,
                 c3 = sg * -0x1.6c087e89a359dp-10, c4 = sg * 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    if (!odd) {                         /* sinf_poly, (n & 1) == 0 */
        double x3 = x * x2;
        double t1 = s2 + x2 * s3;
        double x7 = x3 * x2;
        double t = x + x3 * s1;
        return (float)(t + x7 * t1);
    } else {
        double x4 = x2 * x2;
        double t2 = c3 + x2 * c4;
        double t1 = c0 + x2 * c1;
        double x6 = x4 * x2;
        double t = t1 + x4 * c2;
        return (float)(t + x6 * t2);
    }
}
static uint32_t abstop12(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    return (u >> 20) & 0x7ff;
}
void orbref_sincosf(float y, float *sn, float *cs)
{
    double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) {
            *sn = y;
            *cs = 1.0f;
            return;
        }
        double x2 = x * x;
        *sn = sincosf_poly(x, x2, 0, 0);
        *cs = sincosf_poly(x, x2, 0, 1);
        return;
    }
    /* reduce_fast, !TOINT_INTRINSICS: hpi_inv prescaled by 2^24 */
    double r = x * 0x1.45F306DC9C883p+23;
    int n = ((int32_t)r + 0x800000) >> 24;
    x = x - n * 0x1.921FB54442D18p0;
    static const double sign[4] = {1.0, -1.0, -1.0, 1.0};
    double sg = sign[n & 3];
    *sn = sincosf_poly(x * sg, x * x, (n & 2) != 0, n & 1);
    *cs = sincosf_poly(x * sg, x * x, (n & 2) != 0, (n ^ 1) & 1);
}

/* Pins orbref_sincosf against the libm this library is linked with: the float bit patterns
 * first, first + step, ... <= last are compared with sinf, cosf and sincosf.  Returns the number of
 * inputs on which any of the four values differs.  Test infrastructure. */
extern void sincosf(float, float *, float *);
long orbref_sincosf_check_libm(uint32_t first, uint32_t last, uint32_t step)
{
    long bad = 0;
    if (step == 0) step = 1;
    for (uint64_t u = first; u <= last; u += step) {
        uint32_t b = (uint32_t)u;
        float y, s, c, s2, c2;
        memcpy(&y, &b, 4);
        orbref_sincosf(y, &s, &c);
        volatile float yy = y;          /* keeps the compiler from folding the libm calls */
        float ls = sinf(yy), lc = cosf(yy);
        sincosf(yy, &s2, &c2);
        if (memcmp(&s, &ls, 4) || memcmp(&c, &lc, 4) || memcmp(&s2, &ls, 4) || memcmp(&c2, &lc, 4)) ++bad;
    }
    return bad;
}

void orbref_sincos_deg(float angle_deg, float *c, float *s)
{
    const float factorPI = (float)(3.14159265358979323846 / 180.f); /* :16 (double/float -> double, cast) */
    float a = angle_deg * factorPI;                                 /* :53 */
    orbref_sincosf(a, s, c);                                        /* :54 */
}

/* n angles at once (test convenience): out = n (cos, sin) pairs */
void orbref_sincos_deg_n(const float *angle_deg, int n, float *out)
{
    for (int i = 0; i < n; ++i) orbref_sincos_deg(angle_deg[i], &out[2 * i], &out[2 * i + 1]);
}

void orbref_brief(const uint8_t *blur, int stride, int x, int y, float angle_deg, uint8_t *desc)
{
    float a, b;
    orbref_sincos_deg(angle_deg, &a, &b);
    const uint8_t *center = blur + (size_t)y * stride + x;
    const int8_t *pat = g_pattern;
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int j = 0; j < 8; ++j) {
            float x0 = (float)pat[4 * j], y0 = (float)pat[4 * j + 1];
            float x1 = (float)pat[4 * j + 2], y1 = (float)pat[4 * j + 3];
            int t0 = center[orbref_round_f(x0 * b + y0 * a) * stride + orbref_round_f(x0 * a - y0 * b)];
            int t1 = center[orbref_round_f(x1 * b + y1 * a) * stride + orbref_round_f(x1 * a - y1 * b)];
            val |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ------------------------------------------------------------------------- */
/* operator() -- modules/ORB/ORBExtractor.cpp:495-547, :572-638                 */
/* ------------------------------------------------------------------------- */
int orbref_extract(const orbref_cfg *cfg, const uint8_t *img, int w, int h, int stride,
                   orbref_kp *kps, uint8_t *desc, int cap, int *per_level_counts)
{
    if (!img || w <= 0 || h <= 0) return 0; /* :497 */
    const int L = cfg->n_levels;
    uint8_t *levels[ORBREF_MAX_LEVELS];
    int lw[ORBREF_MAX_LEVELS], lh[ORBREF_MAX_LEVELS];
    for (int l = 0; l < L; ++l) {
        orbref_level_size(cfg, w, h, l, &lw[l], &lh[l]);
        levels[l] = (uint8_t *)malloc((size_t)lw[l] * lh[l] + 1);
    }
    orbref_pyramid(cfg, img, w, h, stride, levels);

    orbref_cand **sel = (orbref_cand **)calloc(L, sizeof *sel);
    float **ang = (float **)calloc(L, sizeof *ang);
    int nsel[ORBREF_MAX_LEVELS];
    int total = 0;
    for (int l = 0; l < L; ++l) {
        nsel[l] = 0;
        int maxc = (lw[l] > 2 * ORBREF_EDGE && lh[l] > 2 * ORBREF_EDGE)
                       ? (lw[l] - 2 * ORBREF_EDGE) * (lh[l] - 2 * ORBREF_EDGE) : 0;
        if (maxc == 0) continue;
        orbref_cand *cand = (orbref_cand *)malloc(sizeof(orbref_cand) * maxc);
        int nc = orbref_level_candidates(cfg, levels[l], lw[l], lh[l], lw[l], cand, maxc);
        sel[l] = (orbref_cand *)malloc(sizeof(orbref_cand) * (nc ? nc : 1));
        nsel[l] = orbref_distribute_octree(cand, nc, ORBREF_EDGE, lw[l] - ORBREF_EDGE, ORBREF_EDGE,
                                           lh[l] - ORBREF_EDGE, cfg->n_features_per_level[l], sel[l], nc);
        free(cand);
        ang[l] = (float *)malloc(sizeof(float) * (nsel[l] ? nsel[l] : 1));
        for (int k = 0; k < nsel[l]; ++k) {
            sel[l][k].x += ORBREF_EDGE; /* :627-628 */
            sel[l][k].y += ORBREF_EDGE;
            /* :636-637 orientation on the un-blurred level */
            ang[l][k] = orbref_ic_angle(cfg, levels[l], lw[l], orbref_round_f(sel[l][k].x), orbref_round_f(sel[l][k].y));
        }
        total += nsel[l];
    }
    if (per_level_counts) for (int l = 0; l < L; ++l) per_level_counts[l] = nsel[l];

    int ret = total;
    if (total > cap) ret = -1;
    if (total > 0 && total <= cap) {
        int off = 0;
        for (int l = 0; l < L; ++l) {
            if (nsel[l] == 0) continue;
            uint8_t *blur = (uint8_t *)malloc((size_t)lw[l] * lh[l]);
            orbref_gaussian_blur7(cfg, levels[l], lw[l], lh[l], lw[l], blur, lw[l]);
            float scale = cfg->scale_factors[l];
            for (int k = 0; k < nsel[l]; ++k) {
                orbref_kp *kp = &kps[off + k];
                float px = sel[l][k].x, py = sel[l][k].y;
                orbref_brief(blur, lw[l], orbref_round_f(px), orbref_round_f(py), ang[l][k], desc + (size_t)(off + k) * 32);
                if (l != 0) { px *= scale; py *= scale; } /* :537-542 */
                kp->x = px; kp->y = py;
                kp->size = cfg->scale_factors[l]; /* :631 */
                kp->angle = ang[l][k];
                kp->response = sel[l][k].response;
                kp->octave = l;
                kp->class_id = -1;
            }
            off += nsel[l];
            free(blur);
        }
    }
    for (int l = 0; l < L; ++l) { free(levels[l]); free(sel[l]); free(ang[l]); }
    free(sel); free(ang);
    return ret;
}

/* ------------------------------------------------------------------------- */
/* Matcher -- modules/ORB/ORBMatcher.cpp                                        */
/* ------------------------------------------------------------------------- */
#define TH_LOW 50
#define TH_HIGH 100
#define HISTO_LENGTH 30

int orbref_hamming(const uint8_t *a, const uint8_t *b)
{
    /* :17-31, the bit-twiddling popcount over 8 x u32 */
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

void orbref_three_maxima(const int *hist_sizes, int n_bins, int *ind1, int *ind2, int *ind3)
{
    /* :594-622; callers pre-set ind1=ind2=ind3=-1 */
    int max1 = 0, max2 = -1, max3 = -2;
    for (int i = 0; i < n_bins; ++i) {
        const int n = hist_sizes[i];
        if (n > max1) { max3 = max2; max2 = max1; max1 = n; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (n > max2) { max3 = max2; max2 = n; *ind3 = *ind2; *ind2 = i; }
        else if (n > max3) { max3 = n; *ind3 = i; }
    }
    if (max2 < max1 / 10) { *ind2 = -1; *ind3 = -1; }
    else if (max3 < max1 / 10) { *ind3 = -1; }
}

typedef struct rothist { int *items[HISTO_LENGTH]; int n[HISTO_LENGTH]; int cap[HISTO_LENGTH]; } rothist;
static void rh_init(rothist *h) { memset(h, 0, sizeof *h); }
static void rh_push(rothist *h, int bin, int v)
{
    if (h->n[bin] == h->cap[bin]) {
        h->cap[bin] = h->cap[bin] ? h->cap[bin] * 2 : 64;
        h->items[bin] = (int *)realloc(h->items[bin], sizeof(int) * h->cap[bin]);
    }
    h->items[bin][h->n[bin]++] = v;
}
static void rh_free(rothist *h) { for (int i = 0; i < HISTO_LENGTH; ++i) free(h->items[i]); }
static int rot_bin(float a1, float a2)
{
    /* factor = 1.f/HISTO_LENGTH (bins 0..12 only) */
    const float factor = 1.f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0) rot += 360.f;
    int bin = orbref_round_f(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

static int fv_lower_bound(const orbref_fv *fv, uint32_t key)
{
    int lo = 0, hi = fv->n_nodes;
    while (lo < hi) { int mid = (lo + hi) / 2; if (fv->node_ids[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

int orbref_search_by_bow(float nn_ratio, int check_orientation,
                         const uint8_t *desc1, const float *angle1, const uint8_t *kf_mp_ok, int n1,
                         const orbref_fv *fv1,
                         const uint8_t *desc2, const float *angle2, int32_t *frame_mp, int n2,
                         const orbref_fv *fv2)
{
    (void)n1; (void)n2;
    int numMatch = 0;
    rothist rh; rh_init(&rh);
    int i1 = 0, i2 = 0;
    while (i1 < fv1->n_nodes && i2 < fv2->n_nodes) {
        if (fv1->node_ids[i1] == fv2->node_ids[i2]) {
            for (int a = fv1->offsets[i1]; a < fv1->offsets[i1 + 1]; ++a) {
                const int idx1 = (int)fv1->indices[a];
                if (!kf_mp_ok[idx1]) continue; /* :143 */
                int bestDist = 256, secondDist = 256, bestIdx2 = -1;
                for (int b = fv2->offsets[i2]; b < fv2->offsets[i2 + 1]; ++b) {
                    const int idx2 = (int)fv2->indices[b];
                    if (frame_mp[idx2] != -1) continue; /* :150 */
                    const int dist = orbref_hamming(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
                    if (dist < bestDist) { secondDist = bestDist; bestDist = dist; bestIdx2 = idx2; }
                    else if (dist < secondDist) secondDist = dist;
                }
                if (bestDist <= TH_LOW && (float)bestDist < nn_ratio * (float)secondDist) { /* :164 */
                    frame_mp[bestIdx2] = idx1;
                    numMatch++;
                    if (check_orientation) rh_push(&rh, rot_bin(angle1[idx1], angle2[bestIdx2]), bestIdx2);
                }
            }
            i1++; i2++;
        } else if (fv1->node_ids[i1] < fv2->node_ids[i2]) {
            i1 = fv_lower_bound(fv1, fv2->node_ids[i2]);
        } else {
            i2 = fv_lower_bound(fv2, fv1->node_ids[i1]);
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orbref_three_maxima(rh.n, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int k = 0; k < rh.n[i]; ++k) { frame_mp[rh.items[i][k]] = -1; numMatch--; }
        }
    }
    rh_free(&rh);
    return numMatch;
}

int orbref_search_for_triangulation(int check_orientation,
                                    const uint8_t *desc1, const float *angle1, const uint8_t *has_mp1, int n1,
                                    const orbref_fv *fv1,
                                    const uint8_t *desc2, const float *angle2, const uint8_t *has_mp2, int n2,
                                    const orbref_fv *fv2, int32_t *matches12)
{
    int numMatch = 0;
    uint8_t *matched2 = (uint8_t *)calloc(n2 ? n2 : 1, 1);
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    rothist rh; rh_init(&rh);
    int i1 = 0, i2 = 0;
    while (i1 < fv1->n_nodes && i2 < fv2->n_nodes) {
        if (fv1->node_ids[i1] == fv2->node_ids[i2]) {
            for (int a = fv1->offsets[i1]; a < fv1->offsets[i1 + 1]; ++a) {
                const int idx1 = (int)fv1->indices[a];
                if (has_mp1[idx1]) continue; /* :452 */
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int b = fv2->offsets[i2]; b < fv2->offsets[i2 + 1]; ++b) {
                    const int idx2 = (int)fv2->indices[b];
                    if (matched2[idx2] || has_mp2[idx2]) continue; /* :466 */
                    const int dist = orbref_hamming(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
                    if (dist < bestDist) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 > 0) { /* :484 -- index 0 is never accepted */
                    matches12[idx1] = bestIdx2;
                    matched2[bestIdx2] = 1;
                    numMatch++;
                    if (check_orientation) rh_push(&rh, rot_bin(angle1[idx1], angle2[bestIdx2]), idx1);
                }
            }
            i1++; i2++;
        } else if (fv1->node_ids[i1] < fv2->node_ids[i2]) {
            i1 = fv_lower_bound(fv1, fv2->node_ids[i2]);
        } else {
            i2 = fv_lower_bound(fv2, fv1->node_ids[i1]);
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orbref_three_maxima(rh.n, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int k = 0; k < rh.n[i]; ++k) { matches12[rh.items[i][k]] = -1; numMatch--; }
        }
    }
    rh_free(&rh);
    free(matched2);
    return numMatch;
}

/* ---- Frame post-processing: Frame.cpp:24-28, Pinhole.cpp:55-83, Fisheye.cpp:110-117 ---- */
void orbref_undistort_point(const orbref_camera *cam, float u_f, float v_f, float *xu, float *yu)
{
    double k[12] = {0};
    for (int i = 0; i < cam->n_dist && i < 12; ++i) k[i] = (double)cam->dist[i];
    const double fx = cam->fx, fy = cam->fy, cx = cam->cx, cy = cam->cy, ifx = 1. / fx, ify = 1. / fy;
    const double u = u_f, v = v_f;
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    /* RR = P(:, 0:3) * I with P = K: the zero entries of K still take part in the sums */
    const double xx = fx * x + 0. * y + cx, yy = 0. * x + fy * y + cy, ww = 1. / (0. * x + 0. * y + 1.);
    *xu = (float)(xx * ww);
    *yu = (float)(yy * ww);
}

void orbref_frame_post(const orbref_camera *cam, orbref_kp *raw, int n, orbref_kp *un)
{
    for (int i = 0; i < n; ++i) {
        if (cam->size_scale) raw[i].size *= cam->size_scale[(size_t)(int)raw[i].y * cam->width + (int)raw[i].x];
        un[i] = raw[i];
        if (cam->undistort && cam->n_dist > 0 && cam->dist[0] != 0.f)
            orbref_undistort_point(cam, raw[i].x, raw[i].y, &un[i].x, &un[i].y);
    }
}

/* ---- DBoW2 vocabulary transform (TemplatedVocabulary.h:1127-1259) ---- */
struct orbref_voc {
    int k, L, scoring, weighting, n_nodes, n_words;
    int32_t *child_start; /* n_nodes + 1 */
    int32_t *child;       /* n_nodes - 1, children of node p in ascending id */
    uint32_t *word_id;
    double *weight;
    uint8_t *desc;
};

orbref_voc *orbref_voc_build(int k, int L, int scoring, int weighting, int n_nodes, const int32_t *parent,
                             const uint8_t *is_leaf, const uint8_t *desc, const double *weight)
{
    orbref_voc *v = (orbref_voc *)calloc(1, sizeof *v);
    v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting; v->n_nodes = n_nodes;
    v->child_start = (int32_t *)calloc((size_t)n_nodes + 1, 4);
    v->child = (int32_t *)malloc(4 * (size_t)(n_nodes > 1 ? n_nodes - 1 : 1));
    v->word_id = (uint32_t *)calloc((size_t)n_nodes, 4); /* Node(): word_id(0), weight(0) */
    v->weight = (double *)calloc((size_t)n_nodes, 8);
    v->desc = (uint8_t *)calloc((size_t)n_nodes, 32);
    for (int i = 1; i < n_nodes; ++i) v->child_start[parent[i] + 1]++;
    for (int i = 0; i < n_nodes; ++i) v->child_start[i + 1] += v->child_start[i];
    int32_t *fill = (int32_t *)calloc((size_t)n_nodes, 4);
    for (int i = 1; i < n_nodes; ++i) { /* :1393 push_back in file order */
        v->child[v->child_start[parent[i]] + fill[parent[i]]++] = i;
        memcpy(v->desc + 32 * (size_t)i, desc + 32 * (size_t)i, 32);
        v->weight[i] = weight[i];
        if (is_leaf[i]) v->word_id[i] = (uint32_t)v->n_words++; /* :1408-1414 */
    }
    free(fill);
    return v;
}
void orbref_voc_free(orbref_voc *v)
{
    if (!v) return;
    free(v->child_start); free(v->child); free(v->word_id); free(v->weight); free(v->desc); free(v);
}
int orbref_voc_n_words(const orbref_voc *v) { return v->n_words; }

void orbref_voc_transform_feature(const orbref_voc *v, const uint8_t *f, int levelsup, uint32_t *word, double *weight,
                                  uint32_t *nid)
{
    const int nid_level = v->L - levelsup;
    if (nid_level <= 0 && nid) *nid = 0; /* :1228 */
    int final_id = 0, current_level = 0;
    do {
        ++current_level;
        const int b = v->child_start[final_id], e = v->child_start[final_id + 1];
        final_id = v->child[b];
        double best_d = (double)orbref_hamming(f, v->desc + 32 * (size_t)final_id);
        for (int c = b + 1; c < e; ++c) {
            const int id = v->child[c];
            const double d = (double)orbref_hamming(f, v->desc + 32 * (size_t)id);
            if (d < best_d) { best_d = d; final_id = id; }
        }
        if (nid && current_level == nid_level) *nid = (uint32_t)final_id;
    } while (v->child_start[final_id] != v->child_start[final_id + 1]); /* !isLeaf() */
    *word = v->word_id[final_id];
    *weight = v->weight[final_id];
}

void orbref_voc_transform(const orbref_voc *v, const uint8_t *desc, int n, int levelsup, uint32_t *bow_ids,
                          double *bow_vals, int *n_words_out, uint32_t *fv_nodes, int32_t *fv_off, uint32_t *fv_idx,
                          int *n_fv)
{
    *n_words_out = 0; *n_fv = 0; fv_off[0] = 0;
    if (v->n_words == 0) return; /* empty() (:1133) */
    /* the two std::maps as sorted arrays; FeatureVector values as per-node growing lists */
    int nb = 0, nf = 0;
    uint32_t **lists = (uint32_t **)calloc((size_t)(n ? n : 1), sizeof *lists);
    int *len = (int *)calloc((size_t)(n ? n : 1), sizeof *len);
    const int accumulate = v->weighting == 0 || v->weighting == 1; /* TF_IDF, TF (:1142) vs IDF, BINARY (:1171) */
    for (int i = 0; i < n; ++i) {
        uint32_t id = 0, nid = 0xFFFFFFFFu; /* nid is uninitialised in the reference; marked here */
        double w = 0;
        orbref_voc_transform_feature(v, desc + 32 * (size_t)i, levelsup, &id, &w, &nid);
        if (!(w > 0)) continue; /* stopped word (:1157) */
        int p = 0;
        while (p < nb && bow_ids[p] < id) ++p; /* lower_bound */
        if (p < nb && bow_ids[p] == id) {
            if (accumulate) bow_vals[p] += w; /* addWeight; addIfNotExist keeps the first */
        } else {
            memmove(bow_ids + p + 1, bow_ids + p, 4 * (size_t)(nb - p));
            memmove(bow_vals + p + 1, bow_vals + p, 8 * (size_t)(nb - p));
            bow_ids[p] = id; bow_vals[p] = w; ++nb;
        }
        p = 0;
        while (p < nf && fv_nodes[p] < nid) ++p;
        if (!(p < nf && fv_nodes[p] == nid)) {
            memmove(fv_nodes + p + 1, fv_nodes + p, 4 * (size_t)(nf - p));
            memmove(lists + p + 1, lists + p, sizeof *lists * (size_t)(nf - p));
            memmove(len + p + 1, len + p, sizeof *len * (size_t)(nf - p));
            fv_nodes[p] = nid; lists[p] = (uint32_t *)malloc(4 * (size_t)n); len[p] = 0; ++nf;
        }
        lists[p][len[p]++] = (uint32_t)i;
    }
    const int must = v->scoring != 5;          /* DotProductScoring is the only one that does not normalise */
    const int l2 = v->scoring == 1;            /* L2Scoring; every other normalising scorer uses L1 */
    if (accumulate && nb > 0 && !must) {       /* :1164-1170 */
        const double nd = (double)nb;
        for (int p = 0; p < nb; ++p) bow_vals[p] /= nd;
    }
    if (must) {                                /* BowVector::normalize (BowVector.cpp:62-90) */
        double norm = 0.0;
        if (!l2) for (int p = 0; p < nb; ++p) norm += fabs(bow_vals[p]);
        else { for (int p = 0; p < nb; ++p) norm += bow_vals[p] * bow_vals[p]; norm = sqrt(norm); }
        if (norm > 0.0) for (int p = 0; p < nb; ++p) bow_vals[p] /= norm;
    }
    int t = 0;
    for (int p = 0; p < nf; ++p) {
        fv_off[p] = t;
        memcpy(fv_idx + t, lists[p], 4 * (size_t)len[p]);
        t += len[p];
        free(lists[p]);
    }
    fv_off[nf] = t;
    free(lists); free(len);
    *n_words_out = nb; *n_fv = nf;
}

/* ---- MapPoint::computeDescriptor: modules/BasicObject/MapPoint.cpp:103-152 ---- */
static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }
int orbref_distinctive_descriptor(const uint8_t *desc, int n)
{
    if (n <= 0) return -1; /* :122 */
    int *dist = (int *)malloc(sizeof(int) * (size_t)n * n), *row = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        dist[i * n + i] = 0;
        for (int j = i + 1; j < n; ++j)
            dist[i * n + j] = dist[j * n + i] = orbref_hamming(desc + 32 * (size_t)i, desc + 32 * (size_t)j);
    }
    int bestMedian = 256, bestIdx = 0;
    for (int i = 0; i < n; ++i) {
        memcpy(row, dist + i * n, sizeof(int) * (size_t)n);
        qsort(row, (size_t)n, sizeof(int), cmp_int);
        const int median = row[(n - 1) / 2];
        if (median < bestMedian) { bestMedian = median; bestIdx = i; }
    }
    free(dist); free(row);
    return bestIdx;
}

/* ---- dense best / second-best: the inner loop of modules/ORB/ORBMatcher.cpp:148-162 ---- */
void orbref_best2(const uint8_t *a, int na, const uint8_t *b, int nb, int32_t *best_idx, uint16_t *best,
                  uint16_t *second)
{
    for (int i = 0; i < na; ++i) {
        int b1 = 256, b2 = 256, bi = -1;
        for (int j = 0; j < nb; ++j) {
            const int d = orbref_hamming(a + 32 * (size_t)i, b + 32 * (size_t)j);
            if (d < b1) { b2 = b1; b1 = d; bi = j; }
            else if (d < b2) b2 = d;
        }
        best_idx[i] = bi; best[i] = (uint16_t)b1; second[i] = (uint16_t)b2;
    }
}

/* ---- Frame grid: modules/BasicObject/Frame.cpp:33-51, :90-127 (GRID_SIZE 40) ---- */
#define GRID_SIZE 40
orbref_grid *orbref_grid_build(const orbref_kp *kps, int n, int img_w, int img_h)
{
    orbref_grid *g = (orbref_grid *)calloc(1, sizeof *g);
    g->img_w = img_w; g->img_h = img_h;
    g->cols = img_w % GRID_SIZE == 0 ? img_w / GRID_SIZE : img_w / GRID_SIZE + 1;
    g->rows = img_h % GRID_SIZE == 0 ? img_h / GRID_SIZE : img_h / GRID_SIZE + 1;
    int nc = g->cols * g->rows;
    g->cell_start = (int32_t *)calloc(nc + 1, sizeof(int32_t));
    g->cell_items = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    int *cell_of = (int *)malloc(sizeof(int) * (n ? n : 1));
    for (int i = 0; i < n; ++i) {
        int x = orbref_floor_f(kps[i].x), y = orbref_floor_f(kps[i].y);
        if (x < 0 || x >= img_w || y < 0 || y >= img_h) { cell_of[i] = -1; continue; }
        cell_of[i] = (x / GRID_SIZE) * g->rows + (y / GRID_SIZE);
        g->cell_start[cell_of[i] + 1]++;
    }
    for (int c = 0; c < nc; ++c) g->cell_start[c + 1] += g->cell_start[c];
    int *fill = (int *)calloc(nc ? nc : 1, sizeof(int));
    for (int i = 0; i < n; ++i) if (cell_of[i] >= 0) g->cell_items[g->cell_start[cell_of[i]] + fill[cell_of[i]]++] = i;
    free(fill); free(cell_of);
    return g;
}
void orbref_grid_free(orbref_grid *g) { if (!g) return; free(g->cell_start); free(g->cell_items); free(g); }

int orbref_features_in_area(const orbref_grid *g, const orbref_kp *kps, float x, float y, float r,
                            int min_level, int max_level, int32_t *out, int cap)
{
    int minCellX = orbref_floor_f(x - r) / GRID_SIZE; if (minCellX < 0) minCellX = 0;
    int maxCellX = orbref_floor_f(x + r) / GRID_SIZE; if (maxCellX > g->cols - 1) maxCellX = g->cols - 1;
    if (minCellX > maxCellX) return 0;
    int minCellY = orbref_floor_f(y - r) / GRID_SIZE; if (minCellY < 0) minCellY = 0;
    int maxCellY = orbref_floor_f(y + r) / GRID_SIZE; if (maxCellY > g->rows - 1) maxCellY = g->rows - 1;
    if (minCellY > maxCellY) return 0;
    const int check = min_level > 0 || max_level >= 0;
    int n = 0;
    for (int cx = minCellX; cx <= maxCellX; ++cx)
        for (int cy = minCellY; cy <= maxCellY; ++cy) {
            int c = cx * g->rows + cy;
            for (int k = g->cell_start[c]; k < g->cell_start[c + 1]; ++k) {
                int idx = g->cell_items[k];
                const orbref_kp *kp = &kps[idx];
                if (check) {
                    if (kp->octave < min_level) continue;
                    if (max_level >= 0 && kp->octave > max_level) continue;
                }
                if (fabsf(kp->x - x) <= r && fabsf(kp->y - y) <= r) { if (n < cap) out[n] = idx; ++n; }
            }
        }
    return n;
}

int orbref_search_for_initialization(float nn_ratio, int check_orientation,
                                     const orbref_kp *kps1, const uint8_t *desc1, int n1,
                                     const orbref_kp *kps2, const uint8_t *desc2, int n2,
                                     int img_w, int img_h,
                                     float *pre, int32_t *matches12, int window_size)
{
    /* modules/ORB/ORBMatcher.cpp:33-116 */
    int numMatches = 0;
    orbref_grid *g = orbref_grid_build(kps2, n2, img_w, img_h);
    int32_t *matches21 = (int32_t *)malloc(sizeof(int32_t) * (n2 ? n2 : 1));
    int *matchedDist = (int *)malloc(sizeof(int) * (n2 ? n2 : 1));
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (n2 ? n2 : 1));
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    for (int i = 0; i < n2; ++i) { matches21[i] = -1; matchedDist[i] = INT_MAX; }
    rothist rh; rh_init(&rh);
    for (int idx1 = 0; idx1 < n1; ++idx1) {
        int level1 = kps1[idx1].octave;
        if (level1 > 0) continue;
        int nc = orbref_features_in_area(g, kps2, pre[2 * idx1], pre[2 * idx1 + 1], (float)window_size, level1, level1, cand, n2);
        if (nc == 0) continue;
        int bestDist = INT_MAX - 1, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int k = 0; k < nc; ++k) {
            int idx2 = cand[k];
            int dist = orbref_hamming(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
            if (matchedDist[idx2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = idx2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW && bestDist < orbref_round_f((float)bestDist2 * nn_ratio)) { /* :74 */
            if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; numMatches--; }
            matches12[idx1] = bestIdx2;
            matches21[bestIdx2] = idx1;
            matchedDist[bestIdx2] = bestDist;
            numMatches++;
            if (check_orientation) rh_push(&rh, rot_bin(kps1[idx1].angle, kps2[bestIdx2].angle), idx1);
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orbref_three_maxima(rh.n, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int k = 0; k < rh.n[i]; ++k) {
                int idx1 = rh.items[i][k];
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; numMatches--; }
            }
        }
    }
    for (int idx1 = 0; idx1 < n1; ++idx1)
        if (matches12[idx1] >= 0) { pre[2 * idx1] = kps2[matches12[idx1]].x; pre[2 * idx1 + 1] = kps2[matches12[idx1]].y; }
    rh_free(&rh);
    free(matches21); free(matchedDist); free(cand);
    orbref_grid_free(g);
    return numMatches;
}


/* ---- SearchByProjection(lastFrame|lastKF, curFrame, th): modules/ORB/ORBMatcher.cpp:203-348 ---- */
int orbref_search_by_projection_frame(int check_orientation, const uint8_t *q_desc, const float *q_xy,
                                      const float *q_radius, const int32_t *q_octave, const float *q_angle,
                                      const uint8_t *q_ok, int nq, const orbref_kp *kps2, const uint8_t *desc2, int n2,
                                      int img_w, int img_h, int32_t *frame_mp)
{
    int numMatch = 0;
    orbref_grid *g = orbref_grid_build(kps2, n2, img_w, img_h);
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (n2 ? n2 : 1));
    rothist rh; rh_init(&rh);
    for (int i = 0; i < nq; ++i) {
        if (!q_ok[i]) continue; /* :213-224 */
        int lastLevel = q_octave[i];
        int nc = orbref_features_in_area(g, kps2, q_xy[2 * i], q_xy[2 * i + 1], q_radius[i], lastLevel - 1, lastLevel + 1, cand, n2);
        if (nc == 0) continue;
        int bestDist = TH_HIGH + 1, bestIdx2 = -1;
        for (int k = 0; k < nc; ++k) {
            int idx2 = cand[k];
            if (frame_mp[idx2] != -1) continue; /* :235 */
            int dist = orbref_hamming(q_desc + 32 * (size_t)i, desc2 + 32 * (size_t)idx2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = idx2; }
        }
        if (bestDist <= TH_HIGH) {
            frame_mp[bestIdx2] = i;
            numMatch++;
            if (check_orientation) rh_push(&rh, rot_bin(q_angle[i], kps2[bestIdx2].angle), bestIdx2);
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orbref_three_maxima(rh.n, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int b = 0; b < HISTO_LENGTH; ++b) {
            if (b == ind1 || b == ind2 || b == ind3) continue;
            for (int k = 0; k < rh.n[b]; ++k) { frame_mp[rh.items[b][k]] = -1; numMatch--; }
        }
    }
    rh_free(&rh); free(cand); orbref_grid_free(g);
    return numMatch;
}

/* ---- SearchByProjection(frame, mapPoints, th): modules/ORB/ORBMatcher.cpp:350-415 ---- */
int orbref_search_by_projection_points(float nn_ratio, const uint8_t *q_desc, const float *q_xy, const float *q_radius,
                                       const int32_t *q_level, const uint8_t *q_ok, int nq, const orbref_kp *kps2,
                                       const uint8_t *desc2, int n2, int img_w, int img_h, int32_t *frame_mp,
                                       int32_t *counters)
{
    int numMatch = 0, numOut = 0, fail1 = 0, fail2 = 0;
    orbref_grid *g = orbref_grid_build(kps2, n2, img_w, img_h);
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (n2 ? n2 : 1));
    for (int i = 0; i < nq; ++i) {
        if (!q_ok[i]) { numOut++; continue; }
        const int predictLevel = q_level[i];
        int nc = orbref_features_in_area(g, kps2, q_xy[2 * i], q_xy[2 * i + 1], q_radius[i], predictLevel - 1, predictLevel, cand, n2);
        if (nc == 0) continue;
        int bestDist = 256, bestLevel = -1, secondDist = 257, secondLevel = -1, bestIdx = -1;
        for (int k = 0; k < nc; ++k) {
            int idx = cand[k];
            if (frame_mp[idx] != -1) continue; /* :383 */
            int dist = orbref_hamming(q_desc + 32 * (size_t)i, desc2 + 32 * (size_t)idx);
            if (dist < bestDist) { secondDist = bestDist; bestDist = dist; secondLevel = bestLevel; bestLevel = kps2[idx].octave; bestIdx = idx; }
            else if (dist < secondDist) { secondDist = dist; secondLevel = kps2[idx].octave; }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == secondLevel && (float)bestDist > nn_ratio * (float)secondDist) { fail1++; continue; }
            frame_mp[bestIdx] = i;
            numMatch++;
        } else fail2++;
    }
    if (counters) { counters[0] = numOut; counters[1] = fail1; counters[2] = fail2; }
    free(cand); orbref_grid_free(g);
    return numMatch;
}

/* ---- KeyFrame::getFeaturesInArea: modules/BasicObject/KeyFrame.cpp:181-211.  Same walk as Frame's, but the window test
 * is STRICT (`abs(kp.pt.x - x) < r`, :204) where Frame.cpp:120 uses `<=`. ---- */
int orbref_keyframe_features_in_area(const orbref_grid *g, const orbref_kp *kps, float x, float y, float r,
                                     int min_level, int max_level, int32_t *out, int cap)
{
    int minCellX = orbref_floor_f(x - r) / GRID_SIZE; if (minCellX < 0) minCellX = 0;
    int maxCellX = orbref_floor_f(x + r) / GRID_SIZE; if (maxCellX > g->cols - 1) maxCellX = g->cols - 1;
    if (minCellX > maxCellX) return 0;
    int minCellY = orbref_floor_f(y - r) / GRID_SIZE; if (minCellY < 0) minCellY = 0;
    int maxCellY = orbref_floor_f(y + r) / GRID_SIZE; if (maxCellY > g->rows - 1) maxCellY = g->rows - 1;
    if (minCellY > maxCellY) return 0;
    const int check = min_level > 0 || max_level >= 0;
    int n = 0;
    for (int cx = minCellX; cx <= maxCellX; ++cx)
        for (int cy = minCellY; cy <= maxCellY; ++cy) {
            int c = cx * g->rows + cy;
            for (int k = g->cell_start[c]; k < g->cell_start[c + 1]; ++k) {
                int idx = g->cell_items[k];
                const orbref_kp *kp = &kps[idx];
                if (check) {
                    if (kp->octave < min_level) continue;
                    if (max_level >= 0 && kp->octave > max_level) continue;
                }
                if (fabsf(kp->x - x) < r && fabsf(kp->y - y) < r) { if (n < cap) out[n] = idx; ++n; }
            }
        }
    return n;
}

/* ---- static SearchByProjection(keyFrame, mapPoints, Map*, th) ("fuse"): modules/ORB/ORBMatcher.cpp:524-592.
 * Restated here is everything of the loop body that does not touch MapPoint / KeyFrame state: for the map point i
 * whose projection passed :535-552 (q_ok), the window (:554-556: radius th * scale(predictLevel), levels
 * predictLevel-1 .. predictLevel, KeyFrame grid), the chi-square gate on the re-projection error (:566-567, a float
 * against the double product 5.991 * sigma2) and the closest descriptor below TH_LOW + 1 (:560, :569-574, first one
 * on ties).  best_idx[i] = -1 where the reference would `continue` (:558) or keep bestIdx1 == -1 (:577).
 * The observation rewiring of :578-589 stays with the caller; it never changes what a later point's window or
 * descriptor are (a point whose descriptor is recomputed by replace(), MapPoint.cpp:261, is from then on observed by
 * the key frame or bad, and is skipped at :534). ---- */
int orbref_search_fuse(const uint8_t *q_desc, const float *q_xy, const float *q_radius, const int32_t *q_level,
                       const uint8_t *q_ok, int nq, const orbref_kp *kps, const uint8_t *desc, int n, int img_w,
                       int img_h, const float *sigma2, int32_t *best_idx, int32_t *best_dist)
{
    int found = 0;
    orbref_grid *g = orbref_grid_build(kps, n, img_w, img_h);
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    for (int i = 0; i < nq; ++i) {
        best_idx[i] = -1; best_dist[i] = TH_LOW + 1;
        if (!q_ok[i]) continue;
        const int predictLevel = q_level[i];
        const float px = q_xy[2 * i], py = q_xy[2 * i + 1];
        int nc = orbref_keyframe_features_in_area(g, kps, px, py, q_radius[i], predictLevel - 1, predictLevel, cand, n);
        if (nc == 0) continue;
        int bestDist = TH_LOW + 1, bestIdx1 = -1;
        for (int k = 0; k < nc; ++k) {
            const int idx1 = cand[k];
            const orbref_kp *kp = &kps[idx1];
            const float squareError2 = (px - kp->x) * (px - kp->x) + (py - kp->y) * (py - kp->y);
            if ((double)squareError2 > 5.991 * (double)sigma2[kp->octave]) continue;
            const int dist = orbref_hamming(desc + 32 * (size_t)idx1, q_desc + 32 * (size_t)i);
            if (dist < bestDist) { bestDist = dist; bestIdx1 = idx1; }
        }
        best_idx[i] = bestIdx1; best_dist[i] = bestDist;
        if (bestIdx1 != -1) found++;
    }
    free(cand); orbref_grid_free(g);
    return found;
}
