/*
 * orc_lk.c -- CPU ORACLE (test infrastructure, NOT product code).  PARITY UNPINNED, see
 * vo_oracle.h.
 *
 * Restates cv::calcOpticalFlowPyrLK as the reference calls it
 * (src/feature.cpp:127-128,136-139: winSize 21x21, maxLevel 3, COUNT+EPS 30/0.01, flags 0,
 * minEigThreshold 1e-3, err vector passed) following OpenCV 4.5.x
 * modules/video/src/lkpyramid.cpp (buildOpticalFlowPyramid, calcSharrDeriv,
 * LKTrackerInvoker) and modules/imgproc/src/pyramids.cpp (pyrDown 8U) -- SURVEY.md App. A1-A3.
 * Compiled with -ffp-contract=off so no FMA is formed in the 2x2 solve.
 */
#include "vo_oracle.h"

#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
static inline int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len) {
        if (p < 0)
            p = -p;
        else
            p = 2 * (len - 1) - p;
    }
    return p;
}

/* pyramids.cpp: PyrDownInvoker<FixPtCast<uchar,8>>: horizontal [1 4 6 4 1] in int, vertical
 * [1 4 6 4 1] then (v + 128) >> 8, source indices reflected on the ROI itself */
void orc_pyr_down(const uint8_t *src, int w, int h, uint8_t *dst)
{
    int dw = (w + 1) / 2, dh = (h + 1) / 2;
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
    int *rows = (int *)malloc(sizeof(int) * (size_t)dw * 5);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int y = 0; y < dh; y++) {
        for (int k = 0; k < 5; k++) {
            int sy = reflect101(2 * y - 2 + k, h);
            const uint8_t *s = src + (size_t)sy * w;
            int *row = rows + (size_t)k * dw;
            for (int x = 0; x < dw; x++) {
                int x0 = reflect101(2 * x - 2, w), x1 = reflect101(2 * x - 1, w);
                int x2 = 2 * x, x3 = reflect101(2 * x + 1, w), x4 = reflect101(2 * x + 2, w);
                row[x] = s[x2] * 6 + (s[x1] + s[x3]) * 4 + s[x0] + s[x4];
            }
        }
        const int *r0 = rows, *r1 = rows + dw, *r2 = rows + 2 * dw, *r3 = rows + 3 * dw,
                  *r4 = rows + 4 * dw;
        for (int x = 0; x < dw; x++) {
            int v = r2[x] * 6 + (r1[x] + r3[x]) * 4 + r0[x] + r4[x];
            dst[(size_t)y * dw + x] = (uint8_t)((v + 128) >> 8);
        }
    }
    free(rows);
    }
}

/* lkpyramid.cpp calcSharrDeriv: rows/cols clamped by REFLECT_101, no normalisation */
void orc_scharr(const uint8_t *src, int w, int h, int16_t *dst)
{
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
    int *t0 = (int *)malloc(sizeof(int) * (size_t)(w + 2));
    int *t1 = (int *)malloc(sizeof(int) * (size_t)(w + 2));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int y = 0; y < h; y++) {
        const uint8_t *s0 = src + (size_t)(y > 0 ? y - 1 : h > 1 ? 1 : 0) * w;
        const uint8_t *s1 = src + (size_t)y * w;
        const uint8_t *s2 = src + (size_t)(y < h - 1 ? y + 1 : h > 1 ? h - 2 : 0) * w;
        int *tr0 = t0 + 1, *tr1 = t1 + 1;
        for (int x = 0; x < w; x++) {
            tr0[x] = (s0[x] + s2[x]) * 3 + s1[x] * 10;
            tr1[x] = s2[x] - s0[x];
        }
        int x0 = (w > 1 ? 1 : 0), x1 = (w > 1 ? w - 2 : 0);
        tr0[-1] = tr0[x0];
        tr0[w] = tr0[x1];
        tr1[-1] = tr1[x0];
        tr1[w] = tr1[x1];
        int16_t *d = dst + (size_t)y * w * 2;
        for (int x = 0; x < w; x++) {
            d[2 * x] = (int16_t)(tr0[x + 1] - tr0[x - 1]);
            d[2 * x + 1] = (int16_t)((tr1[x + 1] + tr1[x - 1]) * 3 + tr1[x] * 10);
        }
    }
    free(t0);
    free(t1);
    }
}

/* one pyramid level stored with a `brd`-pixel border on every side, like OpenCV's buffers */
typedef struct {
    int w, h, brd, stride;
    uint8_t *buf;  /* (h+2brd) x stride, REFLECT_101 border */
    uint8_t *img;  /* pointer to pixel (0,0) inside buf */
} OrcLevel;

static void level_alloc(OrcLevel *L, int w, int h, int brd)
{
    L->w = w;
    L->h = h;
    L->brd = brd;
    L->stride = w + 2 * brd;
    L->buf = (uint8_t *)malloc((size_t)L->stride * (h + 2 * brd));
    L->img = L->buf + (size_t)brd * L->stride + brd;
}

/* copyMakeBorder(level, ..., BORDER_REFLECT_101) */
static void level_fill(OrcLevel *L, const uint8_t *src /* w x h contiguous */)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int y = -L->brd; y < L->h + L->brd; y++) {
        int sy = reflect101(y, L->h);
        uint8_t *d = L->img + (ptrdiff_t)y * L->stride;
        const uint8_t *s = src + (size_t)sy * L->w;
        for (int x = -L->brd; x < L->w + L->brd; x++)
            d[x] = s[reflect101(x, L->w)];
    }
}

/* buildOpticalFlowPyramid(img, pyr, winSize, maxLevel, withDerivatives=false,
 * REFLECT_101, CONSTANT, tryReuse): returns the max level actually built */
static int build_pyramid(const uint8_t *img, int w, int h, int win, int max_level,
                         OrcLevel *levels)
{
    uint8_t *cur = (uint8_t *)malloc((size_t)w * h);
    memcpy(cur, img, (size_t)w * h);
    int cw = w, ch = h, lvl;
    for (lvl = 0; lvl <= max_level; lvl++) {
        if (lvl > 0) {
            int nw = (cw + 1) / 2, nh = (ch + 1) / 2;
            /* lkpyramid.cpp: stop when the next level would not be larger than the window */
            if (nw <= win || nh <= win)
                break;
            uint8_t *nxt = (uint8_t *)malloc((size_t)nw * nh);
            orc_pyr_down(cur, cw, ch, nxt);
            free(cur);
            cur = nxt;
            cw = nw;
            ch = nh;
        }
        level_alloc(&levels[lvl], cw, ch, win);
        level_fill(&levels[lvl], cur);
    }
    free(cur);
    return lvl - 1;
}

/* cvRound(float): round-half-to-even (SSE cvtss2si under the default rounding mode) */
static inline int cv_round_f(float v) { return (int)lrintf(v); }
/* cvFloor(float) as OpenCV's x86 build computes it: i = (int)v is cvttss2si -- INT_MIN ("integer indefinite") for NaN and for
 * every value outside int32 --, then i - (i > v) in two's complement (INT_MIN - 1 wraps to INT_MAX there: -1e30 "floors" to
 * INT_MAX; either is outside every admissibility window).  Spelled out so that no C undefined behaviour takes part (round 6:
 * the sanitizer tier runs the non-finite inputs of tests/adversarial.py through this). */
static inline int cv_floor_f(float v)
{
    int i = (v >= -2147483648.0f && v < 2147483648.0f) ? (int)v : (-2147483647 - 1);
    return (int)((unsigned)i - (unsigned)((float)i > v));
}

#define W_BITS 14
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

static long long g_iter_count;
long long orc_lk_last_iteration_count(void) { return g_iter_count; }
/* optional log of the Gauss-Newton iteration count of every (level, point) of the next call(s): buf [levels][n]
 * (tools/lk_pairing_study.py: what a kernel that runs two features in lock step would pay) */
static int *g_iter_log;
static int g_iter_log_n;
void orc_lk_set_iteration_log(int *buf, int n)
{
    g_iter_log = buf;
    g_iter_log_n = n;
}
/* optional: for every (level, point) the first iteration j after which the window corner is bit-identical to the one
 * after iteration j - p, and that period p (0 = the orbit never repeats exactly): buf [levels][n][2]
 * (tools/lk_cycle_study.py: could iterations of non-converging features be skipped exactly?) */
static int *g_cycle_log;
void orc_lk_set_cycle_log(int *buf) { g_cycle_log = buf; }
/* histogram of the inner-iteration count of every (point, level) solve since the last reset:
 * hist[k] = solves that executed k iterations (k = 0..100); used to size the GPU kernel's loops */
static long long g_iter_hist[101];
void orc_lk_iteration_histogram(long long *hist101, int reset)
{
    for (int i = 0; i <= 100; i++) {
        if (hist101)
            hist101[i] = g_iter_hist[i];
        if (reset)
            g_iter_hist[i] = 0;
    }
}

/* LKTrackerInvoker::operator() for one pyramid level and a range of points */
static void lk_level(const OrcLevel *I, const OrcLevel *J, const int16_t *derivBuf /* padded */,
                     const float *prevPts, float *nextPts, uint8_t *status, float *err, int n,
                     int win, int level, int maxLevel, int maxCount, double epsilon,
                     float minEigThreshold, int accum_mode, int nthreads)
{
    const int dstride = (I->w + 2 * win) * 2; /* int16 units per row of the padded deriv */
    const int16_t *deriv0 = derivBuf + (size_t)win * dstride + win * 2;
    const float halfWin = (float)((win - 1) * 0.5f);
    const float FLT_SCALE = 1.f / (1 << 20);
    long long iters_total = 0;
    (void)nthreads;

#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads) reduction(+ : iters_total)
#endif
    for (int ptidx = 0; ptidx < n; ptidx++) {
        int16_t IWin[32 * 32], dIWin[32 * 32 * 2];
        float prevX = prevPts[2 * ptidx] * (float)(1. / (1 << level));
        float prevY = prevPts[2 * ptidx + 1] * (float)(1. / (1 << level));
        float nextX, nextY;
        if (level == maxLevel) {
            nextX = prevX; /* flags == 0: no OPTFLOW_USE_INITIAL_FLOW */
            nextY = prevY;
        } else {
            nextX = nextPts[2 * ptidx] * 2.f;
            nextY = nextPts[2 * ptidx + 1] * 2.f;
        }
        nextPts[2 * ptidx] = nextX;
        nextPts[2 * ptidx + 1] = nextY;

        prevX -= halfWin;
        prevY -= halfWin;
        int ipx = cv_floor_f(prevX), ipy = cv_floor_f(prevY);
        if (ipx < -win || ipx >= I->w || ipy < -win || ipy >= I->h) {
            if (level == 0) {
                status[ptidx] = 0;
                if (err)
                    err[ptidx] = 0;
            }
            continue;
        }
        float a = prevX - ipx, b = prevY - ipy;
        int iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
        int iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
        int iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

        const int stepI = I->stride, stepJ = J->stride;
        long long iA11 = 0, iA12 = 0, iA22 = 0;
        float fA11 = 0, fA12 = 0, fA22 = 0;
        /* accum_mode 2 / 3: the four f32 lanes of OpenCV's v_float32x4 accumulators qA11 / qA12 / qA22 (see the header) */
        float qA11[4] = {0, 0, 0, 0}, qA12[4] = {0, 0, 0, 0}, qA22[4] = {0, 0, 0, 0};
        const int simd_cols = accum_mode >= 2 ? (win / 8) * 8 : 0; /* columns the 8-pixel SIMD loop covers (16 of 21) */
        for (int y = 0; y < win; y++) {
            const uint8_t *src = I->img + (ptrdiff_t)(y + ipy) * stepI + ipx;
            const int16_t *dsrc = deriv0 + (ptrdiff_t)(y + ipy) * dstride + ipx * 2;
            int16_t *Iptr = IWin + y * win, *dIptr = dIWin + y * win * 2;
            for (int x = 0; x < win; x++, dsrc += 2, dIptr += 2) {
                int ival = DESCALE(src[x] * iw00 + src[x + 1] * iw01 + src[x + stepI] * iw10 +
                                       src[x + stepI + 1] * iw11,
                                   W_BITS - 5);
                int ixval = DESCALE(dsrc[0] * iw00 + dsrc[2] * iw01 + dsrc[dstride] * iw10 +
                                        dsrc[dstride + 2] * iw11,
                                    W_BITS);
                int iyval = DESCALE(dsrc[1] * iw00 + dsrc[3] * iw01 + dsrc[dstride + 1] * iw10 +
                                        dsrc[dstride + 3] * iw11,
                                    W_BITS);
                Iptr[x] = (int16_t)ival;
                dIptr[0] = (int16_t)ixval;
                dIptr[1] = (int16_t)iyval;
                if (accum_mode == 0) {
                    iA11 += (long long)ixval * ixval;
                    iA12 += (long long)ixval * iyval;
                    iA22 += (long long)iyval * iyval;
                } else if (x < simd_cols) {
                    /* lane = x mod 4; fx, fy = v_cvt_f32 of the int16 values; qA = v_muladd(f, f, qA): multiply and add
                     * rounded separately (mode 2, SSE2 / SSE3 baseline) or fused (mode 3, a CV_FMA3 baseline) */
                    const float fx = (float)ixval, fy = (float)iyval;
                    const int l = x & 3;
                    if (accum_mode == 3) {
                        qA22[l] = fmaf(fy, fy, qA22[l]);
                        qA12[l] = fmaf(fx, fy, qA12[l]);
                        qA11[l] = fmaf(fx, fx, qA11[l]);
                    } else {
                        volatile float p22 = fy * fy, p12 = fx * fy, p11 = fx * fx; /* (rounded products) */
                        qA22[l] += p22;
                        qA12[l] += p12;
                        qA11[l] += p11;
                    }
                } else {
                    fA11 += (float)(ixval * ixval);
                    fA12 += (float)(ixval * iyval);
                    fA22 += (float)(iyval * iyval);
                }
            }
        }
        if (accum_mode >= 2) { /* iA += v_reduce_sum(qA): (q0 + q2) + (q1 + q3) */
            fA11 += (qA11[0] + qA11[2]) + (qA11[1] + qA11[3]);
            fA12 += (qA12[0] + qA12[2]) + (qA12[1] + qA12[3]);
            fA22 += (qA22[0] + qA22[2]) + (qA22[1] + qA22[3]);
        }
        float A11, A12, A22;
        if (accum_mode == 0) {
            A11 = (float)iA11 * FLT_SCALE;
            A12 = (float)iA12 * FLT_SCALE;
            A22 = (float)iA22 * FLT_SCALE;
        } else {
            A11 = fA11 * FLT_SCALE;
            A12 = fA12 * FLT_SCALE;
            A22 = fA22 * FLT_SCALE;
        }
        float D = A11 * A22 - A12 * A12;
        float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                       (float)(2 * win * win);
        if (minEig < minEigThreshold || D < FLT_EPSILON) {
            if (level == 0)
                status[ptidx] = 0;
            continue;
        }
        D = 1.f / D;

        nextX -= halfWin;
        nextY -= halfWin;
        float prevDX = 0, prevDY = 0;
        int my_iters = 0;
        float histX[128], histY[128];
        int cyc_j = 0, cyc_p = 0;
        for (int j = 0; j < maxCount; j++) {
            int inx = cv_floor_f(nextX), iny = cv_floor_f(nextY);
            if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
                if (level == 0)
                    status[ptidx] = 0;
                break;
            }
            iters_total++;
            my_iters++;
            a = nextX - inx;
            b = nextY - iny;
            iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
            iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
            iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            long long ib1 = 0, ib2 = 0;
            float fb1 = 0, fb2 = 0;
            /* accum_mode 2 / 3: lanes 0 / 2 (x part) and 1 / 3 (y part) of qb0 and qb1 */
            float qb0x[2] = {0, 0}, qb0y[2] = {0, 0}, qb1x[2] = {0, 0}, qb1y[2] = {0, 0};
            for (int y = 0; y < win; y++) {
                const uint8_t *Jptr = J->img + (ptrdiff_t)(y + iny) * stepJ + inx;
                const int16_t *Iptr = IWin + y * win, *dIptr = dIWin + y * win * 2;
                int dblk[8]; /* the eight residuals of an 8-pixel SIMD block */
                for (int x = 0; x < win; x++, dIptr += 2) {
                    int diff = DESCALE(Jptr[x] * iw00 + Jptr[x + 1] * iw01 + Jptr[x + stepJ] * iw10 +
                                           Jptr[x + stepJ + 1] * iw11,
                                       W_BITS - 5) -
                               Iptr[x];
                    if (accum_mode == 0) {
                        ib1 += (long long)diff * dIptr[0];
                        ib2 += (long long)diff * dIptr[1];
                    } else if (x < simd_cols) {
                        /* v_pack saturates the residual to int16 before the products; then pixels (k, k + 4) of the block
                         * are paired by v_dotprod (_mm_madd_epi16: exact int32 pair sum), converted to f32 and added:
                         * qb0 lanes <- pairs (0, 4) and (1, 5), qb1 lanes <- pairs (2, 6) and (3, 7) */
                        diff = diff > 32767 ? 32767 : diff < -32768 ? -32768 : diff;
                        dblk[x & 7] = diff;
                        if ((x & 7) == 7) {
                            const int16_t *dI = dIptr - 14; /* (Ix, Iy) of pixel 0 of the block */
                            for (int k = 0; k < 4; k++) {
                                const int sx = dblk[k] * dI[2 * k] + dblk[k + 4] * dI[2 * (k + 4)];
                                const int sy = dblk[k] * dI[2 * k + 1] + dblk[k + 4] * dI[2 * (k + 4) + 1];
                                float *qx = k < 2 ? &qb0x[k] : &qb1x[k - 2], *qy = k < 2 ? &qb0y[k] : &qb1y[k - 2];
                                *qx += (float)sx;
                                *qy += (float)sy;
                            }
                        }
                    } else {
                        fb1 += (float)(diff * dIptr[0]);
                        fb2 += (float)(diff * dIptr[1]);
                    }
                }
            }
            if (accum_mode >= 2) {
                /* v_recombine(v_interleave_pairs(qb0 + qb1), 0, qf0, qf1); ib1 += v_reduce_sum(qf0) = (s0 + 0) + (s2 + 0) */
                fb1 += (qb0x[0] + qb1x[0]) + (qb0x[1] + qb1x[1]);
                fb2 += (qb0y[0] + qb1y[0]) + (qb0y[1] + qb1y[1]);
            }
            float b1, b2;
            if (accum_mode == 0) {
                b1 = (float)ib1 * FLT_SCALE;
                b2 = (float)ib2 * FLT_SCALE;
            } else {
                b1 = fb1 * FLT_SCALE;
                b2 = fb2 * FLT_SCALE;
            }
            float dx = (float)((A12 * b2 - A22 * b1) * D);
            float dy = (float)((A12 * b1 - A11 * b2) * D);
            nextX += dx;
            nextY += dy;
            nextPts[2 * ptidx] = nextX + halfWin;
            nextPts[2 * ptidx + 1] = nextY + halfWin;
            if (g_cycle_log && j < 128) {
                histX[j] = nextX;
                histY[j] = nextY;
                for (int pp = 1; pp <= 12 && pp <= j && !cyc_p; pp++)
                    if (memcmp(&histX[j - pp], &nextX, 4) == 0 && memcmp(&histY[j - pp], &nextY, 4) == 0) {
                        cyc_j = j;
                        cyc_p = pp;
                    }
            }
            /* Point2f::ddot -> double */
            if ((double)dx * dx + (double)dy * dy <= epsilon)
                break;
            if (j > 0 && fabs(dx + prevDX) < 0.01 && fabs(dy + prevDY) < 0.01) {
                nextPts[2 * ptidx] -= dx * 0.5f;
                nextPts[2 * ptidx + 1] -= dy * 0.5f;
                break;
            }
            prevDX = dx;
            prevDY = dy;
        }
#ifdef _OPENMP
#pragma omp atomic
#endif
        g_iter_hist[my_iters]++;
        if (g_iter_log && ptidx < g_iter_log_n)
            g_iter_log[(size_t)level * g_iter_log_n + ptidx] = my_iters;
        if (g_cycle_log && g_iter_log && ptidx < g_iter_log_n) {
            g_cycle_log[((size_t)level * g_iter_log_n + ptidx) * 2] = cyc_j;
            g_cycle_log[((size_t)level * g_iter_log_n + ptidx) * 2 + 1] = cyc_p;
        }

        /* status[ptidx] && err && level == 0 && !(flags & OPTFLOW_LK_GET_MIN_EIGENVALS) */
        if (status[ptidx] && err && level == 0) {
            float npx = nextPts[2 * ptidx] - halfWin, npy = nextPts[2 * ptidx + 1] - halfWin;
            int inx = cv_floor_f(npx), iny = cv_floor_f(npy);
            if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
                status[ptidx] = 0;
                continue;
            }
            float aa = npx - inx, bb = npy - iny;
            iw00 = cv_round_f((1.f - aa) * (1.f - bb) * (1 << W_BITS));
            iw01 = cv_round_f(aa * (1.f - bb) * (1 << W_BITS));
            iw10 = cv_round_f((1.f - aa) * bb * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            float errval = 0.f;
            for (int y = 0; y < win; y++) {
                const uint8_t *Jptr = J->img + (ptrdiff_t)(y + iny) * stepJ + inx;
                const int16_t *Iptr = IWin + y * win;
                for (int x = 0; x < win; x++) {
                    int diff = DESCALE(Jptr[x] * iw00 + Jptr[x + 1] * iw01 + Jptr[x + stepJ] * iw10 +
                                           Jptr[x + stepJ + 1] * iw11,
                                       W_BITS - 5) -
                               Iptr[x];
                    errval += (float)abs(diff);
                }
            }
            err[ptidx] = errval * 1.f / (32 * win * win);
        }
    }
    g_iter_count += iters_total;
}

int orc_calc_optical_flow_pyr_lk(const uint8_t *prev, const uint8_t *next, int w, int h,
                                 const float *prev_pts, int n, float *next_pts,
                                 uint8_t *status, float *err, int win, int max_level,
                                 int max_count, double eps, double min_eig_threshold,
                                 int accum_mode, int nthreads)
{
    if (win < 3 || win > 32 || max_level < 0 || max_level > 15)
        return -1;
    g_iter_count = 0;
    if (n == 0)
        return 0;
#ifdef _OPENMP
    if (nthreads <= 0)
        nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    /* criteria sanitising (lkpyramid.cpp SparsePyrLKOpticalFlowImpl::calc) */
    if (max_count < 0)
        max_count = 0;
    if (max_count > 100)
        max_count = 100;
    if (eps < 0.)
        eps = 0.;
    if (eps > 10.)
        eps = 10.;
    double epsilon = eps * eps;

    OrcLevel pI[16], pJ[16];
    int lI = build_pyramid(prev, w, h, win, max_level, pI);
    int lJ = build_pyramid(next, w, h, win, max_level, pJ);
    int maxLevel = lI < lJ ? lI : lJ;

    for (int i = 0; i < n; i++)
        status[i] = 1;
    if (err)
        for (int i = 0; i < n; i++)
            err[i] = 0;

    for (int level = maxLevel; level >= 0; level--) {
        const OrcLevel *I = &pI[level];
        /* derivI = Scharr(prev level); copyMakeBorder(..., BORDER_CONSTANT) = zero border */
        int16_t *d = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)I->w * I->h);
        uint8_t *plain = (uint8_t *)malloc((size_t)I->w * I->h);
        for (int y = 0; y < I->h; y++)
            memcpy(plain + (size_t)y * I->w, I->img + (ptrdiff_t)y * I->stride, (size_t)I->w);
        orc_scharr(plain, I->w, I->h, d);
        free(plain);
        size_t dstride = (size_t)(I->w + 2 * win) * 2;
        int16_t *dpad = (int16_t *)calloc(dstride * (I->h + 2 * win), sizeof(int16_t));
        for (int y = 0; y < I->h; y++)
            memcpy(dpad + (size_t)(y + win) * dstride + win * 2, d + (size_t)y * I->w * 2,
                   sizeof(int16_t) * 2 * (size_t)I->w);
        free(d);
        lk_level(I, &pJ[level], dpad, prev_pts, next_pts, status, err, n, win, level, maxLevel,
                 max_count, epsilon, (float)min_eig_threshold, accum_mode, nthreads);
        free(dpad);
    }
    for (int l = 0; l <= lI; l++)
        free(pI[l].buf);
    for (int l = 0; l <= lJ; l++)
        free(pJ[l].buf);
    return 0;
}
