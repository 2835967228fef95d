/*
 * orc_essential.c -- CPU ORACLE (test infrastructure, NOT product code).  PARITY UNPINNED, see
 * vo_oracle.h.
 *
 * The `mono_rotation` branch of trackingFrame2Frame() (reference src/visualOdometry.cpp:146-157):
 *     E = cv::findEssentialMat(pointsLeft_t0, pointsLeft_t1, focal, pp, cv::RANSAC, 0.999, 1.0, mask);
 *     cv::recoverPose(E, pointsLeft_t0, pointsLeft_t1, rotation, translation_mono, focal, pp, mask);
 * restated from OpenCV 4.5.x [upstream-memory]:
 *   calib3d/src/five-point.cpp  findEssentialMat, EMEstimatorCallback::runKernel / computeError,
 *                               decomposeEssentialMat, recoverPose
 *   calib3d/src/ptsetreg.cpp    RANSACPointSetRegistrator::run / getSubset / findInliers (maxIters 1000)
 *   core/src/mathfuncs.cpp      solvePoly (Durand-Kerner sweeps from the powers of 1+i, 300 iterations)
 *   core/src/lapack.cpp         LUImpl (invert DECOMP_LU), JacobiSVD with the FULL_UV completion
 *   calib3d/src/triangulate.cpp icvTriangulatePoints (f64 inputs)
 *
 * Deliberate, documented differences from the library text (all at rounding level, none in structure):
 *   * getCoeffMat and the determinant polynomial are machine-generated closed forms upstream; here the
 *     same polynomials are formed by plain polynomial products / sums (row 0 = det E, rows 1..9 =
 *     2 E E^T E - trace(E E^T) E row-major), so coefficients agree to rounding only;
 *   * the design matrix row is ordered (x2x1, x2y1, x2, y2x1, y2y1, y2, x1, y1, 1), the order that
 *     makes the null vector the row-major E of computeError's x2^T E x1;
 *   * solvePoly's branch for iterates that coincide bit for bit (num_same_root > 1) only skips the
 *     zero factor; the root-of-unity correction upstream applies there is not restated (unreachable
 *     from distinct starting points in practice).
 */
#include "orc_internal.h"

#include <float.h>
#include <math.h>
#include <string.h>

/* Polynomials in the three unknowns (x, y, z) of E = x X + y Y + z Z + W, by coefficient vectors:
 *   linear  [4]  over (x, y, z, 1)
 *   quadric [10] over (x^2, y^2, z^2, xy, xz, yz, x, y, z, 1)
 *   cubic   [20] over the column order of the 10 x 20 constraint matrix (Stewenius / OpenCV getCoeffMat):
 *     x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy | xz^2 xz x yz^2 yz y z^3 z^2 z 1 */
static void mul_ll(const double *a, const double *b, double *q)
{
    q[0] = a[0] * b[0];
    q[1] = a[1] * b[1];
    q[2] = a[2] * b[2];
    q[3] = a[0] * b[1] + a[1] * b[0];
    q[4] = a[0] * b[2] + a[2] * b[0];
    q[5] = a[1] * b[2] + a[2] * b[1];
    q[6] = a[0] * b[3] + a[3] * b[0];
    q[7] = a[1] * b[3] + a[3] * b[1];
    q[8] = a[2] * b[3] + a[3] * b[2];
    q[9] = a[3] * b[3];
}

/* cubic monomial o collects quadric[QL_Q[k]] * linear[QL_L[k]] for k in [QL_START[o], QL_START[o + 1]) */
static const uint8_t QL_START[21] = {0, 1, 2, 4, 6, 8, 10, 12, 14, 17, 20, 22, 25, 27, 29, 32, 34, 35, 37, 39, 40};
static const uint8_t QL_Q[40] = {0, 1, 0, 3, 1, 3, 0, 4, 0, 6, 1, 5, 1, 7, 3, 4, 5, 3, 6, 7,
                                 2, 4, 4, 6, 8, 6, 9, 2, 5, 5, 7, 8, 7, 9, 2, 2, 8, 8, 9, 9};
static const uint8_t QL_L[40] = {0, 1, 1, 0, 0, 1, 2, 0, 3, 0, 2, 1, 3, 1, 2, 1, 0, 3, 1, 0,
                                 0, 2, 3, 2, 0, 3, 0, 1, 2, 3, 2, 1, 3, 1, 2, 3, 2, 3, 2, 3};

/* c += s * (q * l) */
static void mul_ql_acc(const double *q, const double *l, double s, double *c)
{
    for (int o = 0; o < 20; o++) {
        double t = 0;
        for (int k = QL_START[o]; k < QL_START[o + 1]; k++)
            t += q[QL_Q[k]] * l[QL_L[k]];
        c[o] += s * t;
    }
}

/* core/src/lapack.cpp LUImpl<double> (b: m x n right-hand sides); returns 0 when singular */
static int lu_solve(double *A, int astep, int m, double *b, int bstep, int n)
{
    const double eps = DBL_EPSILON * 100;
    int p = 1;
    for (int i = 0; i < m; i++) {
        int k = i;
        for (int j = i + 1; j < m; j++)
            if (fabs(A[j * astep + i]) > fabs(A[k * astep + i]))
                k = j;
        if (fabs(A[k * astep + i]) < eps)
            return 0;
        if (k != i) {
            for (int j = i; j < m; j++) {
                double t = A[i * astep + j];
                A[i * astep + j] = A[k * astep + j];
                A[k * astep + j] = t;
            }
            for (int j = 0; j < n; j++) {
                double t = b[i * bstep + j];
                b[i * bstep + j] = b[k * bstep + j];
                b[k * bstep + j] = t;
            }
            p = -p;
        }
        double d = -1 / A[i * astep + i];
        for (int j = i + 1; j < m; j++) {
            double alpha = A[j * astep + i] * d;
            for (k = i + 1; k < m; k++)
                A[j * astep + k] += alpha * A[i * astep + k];
            for (k = 0; k < n; k++)
                b[j * bstep + k] += alpha * b[i * bstep + k];
        }
    }
    for (int i = m - 1; i >= 0; i--)
        for (int j = 0; j < n; j++) {
            double s = b[i * bstep + j];
            for (int k = i + 1; k < m; k++)
                s -= A[i * astep + k] * b[k * bstep + j];
            b[i * bstep + j] = s / A[i * astep + i];
        }
    return p;
}

/* cv::solvePoly(coeffs (ascending, real), roots, maxIters = 300); returns the number of roots */
static int solve_poly(const double *c0, int n0, double *re, double *im)
{
    int n = n0;
    for (; n > 1; n--)
        if (fabs(c0[n]) + 0.0 > DBL_EPSILON)
            break;
    double pr = 1, pi = 0;
    for (int i = 0; i < n; i++) {
        re[i] = pr;
        im[i] = pi;
        /* p = p * r, r = (1, 1) */
        double tr = pr * 1 - pi * 1, ti = pr * 1 + pi * 1;
        pr = tr;
        pi = ti;
    }
    const int maxIters = 300;
    for (int iter = 0; iter < maxIters; iter++) {
        double maxDiff = 0;
        for (int i = 0; i < n; i++) {
            pr = re[i];
            pi = im[i];
            double nr = c0[n], ni = 0, dr = c0[n], di = 0;
            for (int j = 0; j < n; j++) {
                /* num = num * p + coeffs[n - j - 1] */
                double tr = nr * pr - ni * pi, ti = nr * pi + ni * pr;
                nr = tr + c0[n - j - 1];
                ni = ti + 0.0;
                if (j != i) {
                    double qr = pr - re[j], qi = pi - im[j];
                    if (qr != 0 || qi != 0) {
                        tr = dr * qr - di * qi;
                        ti = dr * qi + di * qr;
                        dr = tr;
                        di = ti;
                    }
                }
            }
            /* num /= denom (cv::Complex operator /) */
            double t = 1. / (dr * dr + di * di);
            double xr = (nr * dr + ni * di) * t, xi = (-nr * di + ni * dr) * t;
            re[i] = pr - xr;
            im[i] = pi - xi;
            double a = sqrt(xr * xr + xi * xi);
            maxDiff = maxDiff > a ? maxDiff : a;
        }
        if (maxDiff <= 0)
            break;
    }
    for (int i = 0; i < n; i++)
        if (fabs(im[i]) < 1e-100)
            im[i] = 0;
    for (int i = n; i < n0; i++) { /* for( ; n < n0; n++ ) roots[n+1] = roots[n] */
        re[i] = re[i - 1];
        im[i] = im[i - 1];
    }
    return n0;
}

/* polynomial (ascending coefficients) helpers for det B(z) */
static void pmul(const double *a, int na, const double *b, int nb, double *r)
{
    for (int i = 0; i < na + nb - 1; i++)
        r[i] = 0;
    for (int i = 0; i < na; i++)
        for (int j = 0; j < nb; j++)
            r[i + j] += a[i] * b[j];
}

/* EMEstimatorCallback::runKernel: q1, q2 = 5 normalised correspondences (x, y); E out [<=10][9] */
int orc_five_point(const double *q1, const double *q2, double *Es)
{
    /* Q (5 x 9) stored as the 5 rows JacobiSVD rotates (m = 9, n = 5, n1 = 9) */
    double At[9 * 9], W[9], Vt5[25];
    memset(At, 0, sizeof(At));
    for (int i = 0; i < 5; i++) {
        double x1 = q1[2 * i], y1 = q1[2 * i + 1], x2 = q2[2 * i], y2 = q2[2 * i + 1];
        double *r = At + 9 * i;
        r[0] = x2 * x1;
        r[1] = x2 * y1;
        r[2] = x2;
        r[3] = y2 * x1;
        r[4] = y2 * y1;
        r[5] = y2;
        r[6] = x1;
        r[7] = y1;
        r[8] = 1.0;
    }
    orc_jacobi_svd(At, 9, W, Vt5, 5, 9, 5, 9);
    const double *X = At + 9 * 5, *Y = At + 9 * 6, *Z = At + 9 * 7, *Wv = At + 9 * 8;

    /* E(x, y, z) = x X + y Y + z Z + W, entries linear polynomials */
    double E[9][4], EEt[9][10], tr[10], q[10];
    for (int i = 0; i < 9; i++) {
        E[i][0] = X[i];
        E[i][1] = Y[i];
        E[i][2] = Z[i];
        E[i][3] = Wv[i];
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double *d = EEt[3 * i + j];
            mul_ll(E[3 * i], E[3 * j], d);
            for (int k = 1; k < 3; k++) {
                mul_ll(E[3 * i + k], E[3 * j + k], q);
                for (int o = 0; o < 10; o++)
                    d[o] += q[o];
            }
        }
    for (int o = 0; o < 10; o++)
        tr[o] = EEt[0][o] + EEt[4][o] + EEt[8][o];

    double A[10 * 20];
    memset(A, 0, sizeof(A));
    /* row 0: det E */
    {
        static const int perm[6][4] = {{0, 4, 8, 1}, {0, 5, 7, -1}, {1, 5, 6, 1}, {1, 3, 8, -1}, {2, 3, 7, 1}, {2, 4, 6, -1}};
        for (int p = 0; p < 6; p++) {
            mul_ll(E[perm[p][1]], E[perm[p][2]], q);
            mul_ql_acc(q, E[perm[p][0]], (double)perm[p][3], A);
        }
    }
    /* rows 1..9: 2 E E^T E - trace(E E^T) E */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double *row = A + 20 * (1 + 3 * i + j);
            for (int k = 0; k < 3; k++)
                mul_ql_acc(EEt[3 * i + k], E[3 * k + j], 2.0, row);
            mul_ql_acc(tr, E[3 * i + j], -1.0, row);
        }

    /* A = A.colRange(0, 10).inv() * A.colRange(10, 20)   (invert DECOMP_LU, then the product) */
    double L[100], inv[100], G[100];
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 10; j++) {
            L[10 * i + j] = A[20 * i + j];
            inv[10 * i + j] = i == j ? 1.0 : 0.0;
        }
    if (lu_solve(L, 10, 10, inv, 10, 10) == 0)
        memset(inv, 0, sizeof(inv));
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 10; j++) {
            double s = 0;
            for (int k = 0; k < 10; k++)
                s += inv[10 * i + k] * A[20 * k + 10 + j];
            G[10 * i + j] = s;
        }

    /* B(z): rows from (x^2 z, x^2), (y^2 z, y^2), (xyz, xy); descending powers of z per block */
    double b[3 * 13];
    for (int i = 0; i < 3; i++) {
        const double *r1 = G + 10 * (2 * i + 4), *r2 = G + 10 * (2 * i + 5);
        double row1[13] = {0}, row2[13] = {0};
        for (int k = 0; k < 3; k++) {
            row1[1 + k] = r1[k];
            row1[5 + k] = r1[3 + k];
            row2[k] = r2[k];
            row2[4 + k] = r2[3 + k];
        }
        for (int k = 0; k < 4; k++) {
            row1[9 + k] = r1[6 + k];
            row2[8 + k] = r2[6 + k];
        }
        for (int k = 0; k < 13; k++)
            b[13 * i + k] = row1[k] - row2[k];
    }

    /* c(z) = det B(z), ascending coefficients c[0..10] */
    double c[11] = {0};
    {
        /* ascending-order copies of the three columns */
        double p0[3][4], p1[3][4], p2[3][5];
        for (int i = 0; i < 3; i++) {
            for (int k = 0; k < 4; k++) {
                p0[i][k] = b[13 * i + 3 - k];
                p1[i][k] = b[13 * i + 7 - k];
            }
            for (int k = 0; k < 5; k++)
                p2[i][k] = b[13 * i + 12 - k];
        }
        static const int cof[3][2] = {{1, 2}, {0, 2}, {0, 1}};
        for (int i = 0; i < 3; i++) {
            int r = cof[i][0], s = cof[i][1];
            double m1[7], m2[7], minor[7], term[11];
            pmul(p0[r], 4, p1[s], 4, m1);
            pmul(p1[r], 4, p0[s], 4, m2);
            for (int k = 0; k < 7; k++)
                minor[k] = m1[k] - m2[k];
            pmul(p2[i], 5, minor, 7, term);
            for (int k = 0; k < 11; k++)
                c[k] += (i == 1 ? -1.0 : 1.0) * term[k];
        }
    }

    double re[10], im[10];
    solve_poly(c, 10, re, im);

    int count = 0;
    for (int i = 0; i < 10; i++) {
        if (fabs(im[i]) > 1e-10)
            continue;
        double z1 = re[i], z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double bz[9], w[3], u[9], vt[9];
        for (int j = 0; j < 3; j++) {
            const double *br = b + 13 * j;
            bz[3 * j + 0] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
            bz[3 * j + 1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
            bz[3 * j + 2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
        }
        orc_svd(bz, 3, 3, w, u, vt); /* SVD::solveZ: last row of vt */
        if (fabs(vt[8]) < 1e-10)
            continue;
        double x = vt[6] / vt[8], y = vt[7] / vt[8];
        double *Ev = Es + 9 * count, nrm = 0;
        for (int k = 0; k < 9; k++) {
            Ev[k] = X[k] * x + Y[k] * y + Z[k] * z1 + Wv[k];
            nrm += Ev[k] * Ev[k];
        }
        nrm = sqrt(nrm);
        for (int k = 0; k < 9; k++)
            Ev[k] /= nrm;
        count++;
    }
    return count;
}

/* EMEstimatorCallback::computeError: Sampson distance (f64) stored as f32 */
float orc_sampson_error(const double *E, double x1x, double x1y, double x2x, double x2y)
{
    double Ex1[3], Etx2[3];
    for (int r = 0; r < 3; r++) {
        Ex1[r] = E[3 * r] * x1x + E[3 * r + 1] * x1y + E[3 * r + 2] * 1.;
        Etx2[r] = E[r] * x2x + E[3 + r] * x2y + E[6 + r] * 1.;
    }
    double x2tEx1 = x2x * Ex1[0] + x2y * Ex1[1] + 1. * Ex1[2];
    double a = Ex1[0] * Ex1[0], b = Ex1[1] * Ex1[1], c = Etx2[0] * Etx2[0], d = Etx2[1] * Etx2[1];
    return (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
}

/* points.col(0) = (points.col(0) - cx) / fx as OpenCV's MatExpr evaluates it: x * (1/fx) + (-cx * (1/fx)) */
static void normalise_points(const float *p, int n, double fx, double fy, double cx, double cy, double *q)
{
    const double ax = 1. / fx, ay = 1. / fy, bx = -cx * ax, by = -cy * ay;
    for (int i = 0; i < n; i++) {
        q[2 * i] = (double)p[2 * i] * ax + bx;
        q[2 * i + 1] = (double)p[2 * i + 1] * ay + by;
    }
}

static int update_num_iters(double p, double ep, int modelPoints, int maxIters)
{
    p = p > 0. ? p : 0.;
    p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.;
    ep = ep < 1. ? ep : 1.;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, modelPoints);
    if (denom < DBL_MIN)
        return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)rint(num / denom);
}

int orc_find_essential_mat(const float *pts1, const float *pts2, int n, double focal, double ppx, double ppy,
                           double prob, double threshold, double *E, uint8_t *mask_out, double *dbg)
{
    const int modelPoints = 5, maxIters = 1000;
    if (dbg)
        dbg[0] = dbg[1] = dbg[2] = 0;
    if (n < modelPoints)
        return 0;
    double *q1 = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *q2 = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    uint8_t *mask = (uint8_t *)malloc((size_t)n), *best = (uint8_t *)malloc((size_t)n);
    normalise_points(pts1, n, focal, focal, ppx, ppy, q1);
    normalise_points(pts2, n, focal, focal, ppx, ppy, q2);
    threshold /= (focal + focal) / 2;
    const float t2 = (float)(threshold * threshold);
    int niters = maxIters, maxGood = 0, iter = 0, ret = 0;
    double bestE[9] = {0}, Es[90];
    memset(best, 0, (size_t)n);
    if (n == modelPoints) {
        int nm = orc_five_point(q1, q2, Es);
        if (nm > 0) {
            memcpy(E, Es, sizeof(double) * 9); /* bestModel = all models; the caller reshapes the first 3 rows */
            if (mask_out)
                memset(mask_out, 1, (size_t)n);
            ret = 1;
        }
        goto done;
    }
    uint64_t rng = (uint64_t)-1;
    for (iter = 0; iter < niters; iter++) {
        int idx[5];
        double s1[10], s2[10];
        for (int i = 0; i < modelPoints; i++) {
            int idx_i, j;
            for (;;) {
                idx_i = idx[i] = (int)(orc_rng_next(&rng) % (unsigned)n);
                for (j = 0; j < i; j++)
                    if (idx_i == idx[j])
                        break;
                if (j == i)
                    break;
            }
            s1[2 * i] = q1[2 * idx_i];
            s1[2 * i + 1] = q1[2 * idx_i + 1];
            s2[2 * i] = q2[2 * idx_i];
            s2[2 * i + 1] = q2[2 * idx_i + 1];
        }
        int nmodels = orc_five_point(s1, s2, Es);
        for (int m = 0; m < nmodels; m++) {
            const double *Em = Es + 9 * m;
            int good = 0;
            for (int i = 0; i < n; i++) {
                int f = orc_sampson_error(Em, q1[2 * i], q1[2 * i + 1], q2[2 * i], q2[2 * i + 1]) <= t2;
                mask[i] = (uint8_t)f;
                good += f;
            }
            if (good > (maxGood > modelPoints - 1 ? maxGood : modelPoints - 1)) {
                uint8_t *t = mask;
                mask = best;
                best = t;
                memcpy(bestE, Em, sizeof(bestE));
                maxGood = good;
                niters = update_num_iters(prob, (double)(n - good) / n, modelPoints, niters);
                if (dbg)
                    dbg[1] = iter * 10 + m;
            }
        }
    }
    if (maxGood > 0) {
        memcpy(E, bestE, sizeof(bestE));
        if (mask_out)
            memcpy(mask_out, best, (size_t)n);
        ret = 1;
    }
done:
    if (dbg) {
        dbg[0] = iter;
        dbg[2] = maxGood;
    }
    free(q1);
    free(q2);
    free(mask);
    free(best);
    return ret;
}

static double det3(const double *M)
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

static void mat3_mul(const double *A, const double *B, double *C)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++)
                s += A[3 * i + k] * B[3 * k + j];
            C[3 * i + j] = s;
        }
}

/* five-point.cpp decomposeEssentialMat */
void orc_decompose_essential_mat(const double *E, double *R1, double *R2, double *t)
{
    double w[3], U[9], Vt[9], T[9];
    orc_svd(E, 3, 3, w, U, Vt);
    if (det3(U) < 0)
        for (int i = 0; i < 9; i++)
            U[i] *= -1.;
    if (det3(Vt) < 0)
        for (int i = 0; i < 9; i++)
            Vt[i] *= -1.;
    static const double Wm[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    mat3_mul(U, Wm, T);
    mat3_mul(T, Vt, R1);
    mat3_mul(U, Wt, T);
    mat3_mul(T, Vt, R2);
    for (int i = 0; i < 3; i++)
        t[i] = U[3 * i + 2] * 1.0;
}

/* icvTriangulatePoints for P0 = [I | 0] and P = [R | t], f64 normalised points; X: homogeneous (4) */
static void triangulate_d(const double *P0, const double *P1, double x0, double y0, double x1, double y1, double *X)
{
    double A[16], w[4], u[16], vt[16];
    const double *P[2] = {P0, P1};
    const double x[2] = {x0, x1}, y[2] = {y0, y1};
    for (int j = 0; j < 2; j++)
        for (int k = 0; k < 4; k++) {
            A[(j * 2 + 0) * 4 + k] = x[j] * P[j][2 * 4 + k] - P[j][0 * 4 + k];
            A[(j * 2 + 1) * 4 + k] = y[j] * P[j][2 * 4 + k] - P[j][1 * 4 + k];
        }
    orc_svd(A, 4, 4, w, u, vt);
    for (int k = 0; k < 4; k++)
        X[k] = vt[12 + k];
}

/* cv::recoverPose(E, points1, points2, R, t, focal, pp, mask) (distanceThresh = 50).
 * mask (n, in/out, may be NULL = no input mask).  Returns the number of points that pass the cheirality check. */
int orc_recover_pose(const double *E, const float *pts1, const float *pts2, int n, double focal, double ppx,
                     double ppy, double *R, double *t, uint8_t *mask)
{
    const double dist = 50.0;
    double *q1 = (double *)malloc(sizeof(double) * 2 * (size_t)(n > 0 ? n : 1));
    double *q2 = (double *)malloc(sizeof(double) * 2 * (size_t)(n > 0 ? n : 1));
    uint8_t *m4 = (uint8_t *)malloc(4 * (size_t)(n > 0 ? n : 1));
    normalise_points(pts1, n, focal, focal, ppx, ppy, q1);
    normalise_points(pts2, n, focal, focal, ppx, ppy, q2);
    double R1[9], R2[9], tv[3];
    orc_decompose_essential_mat(E, R1, R2, tv);
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    int good[4] = {0, 0, 0, 0};
    for (int c = 0; c < 4; c++) {
        const double *Rc = (c & 1) ? R2 : R1;
        const double sg = c >= 2 ? -1.0 : 1.0;
        double P[12];
        for (int r = 0; r < 3; r++) {
            for (int k = 0; k < 3; k++)
                P[4 * r + k] = Rc[3 * r + k] * 1.0;
            P[4 * r + 3] = sg * tv[r] * 1.0;
        }
        for (int i = 0; i < n; i++) {
            double Q[4];
            triangulate_d(P0, P, q1[2 * i], q1[2 * i + 1], q2[2 * i], q2[2 * i + 1], Q);
            int ok = Q[2] * Q[3] > 0;
            /* Q.row(k) /= Q.row(3): cv::divide gives 0 for a zero divisor */
            double w = Q[3];
            for (int k = 0; k < 4; k++)
                Q[k] = w != 0 ? Q[k] / w : 0;
            ok = (Q[2] < dist) && ok;
            double z2 = P[8] * Q[0] + P[9] * Q[1] + P[10] * Q[2] + P[11] * Q[3];
            ok = (z2 > 0) && ok;
            ok = (z2 < dist) && ok;
            /* compare results are 0 / 255; bitwise_and with the caller's mask keeps its bits */
            uint8_t mv = (uint8_t)((ok ? 255 : 0) & (mask ? mask[i] : 255));
            m4[(size_t)c * n + i] = mv;
            good[c] += mv != 0;
        }
    }
    int sel;
    if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3])
        sel = 0;
    else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3])
        sel = 1;
    else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3])
        sel = 2;
    else
        sel = 3;
    memcpy(R, (sel & 1) ? R2 : R1, sizeof(double) * 9);
    for (int r = 0; r < 3; r++)
        t[r] = sel >= 2 ? -tv[r] : tv[r];
    if (mask)
        for (int i = 0; i < n; i++)
            mask[i] = m4[(size_t)sel * n + i];
    int g = good[sel];
    free(q1);
    free(q2);
    free(m4);
    return g;
}
