/*
 * orc_geom.c -- CPU ORACLE (test infrastructure, NOT product code).  PARITY UNPINNED, see
 * vo_oracle.h.
 *
 * Small dense f64 linear algebra + projective geometry restated from OpenCV 4.5.x:
 *   core/src/lapack.cpp    JacobiSVDImpl_<double>, SVBkSbImpl_, _SVDcompute, solve/invert(SVD)
 *   calib3d/src/triangulate.cpp   icvTriangulatePoints
 *   calib3d/src/fundam.cpp        convertPointsFromHomogeneous (cn == 4, f32)
 *   calib3d/src/calibration.cpp   cvRodrigues2, cvProjectPoints2 (zero distortion)
 * Call sites in the reference: src/main.cpp:169-171, src/visualOdometry.cpp:176-178,188.
 */
#include "orc_internal.h"

#include <float.h>
#include <math.h>
#include <string.h>

/* ---- cv::RNG ------------------------------------------------------------------------ */
uint32_t orc_rng_next(uint64_t *state)
{
    *state = (uint64_t)(uint32_t)(*state) * 4164903690U + (uint32_t)(*state >> 32);
    return (uint32_t)(*state);
}

/* OpenCV calls std::hypot here.  Its last-ulp behaviour is libm specific (glibc changed it in 2.35,
 * device libms differ again), and EPnP's 5-point null space amplifies a 1-ulp difference in a
 * Givens angle into ~1e-6 px hypothesis differences.  The restatement therefore pins the
 * definition to three correctly rounded IEEE operations: sqrt(fma(a, a, b*b)). */
static inline double orc_hypot(double a, double b) { return sqrt(fma(a, a, b * b)); }

/* ---- JacobiSVDImpl_<double>(At, astep, W, Vt, vstep, m, n, n1, DBL_MIN, DBL_EPSILON*10) ---
 * At: n rows of length m (row i = column i of A). On exit rows of At are the left singular
 * vectors (first n1 rows normalised), W sorted descending, Vt rows = right singular vectors. */
void orc_jacobi_svd(double *At, int astep, double *_W, double *Vt, int vstep, int m, int n, int n1)
{
    const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
    double W[ORC_SVD_MAXN];
    int i, j, k, iter, max_iter = m > 30 ? m : 30;
    double c, s, sd;

    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            double t = At[i * astep + k];
            sd += t * t;
        }
        W[i] = sd;
        if (Vt) {
            for (k = 0; k < n; k++)
                Vt[i * vstep + k] = 0;
            Vt[i * vstep + i] = 1;
        }
    }

    for (iter = 0; iter < max_iter; iter++) {
        int changed = 0;
        for (i = 0; i < n - 1; i++)
            for (j = i + 1; j < n; j++) {
                double *Ai = At + i * astep, *Aj = At + j * astep;
                double a = W[i], p = 0, b = W[j];
                for (k = 0; k < m; k++)
                    p += Ai[k] * Aj[k];
                if (fabs(p) <= eps * sqrt(a * b))
                    continue;
                p *= 2;
                double beta = a - b, gamma = orc_hypot(p, beta);
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (k = 0; k < m; k++) {
                    double t0 = c * Ai[k] + s * Aj[k];
                    double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0;
                    Aj[k] = t1;
                    a += t0 * t0;
                    b += t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = 1;
                if (Vt) {
                    double *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;
                    for (k = 0; k < n; k++) {
                        double t0 = c * Vi[k] + s * Vj[k];
                        double t1 = -s * Vi[k] + c * Vj[k];
                        Vi[k] = t0;
                        Vj[k] = t1;
                    }
                }
            }
        if (!changed)
            break;
    }

    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            double t = At[i * astep + k];
            sd += t * t;
        }
        W[i] = sqrt(sd);
    }

    for (i = 0; i < n - 1; i++) {
        j = i;
        for (k = i + 1; k < n; k++)
            if (W[j] < W[k])
                j = k;
        if (i != j) {
            double t = W[i];
            W[i] = W[j];
            W[j] = t;
            if (Vt) {
                for (k = 0; k < m; k++) {
                    t = At[i * astep + k];
                    At[i * astep + k] = At[j * astep + k];
                    At[j * astep + k] = t;
                }
                for (k = 0; k < n; k++) {
                    t = Vt[i * vstep + k];
                    Vt[i * vstep + k] = Vt[j * vstep + k];
                    Vt[j * vstep + k] = t;
                }
            }
        }
    }
    for (i = 0; i < n; i++)
        _W[i] = W[i];
    if (!Vt)
        return;

    uint64_t rng = 0x12345678;
    for (i = 0; i < n1; i++) {
        sd = i < n ? W[i] : 0;
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            /* zero singular value: random vector, orthogonalised against the previous ones */
            const double val0 = 1. / m;
            for (k = 0; k < m; k++) {
                double val = (orc_rng_next(&rng) & 256) != 0 ? val0 : -val0;
                At[i * astep + k] = val;
            }
            for (iter = 0; iter < 2; iter++) {
                for (j = 0; j < i; j++) {
                    sd = 0;
                    for (k = 0; k < m; k++)
                        sd += At[i * astep + k] * At[j * astep + k];
                    double asum = 0;
                    for (k = 0; k < m; k++) {
                        double t = At[i * astep + k] - sd * At[j * astep + k];
                        At[i * astep + k] = t;
                        asum += fabs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (k = 0; k < m; k++)
                        At[i * astep + k] *= asum;
                }
            }
            sd = 0;
            for (k = 0; k < m; k++) {
                double t = At[i * astep + k];
                sd += t * t;
            }
            sd = sqrt(sd);
        }
        s = sd > minval ? 1 / sd : 0.;
        for (k = 0; k < m; k++)
            At[i * astep + k] *= s;
    }
}

/* cv::SVD::compute(A, w, u, vt) for m >= n (row-major A m x n; u m x n; vt n x n) */
void orc_svd(const double *A, int m, int n, double *w, double *u, double *vt)
{
    double At[ORC_SVD_MAXN * ORC_SVD_MAXM];
    for (int i = 0; i < n; i++)
        for (int k = 0; k < m; k++)
            At[i * m + k] = A[k * n + i];
    orc_jacobi_svd(At, m, w, vt, n, m, n, n);
    if (u)
        for (int i = 0; i < n; i++)
            for (int k = 0; k < m; k++)
                u[k * n + i] = At[i * m + k];
}

/* cv::solve(A, b, x, DECOMP_SVD), one right-hand side: JacobiSVD on At + SVBkSb (uT, vT) */
void orc_solve_svd(const double *A, int m, int n, const double *b, double *x)
{
    double At[ORC_SVD_MAXN * ORC_SVD_MAXM], w[ORC_SVD_MAXN], vt[ORC_SVD_MAXN * ORC_SVD_MAXN];
    for (int i = 0; i < n; i++)
        for (int k = 0; k < m; k++)
            At[i * m + k] = A[k * n + i];
    orc_jacobi_svd(At, m, w, vt, n, m, n, n);
    double threshold = 0;
    for (int i = 0; i < n; i++)
        x[i] = 0;
    for (int i = 0; i < n; i++)
        threshold += w[i];
    threshold *= DBL_EPSILON * 2;
    for (int i = 0; i < n; i++) {
        double wi = w[i];
        if (fabs(wi) <= threshold)
            continue;
        wi = 1 / wi;
        double s = 0;
        for (int j = 0; j < m; j++)
            s += At[i * m + j] * b[j];
        s *= wi;
        for (int j = 0; j < n; j++)
            x[j] = x[j] + s * vt[i * n + j];
    }
}

/* cv::invert(A, Ainv, DECOMP_SVD) for square n x n: SVD::compute + backSubst(w,u,vt,Mat()) */
void orc_invert_svd(const double *A, int n, double *Ainv)
{
    double w[ORC_SVD_MAXN], u[ORC_SVD_MAXN * ORC_SVD_MAXN], vt[ORC_SVD_MAXN * ORC_SVD_MAXN];
    double buffer[ORC_SVD_MAXN];
    orc_svd(A, n, n, w, u, vt);
    double threshold = 0;
    for (int i = 0; i < n * n; i++)
        Ainv[i] = 0;
    for (int i = 0; i < n; i++)
        threshold += w[i];
    threshold *= DBL_EPSILON * 2;
    for (int i = 0; i < n; i++) {
        double wi = w[i];
        if (fabs(wi) <= threshold)
            continue;
        wi = 1 / wi;
        for (int j = 0; j < n; j++)
            buffer[j] = u[j * n + i] * wi;
        /* MatrAXPY(n, nb, buffer, 0, v, vdelta1, x, ldx): x[r][:] += v_i[r] * buffer[:] */
        for (int r = 0; r < n; r++) {
            double sv = vt[i * n + r];
            for (int j = 0; j < n; j++)
                Ainv[r * n + j] = Ainv[r * n + j] + sv * buffer[j];
        }
    }
}

/* ---- triangulate.cpp icvTriangulatePoints + cv::triangulatePoints (f32 in -> f32 out) ---- */
void orc_triangulate_points(const float *P_l, const float *P_r, const float *pts_l,
                            const float *pts_r, int n, float *points4d)
{
    const float *P[2] = {P_l, P_r};
    const float *pts[2] = {pts_l, pts_r};
    for (int i = 0; i < n; i++) {
        double A[16], w[4], u[16], vt[16];
        for (int j = 0; j < 2; j++) {
            double x = pts[j][2 * i], y = pts[j][2 * i + 1];
            for (int k = 0; k < 4; k++) {
                A[(j * 2 + 0) * 4 + k] = x * (double)P[j][2 * 4 + k] - (double)P[j][0 * 4 + k];
                A[(j * 2 + 1) * 4 + k] = y * (double)P[j][2 * 4 + k] - (double)P[j][1 * 4 + k];
            }
        }
        orc_svd(A, 4, 4, w, u, vt);
        points4d[0 * n + i] = (float)vt[3 * 4 + 0];
        points4d[1 * n + i] = (float)vt[3 * 4 + 1];
        points4d[2 * n + i] = (float)vt[3 * 4 + 2];
        points4d[3 * n + i] = (float)vt[3 * 4 + 3];
    }
}

/* fundam.cpp convertPointsFromHomogeneous, 4-channel float branch */
void orc_convert_points_from_homogeneous(const float *p, int n, float *out)
{
    for (int i = 0; i < n; i++) {
        float scale = p[4 * i + 3] != 0.f ? 1.f / p[4 * i + 3] : 1.f;
        out[3 * i] = p[4 * i] * scale;
        out[3 * i + 1] = p[4 * i + 1] * scale;
        out[3 * i + 2] = p[4 * i + 2] * scale;
    }
}

/* main.cpp:170-171: triangulatePoints -> .t() -> convertPointsFromHomogeneous */
void orc_triangulate(const float *P_l, const float *P_r, const float *pts_l, const float *pts_r,
                     int n, float *xyz)
{
    if (n <= 0)
        return;
    float *p4 = (float *)malloc(sizeof(float) * 4 * (size_t)n);
    float *p4t = (float *)malloc(sizeof(float) * 4 * (size_t)n);
    orc_triangulate_points(P_l, P_r, pts_l, pts_r, n, p4);
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 4; k++)
            p4t[4 * i + k] = p4[k * n + i];
    orc_convert_points_from_homogeneous(p4t, n, xyz);
    free(p4);
    free(p4t);
}

/* ---- calibration.cpp cvRodrigues2 ------------------------------------------------------- */
void orc_rodrigues_vec2mat(const double *rv, double *R, double *J /* 3x9 or NULL */)
{
    double rx = rv[0], ry = rv[1], rz = rv[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        memset(R, 0, sizeof(double) * 9);
        R[0] = R[4] = R[8] = 1;
        if (J) {
            memset(J, 0, sizeof(double) * 27);
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    double c = cos(theta), s = sin(theta), c1 = 1. - c;
    double itheta = theta ? 1. / theta : 0.;
    rx *= itheta;
    ry *= itheta;
    rz *= itheta;
    double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    /* R = cos(theta)*I + (1 - cos(theta))*r*rT + sin(theta)*[r_x] */
    for (int k = 0; k < 9; k++)
        R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    if (J) {
        double drrt[27] = {rx + rx, ry, rz, ry, 0,       0,  rz, 0,  0,       0, rx, 0,  rx, ry + ry,
                           rz,      0,  rz, 0,  0,       0,  rx, 0,  0,       ry, rx, ry, rz + rz};
        const double d_r_x_[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0,
                                   0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
        for (int i = 0; i < 3; i++) {
            double ri = i == 0 ? rx : i == 1 ? ry : rz;
            double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            double a3 = (c - s * itheta) * ri, a4 = s * itheta;
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] +
                               a4 * d_r_x_[i * 9 + k];
        }
    }
}

void orc_rodrigues_mat2vec(const double *Rin, double *rv)
{
    double R[9], w[3], u[9], vt[9];
    /* checkRange(R, true, NULL, -100, 100) */
    for (int k = 0; k < 9; k++)
        if (!(Rin[k] > -100 && Rin[k] < 100)) {
            rv[0] = rv[1] = rv[2] = 0;
            return;
        }
    orc_svd(Rin, 3, 3, w, u, vt);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[i * 3 + j] = u[i * 3 + 0] * vt[0 * 3 + j] + u[i * 3 + 1] * vt[1 * 3 + j] +
                           u[i * 3 + 2] * vt[2 * 3 + j];
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        double t;
        if (c > 0)
            rx = ry = rz = 0;
        else {
            t = (R[0] + 1) * 0.5;
            rx = sqrt(t > 0. ? t : 0.);
            t = (R[4] + 1) * 0.5;
            ry = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            rz = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0))
                rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta;
            ry *= theta;
            rz *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth;
        ry *= vth;
        rz *= vth;
    }
    rv[0] = rx;
    rv[1] = ry;
    rv[2] = rz;
}

/* ---- calibration.cpp cvProjectPoints2, distortion == 0 (all k terms are exact zeros) ----- */
void orc_project_points_d(const double *M /* n x 3 */, int n, const double *rvec,
                          const double *tvec, const double *A /* fx fy cx cy */, double *m,
                          double *dpdr /* 2n x 3, row stride jstride */,
                          double *dpdt /* 2n x 3 */, int jstride)
{
    double R[9], dRdr[27];
    orc_rodrigues_vec2mat(rvec, R, (dpdr ? dRdr : NULL));
    const double fx = A[0], fy = A[1], cx = A[2], cy = A[3];
    for (int i = 0; i < n; i++) {
        double X = M[3 * i], Y = M[3 * i + 1], Z = M[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + tvec[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + tvec[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + tvec[2];
        z = z ? 1. / z : 1;
        x *= z;
        y *= z;
        /* cdist = icdist2 = 1, tilt = identity -> xd = x, yd = y */
        m[2 * i] = x * fx + cx;
        m[2 * i + 1] = y * fy + cy;
        if (dpdt) {
            double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
            for (int j = 0; j < 3; j++) {
                dpdt[(2 * i) * jstride + j] = fx * dxdt[j];
                dpdt[(2 * i + 1) * jstride + j] = fy * dydt[j];
            }
        }
        if (dpdr) {
            double dx0dr[3] = {X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2],
                               X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11],
                               X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20]};
            double dy0dr[3] = {X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5],
                               X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14],
                               X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23]};
            double dz0dr[3] = {X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8],
                               X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17],
                               X * dRdr[24] + Y * dRdr[25] + Z * dRdr[26]};
            for (int j = 0; j < 3; j++) {
                double dxdr = z * (dx0dr[j] - x * dz0dr[j]);
                double dydr = z * (dy0dr[j] - y * dz0dr[j]);
                dpdr[(2 * i) * jstride + j] = fx * dxdr;
                dpdr[(2 * i + 1) * jstride + j] = fy * dydr;
            }
        }
    }
}

/* cv::projectPoints with 32F object points -> 32F image points */
void orc_project_points(const float *xyz, int n, const double *rvec, const double *tvec,
                        const float *K, float *uv_out)
{
    if (n <= 0)
        return;
    double A[4] = {(double)K[0], (double)K[4], (double)K[2], (double)K[5]};
    double *M = (double *)malloc(sizeof(double) * 3 * (size_t)n);
    double *m = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    for (int i = 0; i < 3 * n; i++)
        M[i] = (double)xyz[i];
    orc_project_points_d(M, n, rvec, tvec, A, m, NULL, NULL, 0);
    for (int i = 0; i < 2 * n; i++)
        uv_out[i] = (float)m[i];
    free(M);
    free(m);
}
