// vo_fivept.h -- five-point essential-matrix solver and pose recovery for the device side of the
// `mono_rotation` branch of trackingFrame2Frame() (reference src/visualOdometry.cpp:146-157:
// cv::findEssentialMat(..., RANSAC, 0.999, 1.0, mask) + cv::recoverPose(...)).
//
// Follows OpenCV 4.5.x calib3d/src/five-point.cpp (EMEstimatorCallback::runKernel / computeError,
// decomposeEssentialMat, recoverPose) and core's solvePoly / LU / JacobiSVD operation by operation:
//   5 x 9 design matrix -> null space (JacobiSVD FULL_UV completion) -> E = x X + y Y + z Z + W
//   -> ten cubic constraints (det E = 0, 2 E E^T E - tr(E E^T) E = 0) as a 10 x 20 matrix
//   -> inverse of the left 10 x 10 block (LU, partial pivoting) times the right block
//   -> 3 x 3 polynomial matrix B(z), det B(z) = degree-10 polynomial
//   -> cv::solvePoly (Durand-Kerner sweeps from the powers of 1 + i, 300 iterations)
//   -> per real root: (x, y) from the null vector of B(z) (3 x 3 SVD), E normalised to unit norm.
// VO_HD like the rest of the pose math, so tests/host_check can run the exact device code on the CPU.
// Attribution: follows the operation order of OpenCV's modules/calib3d/src/five-point.cpp (Nister's five-point algorithm,
// derived from Bo Li's implementation; BSD-style notice in that file; OpenCV is Apache-2.0) -- see NOTICE.  Written for
// this repository; no OpenCV source is included.
#pragma once

#include "vo_linalg.h"

namespace vo {

// polynomials in (x, y, z) by coefficient vectors:
//   linear [4] over (x, y, z, 1); quadric [10] over (x^2, y^2, z^2, xy, xz, yz, x, y, z, 1);
//   cubic [20] over the columns of the constraint matrix:
//     x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy | xz^2 xz x yz^2 yz y z^3 z^2 z 1
VO_HD void em_mul_ll(const double *a, const double *b, double *q)
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

// c += s * (q * l): cubic monomial o collects q[QQ[k]] * l[QL[k]] for k in [ST[o], ST[o + 1])
VO_HD void em_mul_ql_acc(const double *q, const double *l, double s, double *c)
{
    constexpr int ST[21] = {0, 1, 2, 4, 6, 8, 10, 12, 14, 17, 20, 22, 25, 27, 29, 32, 34, 35, 37, 39, 40};
    constexpr int QQ[40] = {0, 1, 0, 3, 1, 3, 0, 4, 0, 6, 1, 5, 1, 7, 3, 4, 5, 3, 6, 7,
                            2, 4, 4, 6, 8, 6, 9, 2, 5, 5, 7, 8, 7, 9, 2, 2, 8, 8, 9, 9};
    constexpr int QL[40] = {0, 1, 1, 0, 0, 1, 2, 0, 3, 0, 2, 1, 3, 1, 2, 1, 0, 3, 1, 0,
                            0, 2, 3, 2, 0, 3, 0, 1, 2, 3, 2, 1, 3, 1, 2, 3, 2, 3, 2, 3};
#pragma unroll
    for (int o = 0; o < 20; o++) {
        double t = 0;
#pragma unroll
        for (int k = ST[o]; k < ST[o + 1]; k++)
            t += q[QQ[k]] * l[QL[k]];
        c[o] += s * t;
    }
}

// core/src/lapack.cpp LUImpl<double>: A (m x m, consumed), b (m x n) -> solution in b; false if singular
template <int MM, int NN>
VO_HD bool em_lu_solve(double *A, double *b)
{
    const double eps = DBL_EPSILON * 100;
    for (int i = 0; i < MM; i++) {
        int k = i;
        for (int j = i + 1; j < MM; j++)
            if (fabs(A[j * MM + i]) > fabs(A[k * MM + i]))
                k = j;
        if (fabs(A[k * MM + i]) < eps)
            return false;
        if (k != i) {
            for (int j = i; j < MM; j++) {
                double t = A[i * MM + j];
                A[i * MM + j] = A[k * MM + j];
                A[k * MM + j] = t;
            }
            for (int j = 0; j < NN; j++) {
                double t = b[i * NN + j];
                b[i * NN + j] = b[k * NN + j];
                b[k * NN + j] = t;
            }
        }
        const double d = -1 / A[i * MM + i];
        for (int j = i + 1; j < MM; j++) {
            const double alpha = A[j * MM + i] * d;
            for (k = i + 1; k < MM; k++)
                A[j * MM + k] += alpha * A[i * MM + k];
            for (k = 0; k < NN; k++)
                b[j * NN + k] += alpha * b[i * NN + k];
        }
    }
    for (int i = MM - 1; i >= 0; i--)
        for (int j = 0; j < NN; j++) {
            double s = b[i * NN + j];
            for (int k = i + 1; k < MM; k++)
                s -= A[i * MM + k] * b[k * NN + j];
            b[i * NN + j] = s / A[i * MM + i];
        }
    return true;
}

// One Durand-Kerner sweep of cv::solvePoly over the N current root estimates (updated in place, each update
// already sees the roots updated before it).  With a compile-time N every array index is static after
// unrolling, so the 2 N root components and the coefficients stay in registers; returns maxDiff.
template <int N>
VO_HD double em_dk_sweep(const double *c0, double *re, double *im)
{
    double maxDiff = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const double pr = re[i], pi = im[i];
        double nr = c0[N], ni = 0, dr = c0[N], di = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            double tr = nr * pr - ni * pi, ti = nr * pi + ni * pr; // num = num * p + coeffs[n - j - 1]
            nr = tr + c0[N - j - 1];
            ni = ti + 0.0;
            if (j != i) {
                const double qr = pr - re[j], qi = pi - im[j];
                if (qr != 0 || qi != 0) {
                    tr = dr * qr - di * qi;
                    ti = dr * qi + di * qr;
                    dr = tr;
                    di = ti;
                }
            }
        }
        const double t = 1. / (dr * dr + di * di); // num /= denom (cv::Complex operator /)
        const double xr = (nr * dr + ni * di) * t, xi = (-nr * di + ni * dr) * t;
        re[i] = pr - xr;
        im[i] = pi - xi;
        const double a = sqrt(xr * xr + xi * xi);
        maxDiff = maxDiff > a ? maxDiff : a;
    }
    return maxDiff;
}

template <int N>
VO_HD void em_dk_run(const double *c0, double *re, double *im)
{
    double pr = 1, pi = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        re[i] = pr;
        im[i] = pi;
        const double tr = pr * 1 - pi * 1, ti = pr * 1 + pi * 1; // p *= (1, 1)
        pr = tr;
        pi = ti;
    }
    for (int iter = 0; iter < 300; iter++)
        if (em_dk_sweep<N>(c0, re, im) <= 0)
            break;
#pragma unroll
    for (int i = 0; i < N; i++)
        if (fabs(im[i]) < 1e-100)
            im[i] = 0;
#pragma unroll
    for (int i = N; i < 10; i++) { // for( ; n < n0; n++ ) roots[n+1] = roots[n]
        re[i] = re[i - 1];
        im[i] = im[i - 1];
    }
}

// cv::solvePoly on real ascending coefficients c[0..10] (maxIters = 300): Durand-Kerner from the powers of
// 1 + i; roots (re, im)[10].  Leading coefficients below DBL_EPSILON lower the degree first, like OpenCV.  The
// branch for iterates that coincide bit for bit only skips the zero factor; OpenCV additionally takes a root of
// the correction there (unreachable from the distinct starting points in practice).
VO_HD void em_solve_poly10(const double *c0, double *re, double *im)
{
    int n = 10;
    for (; n > 1; n--)
        if (fabs(c0[n]) > DBL_EPSILON)
            break;
    switch (n) {
    case 10: em_dk_run<10>(c0, re, im); break;
    case 9: em_dk_run<9>(c0, re, im); break;
    case 8: em_dk_run<8>(c0, re, im); break;
    case 7: em_dk_run<7>(c0, re, im); break;
    case 6: em_dk_run<6>(c0, re, im); break;
    case 5: em_dk_run<5>(c0, re, im); break;
    case 4: em_dk_run<4>(c0, re, im); break;
    case 3: em_dk_run<3>(c0, re, im); break;
    case 2: em_dk_run<2>(c0, re, im); break;
    default: em_dk_run<1>(c0, re, im); break;
    }
}

// ascending-coefficient polynomial product r = a * b
template <int NA, int NB>
VO_HD void em_pmul(const double *a, const double *b, double *r)
{
    for (int i = 0; i < NA + NB - 1; i++)
        r[i] = 0;
    for (int i = 0; i < NA; i++)
        for (int j = 0; j < NB; j++)
            r[i + j] += a[i] * b[j];
}

// EMEstimatorCallback::runKernel: q1 / q2 = 5 normalised correspondences (x, y); Es [<= 10][9] row-major.
// Returns the number of models.
VO_HD_NOINLINE int five_point_solve(const double *q1, const double *q2, double *Es)
{
    double At[9 * 9], W5[5];
    for (int i = 0; i < 81; i++)
        At[i] = 0;
    for (int i = 0; i < 5; i++) {
        const double x1 = q1[2 * i], y1 = q1[2 * i + 1], x2 = q2[2 * i], y2 = q2[2 * i + 1];
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
    jacobi_svd<9, 5, false, 1, 9>(At, W5, nullptr);
    const double *X = At + 9 * 5, *Y = At + 9 * 6, *Z = At + 9 * 7, *Wv = At + 9 * 8;

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
            em_mul_ll(E[3 * i], E[3 * j], d);
            for (int k = 1; k < 3; k++) {
                em_mul_ll(E[3 * i + k], E[3 * j + k], q);
                for (int o = 0; o < 10; o++)
                    d[o] += q[o];
            }
        }
    for (int o = 0; o < 10; o++)
        tr[o] = EEt[0][o] + EEt[4][o] + EEt[8][o];

    double A[10 * 20];
    for (int i = 0; i < 200; i++)
        A[i] = 0;
    {
        const int perm[6][4] = {{0, 4, 8, 1}, {0, 5, 7, -1}, {1, 5, 6, 1}, {1, 3, 8, -1}, {2, 3, 7, 1}, {2, 4, 6, -1}};
        for (int p = 0; p < 6; p++) {
            em_mul_ll(E[perm[p][1]], E[perm[p][2]], q);
            em_mul_ql_acc(q, E[perm[p][0]], (double)perm[p][3], A);
        }
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double *row = A + 20 * (1 + 3 * i + j);
            for (int k = 0; k < 3; k++)
                em_mul_ql_acc(EEt[3 * i + k], E[3 * k + j], 2.0, row);
            em_mul_ql_acc(tr, E[3 * i + j], -1.0, row);
        }

    // rows 4..9 of inv(A[:, 0:10]) * A[:, 10:20]
    double L[100], inv[100], G[60];
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 10; j++) {
            L[10 * i + j] = A[20 * i + j];
            inv[10 * i + j] = i == j ? 1.0 : 0.0;
        }
    if (!em_lu_solve<10, 10>(L, inv))
        for (int i = 0; i < 100; i++)
            inv[i] = 0;
    for (int i = 4; i < 10; i++)
        for (int j = 0; j < 10; j++) {
            double s = 0;
            for (int k = 0; k < 10; k++)
                s += inv[10 * i + k] * A[20 * k + 10 + j];
            G[10 * (i - 4) + j] = s;
        }

    double b[3 * 13];
    for (int i = 0; i < 3; i++) {
        const double *r1 = G + 10 * (2 * i), *r2 = G + 10 * (2 * i + 1);
        double row1[13], row2[13];
        for (int k = 0; k < 13; k++)
            row1[k] = row2[k] = 0;
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

    double c[11];
    for (int k = 0; k < 11; k++)
        c[k] = 0;
    {
        double p0[3][4], p1[3][4], p2[3][5];
        for (int i = 0; i < 3; i++) {
            for (int k = 0; k < 4; k++) {
                p0[i][k] = b[13 * i + 3 - k];
                p1[i][k] = b[13 * i + 7 - k];
            }
            for (int k = 0; k < 5; k++)
                p2[i][k] = b[13 * i + 12 - k];
        }
        const int cof[3][2] = {{1, 2}, {0, 2}, {0, 1}};
        for (int i = 0; i < 3; i++) {
            const int r = cof[i][0], s = cof[i][1];
            double m1[7], m2[7], minor[7], term[11];
            em_pmul<4, 4>(p0[r], p1[s], m1);
            em_pmul<4, 4>(p1[r], p0[s], m2);
            for (int k = 0; k < 7; k++)
                minor[k] = m1[k] - m2[k];
            em_pmul<5, 7>(p2[i], minor, term);
            for (int k = 0; k < 11; k++)
                c[k] += (i == 1 ? -1.0 : 1.0) * term[k];
        }
    }

    double re[10], im[10];
    em_solve_poly10(c, re, im);

    int count = 0;
    for (int i = 0; i < 10; i++) {
        if (fabs(im[i]) > 1e-10)
            continue;
        const double z1 = re[i], z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double bzt[9], w[3], vt[9];
        for (int j = 0; j < 3; j++) { // stored transposed for the one-sided Jacobi: bzt[col][row]
            const double *br = b + 13 * j;
            bzt[0 * 3 + j] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
            bzt[1 * 3 + j] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
            bzt[2 * 3 + j] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
        }
        jacobi_svd<3, 3, true>(bzt, w, vt);
        if (fabs(vt[8]) < 1e-10)
            continue;
        const double x = vt[6] / vt[8], y = vt[7] / vt[8];
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

// EMEstimatorCallback::computeError: Sampson distance in f64, stored as f32
VO_HD float em_sampson_error(const double *E, double x1x, double x1y, double x2x, double x2y)
{
    double Ex1[3], Etx2[3];
    for (int r = 0; r < 3; r++) {
        Ex1[r] = E[3 * r] * x1x + E[3 * r + 1] * x1y + E[3 * r + 2] * 1.;
        Etx2[r] = E[r] * x2x + E[3 + r] * x2y + E[6 + r] * 1.;
    }
    const double x2tEx1 = x2x * Ex1[0] + x2y * Ex1[1] + 1. * Ex1[2];
    const double a = Ex1[0] * Ex1[0], b = Ex1[1] * Ex1[1], c = Etx2[0] * Etx2[0], d = Etx2[1] * Etx2[1];
    return (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
}

VO_HD double em_det3(const double *M)
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

VO_HD void em_mat3_mul(const double *A, const double *B, double *C)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++)
                s += A[3 * i + k] * B[3 * k + j];
            C[3 * i + j] = s;
        }
}

// five-point.cpp decomposeEssentialMat
VO_HD void em_decompose(const double *E, double *R1, double *R2, double *t)
{
    double Ut[9], w[3], Vt[9], U[9], T[9];
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++)
            Ut[i * 3 + k] = E[k * 3 + i];
    jacobi_svd<3, 3, true>(Ut, w, Vt);
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++)
            U[k * 3 + i] = Ut[i * 3 + k];
    if (em_det3(U) < 0)
        for (int i = 0; i < 9; i++)
            U[i] *= -1.;
    if (em_det3(Vt) < 0)
        for (int i = 0; i < 9; i++)
            Vt[i] *= -1.;
    const double Wm[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    em_mat3_mul(U, Wm, T);
    em_mat3_mul(T, Vt, R1);
    em_mat3_mul(U, Wt, T);
    em_mat3_mul(T, Vt, R2);
    for (int i = 0; i < 3; i++)
        t[i] = U[3 * i + 2] * 1.0;
}

// recoverPose's cheirality test of one correspondence against P0 = [I | 0], P = [R | t] (3 x 4 row-major):
// triangulate (DLT, f64), depth positive and below `dist` in both cameras
VO_HD bool em_cheirality(const double *P, double x0, double y0, double x1, double y1, double dist)
{
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    double At[16], w[4], vt[16];
    const double x[2] = {x0, x1}, y[2] = {y0, y1};
    for (int j = 0; j < 2; j++) {
        const double *Pj = j == 0 ? P0 : P;
        for (int k = 0; k < 4; k++) {
            At[k * 4 + (j * 2 + 0)] = x[j] * Pj[8 + k] - Pj[k];
            At[k * 4 + (j * 2 + 1)] = y[j] * Pj[8 + k] - Pj[4 + k];
        }
    }
    jacobi_svd<4, 4, true>(At, w, vt);
    double Q[4] = {vt[12], vt[13], vt[14], vt[15]};
    bool ok = Q[2] * Q[3] > 0;
    const double ww = Q[3];
    for (int k = 0; k < 4; k++)
        Q[k] = ww != 0 ? Q[k] / ww : 0;
    ok = (Q[2] < dist) && ok;
    const double z2 = P[8] * Q[0] + P[9] * Q[1] + P[10] * Q[2] + P[11] * Q[3];
    ok = (z2 > 0) && ok;
    ok = (z2 < dist) && ok;
    return ok;
}

} // namespace vo
