// vo_epnp.h -- 5-point EPnP minimal solver used as the RANSAC hypothesis kernel.
//
// Drop-in for what cv::solvePnPRansac runs per iteration in the reference
// (visualOdometry.cpp:176-178 -> PnPRansacCallback::runKernel -> solvePnP(SOLVEPNP_EPNP) on a
// 5-point subset): control points by PCA, barycentric coordinates, 10x12 M, null space of M^T M,
// three beta approximations each refined by 5 Gauss-Newton steps, Horn absolute orientation,
// best-of-three by reprojection error; then R -> rvec.  VO_HD so tests/host_check can run the very
// same code on the CPU (unit test only, never a product fallback).
// Attribution: follows the operation order of OpenCV's modules/calib3d/src/epnp.cpp (EPnP, Lepetit / Moreno-Noguer /
// Fua; BSD-style notice in that file; OpenCV is Apache-2.0) so that results match the reference's solvePnPRansac calls --
// see NOTICE.  Written for this repository; no OpenCV source is included.
#pragma once

#include "vo_linalg.h"

#ifndef VO_EPNP_STAMP
#define VO_EPNP_STAMP(i)
#endif

namespace vo {

struct Epnp5 {
    double uc, vc, fu, fv;
    double pws[15], us[10], alphas[20], pcs[15];
    double cws[4][3], ccs[4][3];
};

VO_HD double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
VO_HD double dist2(const double *p1, const double *p2)
{
    return (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) +
           (p1[2] - p2[2]) * (p1[2] - p2[2]);
}

// Householder QR least squares, 6x4 (the EPnP authors' qr_solve incl. its pivot-scan behaviour)
VO_HD void qr_solve_6x4(double *pA, double *pb, double *pX)
{
    const int nr = 6, nc = 4;
    double A1[4], A2[4];
#pragma unroll
    for (int k = 0; k < nc; k++) {
        double eta = fabs(pA[k * nc + k]);
#pragma unroll
        for (int i = k + 1; i < nr; i++) { // scans rows k .. nr-2
            double elt = fabs(pA[(i - 1) * nc + k]);
            if (eta < elt)
                eta = elt;
        }
        if (eta == 0) {
            return;
        }
        double sum2 = 0.0, inv_eta = 1. / eta;
#pragma unroll
        for (int i = k; i < nr; i++) {
            pA[i * nc + k] *= inv_eta;
            sum2 += pA[i * nc + k] * pA[i * nc + k];
        }
        double sigma = sqrt(sum2);
        if (pA[k * nc + k] < 0)
            sigma = -sigma;
        pA[k * nc + k] += sigma;
        A1[k] = sigma * pA[k * nc + k];
        A2[k] = -eta * sigma;
#pragma unroll
        for (int j = k + 1; j < nc; j++) {
            double sum = 0;
#pragma unroll
            for (int i = k; i < nr; i++)
                sum += pA[i * nc + k] * pA[i * nc + j];
            double tau = sum / A1[k];
#pragma unroll
            for (int i = k; i < nr; i++)
                pA[i * nc + j] -= tau * pA[i * nc + k];
        }
    }
#pragma unroll
    for (int j = 0; j < nc; j++) { // b <- Qt b
        double tau = 0;
#pragma unroll
        for (int i = j; i < nr; i++)
            tau += pA[i * nc + j] * pb[i];
        tau /= A1[j];
#pragma unroll
        for (int i = j; i < nr; i++)
            pb[i] -= tau * pA[i * nc + j];
    }
    pX[nc - 1] = pb[nc - 1] / A2[nc - 1]; // X = R^-1 b
#pragma unroll
    for (int i = nc - 2; i >= 0; i--) {
        double sum = 0;
#pragma unroll
        for (int j = i + 1; j < nc; j++)
            sum += pA[i * nc + j] * pX[j];
        pX[i] = (pb[i] - sum) / A2[i];
    }
}

VO_HD void epnp_gauss_newton(const double *L, const double *rho, double *betas)
{
    double a[24], b[6], x[4] = {0, 0, 0, 0};
    for (int it = 0; it < 5; it++) {
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const double *rowL = L + i * 10;
            double *rowA = a + i * 4;
            rowA[0] = 2 * rowL[0] * betas[0] + rowL[1] * betas[1] + rowL[3] * betas[2] + rowL[6] * betas[3];
            rowA[1] = rowL[1] * betas[0] + 2 * rowL[2] * betas[1] + rowL[4] * betas[2] + rowL[7] * betas[3];
            rowA[2] = rowL[3] * betas[0] + rowL[4] * betas[1] + 2 * rowL[5] * betas[2] + rowL[8] * betas[3];
            rowA[3] = rowL[6] * betas[0] + rowL[7] * betas[1] + rowL[8] * betas[2] + 2 * rowL[9] * betas[3];
            b[i] = rho[i] - (rowL[0] * betas[0] * betas[0] + rowL[1] * betas[0] * betas[1] +
                             rowL[2] * betas[1] * betas[1] + rowL[3] * betas[0] * betas[2] +
                             rowL[4] * betas[1] * betas[2] + rowL[5] * betas[2] * betas[2] +
                             rowL[6] * betas[0] * betas[3] + rowL[7] * betas[1] * betas[3] +
                             rowL[8] * betas[2] * betas[3] + rowL[9] * betas[3] * betas[3]);
        }
        qr_solve_6x4(a, b, x);
#pragma unroll
        for (int i = 0; i < 4; i++)
            betas[i] += x[i];
    }
}

// camera-frame control points from betas, point cloud, sign, Horn alignment, reprojection error
template <int S>
VO_HD double epnp_compute_R_and_t(Epnp5 &e, const double *ut, const double *betas, double *R /*9*/,
                                  double *t)
{
    const int n = 5;
#pragma unroll
    for (int i = 0; i < 4; i++)
        e.ccs[i][0] = e.ccs[i][1] = e.ccs[i][2] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double *v = ut + 12 * (11 - i) * S;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < 3; k++)
                e.ccs[j][k] += betas[i] * v[(3 * j + k) * S];
    }
#pragma unroll
    for (int i = 0; i < n; i++) {
        const double *a = e.alphas + 4 * i;
        double *pc = e.pcs + 3 * i;
#pragma unroll
        for (int j = 0; j < 3; j++)
            pc[j] = a[0] * e.ccs[0][j] + a[1] * e.ccs[1][j] + a[2] * e.ccs[2][j] + a[3] * e.ccs[3][j];
    }
    if (e.pcs[2] < 0.0) { // solve_for_sign
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                e.ccs[i][j] = -e.ccs[i][j];
#pragma unroll
        for (int i = 0; i < 3 * n; i++)
            e.pcs[i] = -e.pcs[i];
    }
    // estimate_R_and_t
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < n; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            pc0[j] += e.pcs[3 * i + j];
            pw0[j] += e.pws[3 * i + j];
        }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        pc0[j] /= n;
        pw0[j] /= n;
    }
    double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < n; i++) {
        const double *pc = e.pcs + 3 * i, *pw = e.pws + 3 * i;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
            abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
            abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
        }
    }
    double At[9], wd[3], vt[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++)
            At[i * 3 + k] = abt[k * 3 + i];
    jacobi_svd<3, 3, true>(At, wd, vt);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) // (U V^T)[i][j], U[i][k] = At[k][i], V[j][k] = vt[k][j]
            R[i * 3 + j] = At[0 * 3 + i] * vt[0 * 3 + j] + At[1 * 3 + i] * vt[1 * 3 + j] +
                           At[2 * 3 + i] * vt[2 * 3 + j];
    const double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] -
                       R[2] * R[4] * R[6] - R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
    if (det < 0) {
        R[6] = -R[6];
        R[7] = -R[7];
        R[8] = -R[8];
    }
    t[0] = pc0[0] - dot3(R + 0, pw0);
    t[1] = pc0[1] - dot3(R + 3, pw0);
    t[2] = pc0[2] - dot3(R + 6, pw0);
    // reprojection_error
    double sum2 = 0.0;
#pragma unroll
    for (int i = 0; i < n; i++) {
        const double *pw = e.pws + 3 * i;
        double Xc = dot3(R + 0, pw) + t[0];
        double Yc = dot3(R + 3, pw) + t[1];
        double inv_Zc = 1.0 / (dot3(R + 6, pw) + t[2]);
        double ue = e.uc + e.fu * Xc * inv_Zc;
        double ve = e.vc + e.fv * Yc * inv_Zc;
        double u = e.us[2 * i], v = e.us[2 * i + 1];
        sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / n;
}

// xyz5[15], uv5[10]: the subset (f32, as RANSAC's getSubset copies them); Kf: 3x3 f32 row-major.
// Outputs rvec[3], tvec[3] (f64).
// S / ut_mem: the 12 x 12 matrix M^T M (and its 12 column norms) is the one object of this solver
// that does not fit in registers next to everything else; the caller provides (144 + 12) * S doubles
// for it, element idx at ut_mem[idx * S] (device: lane-interleaved LDS, S = workgroup size; host: a
// private array, S = 1).
// The solver in three pieces, so that the 12 x 12 SVD between the first two can also be run by a whole DPP row
// (vo_svd_wide.h, epnp_wide_kernel): epnp5_prepare -- through M^T M in `ut`; the SVD; epnp5_finish -- the rest.
#define VO_UT(idx) ut[(idx) * S]
template <int S>
VO_HD void epnp5_prepare(const float *xyz5, const float *uv5, const float *Kf, Epnp5 &e, double *ut)
{
    const int n = 5;
    VO_EPNP_STAMP(0);
    e.fu = (double)Kf[0];
    e.fv = (double)Kf[4];
    e.uc = (double)Kf[2];
    e.vc = (double)Kf[5];
    const double ifx = 1. / e.fu, ify = 1. / e.fv;
#pragma unroll
    for (int i = 0; i < n; i++) {
        // undistortPoints (zero distortion) -> f32 normalised coords -> back to pixels in f64
        double x = ((double)uv5[2 * i] - e.uc) * ifx, y = ((double)uv5[2 * i + 1] - e.vc) * ify;
        float xn = (float)x, yn = (float)y;
        e.pws[3 * i] = xyz5[3 * i];
        e.pws[3 * i + 1] = xyz5[3 * i + 1];
        e.pws[3 * i + 2] = xyz5[3 * i + 2];
        e.us[2 * i] = xn * e.fu + e.uc;
        e.us[2 * i + 1] = yn * e.fv + e.vc;
    }
    // ---- choose_control_points
    e.cws[0][0] = e.cws[0][1] = e.cws[0][2] = 0;
#pragma unroll
    for (int i = 0; i < n; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            e.cws[0][j] += e.pws[3 * i + j];
#pragma unroll
    for (int j = 0; j < 3; j++)
        e.cws[0][j] /= n;
    {
        double PW0[15], ptp[9], dc[3], vt[9];
#pragma unroll
        for (int i = 0; i < n; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                PW0[3 * i + j] = e.pws[3 * i + j] - e.cws[0][j];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = i; j < 3; j++) {
                double s = 0;
#pragma unroll
                for (int k = 0; k < n; k++)
                    s += PW0[k * 3 + i] * PW0[k * 3 + j];
                ptp[i * 3 + j] = s;
            }
        ptp[3] = ptp[1];
        ptp[6] = ptp[2];
        ptp[7] = ptp[5];
        // symmetric: At == A^T == A; rows of At after the SVD = U^T
        jacobi_svd<3, 3, true>(ptp, dc, vt);
#pragma unroll
        for (int i = 1; i < 4; i++) {
            double k = sqrt(dc[i - 1] / n);
#pragma unroll
            for (int j = 0; j < 3; j++)
                e.cws[i][j] = e.cws[0][j] + k * ptp[3 * (i - 1) + j];
        }
    }
    // ---- compute_barycentric_coordinates
    {
        double cc[9], ci[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 1; j < 4; j++)
                cc[3 * i + j - 1] = e.cws[j][i] - e.cws[0][i];
        invert_svd<3>(cc, ci);
#pragma unroll
        for (int i = 0; i < n; i++) {
            const double *pi = e.pws + 3 * i;
            double *a = e.alphas + 4 * i;
#pragma unroll
            for (int j = 0; j < 3; j++)
                a[1 + j] = ci[3 * j] * (pi[0] - e.cws[0][0]) + ci[3 * j + 1] * (pi[1] - e.cws[0][1]) +
                           ci[3 * j + 2] * (pi[2] - e.cws[0][2]);
            a[0] = 1.0f - a[1] - a[2] - a[3];
        }
    }
    // ---- M (10 x 12), M^T M, its eigen-basis through the SVD
    // M^T M is accumulated point by point (rows 2p and 2p+1 of M at a time): for every (i, j) the
    // partial sums are formed in the same order k = 0 .. 9 as the plain triple loop, so the result
    // is bit-identical, but M itself (120 doubles) never has to exist.
    VO_EPNP_STAMP(1);
    {
#pragma unroll
        for (int i = 0; i < 12; i++)
#pragma unroll
            for (int j = i; j < 12; j++)
                VO_UT(i * 12 + j) = 0;
#pragma unroll 1
        for (int p = 0; p < n; p++) {
            const double *as = e.alphas + 4 * p;
            const double u = e.us[2 * p], v = e.us[2 * p + 1];
            double M1[12], M2[12];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                M1[3 * q] = as[q] * e.fu;
                M1[3 * q + 1] = 0.0;
                M1[3 * q + 2] = as[q] * (e.uc - u);
                M2[3 * q] = 0.0;
                M2[3 * q + 1] = as[q] * e.fv;
                M2[3 * q + 2] = as[q] * (e.vc - v);
            }
#pragma unroll
            for (int i = 0; i < 12; i++)
#pragma unroll
                for (int j = i; j < 12; j++) {
                    double acc = VO_UT(i * 12 + j);
                    acc += M1[i] * M1[j];
                    acc += M2[i] * M2[j];
                    VO_UT(i * 12 + j) = acc;
                }
        }
#pragma unroll
        for (int i = 0; i < 12; i++)
#pragma unroll
            for (int j = 0; j < i; j++)
                VO_UT(i * 12 + j) = VO_UT(j * 12 + i);
    }
    VO_EPNP_STAMP(2);
}

// rows 11, 10, 9, 8 of `ut` = the null-space basis (rows of U^T sorted by descending singular value)
// ---- L_6x10, rho
template <int S>
VO_HD void epnp5_L_rho(const Epnp5 &e, const double *ut, double *L /*60*/, double *rho /*6*/)
{
    double dv[4][6][3];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int a = 0, b = 1;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const int va = 12 * (11 - i) + 3 * a, vb = 12 * (11 - i) + 3 * b; // null vector i = row 11 - i
            dv[i][j][0] = VO_UT(va) - VO_UT(vb);
            dv[i][j][1] = VO_UT(va + 1) - VO_UT(vb + 1);
            dv[i][j][2] = VO_UT(va + 2) - VO_UT(vb + 2);
            b++;
            if (b > 3) {
                a++;
                b = a + 1;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double *row = L + 10 * i;
        row[0] = dot3(dv[0][i], dv[0][i]);
        row[1] = 2.0f * dot3(dv[0][i], dv[1][i]);
        row[2] = dot3(dv[1][i], dv[1][i]);
        row[3] = 2.0f * dot3(dv[0][i], dv[2][i]);
        row[4] = 2.0f * dot3(dv[1][i], dv[2][i]);
        row[5] = dot3(dv[2][i], dv[2][i]);
        row[6] = 2.0f * dot3(dv[0][i], dv[3][i]);
        row[7] = 2.0f * dot3(dv[1][i], dv[3][i]);
        row[8] = 2.0f * dot3(dv[2][i], dv[3][i]);
        row[9] = dot3(dv[3][i], dv[3][i]);
    }
    rho[0] = dist2(e.cws[0], e.cws[1]);
    rho[1] = dist2(e.cws[0], e.cws[2]);
    rho[2] = dist2(e.cws[0], e.cws[3]);
    rho[3] = dist2(e.cws[1], e.cws[2]);
    rho[4] = dist2(e.cws[1], e.cws[3]);
    rho[5] = dist2(e.cws[2], e.cws[3]);
}

// One of the three beta approximations (A = 0: [B11 B12 B13 B14], 1: [B11 B12 B22], 2: [B11 B12 B22 B13 B23]), its five
// Gauss-Newton steps, R and t; returns the reprojection error.  The three are independent of each other (e.ccs / e.pcs are
// scratch of epnp_compute_R_and_t): the monolithic solver runs them one after the other, epnp_approx_kernel side by side.
template <int S, int A>
VO_HD double epnp5_approx(Epnp5 &e, const double *ut, const double *L, const double *rho, double *R /*9*/, double *t /*3*/)
{
    double betas[4];
    if (A == 0) {
        double l[24], b4[4];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            l[i * 4 + 0] = L[i * 10 + 0];
            l[i * 4 + 1] = L[i * 10 + 1];
            l[i * 4 + 2] = L[i * 10 + 3];
            l[i * 4 + 3] = L[i * 10 + 6];
        }
        solve_svd<6, 4>(l, rho, b4);
        if (b4[0] < 0) {
            betas[0] = sqrt(-b4[0]);
            betas[1] = -b4[1] / betas[0];
            betas[2] = -b4[2] / betas[0];
            betas[3] = -b4[3] / betas[0];
        } else {
            betas[0] = sqrt(b4[0]);
            betas[1] = b4[1] / betas[0];
            betas[2] = b4[2] / betas[0];
            betas[3] = b4[3] / betas[0];
        }
    } else if (A == 1) {
        double l[18], b3[3];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            l[i * 3 + 0] = L[i * 10 + 0];
            l[i * 3 + 1] = L[i * 10 + 1];
            l[i * 3 + 2] = L[i * 10 + 2];
        }
        solve_svd<6, 3>(l, rho, b3);
        if (b3[0] < 0) {
            betas[0] = sqrt(-b3[0]);
            betas[1] = (b3[2] < 0) ? sqrt(-b3[2]) : 0.0;
        } else {
            betas[0] = sqrt(b3[0]);
            betas[1] = (b3[2] > 0) ? sqrt(b3[2]) : 0.0;
        }
        if (b3[1] < 0)
            betas[0] = -betas[0];
        betas[2] = 0.0;
        betas[3] = 0.0;
    } else {
        double l[30], b5[5];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            l[i * 5 + 0] = L[i * 10 + 0];
            l[i * 5 + 1] = L[i * 10 + 1];
            l[i * 5 + 2] = L[i * 10 + 2];
            l[i * 5 + 3] = L[i * 10 + 3];
            l[i * 5 + 4] = L[i * 10 + 4];
        }
        solve_svd<6, 5>(l, rho, b5);
        if (b5[0] < 0) {
            betas[0] = sqrt(-b5[0]);
            betas[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0;
        } else {
            betas[0] = sqrt(b5[0]);
            betas[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0;
        }
        if (b5[1] < 0)
            betas[0] = -betas[0];
        betas[2] = b5[3] / betas[0];
        betas[3] = 0.0;
    }
    epnp_gauss_newton(L, rho, betas);
    return epnp_compute_R_and_t<S>(e, ut, betas, R, t);
}

// best of the three by reprojection error (first strictly smaller wins), R -> rvec; selected with compile-time indices so
// that the candidates stay in registers
VO_HD void epnp5_select(const double *rep, const double *R0, const double *R1, const double *R2, const double *t0,
                        const double *t1, const double *t2, double *rvec, double *tvec)
{
    const bool use1 = rep[1] < rep[0];
    const double rep01 = use1 ? rep[1] : rep[0];
    const bool use2 = rep[2] < rep01;
    double Rb[9], tb[3];
#pragma unroll
    for (int k = 0; k < 9; k++)
        Rb[k] = use2 ? R2[k] : use1 ? R1[k] : R0[k];
#pragma unroll
    for (int k = 0; k < 3; k++)
        tb[k] = use2 ? t2[k] : use1 ? t1[k] : t0[k];
    rodrigues_m2v(Rb, rvec);
    tvec[0] = tb[0];
    tvec[1] = tb[1];
    tvec[2] = tb[2];
}

template <int S>
VO_HD void epnp5_finish(Epnp5 &e, const double *ut, double *rvec, double *tvec)
{
    VO_EPNP_STAMP(3);
    double L[60], rho[6];
    epnp5_L_rho<S>(e, ut, L, rho);
    VO_EPNP_STAMP(4);
    double Rs[3][9], ts[3][3], rep[3];
    rep[0] = epnp5_approx<S, 0>(e, ut, L, rho, Rs[0], ts[0]);
    VO_EPNP_STAMP(5);
    rep[1] = epnp5_approx<S, 1>(e, ut, L, rho, Rs[1], ts[1]);
    VO_EPNP_STAMP(6);
    rep[2] = epnp5_approx<S, 2>(e, ut, L, rho, Rs[2], ts[2]);
    VO_EPNP_STAMP(7);
    epnp5_select(rep, Rs[0], Rs[1], Rs[2], ts[0], ts[1], ts[2], rvec, tvec);
    VO_EPNP_STAMP(8);
}
#undef VO_UT

template <int S>
VO_HD_NOINLINE void epnp5_solve_t(const float *xyz5, const float *uv5, const float *Kf, double *rvec,
                                  double *tvec, double *ut)
{
    Epnp5 e;
    double d12[12];
    epnp5_prepare<S>(xyz5, uv5, Kf, e, ut);
    jacobi_svd<12, 12, false, S>(ut, d12, nullptr);
    epnp5_finish<S>(e, ut, rvec, tvec);
}

// private-array form (host unit tests, and any caller without a staging area)
VO_HD_NOINLINE void epnp5_solve(const float *xyz5, const float *uv5, const float *Kf, double *rvec,
                                double *tvec)
{
    double ut[144 + 12];
    epnp5_solve_t<1>(xyz5, uv5, Kf, rvec, tvec, ut);
}

} // namespace vo
