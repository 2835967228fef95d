// vo_p3p.h -- solvePnPRansac with EXACTLY FOUR correspondences (device side, VO_HD: also compiled by g++ into
// tests/host_check).
//
// cv::solvePnPRansac as trackingFrame2Frame calls it (reference src/visualOdometry.cpp:176-178) switches its minimal
// solver when npoints == 4 (model_points = 4, SOLVEPNP_P3P) and, model_points being npoints, returns
// solvePnP(opoints, ipoints, K, dist, rvec, tvec, useExtrinsicGuess, SOLVEPNP_P3P) directly: no RANSAC loop, no
// Levenberg-Marquardt refinement, all four points inliers; `false` with rvec / tvec untouched when there is no solution.
// Round 2 reported this case as "not provided" (status -2); this is the solver.
//
// Algorithm = OpenCV's p3p class (Gao, Hou, Tang, Cheng, PAMI 2003; calib3d/src/p3p.cpp + polynom_solver.cpp) and the
// tail of solveP3P (calib3d/src/solvepnp.cpp): the first three points give up to four poses (quartic by Ferrari's
// closed form through the resolvent cubic, Horn's quaternion alignment through a cyclic Jacobi of a symmetric 4 x 4),
// ranked by the fourth point's error in normalised coordinates, then re-ranked by the squared pixel error of all four
// points after Rodrigues + projectPoints; the first one is the answer.  Operation order follows the library so that
// the result tracks the CPU path to rounding.  The cubic's pow(x, 1/3.) / acos / cos are vo_math.h's (IEEE +, -, *, /, sqrt
// only): the device computes the same bits as the host build of this header (tested bit for bit on the MI355X), and the
// host build is held to the CPU checker -- which calls glibc's, as OpenCV does -- on the CPU (tests/test_p3p.py).
// Attribution: follows the operation order of OpenCV's calib3d p3p.cpp / polynom_solver.cpp (Apache-2.0) -- see NOTICE.
// Written for this repository; no OpenCV source is included.
#pragma once

#include "vo_linalg.h"

namespace vo {

VO_HD int p3p_deg2(double a, double b, double c, double &x1, double &x2)
{
    const double delta = b * b - 4 * a * c;
    if (delta < 0)
        return 0;
    const double inv_2a = 0.5 / a;
    if (delta == 0) {
        x1 = x2 = -b * inv_2a;
        return 1;
    }
    const double sq = sqrt(delta);
    x1 = (-b + sq) * inv_2a;
    x2 = (-b - sq) * inv_2a;
    return 2;
}

VO_HD int p3p_deg3(double a, double b, double c, double d, double &x0, double &x1, double &x2)
{
    const double PI = 3.14159265358979323846;
    if (a == 0) {
        if (b == 0) {
            if (c == 0)
                return 0;
            x0 = -d / c;
            return 1;
        }
        x2 = 0;
        return p3p_deg2(b, c, d, x0, x1);
    }
    const double inv_a = 1. / a;
    const double b_a = inv_a * b, b_a2 = b_a * b_a, c_a = inv_a * c, d_a = inv_a * d;
    const double Q = (3 * c_a - b_a2) / 9;
    const double R = (9 * b_a * c_a - 27 * d_a - 2 * b_a * b_a2) / 54;
    const double Q3 = Q * Q * Q;
    const double D = Q3 + R * R;
    const double b_a_3 = (1. / 3.) * b_a;
    if (Q == 0) {
        if (R == 0) {
            x0 = x1 = x2 = -b_a_3;
            return 3;
        }
        x0 = vo_cbrt(2 * R) - b_a_3; // (library text: pow(2 * R, 1 / 3.0))
        return 1;
    }
    if (D <= 0) { // three real roots
        const double theta = vo_acos(R / sqrt(-Q3));
        const double sqrt_Q = sqrt(-Q);
        x0 = 2 * sqrt_Q * vo_cos(theta / 3.0) - b_a_3;
        x1 = 2 * sqrt_Q * vo_cos((theta + 2 * PI) / 3.0) - b_a_3;
        x2 = 2 * sqrt_Q * vo_cos((theta + 4 * PI) / 3.0) - b_a_3;
        return 3;
    }
    const double AD = vo_cbrt(fabs(R) + sqrt(D)) * (R > 0 ? 1 : (R < 0 ? -1 : 0)); // (pow(.., 1.0 / 3.0))
    const double BD = (AD == 0) ? 0 : -Q / AD;
    x0 = AD + BD - b_a_3;
    return 1;
}

VO_HD int p3p_deg4(double a, double b, double c, double d, double e, double *x)
{
    if (a == 0) {
        x[3] = 0;
        return p3p_deg3(b, c, d, e, x[0], x[1], x[2]);
    }
    const double inv_a = 1. / a;
    b *= inv_a;
    c *= inv_a;
    d *= inv_a;
    e *= inv_a;
    const double b2 = b * b, bc = b * c, b3 = b2 * b;
    double r0 = 0, r1 = 0, r2 = 0;
    if (p3p_deg3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, r0, r1, r2) == 0)
        return 0;
    const double R2 = 0.25 * b2 - c + r0;
    if (R2 < 0)
        return 0;
    const double R = sqrt(R2), inv_R = 1. / R;
    double D2, E2;
    if (R < 10E-12) {
        const double temp = r0 * r0 - 4 * e;
        if (temp < 0)
            D2 = E2 = -1;
        else {
            const double st = sqrt(temp);
            D2 = 0.75 * b2 - 2 * c + 2 * st;
            E2 = D2 - 4 * st;
        }
    } else {
        const double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * inv_R * (4 * bc - 8 * d - b3);
        D2 = u + v;
        E2 = u - v;
    }
    const double b_4 = 0.25 * b, R_2 = 0.5 * R;
    int nr = 0;
    if (D2 >= 0) {
        const double Dv = sqrt(D2);
        nr = 2;
        x[0] = R_2 + 0.5 * Dv - b_4;
        x[1] = x[0] - Dv;
    }
    if (E2 >= 0) {
        const double Ev = sqrt(E2);
        if (nr == 0) {
            x[0] = -R_2 + 0.5 * Ev - b_4;
            x[1] = x[0] - Ev;
            nr = 2;
        } else {
            x[2] = -R_2 + 0.5 * Ev - b_4;
            x[3] = x[2] - Ev;
            nr = 4;
        }
    }
    return nr;
}

// ray lengths |PA|, |PB|, |PC| (up to 4 triples, L[k * 3 + j]) from dist = (|BC|, |AC|, |AB|) and cosines of (BPC, APC, APB)
VO_HD int p3p_lengths(double *L, const double *dist, const double *cosines)
{
    const double p = cosines[0] * 2, q = cosines[1] * 2, r = cosines[2] * 2;
    const double inv_d22 = 1. / (dist[2] * dist[2]);
    const double a = inv_d22 * (dist[0] * dist[0]), b = inv_d22 * (dist[1] * dist[1]);
    const double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
    const double pr = p * r, pqr = q * pr;
    if (p2 + q2 + r2 - pqr - 1 == 0)
        return 0;
    const double ab = a * b, a_2 = 2 * a;
    const double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
    if (A == 0)
        return 0;
    const double a_4 = 4 * a;
    const double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
    const double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
    const double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
    const double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
    const double temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr);
    const double b0 = b * temp * temp;
    if (b0 == 0)
        return 0;
    double roots[4] = {0, 0, 0, 0};
    const int n = p3p_deg4(A, B, C, D, E, roots);
    if (n == 0)
        return 0;
    int ns = 0;
    const double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q, inv_b0 = 1. / b0;
    for (int i = 0; i < n; i++) {
        const double x = roots[i];
        if (x <= 0)
            continue;
        const double x2 = x * x;
        const double b1 =
            ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) *
            (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
              (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) + pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) *
                 x2 +
             (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) + r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
              pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) *
                 x +
             2 * r3q * (a_2 - b - a2 + ab - 1) + pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
             p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));
        if (b1 <= 0)
            continue;
        const double y = inv_b0 * b1;
        const double v = x2 + y * y - x * y * r;
        if (v <= 0)
            continue;
        const double Z = dist[2] / sqrt(v);
        L[ns * 3 + 0] = x * Z;
        L[ns * 3 + 1] = y * Z;
        L[ns * 3 + 2] = Z;
        ns++;
    }
    return ns;
}

// eigen-decomposition of a symmetric 4 x 4 by cyclic Jacobi rotations: eigenvalues D, eigenvectors = columns of U
VO_HD void p3p_jacobi4(double *A, double *D, double *U)
{
    double B[4], Z[4] = {0, 0, 0, 0};
    for (int i = 0; i < 16; i++)
        U[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 4; i++)
        D[i] = B[i] = A[5 * i];
    for (int iter = 0; iter < 50; iter++) {
        const double sum = fabs(A[1]) + fabs(A[2]) + fabs(A[3]) + fabs(A[6]) + fabs(A[7]) + fabs(A[11]);
        if (sum == 0.0)
            return;
        const double tresh = (iter < 3) ? 0.2 * sum / 16. : 0.0;
        for (int i = 0; i < 3; i++) {
            for (int j = i + 1; j < 4; j++) {
                double &aij = A[i * 4 + j];
                const double Aij = aij;
                const double eps_machine = 100.0 * fabs(Aij);
                if (iter > 3 && fabs(D[i]) + eps_machine == fabs(D[i]) && fabs(D[j]) + eps_machine == fabs(D[j])) {
                    aij = 0.0;
                } else if (fabs(Aij) > tresh) {
                    double hh = D[j] - D[i], t;
                    if (fabs(hh) + eps_machine == fabs(hh)) {
                        t = Aij / hh;
                    } else {
                        const double theta = 0.5 * hh / Aij;
                        t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                        if (theta < 0.0)
                            t = -t;
                    }
                    hh = t * Aij;
                    Z[i] -= hh;
                    Z[j] += hh;
                    D[i] -= hh;
                    D[j] += hh;
                    aij = 0.0;
                    const double c = 1.0 / sqrt(1 + t * t), s = t * c, tau = s / (1.0 + c);
                    for (int k = 0; k <= i - 1; k++) {
                        const double g = A[k * 4 + i], h = A[k * 4 + j];
                        A[k * 4 + i] = g - s * (h + g * tau);
                        A[k * 4 + j] = h + s * (g - h * tau);
                    }
                    for (int k = i + 1; k <= j - 1; k++) {
                        const double g = A[i * 4 + k], h = A[k * 4 + j];
                        A[i * 4 + k] = g - s * (h + g * tau);
                        A[k * 4 + j] = h + s * (g - h * tau);
                    }
                    for (int k = j + 1; k < 4; k++) {
                        const double g = A[i * 4 + k], h = A[j * 4 + k];
                        A[i * 4 + k] = g - s * (h + g * tau);
                        A[j * 4 + k] = h + s * (g - h * tau);
                    }
                    for (int k = 0; k < 4; k++) {
                        const double g = U[k * 4 + i], h = U[k * 4 + j];
                        U[k * 4 + i] = g - s * (h + g * tau);
                        U[k * 4 + j] = h + s * (g - h * tau);
                    }
                }
            }
        }
        for (int i = 0; i < 4; i++) {
            B[i] += Z[i];
            D[i] = B[i];
            Z[i] = 0;
        }
    }
}

// Horn: R (row-major 9), T with  Mend[i] = R * P[i] + T  for the three points P (rows of P, 3 x 3)
VO_HD void p3p_align(const double *Mend /* 3 x 3 */, const double *P /* 3 x 3 */, double *R, double *T)
{
    double Cs[3], Ce[3], s[9];
    for (int i = 0; i < 3; i++) {
        Ce[i] = (Mend[i] + Mend[3 + i] + Mend[6 + i]) / 3;
        Cs[i] = (P[i] + P[3 + i] + P[6 + i]) / 3;
    }
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++)
            s[i * 3 + j] = (P[i] * Mend[j] + P[3 + i] * Mend[3 + j] + P[6 + i] * Mend[6 + j]) / 3 - Ce[j] * Cs[i];
    double Qs[16], evs[4], U[16];
    Qs[0] = s[0] + s[4] + s[8];
    Qs[5] = s[0] - s[4] - s[8];
    Qs[10] = s[4] - s[8] - s[0];
    Qs[15] = s[8] - s[0] - s[4];
    Qs[4] = Qs[1] = s[5] - s[7];
    Qs[8] = Qs[2] = s[6] - s[2];
    Qs[12] = Qs[3] = s[1] - s[3];
    Qs[9] = Qs[6] = s[3] + s[1];
    Qs[13] = Qs[7] = s[6] + s[2];
    Qs[14] = Qs[11] = s[7] + s[5];
    p3p_jacobi4(Qs, evs, U);
    int iev = 0;
    double ev_max = evs[0];
    for (int i = 1; i < 4; i++)
        if (evs[i] > ev_max) {
            ev_max = evs[i];
            iev = i;
        }
    const double q0 = U[iev], q1 = U[4 + iev], q2 = U[8 + iev], q3 = U[12 + iev];
    const double q02 = q0 * q0, q12 = q1 * q1, q22 = q2 * q2, q32 = q3 * q3;
    const double q0_1 = q0 * q1, q0_2 = q0 * q2, q0_3 = q0 * q3, q1_2 = q1 * q2, q1_3 = q1 * q3, q2_3 = q2 * q3;
    R[0] = q02 + q12 - q22 - q32;
    R[1] = 2. * (q1_2 - q0_3);
    R[2] = 2. * (q1_3 + q0_2);
    R[3] = 2. * (q1_2 + q0_3);
    R[4] = q02 + q22 - q12 - q32;
    R[5] = 2. * (q2_3 - q0_1);
    R[6] = 2. * (q1_3 - q0_2);
    R[7] = 2. * (q2_3 + q0_1);
    R[8] = q02 + q32 - q12 - q22;
    for (int i = 0; i < 3; i++)
        T[i] = Ce[i] - (R[i * 3] * Cs[0] + R[i * 3 + 1] * Cs[1] + R[i * 3 + 2] * Cs[2]);
}

// p3p::solve with the fourth point (p4p): pixel coordinates uv [4][2], object points obj [4][3] (f64);
// R [4][9], t [4][3] sorted by the fourth point's squared error in normalised coordinates; returns the count
VO_HD int p3p_poses(double fx, double fy, double cx, double cy, const double *uv, const double *obj, double *R, double *t)
{
    const double inv_fx = 1. / fx, inv_fy = 1. / fy, cx_fx = cx / fx, cy_fy = cy / fy;
    double mu[4], mv[4], mk[3];
    for (int i = 0; i < 3; i++) {
        mu[i] = inv_fx * uv[2 * i] - cx_fx;
        mv[i] = inv_fy * uv[2 * i + 1] - cy_fy;
        const double norm = sqrt(mu[i] * mu[i] + mv[i] * mv[i] + 1);
        mk[i] = 1. / norm;
        mu[i] *= mk[i];
        mv[i] *= mk[i];
    }
    mu[3] = inv_fx * uv[6] - cx_fx;
    mv[3] = inv_fy * uv[7] - cy_fy;
    const double *P0 = obj, *P1 = obj + 3, *P2 = obj + 6, *P3 = obj + 9;
    double dist[3], cosines[3];
    dist[0] = sqrt((P1[0] - P2[0]) * (P1[0] - P2[0]) + (P1[1] - P2[1]) * (P1[1] - P2[1]) + (P1[2] - P2[2]) * (P1[2] - P2[2]));
    dist[1] = sqrt((P0[0] - P2[0]) * (P0[0] - P2[0]) + (P0[1] - P2[1]) * (P0[1] - P2[1]) + (P0[2] - P2[2]) * (P0[2] - P2[2]));
    dist[2] = sqrt((P0[0] - P1[0]) * (P0[0] - P1[0]) + (P0[1] - P1[1]) * (P0[1] - P1[1]) + (P0[2] - P1[2]) * (P0[2] - P1[2]));
    cosines[0] = mu[1] * mu[2] + mv[1] * mv[2] + mk[1] * mk[2];
    cosines[1] = mu[0] * mu[2] + mv[0] * mv[2] + mk[0] * mk[2];
    cosines[2] = mu[0] * mu[1] + mv[0] * mv[1] + mk[0] * mk[1];
    double L[12];
    for (int i = 0; i < 12; i++)
        L[i] = 0;
    const int n = p3p_lengths(L, dist, cosines);
    double err[4];
    for (int i = 0; i < n; i++) {
        double M[9];
        for (int k = 0; k < 3; k++) {
            M[k * 3 + 0] = L[i * 3 + k] * mu[k];
            M[k * 3 + 1] = L[i * 3 + k] * mv[k];
            M[k * 3 + 2] = L[i * 3 + k] * mk[k];
        }
        double *Ri = R + 9 * i, *ti = t + 3 * i;
        p3p_align(M, obj, Ri, ti);
        const double X3p = Ri[0] * P3[0] + Ri[1] * P3[1] + Ri[2] * P3[2] + ti[0];
        const double Y3p = Ri[3] * P3[0] + Ri[4] * P3[1] + Ri[5] * P3[2] + ti[1];
        const double Z3p = Ri[6] * P3[0] + Ri[7] * P3[1] + Ri[8] * P3[2] + ti[2];
        const double mu3p = X3p / Z3p, mv3p = Y3p / Z3p;
        err[i] = (mu3p - mu[3]) * (mu3p - mu[3]) + (mv3p - mv[3]) * (mv3p - mv[3]);
    }
    for (int i = 1; i < n; i++) // stable insertion sort
        for (int j = i; j > 0 && err[j - 1] > err[j]; j--) {
            const double e = err[j];
            err[j] = err[j - 1];
            err[j - 1] = e;
            for (int k = 0; k < 9; k++) {
                const double v = R[9 * j + k];
                R[9 * j + k] = R[9 * (j - 1) + k];
                R[9 * (j - 1) + k] = v;
            }
            for (int k = 0; k < 3; k++) {
                const double v = t[3 * j + k];
                t[3 * j + k] = t[3 * (j - 1) + k];
                t[3 * (j - 1) + k] = v;
            }
        }
    return n;
}

// solvePnP(SOLVEPNP_P3P) on four f32 correspondences = solveP3P's first solution.  Returns the number of solutions;
// 0 leaves rvec / tvec untouched (solvePnP returned false).
VO_HD int p3p4_solve(const float *xyz4, const float *uv4, const float *K, double *rvec, double *tvec)
{
    const double fx = (double)K[0], fy = (double)K[4], cx = (double)K[2], cy = (double)K[5];
    const double ifx = 1. / fx, ify = 1. / fy;
    double pix[8], obj[12], img[8];
    for (int i = 0; i < 4; i++) {
        // undistortPoints (zero distortion) stores f32 normalised coordinates; p3p::extract_points re-applies K in f64
        const float xn = (float)(((double)uv4[2 * i] - cx) * ifx), yn = (float)(((double)uv4[2 * i + 1] - cy) * ify);
        pix[2 * i] = xn * fx + cx;
        pix[2 * i + 1] = yn * fy + cy;
        for (int k = 0; k < 3; k++)
            obj[3 * i + k] = (double)xyz4[3 * i + k];
        img[2 * i] = (double)uv4[2 * i];
        img[2 * i + 1] = (double)uv4[2 * i + 1];
    }
    double Rs[36], ts[12], rv[12], err[4];
    const int n = p3p_poses(fx, fy, cx, cy, pix, obj, Rs, ts);
    for (int i = 0; i < n; i++) {
        rodrigues_m2v(Rs + 9 * i, rv + 3 * i);
        double Rm[9];
        rodrigues_v2m(rv + 3 * i, Rm, nullptr); // projectPoints starts from the VECTOR
        double e = 0;
        for (int k = 0; k < 4; k++) {
            double p[2];
            project_point(Rm, ts + 3 * i, nullptr, fx, fy, cx, cy, obj[3 * k], obj[3 * k + 1], obj[3 * k + 2], p, nullptr,
                          nullptr);
            e += (img[2 * k] - p[0]) * (img[2 * k] - p[0]);
            e += (img[2 * k + 1] - p[1]) * (img[2 * k + 1] - p[1]);
        }
        err[i] = e;
    }
    int best = 0; // the first element of the stable ascending sort = the first minimum
    for (int i = 1; i < n; i++)
        if (err[i] < err[best])
            best = i;
    if (n > 0)
        for (int k = 0; k < 3; k++) {
            rvec[k] = rv[3 * best + k];
            tvec[k] = ts[3 * best + k];
        }
    return n;
}

} // namespace vo
