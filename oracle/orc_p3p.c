/*
 * orc_p3p.c -- CPU ORACLE (test infrastructure, NOT product code).  PARITY UNPINNED: restated from knowledge of
 * upstream OpenCV 4.5.x (calib3d/src/p3p.cpp, polynom_solver.cpp, solvepnp.cpp solveP3P); OpenCV is not in this image.
 *
 * Why it exists: cv::solvePnPRansac as called by trackingFrame2Frame (reference src/visualOdometry.cpp:176-178) with
 * EXACTLY FOUR correspondences switches its minimal solver -- `else if (npoints == 4) { model_points = 4;
 * ransac_kernel_method = SOLVEPNP_P3P; }` -- and, model_points being npoints, returns
 * solvePnP(opoints, ipoints, K, dist, rvec, tvec, useExtrinsicGuess, SOLVEPNP_P3P) directly: no RANSAC loop, no
 * refinement, all four points reported as inliers, `false` (rvec / tvec untouched, inliers released) when P3P finds no
 * solution.
 *
 * Algorithm (Gao, Hou, Tang, Cheng: "Complete solution classification for the perspective-three-point problem",
 * PAMI 25(8), 2003, as implemented by OpenCV's p3p class): the first three points give up to four candidate poses
 * (quartic in the ratio of two ray lengths, Horn's closed-form absolute orientation for each root); the fourth point
 * ranks them by its reprojection error in normalised coordinates; solveP3P then re-ranks by the pixel reprojection
 * error over all four points (stable insertion sorts both) and solvePnP takes the first.
 *
 * Pinned here by: planted poses recovered to 1e-9 (tests/test_p3p.py), every candidate satisfying the three-point
 * constraints exactly, the quartic's roots against numpy.roots.
 */
#include "orc_internal.h"

#include <math.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- polynom_solver.cpp ---- */
static int solve_deg2(double a, double b, double c, double *x1, double *x2)
{
    double delta = b * b - 4 * a * c;
    if (delta < 0)
        return 0;
    double inv_2a = 0.5 / a;
    if (delta == 0) {
        *x1 = -b * inv_2a;
        *x2 = *x1;
        return 1;
    }
    double sqrt_delta = sqrt(delta);
    *x1 = (-b + sqrt_delta) * inv_2a;
    *x2 = (-b - sqrt_delta) * inv_2a;
    return 2;
}

/* Cardano / trigonometric form (MathWorld "Cubic Equation") */
static int solve_deg3(double a, double b, double c, double d, double *x0, double *x1, double *x2)
{
    if (a == 0) {
        if (b == 0) {
            if (c == 0)
                return 0;
            *x0 = -d / c;
            return 1;
        }
        *x2 = 0;
        return solve_deg2(b, c, d, x0, x1);
    }
    double inv_a = 1. / a;
    double b_a = inv_a * b, b_a2 = b_a * b_a;
    double c_a = inv_a * c;
    double d_a = inv_a * d;
    double Q = (3 * c_a - b_a2) / 9;
    double R = (9 * b_a * c_a - 27 * d_a - 2 * b_a * b_a2) / 54;
    double Q3 = Q * Q * Q;
    double D = Q3 + R * R;
    double b_a_3 = (1. / 3.) * b_a;
    if (Q == 0) {
        if (R == 0) {
            *x0 = *x1 = *x2 = -b_a_3;
            return 3;
        }
        *x0 = pow(2 * R, 1 / 3.0) - b_a_3;
        return 1;
    }
    if (D <= 0) {
        double theta = acos(R / sqrt(-Q3));
        double sqrt_Q = sqrt(-Q);
        *x0 = 2 * sqrt_Q * cos(theta / 3.0) - b_a_3;
        *x1 = 2 * sqrt_Q * cos((theta + 2 * M_PI) / 3.0) - b_a_3;
        *x2 = 2 * sqrt_Q * cos((theta + 4 * M_PI) / 3.0) - b_a_3;
        return 3;
    }
    double AD = pow(fabs(R) + sqrt(D), 1.0 / 3.0) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
    double BD = (AD == 0) ? 0 : -Q / AD;
    *x0 = AD + BD - b_a_3;
    return 1;
}

/* Ferrari via the resolvent cubic (MathWorld "Quartic Equation") */
int orc_solve_deg4(double a, double b, double c, double d, double e, double *x /* [4] */)
{
    if (a == 0) {
        x[3] = 0;
        return solve_deg3(b, c, d, e, &x[0], &x[1], &x[2]);
    }
    double inv_a = 1. / a;
    b *= inv_a;
    c *= inv_a;
    d *= inv_a;
    e *= inv_a;
    double b2 = b * b, bc = b * c, b3 = b2 * b;
    double r0, r1, r2;
    int n = solve_deg3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, &r0, &r1, &r2);
    if (n == 0)
        return 0;
    double R2 = 0.25 * b2 - c + r0, R;
    if (R2 < 0)
        return 0;
    R = sqrt(R2);
    double inv_R = 1. / R;
    int nb_real_roots = 0;
    double D2, E2;
    if (R < 10E-12) {
        double temp = r0 * r0 - 4 * e;
        if (temp < 0)
            D2 = E2 = -1;
        else {
            double sqrt_temp = sqrt(temp);
            D2 = 0.75 * b2 - 2 * c + 2 * sqrt_temp;
            E2 = D2 - 4 * sqrt_temp;
        }
    } else {
        double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * inv_R * (4 * bc - 8 * d - b3);
        D2 = u + v;
        E2 = u - v;
    }
    double b_4 = 0.25 * b, R_2 = 0.5 * R;
    if (D2 >= 0) {
        double D = sqrt(D2);
        nb_real_roots = 2;
        double D_2 = 0.5 * D;
        x[0] = R_2 + D_2 - b_4;
        x[1] = x[0] - D;
    }
    if (E2 >= 0) {
        double E = sqrt(E2);
        double E_2 = 0.5 * E;
        if (nb_real_roots == 0) {
            x[0] = -R_2 + E_2 - b_4;
            x[1] = x[0] - E;
            nb_real_roots = 2;
        } else {
            x[2] = -R_2 + E_2 - b_4;
            x[3] = x[2] - E;
            nb_real_roots = 4;
        }
    }
    return nb_real_roots;
}

/* ---- p3p.cpp ---- */
/* lengths |PA|, |PB|, |PC| from the pairwise distances |BC|, |AC|, |AB| and the cosines of the angles BPC, APC, APB */
static int solve_for_lengths(double lengths[4][3], const double distances[3], const double cosines[3])
{
    double p = cosines[0] * 2;
    double q = cosines[1] * 2;
    double r = cosines[2] * 2;
    double inv_d22 = 1. / (distances[2] * distances[2]);
    double a = inv_d22 * (distances[0] * distances[0]);
    double b = inv_d22 * (distances[1] * distances[1]);
    double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
    double pr = p * r, pqr = q * pr;
    if (p2 + q2 + r2 - pqr - 1 == 0) /* reality condition: the four points must not be coplanar */
        return 0;
    double ab = a * b, a_2 = 2 * a;
    double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
    if (A == 0)
        return 0;
    double a_4 = 4 * a;
    double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
    double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
    double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
    double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
    double temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr);
    double b0 = b * temp * temp;
    if (b0 == 0)
        return 0;
    double real_roots[4];
    int n = orc_solve_deg4(A, B, C, D, E, real_roots);
    if (n == 0)
        return 0;
    int nb_solutions = 0;
    double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q;
    double inv_b0 = 1. / b0;
    for (int i = 0; i < n; i++) {
        double x = real_roots[i];
        if (x <= 0)
            continue;
        double x2 = x * x;
        double b1 = ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) *
                    (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
                      (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) +
                       pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) *
                         x2 +
                     (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) +
                      r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) + pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) *
                         x +
                     2 * r3q * (a_2 - b - a2 + ab - 1) + pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
                     p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));
        if (b1 <= 0)
            continue;
        double y = inv_b0 * b1;
        double v = x2 + y * y - x * y * r;
        if (v <= 0)
            continue;
        double Z = distances[2] / sqrt(v);
        double X = x * Z;
        double Y = y * Z;
        lengths[nb_solutions][0] = X;
        lengths[nb_solutions][1] = Y;
        lengths[nb_solutions][2] = Z;
        nb_solutions++;
    }
    return nb_solutions;
}

/* cyclic Jacobi of a symmetric 4 x 4 (Numerical Recipes' jacobi), eigenvalues D, eigenvectors in the columns of U */
static int jacobi_4x4(double *A, double *D, double *U)
{
    double B[4], Z[4] = {0, 0, 0, 0};
    static const double Id[16] = {1., 0., 0., 0., 0., 1., 0., 0., 0., 0., 1., 0., 0., 0., 0., 1.};
    memcpy(U, Id, 16 * sizeof(double));
    B[0] = A[0];
    B[1] = A[5];
    B[2] = A[10];
    B[3] = A[15];
    memcpy(D, B, 4 * sizeof(double));
    for (int iter = 0; iter < 50; iter++) {
        double sum = fabs(A[1]) + fabs(A[2]) + fabs(A[3]) + fabs(A[6]) + fabs(A[7]) + fabs(A[11]);
        if (sum == 0.0)
            return 1;
        double tresh = (iter < 3) ? 0.2 * sum / 16. : 0.0;
        for (int i = 0; i < 3; i++) {
            double *pAij = A + 5 * i + 1;
            for (int j = i + 1; j < 4; j++) {
                double Aij = *pAij;
                double eps_machine = 100.0 * fabs(Aij);
                if (iter > 3 && fabs(D[i]) + eps_machine == fabs(D[i]) && fabs(D[j]) + eps_machine == fabs(D[j]))
                    *pAij = 0.0;
                else if (fabs(Aij) > tresh) {
                    double hh = D[j] - D[i], t;
                    if (fabs(hh) + eps_machine == fabs(hh))
                        t = Aij / hh;
                    else {
                        double theta = 0.5 * hh / Aij;
                        t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                        if (theta < 0.0)
                            t = -t;
                    }
                    hh = t * Aij;
                    Z[i] -= hh;
                    Z[j] += hh;
                    D[i] -= hh;
                    D[j] += hh;
                    *pAij = 0.0;
                    double c = 1.0 / sqrt(1 + t * t);
                    double s = t * c;
                    double tau = s / (1.0 + c);
                    for (int k = 0; k <= i - 1; k++) {
                        double g = A[k * 4 + i], h = A[k * 4 + j];
                        A[k * 4 + i] = g - s * (h + g * tau);
                        A[k * 4 + j] = h + s * (g - h * tau);
                    }
                    for (int k = i + 1; k <= j - 1; k++) {
                        double g = A[i * 4 + k], h = A[k * 4 + j];
                        A[i * 4 + k] = g - s * (h + g * tau);
                        A[k * 4 + j] = h + s * (g - h * tau);
                    }
                    for (int k = j + 1; k < 4; k++) {
                        double g = A[i * 4 + k], h = A[j * 4 + k];
                        A[i * 4 + k] = g - s * (h + g * tau);
                        A[j * 4 + k] = h + s * (g - h * tau);
                    }
                    for (int k = 0; k < 4; k++) {
                        double g = U[k * 4 + i], h = U[k * 4 + j];
                        U[k * 4 + i] = g - s * (h + g * tau);
                        U[k * 4 + j] = h + s * (g - h * tau);
                    }
                }
                pAij++;
            }
        }
        for (int i = 0; i < 4; i++)
            B[i] += Z[i];
        memcpy(D, B, 4 * sizeof(double));
        memset(Z, 0, 4 * sizeof(double));
    }
    return 0;
}

/* Horn's closed-form absolute orientation (unit quaternion = dominant eigenvector of a 4 x 4): R, T with
 * M_end[i] = R * (Xi, Yi, Zi) + T */
static int align(double M_end[3][3], double X0, double Y0, double Z0, double X1, double Y1, double Z1, double X2,
                 double Y2, double Z2, double R[3][3], double T[3])
{
    double C_start[3], C_end[3];
    for (int i = 0; i < 3; i++)
        C_end[i] = (M_end[0][i] + M_end[1][i] + M_end[2][i]) / 3;
    C_start[0] = (X0 + X1 + X2) / 3;
    C_start[1] = (Y0 + Y1 + Y2) / 3;
    C_start[2] = (Z0 + Z1 + Z2) / 3;
    double s[3 * 3];
    for (int j = 0; j < 3; j++) {
        s[0 * 3 + j] = (X0 * M_end[0][j] + X1 * M_end[1][j] + X2 * M_end[2][j]) / 3 - C_end[j] * C_start[0];
        s[1 * 3 + j] = (Y0 * M_end[0][j] + Y1 * M_end[1][j] + Y2 * M_end[2][j]) / 3 - C_end[j] * C_start[1];
        s[2 * 3 + j] = (Z0 * M_end[0][j] + Z1 * M_end[1][j] + Z2 * M_end[2][j]) / 3 - C_end[j] * C_start[2];
    }
    double Qs[16], evs[4], U[16];
    Qs[0 * 4 + 0] = s[0 * 3 + 0] + s[1 * 3 + 1] + s[2 * 3 + 2];
    Qs[1 * 4 + 1] = s[0 * 3 + 0] - s[1 * 3 + 1] - s[2 * 3 + 2];
    Qs[2 * 4 + 2] = s[1 * 3 + 1] - s[2 * 3 + 2] - s[0 * 3 + 0];
    Qs[3 * 4 + 3] = s[2 * 3 + 2] - s[0 * 3 + 0] - s[1 * 3 + 1];
    Qs[1 * 4 + 0] = Qs[0 * 4 + 1] = s[1 * 3 + 2] - s[2 * 3 + 1];
    Qs[2 * 4 + 0] = Qs[0 * 4 + 2] = s[2 * 3 + 0] - s[0 * 3 + 2];
    Qs[3 * 4 + 0] = Qs[0 * 4 + 3] = s[0 * 3 + 1] - s[1 * 3 + 0];
    Qs[2 * 4 + 1] = Qs[1 * 4 + 2] = s[1 * 3 + 0] + s[0 * 3 + 1];
    Qs[3 * 4 + 1] = Qs[1 * 4 + 3] = s[2 * 3 + 0] + s[0 * 3 + 2];
    Qs[3 * 4 + 2] = Qs[2 * 4 + 3] = s[2 * 3 + 1] + s[1 * 3 + 2];
    jacobi_4x4(Qs, evs, U);
    int i_ev = 0;
    double ev_max = evs[i_ev];
    for (int i = 1; i < 4; i++)
        if (evs[i] > ev_max)
            ev_max = evs[i_ev = i];
    double q[4];
    for (int i = 0; i < 4; i++)
        q[i] = U[i * 4 + i_ev];
    double q02 = q[0] * q[0], q12 = q[1] * q[1], q22 = q[2] * q[2], q32 = q[3] * q[3];
    double q0_1 = q[0] * q[1], q0_2 = q[0] * q[2], q0_3 = q[0] * q[3];
    double q1_2 = q[1] * q[2], q1_3 = q[1] * q[3];
    double q2_3 = q[2] * q[3];
    R[0][0] = q02 + q12 - q22 - q32;
    R[0][1] = 2. * (q1_2 - q0_3);
    R[0][2] = 2. * (q1_3 + q0_2);
    R[1][0] = 2. * (q1_2 + q0_3);
    R[1][1] = q02 + q22 - q12 - q32;
    R[1][2] = 2. * (q2_3 - q0_1);
    R[2][0] = 2. * (q1_3 - q0_2);
    R[2][1] = 2. * (q2_3 + q0_1);
    R[2][2] = q02 + q32 - q12 - q22;
    for (int i = 0; i < 3; i++)
        T[i] = C_end[i] - (R[i][0] * C_start[0] + R[i][1] * C_start[1] + R[i][2] * C_start[2]);
    return 1;
}

/* p3p::solve(R[4], t[4], mu0 ... Z3, p4p): image points in PIXELS, up to four poses; with p4p they come sorted by the
 * squared error of the fourth point in normalised coordinates */
int orc_p3p_solve(const double *K4 /* fx fy cx cy */, const double *uv /* 4 x 2 */, const double *xyz /* 4 x 3 */,
                  int p4p, double *R_out /* [4][9] */, double *t_out /* [4][3] */)
{
    const double inv_fx = 1. / K4[0], inv_fy = 1. / K4[1], cx_fx = K4[2] / K4[0], cy_fy = K4[3] / K4[1];
    double mu[4], mv[4], mk[3];
    for (int i = 0; i < 3; i++) {
        mu[i] = inv_fx * uv[2 * i] - cx_fx;
        mv[i] = inv_fy * uv[2 * i + 1] - cy_fy;
        double norm = sqrt(mu[i] * mu[i] + mv[i] * mv[i] + 1);
        mk[i] = 1. / norm;
        mu[i] *= mk[i];
        mv[i] *= mk[i];
    }
    mu[3] = inv_fx * uv[6] - cx_fx;
    mv[3] = inv_fy * uv[7] - cy_fy;
    const double *P0 = xyz, *P1 = xyz + 3, *P2 = xyz + 6, *P3 = xyz + 9;
    double distances[3];
    distances[0] = sqrt((P1[0] - P2[0]) * (P1[0] - P2[0]) + (P1[1] - P2[1]) * (P1[1] - P2[1]) + (P1[2] - P2[2]) * (P1[2] - P2[2]));
    distances[1] = sqrt((P0[0] - P2[0]) * (P0[0] - P2[0]) + (P0[1] - P2[1]) * (P0[1] - P2[1]) + (P0[2] - P2[2]) * (P0[2] - P2[2]));
    distances[2] = sqrt((P0[0] - P1[0]) * (P0[0] - P1[0]) + (P0[1] - P1[1]) * (P0[1] - P1[1]) + (P0[2] - P1[2]) * (P0[2] - P1[2]));
    double cosines[3];
    cosines[0] = mu[1] * mu[2] + mv[1] * mv[2] + mk[1] * mk[2];
    cosines[1] = mu[0] * mu[2] + mv[0] * mv[2] + mk[0] * mk[2];
    cosines[2] = mu[0] * mu[1] + mv[0] * mv[1] + mk[0] * mk[1];
    double lengths[4][3];
    memset(lengths, 0, sizeof(lengths));
    int n = solve_for_lengths(lengths, distances, cosines);
    int nb_solutions = 0;
    double reproj_errors[4], R[4][3][3], t[4][3];
    for (int i = 0; i < n; i++) {
        double M_orig[3][3];
        for (int k = 0; k < 3; k++) {
            M_orig[k][0] = lengths[i][k] * mu[k];
            M_orig[k][1] = lengths[i][k] * mv[k];
            M_orig[k][2] = lengths[i][k] * mk[k];
        }
        if (!align(M_orig, P0[0], P0[1], P0[2], P1[0], P1[1], P1[2], P2[0], P2[1], P2[2], R[nb_solutions], t[nb_solutions]))
            continue;
        if (p4p) {
            double(*Rn)[3] = R[nb_solutions];
            double X3p = Rn[0][0] * P3[0] + Rn[0][1] * P3[1] + Rn[0][2] * P3[2] + t[nb_solutions][0];
            double Y3p = Rn[1][0] * P3[0] + Rn[1][1] * P3[1] + Rn[1][2] * P3[2] + t[nb_solutions][1];
            double Z3p = Rn[2][0] * P3[0] + Rn[2][1] * P3[1] + Rn[2][2] * P3[2] + t[nb_solutions][2];
            double mu3p = X3p / Z3p;
            double mv3p = Y3p / Z3p;
            reproj_errors[nb_solutions] = (mu3p - mu[3]) * (mu3p - mu[3]) + (mv3p - mv[3]) * (mv3p - mv[3]);
        }
        nb_solutions++;
    }
    if (p4p) { /* stable insertion sort by the fourth point's error */
        for (int i = 1; i < nb_solutions; i++)
            for (int j = i; j > 0 && reproj_errors[j - 1] > reproj_errors[j]; j--) {
                double e = reproj_errors[j], Rt[3][3], tt[3];
                reproj_errors[j] = reproj_errors[j - 1];
                reproj_errors[j - 1] = e;
                memcpy(Rt, R[j], sizeof(Rt));
                memcpy(R[j], R[j - 1], sizeof(Rt));
                memcpy(R[j - 1], Rt, sizeof(Rt));
                memcpy(tt, t[j], sizeof(tt));
                memcpy(t[j], t[j - 1], sizeof(tt));
                memcpy(t[j - 1], tt, sizeof(tt));
            }
    }
    for (int i = 0; i < nb_solutions; i++) {
        memcpy(R_out + 9 * i, R[i], sizeof(double) * 9);
        memcpy(t_out + 3 * i, t[i], sizeof(double) * 3);
    }
    return nb_solutions;
}

/* solvePnP(opoints f32, ipoints f32, K, dist = 0, rvec, tvec, -, SOLVEPNP_P3P) with 4 points = solveP3P's first
 * solution: undistortPoints to f32 normalised coordinates, p3p::extract_points back to pixels in f64, p3p::solve, then
 * per solution Rodrigues + projectPoints in f64 and a stable sort by the summed squared pixel error over the four
 * points.  Returns the number of solutions (0: rvec / tvec untouched); rvecs / tvecs (optional, [4][3]) get all of them
 * in their final order. */
int orc_solve_p3p(const float *xyz, const float *uv, const float *K, double *rvec, double *tvec, double *rvecs,
                  double *tvecs)
{
    const double fx = (double)K[0], fy = (double)K[4], cx = (double)K[2], cy = (double)K[5];
    const double ifx = 1. / fx, ify = 1. / fy;
    double pix[8], obj[12], img[8];
    for (int i = 0; i < 4; i++) {
        /* cvUndistortPointsInternal with zero distortion: x = (u - cx) * ifx, stored as float */
        double x = ((double)uv[2 * i] - cx) * ifx, y = ((double)uv[2 * i + 1] - cy) * ify;
        float xn = (float)x, yn = (float)y;
        pix[2 * i] = xn * fx + cx; /* p3p::extract_points */
        pix[2 * i + 1] = yn * fy + cy;
        for (int k = 0; k < 3; k++)
            obj[3 * i + k] = (double)xyz[3 * i + k];
        img[2 * i] = (double)uv[2 * i];
        img[2 * i + 1] = (double)uv[2 * i + 1];
    }
    const double K4[4] = {fx, fy, cx, cy};
    double Rs[4][9], ts[4][3], rv[4][3], err[4];
    int n = orc_p3p_solve(K4, pix, obj, 1, &Rs[0][0], &ts[0][0]);
    for (int i = 0; i < n; i++) {
        orc_rodrigues_mat2vec(Rs[i], rv[i]);
        double proj[8];
        orc_project_points_d(obj, 4, rv[i], ts[i], K4, proj, NULL, NULL, 0);
        double e = 0;
        for (int k = 0; k < 8; k++)
            e += (img[k] - proj[k]) * (img[k] - proj[k]);
        err[i] = e;
    }
    for (int i = 1; i < n; i++)
        for (int j = i; j > 0 && err[j - 1] > err[j]; j--) {
            double e = err[j], a[3];
            err[j] = err[j - 1];
            err[j - 1] = e;
            memcpy(a, rv[j], sizeof(a));
            memcpy(rv[j], rv[j - 1], sizeof(a));
            memcpy(rv[j - 1], a, sizeof(a));
            memcpy(a, ts[j], sizeof(a));
            memcpy(ts[j], ts[j - 1], sizeof(a));
            memcpy(ts[j - 1], a, sizeof(a));
        }
    if (rvecs && tvecs)
        for (int i = 0; i < n; i++) {
            memcpy(rvecs + 3 * i, rv[i], sizeof(double) * 3);
            memcpy(tvecs + 3 * i, ts[i], sizeof(double) * 3);
        }
    if (n > 0) {
        memcpy(rvec, rv[0], sizeof(double) * 3);
        memcpy(tvec, ts[0], sizeof(double) * 3);
    }
    return n;
}
