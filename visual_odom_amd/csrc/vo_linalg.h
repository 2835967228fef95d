// vo_linalg.h -- small dense f64 linear algebra for the device side of the pose solve.
//
// Everything here is VO_HD (__host__ __device__) so the exact code the kernels run can also be
// compiled by g++ into tests/host_check (a CPU unit test of the device functions -- it is not a
// fallback path and libvo_hip never dispatches to it).
//
// The routines follow the operation order of OpenCV's JacobiSVDImpl_ / SVBkSb
// (core/src/lapack.cpp) which the reference reaches through cv::triangulatePoints and
// cv::solvePnPRansac (reference main.cpp:170, visualOdometry.cpp:176), so that with
// -ffp-contract=off the device results track the CPU path to the last bits.
// Attribution: JacobiSVD / back-substitution follow the operation order of OpenCV's modules/core/src/lapack.cpp
// (Apache-2.0) -- see NOTICE.  Written for this repository; no OpenCV source is included.
#pragma once

#include "vo_math.h" // VO_HD, and the sin / cos / acos Rodrigues uses: IEEE operations only, the same bits on gfx950 and g++

namespace vo {

// sqrt(a^2 + b^2) from three correctly rounded IEEE operations (mul, fma, sqrt): bit-identical on
// the host and on gfx950, unlike libm / ocml hypot whose last ulp is implementation specific.
VO_HD double vo_hypot(double a, double b) { return sqrt(fma(a, a, b * b)); }

// One-sided (Hestenes) Jacobi SVD of the M x N matrix whose COLUMNS are the rows of At
// (At is N rows of length M, row stride M).  On return: W[N] descending singular values,
// rows of At = left singular vectors (normalised), rows of Vt = right singular vectors.
// WANT_V = false skips accumulating V (At rows are still sorted), used for EPnP's 12x12.
// STRIDE: element (i, k) of At lives at At[(i * M + k) * STRIDE] -- 1 for a private array, the
// workgroup size for a lane-interleaved LDS array (EPnP's 12 x 12: 288 registers' worth of matrix
// that would otherwise push the kernel into scratch memory).  Small problems (N <= 6) are fully
// unrolled so that every access has a compile-time index and the matrices live in registers.
// N1 > N (cv::SVD::FULL_UV of a wide matrix): At has N1 rows, and rows N .. N1-1 are completed to an
// orthonormal basis by the same pseudo-random fill OpenCV uses for zero singular values.
template <int M, int N, bool WANT_V, int STRIDE = 1, int N1 = N>
VO_HD void jacobi_svd(double *At, double *Wout, double *Vt)
{
    constexpr int UNR = N <= 6 ? N : 1; // unroll factor of the pair loops (1 = keep rolled)
#define VO_AT(i, k) At[((i) * M + (k)) * STRIDE]
    // squared column norms: private registers when unrolled, behind the matrix (N more strided
    // elements: the caller provides (M * N + N) * STRIDE doubles) when the pair loops stay rolled
#define VO_W(i) (*(STRIDE > 1 ? &At[(M * N + (i)) * STRIDE] : &W[i]))
    const double eps = DBL_EPSILON * 10;
    const double minval = DBL_MIN;
    double W[N];
    const int max_iter = M > 30 ? M : 30;

#pragma unroll
    for (int i = 0; i < N; i++) { // W[] is small: always compile-time indexed
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; k++) {
            double t = VO_AT(i, k);
            sd += t * t;
        }
        VO_W(i) = sd;
        if (WANT_V) {
            for (int k = 0; k < N; k++)
                Vt[i * N + k] = 0;
            Vt[i * N + i] = 1;
        }
    }

    // The pair loops are fully unrolled so that every At / W / Vt access has a compile-time index:
    // the matrices then live in registers (VGPRs + AGPRs) instead of scratch memory, which is what
    // the pose kernels' run time used to be made of (1.7 GB of scratch writes per EPnP launch).
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
#pragma unroll UNR
        for (int i = 0; i < N - 1; i++)
#pragma unroll UNR
            for (int j = i + 1; j < N; j++) {
                double a = VO_W(i), p = 0, b = VO_W(j);
#pragma unroll
                for (int k = 0; k < M; k++)
                    p += VO_AT(i, k) * VO_AT(j, k);
                if (fabs(p) <= eps * sqrt(a * b))
                    continue;
                p *= 2;
                double beta = a - b, gamma = vo_hypot(p, beta);
                double c, s;
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
#pragma unroll
                for (int k = 0; k < M; k++) {
                    const double ai = VO_AT(i, k), aj = VO_AT(j, k);
                    double t0 = c * ai + s * aj;
                    double t1 = -s * ai + c * aj;
                    VO_AT(i, k) = t0;
                    VO_AT(j, k) = t1;
                    a += t0 * t0;
                    b += t1 * t1;
                }
                VO_W(i) = a;
                VO_W(j) = b;
                changed = true;
                if (WANT_V) {
                    double *Vi = Vt + i * N, *Vj = Vt + j * N;
#pragma unroll
                    for (int k = 0; k < N; k++) {
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

#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; k++) {
            double t = VO_AT(i, k);
            sd += t * t;
        }
        VO_W(i) = sqrt(sd);
    }

    // selection sort, descending; rows of At / Vt follow.  OpenCV swaps row i with the FIRST index
    // of the running maximum; the same swap is expressed with compile-time row indices (predicated
    // on c == j) so that the rows stay in registers.
#pragma unroll UNR
    for (int i = 0; i < N - 1; i++) {
        int j = i;
        double wj = VO_W(i);
#pragma unroll
        for (int k = i + 1; k < N; k++)
            if (wj < VO_W(k)) {
                j = k;
                wj = VO_W(k);
            }
#pragma unroll UNR
        for (int c = i + 1; c < N; c++) {
            if (c == j) {
                double t = VO_W(i);
                VO_W(i) = VO_W(c);
                VO_W(c) = t;
#pragma unroll
                for (int k = 0; k < M; k++) {
                    t = VO_AT(i, k);
                    VO_AT(i, k) = VO_AT(c, k);
                    VO_AT(c, k) = t;
                }
                if (WANT_V) {
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        t = Vt[i * N + k];
                        Vt[i * N + k] = Vt[c * N + k];
                        Vt[c * N + k] = t;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < N; i++)
        Wout[i] = VO_W(i);

    // left singular vectors: normalise; exactly-zero singular values get a deterministic
    // pseudo-random vector orthogonalised against the previous ones (cv::RNG(0x12345678) stream)
    uint64_t rng = 0x12345678;
#pragma unroll UNR
    for (int i = 0; i < N1; i++) {
        double sd = i < N ? VO_W(i < N ? i : 0) : 0.;
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const double val0 = 1. / M;
            for (int k = 0; k < M; k++) {
                rng = (uint64_t)(uint32_t)rng * 4164903690U + (uint32_t)(rng >> 32);
                VO_AT(i, k) = ((uint32_t)rng & 256) != 0 ? val0 : -val0;
            }
            for (int iter = 0; iter < 2; iter++)
                for (int j = 0; j < i; j++) {
                    sd = 0;
                    for (int k = 0; k < M; k++)
                        sd += VO_AT(i, k) * VO_AT(j, k);
                    double asum = 0;
                    for (int k = 0; k < M; k++) {
                        double t = VO_AT(i, k) - sd * VO_AT(j, k);
                        VO_AT(i, k) = t;
                        asum += fabs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (int k = 0; k < M; k++)
                        VO_AT(i, k) *= asum;
                }
            sd = 0;
            for (int k = 0; k < M; k++) {
                double t = VO_AT(i, k);
                sd += t * t;
            }
            sd = sqrt(sd);
        }
        double s = sd > minval ? 1 / sd : 0.;
        for (int k = 0; k < M; k++)
            VO_AT(i, k) *= s;
    }
}
#undef VO_W
#undef VO_AT

// least squares / linear solve through the SVD (cv::solve(..., DECOMP_SVD), one rhs).
// A is M x N row-major (M >= N), consumed.
template <int M, int N>
VO_HD void solve_svd(const double *A, const double *b, double *x)
{
    double At[N * M], w[N], vt[N * N];
    for (int i = 0; i < N; i++)
        for (int k = 0; k < M; k++)
            At[i * M + k] = A[k * N + i];
    jacobi_svd<M, N, true>(At, w, vt);
    double threshold = 0;
    for (int i = 0; i < N; i++)
        x[i] = 0;
    for (int i = 0; i < N; i++)
        threshold += w[i];
    threshold *= DBL_EPSILON * 2;
    for (int i = 0; i < N; i++) {
        double wi = w[i];
        if (fabs(wi) <= threshold)
            continue;
        wi = 1 / wi;
        double s = 0;
        for (int j = 0; j < M; j++)
            s += At[i * M + j] * b[j];
        s *= wi;
        for (int j = 0; j < N; j++)
            x[j] = x[j] + s * vt[i * N + j];
    }
}

// pseudo-inverse of a square N x N matrix through the SVD (cv::invert(..., DECOMP_SVD))
template <int N>
VO_HD void invert_svd(const double *A, double *Ainv)
{
    double At[N * N], w[N], vt[N * N], buffer[N];
    for (int i = 0; i < N; i++)
        for (int k = 0; k < N; k++)
            At[i * N + k] = A[k * N + i];
    jacobi_svd<N, N, true>(At, w, vt);
    double threshold = 0;
    for (int i = 0; i < N * N; i++)
        Ainv[i] = 0;
    for (int i = 0; i < N; i++)
        threshold += w[i];
    threshold *= DBL_EPSILON * 2;
    for (int i = 0; i < N; i++) {
        double wi = w[i];
        if (fabs(wi) <= threshold)
            continue;
        wi = 1 / wi;
        for (int j = 0; j < N; j++)
            buffer[j] = At[i * N + j] * wi; // u[j][i] = At[i][j]
        for (int r = 0; r < N; r++) {
            double sv = vt[i * N + r];
            for (int j = 0; j < N; j++)
                Ainv[r * N + j] = Ainv[r * N + j] + sv * buffer[j];
        }
    }
}

// rotation vector -> matrix, optional 3x9 Jacobian dR_k/dr_i (cvRodrigues2)
VO_HD void rodrigues_v2m(const double *rv, double *R, double *J)
{
    double rx = rv[0], ry = rv[1], rz = rv[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        for (int k = 0; k < 9; k++)
            R[k] = 0;
        R[0] = R[4] = R[8] = 1;
        if (J) {
            for (int k = 0; k < 27; k++)
                J[k] = 0;
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    double c = vo_cos(theta), s = vo_sin(theta), c1 = 1. - c;
    double itheta = theta ? 1. / theta : 0.;
    rx *= itheta;
    ry *= itheta;
    rz *= itheta;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; k++)
        R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    if (J) {
        const double drrt[27] = {rx + rx, ry, rz, ry, 0, 0, rz, 0, 0,
                                 0, rx, 0, rx, ry + ry, rz, 0, rz, 0,
                                 0, 0, rx, 0, 0, ry, rx, ry, rz + rz};
        const double d_r_x_[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0,
                                   0, 0, 1, 0, 0, 0, -1, 0, 0,
                                   0, -1, 0, 1, 0, 0, 0, 0, 0};
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

// rotation matrix -> vector (cvRodrigues2, matrix branch incl. the SVD re-orthogonalisation)
VO_HD void rodrigues_m2v(const double *Rin, double *rv)
{
    for (int k = 0; k < 9; k++)
        if (!(Rin[k] > -100 && Rin[k] < 100)) {
            rv[0] = rv[1] = rv[2] = 0;
            return;
        }
    double At[9], w[3], vt[9], R[9];
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++)
            At[i * 3 + k] = Rin[k * 3 + i];
    jacobi_svd<3, 3, true>(At, w, vt);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) // U[i][k] = At[k][i]
            R[i * 3 + j] = At[0 * 3 + i] * vt[0 * 3 + j] + At[1 * 3 + i] * vt[1 * 3 + j] +
                           At[2 * 3 + i] * vt[2 * 3 + j];
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = vo_acos(c);
    if (s < 1e-5) {
        if (c > 0)
            rx = ry = rz = 0;
        else {
            double t = (R[0] + 1) * 0.5;
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

// pin-hole projection of one point, zero distortion (cvProjectPoints2 with k == 0):
// uv[2]; optional rows of the 2x6 Jacobian d(u,v)/d(r,t) given dRdr (27)
VO_HD void project_point(const double *R, const double *t, const double *dRdr, double fx, double fy,
                         double cx, double cy, double X, double Y, double Z, double *uv,
                         double *Ju /*6 or null*/, double *Jv /*6 or null*/)
{
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z;
    y *= z;
    uv[0] = x * fx + cx;
    uv[1] = y * fy + cy;
    if (Ju) {
        const double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
        for (int j = 0; j < 3; j++) {
            Ju[3 + j] = fx * dxdt[j];
            Jv[3 + j] = fy * dydt[j];
        }
        const double dx0dr[3] = {X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2],
                                 X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11],
                                 X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20]};
        const double dy0dr[3] = {X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5],
                                 X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14],
                                 X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23]};
        const double dz0dr[3] = {X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8],
                                 X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17],
                                 X * dRdr[24] + Y * dRdr[25] + Z * dRdr[26]};
        for (int j = 0; j < 3; j++) {
            double dxdr = z * (dx0dr[j] - x * dz0dr[j]);
            double dydr = z * (dy0dr[j] - y * dz0dr[j]);
            Ju[j] = fx * dxdr;
            Jv[j] = fy * dydr;
        }
    }
}

} // namespace vo
