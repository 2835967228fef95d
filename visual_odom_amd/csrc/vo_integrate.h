// vo_integrate.h -- tail of the reference's frame loop (main.cpp:196-208), shared by the host entry point
// vo_integrate_odometry() and the device-side sequence loop (seq.hip):
//   rotationMatrixToEulerAngles   utils.cpp:107-131  (f64 arithmetic stored to float; x and z swapped w.r.t. MATLAB)
//   the |euler| < 0.1 rad gate    main.cpp:201-207
//   integrateOdometryStereo       utils.cpp:57-91    (frame_pose <- frame_pose * inv([R|t; 0 0 0 1]) iff 0.05 < |t| < 10)
// Plain f64 +,-,*,/ and sqrt in a fixed order (contraction off on both sides), so host and device agree bit for bit
// on the pose; atan2 only feeds the float-rounded gate comparison.
#pragma once

#include <float.h>
#include <math.h>

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#include <hip/hip_runtime.h>
#define VO_INTEG_HD __host__ __device__ inline
#else
#define VO_INTEG_HD static inline
#endif

namespace vo {

// pose: 4x4 row-major f64 in/out; R: 3x3 row-major; t: 3.  euler_out (optional): float[3].
// Returns 1 when the motion was integrated, 0 when a gate rejected it (pose unchanged).
VO_INTEG_HD int integrate_odometry(double *pose, const double *R, const double *t, float *euler_out)
{
    const float sy = (float)sqrt(R[0] * R[0] + R[3] * R[3]);
    float ex, ey, ez;
    if (!(sy < 1e-6)) {
        ex = (float)atan2(R[7], R[8]);
        ey = (float)atan2(-R[6], (double)sy);
        ez = (float)atan2(R[3], R[0]);
    } else {
        ex = (float)atan2(-R[5], R[4]);
        ey = (float)atan2(-R[6], (double)sy);
        ez = 0.f;
    }
    if (euler_out) {
        euler_out[0] = ex;
        euler_out[1] = ey;
        euler_out[2] = ez;
    }
    if (!(fabsf(ey) < 0.1f && fabsf(ex) < 0.1f && fabsf(ez) < 0.1f))
        return 0; // "Too large rotation" (main.cpp:201-207)
    const double scale = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    if (!(scale > 0.05 && scale < 10))
        return 0; // utils.cpp:80-90
    // inverse of the 4x4 [R|t; 0 0 0 1] as cv::Mat::inv() (DECOMP_LU) forms it for n > 3 (round 6; Gauss-Jordan before:
    // <= 1e-12 per step away): cv::invert copies the matrix, sets dst = I and calls hal::LU64f -- LUImpl, Gaussian elimination
    // with partial pivoting on [A | I], pivots below DBL_EPSILON * 100 are "singular", then back substitution; a singular
    // matrix returns dst = 0, and the reference multiplies by it all the same (utils.cpp:78-84: frame_pose becomes 0).
    // (OpenCV's LAPACK HAL declines matrices below 100 rows, so builds with and without LAPACK run this routine.)
    double A[4][4], a[4][8]; // a[.][4 + j]: the right-hand sides (I, finally the inverse)
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            A[i][j] = i < 3 ? (j < 3 ? R[3 * i + j] : t[i]) : (j == 3 ? 1.0 : 0.0);
            a[i][4 + j] = i == j ? 1.0 : 0.0;
        }
    bool singular = false;
    for (int i = 0; i < 4 && !singular; i++) {
        int k = i;
        for (int j = i + 1; j < 4; j++)
            if (fabs(A[j][i]) > fabs(A[k][i]))
                k = j;
        if (fabs(A[k][i]) < DBL_EPSILON * 100) {
            singular = true;
            break;
        }
        if (k != i) {
            for (int j = i; j < 4; j++) {
                const double tmp = A[i][j];
                A[i][j] = A[k][j];
                A[k][j] = tmp;
            }
            for (int j = 0; j < 4; j++) {
                const double tmp = a[i][4 + j];
                a[i][4 + j] = a[k][4 + j];
                a[k][4 + j] = tmp;
            }
        }
        const double d = -1 / A[i][i];
        for (int j = i + 1; j < 4; j++) {
            const double alpha = A[j][i] * d;
            for (int c = i + 1; c < 4; c++)
                A[j][c] += alpha * A[i][c];
            for (int c = 0; c < 4; c++)
                a[j][4 + c] += alpha * a[i][4 + c];
        }
    }
    if (singular) {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++)
                a[i][4 + j] = 0.0;
    } else {
        for (int i = 3; i >= 0; i--)
            for (int j = 0; j < 4; j++) {
                double sum = a[i][4 + j];
                for (int c = i + 1; c < 4; c++)
                    sum -= A[i][c] * a[c][4 + j];
                a[i][4 + j] = sum / A[i][i];
            }
    }
    double out[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double sum = 0;
            for (int k = 0; k < 4; k++)
                sum += pose[4 * i + k] * a[k][4 + j];
            out[4 * i + j] = sum;
        }
    for (int k = 0; k < 16; k++)
        pose[k] = out[k];
    return 1;
}

} // namespace vo
