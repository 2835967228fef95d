// vo_integrate.h -- tail of the reference's frame loop (main.cpp:196-208), shared by the host entry point
// vo_integrate_odometry() and the device-side sequence loop (seq.hip):
//   rotationMatrixToEulerAngles   utils.cpp:107-131  (f64 arithmetic stored to float; x and z swapped w.r.t. MATLAB)
//   the |euler| < 0.1 rad gate    main.cpp:201-207
//   integrateOdometryStereo       utils.cpp:57-91    (frame_pose <- frame_pose * inv([R|t; 0 0 0 1]) iff 0.05 < |t| < 10)
// Plain f64 +,-,*,/ and sqrt in a fixed order (contraction off on both sides), so host and device agree bit for bit
// on the pose; atan2 only feeds the float-rounded gate comparison.
#pragma once

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
    // inverse of the 4x4 [R|t; 0 0 0 1] by Gauss-Jordan elimination with partial pivoting (what
    // cv::Mat::inv() DECOMP_LU amounts to for a well-conditioned 4x4)
    double a[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++)
            a[i][j] = j < 4 ? (i < 3 ? (j < 3 ? R[3 * i + j] : t[i]) : (j == 3 ? 1.0 : 0.0)) : (j - 4 == i ? 1.0 : 0.0);
    for (int col = 0; col < 4; col++) {
        int piv = col;
        for (int r = col + 1; r < 4; r++)
            if (fabs(a[r][col]) > fabs(a[piv][col]))
                piv = r;
        if (fabs(a[piv][col]) < 1e-300)
            return 0;
        if (piv != col)
            for (int j = 0; j < 8; j++) {
                const double tmp = a[col][j];
                a[col][j] = a[piv][j];
                a[piv][j] = tmp;
            }
        const double d = 1.0 / a[col][col];
        for (int j = 0; j < 8; j++)
            a[col][j] *= d;
        for (int r = 0; r < 4; r++)
            if (r != col) {
                const double f = a[r][col];
                for (int j = 0; j < 8; j++)
                    a[r][j] -= f * a[col][j];
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
