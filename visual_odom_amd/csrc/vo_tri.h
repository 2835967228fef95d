// vo_tri.h -- per-point stereo triangulation (DLT), drop-in for the
// cv::triangulatePoints + cv::convertPointsFromHomogeneous pair of reference main.cpp:169-171.
#pragma once

#include "vo_linalg.h"

namespace vo {

// P_l / P_r: 3x4 f32 row-major.  (xl,yl),(xr,yr): f32 pixel coordinates.  xyz: 3 f32.
VO_HD void triangulate_one(const float *Pl, const float *Pr, float xl, float yl, float xr, float yr,
                           float *xyz)
{
    double At[16], w[4], vt[16];
    const double x[2] = {(double)xl, (double)xr}, y[2] = {(double)yl, (double)yr};
    for (int j = 0; j < 2; j++) {
        const float *P = j == 0 ? Pl : Pr;
        for (int k = 0; k < 4; k++) {
            // A[(2j)][k], A[(2j+1)][k]; stored transposed: At[k][row]
            At[k * 4 + (j * 2 + 0)] = x[j] * (double)P[8 + k] - (double)P[k];
            At[k * 4 + (j * 2 + 1)] = y[j] * (double)P[8 + k] - (double)P[4 + k];
        }
    }
    jacobi_svd<4, 4, true>(At, w, vt);
    // homogeneous point = right singular vector of the smallest singular value, stored as f32
    // (triangulatePoints output depth follows the Point2f inputs), then /w in f32
    float X = (float)vt[12], Y = (float)vt[13], Z = (float)vt[14], W = (float)vt[15];
    float scale = W != 0.f ? 1.f / W : 1.f;
    xyz[0] = X * scale;
    xyz[1] = Y * scale;
    xyz[2] = Z * scale;
}

} // namespace vo
