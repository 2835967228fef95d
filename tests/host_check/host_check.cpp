// host_check.cpp -- TEST ONLY.  Compiles the VO_HD device-side math headers (vo_linalg.h,
// vo_epnp.h, vo_tri.h) with g++ so the CPU test-suite can compare the *same source* the kernels
// run against the oracle without a GPU.  Not a product path: libvo_hip never links this.
#include "../../visual_odom_amd/csrc/vo_epnp.h"
#include "../../visual_odom_amd/csrc/vo_linalg.h"
#include "../../visual_odom_amd/csrc/vo_tri.h"

extern "C" {
void hc_epnp5(const float *xyz5, const float *uv5, const float *K, double *rvec, double *tvec)
{
    vo::epnp5_solve(xyz5, uv5, K, rvec, tvec);
}
void hc_rodrigues_v2m(const double *r, double *R, double *J) { vo::rodrigues_v2m(r, R, J); }
void hc_rodrigues_m2v(const double *R, double *r) { vo::rodrigues_m2v(R, r); }
void hc_triangulate(const float *Pl, const float *Pr, const float *pl, const float *pr, int n, float *xyz)
{
    for (int i = 0; i < n; i++)
        vo::triangulate_one(Pl, Pr, pl[2 * i], pl[2 * i + 1], pr[2 * i], pr[2 * i + 1], xyz + 3 * i);
}
void hc_solve6(const double *A, const double *b, double *x) { vo::solve_svd<6, 6>(A, b, x); }
}
