// host_check.cpp -- TEST ONLY.  Compiles the VO_HD device-side math headers (vo_linalg.h,
// vo_epnp.h, vo_tri.h, vo_lkmath.h) with g++ so the CPU test-suite can compare the *same source* the kernels
// run against the oracle without a GPU.  Not a product path: libvo_hip never links this.
#include <stdint.h>
#include <string.h>

#include "../../visual_odom_amd/csrc/vo_epnp.h"
#include "../../visual_odom_amd/csrc/vo_fivept.h"
#include "../../visual_odom_amd/csrc/vo_linalg.h"
#include "../../visual_odom_amd/csrc/vo_lkmath.h"
#include "../../visual_odom_amd/csrc/vo_p3p.h"
#include "../../visual_odom_amd/csrc/vo_tri.h"

extern "C" {
void hc_epnp5(const float *xyz5, const float *uv5, const float *K, double *rvec, double *tvec)
{
    vo::epnp5_solve(xyz5, uv5, K, rvec, tvec);
}
// solvePnPRansac's four-point case (vo_p3p.h): number of P3P solutions, rvec / tvec of the first sorted one
int hc_p3p4(const float *xyz4, const float *uv4, const float *K, double *rvec, double *tvec)
{
    return vo::p3p4_solve(xyz4, uv4, K, rvec, tvec);
}
int hc_p3p_deg4(double a, double b, double c, double d, double e, double *x) { return vo::p3p_deg4(a, b, c, d, e, x); }
// vo_math.h: what == 0 cbrt, 1 acos, 2 cos, 3 sin, 4 the Levenberg-Marquardt lambda table over n values
void hc_math(int what, const double *x, int n, double *y)
{
    for (int i = 0; i < n; i++)
        y[i] = what == 0 ? vo::vo_cbrt(x[i]) : what == 1 ? vo::vo_acos(x[i]) : what == 2 ? vo::vo_cos(x[i])
               : what == 3 ? vo::vo_sin(x[i]) : vo::vo_lm_lambda((int)x[i]);
}
void hc_rodrigues_v2m(const double *r, double *R, double *J) { vo::rodrigues_v2m(r, R, J); }
void hc_rodrigues_m2v(const double *R, double *r) { vo::rodrigues_m2v(R, r); }
void hc_triangulate(const float *Pl, const float *Pr, const float *pl, const float *pr, int n, float *xyz)
{
    for (int i = 0; i < n; i++)
        vo::triangulate_one(Pl, Pr, pl[2 * i], pl[2 * i + 1], pr[2 * i], pr[2 * i + 1], xyz + 3 * i);
}
// five-point essential matrix + pose recovery pieces (vo_fivept.h)
int hc_five_point(const double *q1, const double *q2, double *Es) { return vo::five_point_solve(q1, q2, Es); }
float hc_sampson(const double *E, const double *x4) { return vo::em_sampson_error(E, x4[0], x4[1], x4[2], x4[3]); }
void hc_decompose(const double *E, double *R1, double *R2, double *t) { vo::em_decompose(E, R1, R2, t); }
int hc_cheirality(const double *P, const double *x4, double dist)
{
    return vo::em_cheirality(P, x4[0], x4[1], x4[2], x4[3], dist) ? 1 : 0;
}
void hc_solve6(const double *A, const double *b, double *x) { vo::solve_svd<6, 6>(A, b, x); }

// LK packed pixel arithmetic (vo_lkmath.h): n row segments, each 8 + 8 bytes / 8 + 8 Scharr dwords
void hc_bilinear7_u8(const uint8_t *top8, const uint8_t *bot8, const int *w4, int n, int16_t *val7)
{
    for (int i = 0; i < n; i++) {
        uint32_t t[2], b[2], out[4];
        memcpy(t, top8 + 8 * i, 8);
        memcpy(b, bot8 + 8 * i, 8);
        const int *w = w4 + 4 * i;
        vo::bilinear7_u8(t[0], t[1], b[0], b[1], vo::pack_w(w[0], w[1]), vo::pack_w(w[2], w[3]), out);
        for (int k = 0; k < 7; k++)
            val7[7 * i + k] = (int16_t)((out[k / 2] >> (16 * (k & 1))) & 0xffff);
        val7[7 * i + 6] = (int16_t)(out[3] & 0xffff);
        if ((out[3] >> 16) != 0)
            val7[7 * i] = 0x7fff; // the unused 8th slot must stay zero
    }
}
void hc_bilinear7_deriv(const uint32_t *dt8, const uint32_t *db8, const int *w4, int n, int16_t *ix7, int16_t *iy7)
{
    for (int i = 0; i < n; i++) {
        uint32_t ix[4], iy[4];
        const int *w = w4 + 4 * i;
        vo::bilinear7_deriv(dt8 + 8 * i, db8 + 8 * i, vo::pack_w(w[0], w[1]), vo::pack_w(w[2], w[3]), ix, iy);
        for (int k = 0; k < 7; k++) {
            ix7[7 * i + k] = (int16_t)((ix[k / 2] >> (16 * (k & 1))) & 0xffff);
            iy7[7 * i + k] = (int16_t)((iy[k / 2] >> (16 * (k & 1))) & 0xffff);
        }
        if ((ix[3] >> 16) != 0 || (iy[3] >> 16) != 0)
            ix7[7 * i] = 0x7fff;
    }
}
// b1 += diff * Ix over packed pairs, exactly as the kernel's inner loop
void hc_diff_dot(const int16_t *val7, const int16_t *I7, const int16_t *ix7, int n, int *b1)
{
    for (int i = 0; i < n; i++) {
        int acc = 0;
        for (int m = 0; m < 4; m++) {
            auto pk = [&](const int16_t *a) {
                uint32_t lo = (uint16_t)a[7 * i + 2 * m], hi = 2 * m + 1 < 7 ? (uint16_t)a[7 * i + 2 * m + 1] : 0u;
                return lo | (hi << 16);
            };
            acc = vo::sdot2(vo::pk_sub_i16(pk(val7), pk(I7)), pk(ix7), acc);
        }
        b1[i] = acc;
    }
}
uint32_t hc_scharr4(const int *p8)
{
    return vo::scharr4_packed(p8[0], p8[1], p8[2], p8[3], p8[4], p8[5], p8[6], p8[7]);
}
}
