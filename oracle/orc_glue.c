/*
 * orc_glue.c -- CPU ORACLE (test infrastructure, NOT product code).  PARITY UNPINNED for the
 * OpenCV parts, see vo_oracle.h; the glue below follows the reference's own sources line by
 * line and is exact.
 *
 *   src/feature.cpp:76-148      deleteUnmatchFeaturesCircle, circularMatching
 *   src/visualOdometry.cpp:44-77,119-125  checkValidMatch, removeInvalidPoints
 *   src/feature.cpp:206-253 + src/bucket.cpp:14-51   bucketingFeatures (quirks B1-B3)
 *   src/utils.cpp:57-131        integrateOdometryStereo, rotationMatrixToEulerAngles
 */
#include "orc_internal.h"

#include <float.h>
#include <math.h>
#include <string.h>

/* accumulation order of the LK sums inside orc_circular_matching (vo_oracle.h: 0 exact, the checker's default; 2 / 3 OpenCV's
 * x86 order -- bench.py's native CPU baseline times mode 2) */
static int g_glue_accum_mode = 0;
void orc_set_circular_matching_accum_mode(int mode) { g_glue_accum_mode = mode < 0 || mode > 3 ? 0 : mode; }


/* feature.cpp:118-148 */
int orc_circular_matching(const uint8_t *l0, const uint8_t *r0, const uint8_t *l1,
                          const uint8_t *r1, int w, int h, float *p0, int n, float *p1, float *p2,
                          float *p3, float *p0r, int *ages, int *n_ages, uint8_t *status4,
                          int *keep_idx, int nthreads)
{
    return orc_circular_matching_lvl(l0, r0, l1, r1, w, h, p0, n, p1, p2, p3, p0r, ages, n_ages, status4, keep_idx,
                                     nthreads, 3);
}

/* the same with calcOpticalFlowPyrLK's maxLevel as a parameter (the reference hard-codes 3, feature.cpp:136-139;
 * BASELINE config 4 asks for one more pyramid level) */
int orc_circular_matching_lvl(const uint8_t *l0, const uint8_t *r0, const uint8_t *l1,
                              const uint8_t *r1, int w, int h, float *p0, int n, float *p1, float *p2,
                              float *p3, float *p0r, int *ages, int *n_ages, uint8_t *status4,
                              int *keep_idx, int nthreads, int max_level)
{
    uint8_t *st = (uint8_t *)malloc(4 * (size_t)(n > 0 ? n : 1));
    float *err = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    uint8_t *s0 = st, *s1 = st + n, *s2 = st + 2 * n, *s3 = st + 3 * n;
    long long it = 0;
    /* feature.cpp:136-139: win 21, maxLevel 3, COUNT+EPS(30, 0.01), flags 0, minEig 0.001 */
    orc_calc_optical_flow_pyr_lk(l0, r0, w, h, p0, n, p1, s0, err, 21, max_level, 30, 0.01, 0.001, g_glue_accum_mode, nthreads);
    it += orc_lk_last_iteration_count();
    orc_calc_optical_flow_pyr_lk(r0, r1, w, h, p1, n, p2, s1, err, 21, max_level, 30, 0.01, 0.001, g_glue_accum_mode, nthreads);
    it += orc_lk_last_iteration_count();
    orc_calc_optical_flow_pyr_lk(r1, l1, w, h, p2, n, p3, s2, err, 21, max_level, 30, 0.01, 0.001, g_glue_accum_mode, nthreads);
    it += orc_lk_last_iteration_count();
    orc_calc_optical_flow_pyr_lk(l1, l0, w, h, p3, n, p0r, s3, err, 21, max_level, 30, 0.01, 0.001, g_glue_accum_mode, nthreads);
    it += orc_lk_last_iteration_count();
    (void)it;
    if (status4)
        memcpy(status4, st, 4 * (size_t)n);

    /* feature.cpp:76-116 deleteUnmatchFeaturesCircle: ages += 1 for ALL entries of ages, then
     * erase-compaction (order preserving) of the 5 point arrays and of ages by the same index */
    int na = n_ages ? *n_ages : 0;
    for (int i = 0; i < na; i++)
        ages[i] += 1;
    int m = 0;
    int *age_keep = ages ? (int *)malloc(sizeof(int) * (size_t)(na > 0 ? na : 1)) : NULL;
    int ma = 0;
    for (int i = 0; i < n; i++) {
        /* feature.cpp:96-99 -- points0_return is NOT tested */
        int bad = (s3[i] == 0) || (p3[2 * i] < 0) || (p3[2 * i + 1] < 0) || (s2[i] == 0) ||
                  (p2[2 * i] < 0) ||
                  (p2[2 * i + 1] < 0) || (s1[i] == 0) || (p1[2 * i] < 0) || (p1[2 * i + 1] < 0) ||
                  (s0[i] == 0) || (p0[2 * i] < 0) || (p0[2 * i + 1] < 0);
        if (!bad) {
            p0[2 * m] = p0[2 * i];
            p0[2 * m + 1] = p0[2 * i + 1];
            p1[2 * m] = p1[2 * i];
            p1[2 * m + 1] = p1[2 * i + 1];
            p2[2 * m] = p2[2 * i];
            p2[2 * m + 1] = p2[2 * i + 1];
            p3[2 * m] = p3[2 * i];
            p3[2 * m + 1] = p3[2 * i + 1];
            p0r[2 * m] = p0r[2 * i];
            p0r[2 * m + 1] = p0r[2 * i + 1];
            if (keep_idx)
                keep_idx[m] = i;
            m++;
        }
        /* ages.erase(ages.begin() + (i - indexCorrection)) for removed i: element i of ages */
        if (ages && i < na && !bad)
            age_keep[ma++] = ages[i];
    }
    if (ages) {
        /* entries of ages beyond n (quirk B3: ages may be longer than points) are kept */
        for (int i = n; i < na; i++)
            age_keep[ma++] = ages[i];
        memcpy(ages, age_keep, sizeof(int) * (size_t)ma);
        *n_ages = ma;
        free(age_keep);
    }
    free(st);
    free(err);
    return m;
}

/* visualOdometry.cpp:44-61 checkValidMatch + :63-77 removeInvalidPoints (x4, :122-125) */
int orc_check_valid_and_remove(float *pl0, float *pr0, float *pl1, float *pr1, const float *pl0r,
                               int m, int threshold, uint8_t *valid)
{
    int k = 0;
    for (int i = 0; i < m; i++) {
        /* int offset = std::max(std::abs(float), std::abs(float)) : float -> int truncation */
        float ax = fabsf(pl0[2 * i] - pl0r[2 * i]), ay = fabsf(pl0[2 * i + 1] - pl0r[2 * i + 1]);
        int offset = (int)(ax < ay ? ay : ax); /* std::max(a, b) = (a < b) ? b : a */
        int ok = !(offset > threshold);
        if (valid)
            valid[i] = (uint8_t)ok;
        if (ok) {
            pl0[2 * k] = pl0[2 * i];
            pl0[2 * k + 1] = pl0[2 * i + 1];
            pr0[2 * k] = pr0[2 * i];
            pr0[2 * k + 1] = pr0[2 * i + 1];
            pl1[2 * k] = pl1[2 * i];
            pl1[2 * k + 1] = pl1[2 * i + 1];
            pr1[2 * k] = pr1[2 * i];
            pr1[2 * k + 1] = pr1[2 * i + 1];
            k++;
        }
    }
    return k;
}

/* feature.cpp:206-253 with Bucket::add_feature / get_features (bucket.cpp:14-51) */
int orc_bucketing_features(int rows, int cols, float *points, int *ages, int *n_points, int *n_ages,
                           int cap, int bucket_size, int features_per_bucket)
{
    int bh = rows / bucket_size, bw = cols / bucket_size;
    int nb = (bh + 1) * (bw + 1); /* `<=` loops allocate (bh+1)*(bw+1) buckets */
    int fpb = features_per_bucket;
    float *bp = (float *)malloc(sizeof(float) * 2 * (size_t)nb * fpb);
    int *ba = (int *)malloc(sizeof(int) * (size_t)nb * fpb);
    int *bn = (int *)calloc((size_t)nb, sizeof(int));
    int np = *n_points;
    (void)n_ages;
    for (int i = 0; i < np; i++) {
        /* int = float / int : float division then truncation */
        float qy = points[2 * i + 1] / (float)bucket_size, qx = points[2 * i] / (float)bucket_size;
        /* NaN / infinite / huge quotients: (int) of them is undefined in C (x86 gives INT_MIN), and the reference then
         * indexes its vector out of range whatever the conversion gave -- such a feature is ignored, decided on the floats */
        if (!(fabsf(qy) < 32768.f && fabsf(qx) < 32768.f))
            continue;
        int hidx = (int)qy, widx = (int)qx;
        int idx = hidx * bw + widx; /* aliasing quirk B2: stride bw, widx in [0, bw] */
        if (idx < 0 || idx >= nb)
            continue; /* the reference would index out of bounds (UB); never hit for in-image pts */
        int age = ages[i];
        if (age < 10) { /* bucket.cpp:16-17 */
            if (bn[idx] < fpb) {
                bp[2 * (idx * fpb + bn[idx])] = points[2 * i];
                bp[2 * (idx * fpb + bn[idx]) + 1] = points[2 * i + 1];
                ba[idx * fpb + bn[idx]] = age;
                bn[idx]++;
            } else {
                /* bucket.cpp:26-41: compares the INCOMING age with age_min in a loop; the first
                 * i with age < age_min wins, age_min then equals age so no later i can win */
                int age_min = ba[idx * fpb + 0], age_min_idx = 0;
                for (int q = 0; q < bn[idx]; q++)
                    if (age < age_min) {
                        age_min = age;
                        age_min_idx = q;
                    }
                bp[2 * (idx * fpb + age_min_idx)] = points[2 * i];
                bp[2 * (idx * fpb + age_min_idx) + 1] = points[2 * i + 1];
                ba[idx * fpb + age_min_idx] = age;
            }
        }
    }
    int out = 0;
    for (int hh = 0; hh <= bh; hh++)
        for (int ww = 0; ww <= bw; ww++) {
            int idx = hh * bw + ww; /* quirk B2: aliased buckets are emitted twice */
            for (int q = 0; q < bn[idx]; q++) {
                if (out >= cap) {
                    free(bp);
                    free(ba);
                    free(bn);
                    return -1;
                }
                points[2 * out] = bp[2 * (idx * fpb + q)];
                points[2 * out + 1] = bp[2 * (idx * fpb + q) + 1];
                ages[out] = ba[idx * fpb + q];
                out++;
            }
        }
    *n_points = out;
    *n_ages = out;
    free(bp);
    free(ba);
    free(bn);
    return out;
}

/* utils.cpp:107-131 */
void orc_rotation_matrix_to_euler(const double *R, float *e)
{
    float sy = (float)sqrt(R[0] * R[0] + R[3] * R[3]);
    int singular = sy < 1e-6;
    float x, y, z;
    if (!singular) {
        x = (float)atan2(R[7], R[8]);
        y = (float)atan2(-R[6], sy);
        z = (float)atan2(R[3], R[0]);
    } else {
        x = (float)atan2(-R[5], R[4]);
        y = (float)atan2(-R[6], sy);
        z = 0;
    }
    e[0] = x;
    e[1] = y;
    e[2] = z;
}

/* utils.cpp:57-91: frame_pose = frame_pose * inv([R t; 0 0 0 1]) iff 0.05 < |t| < 10.
 * cv::Mat::inv() (DECOMP_LU) restated as the LU elimination cv::invert runs for n > 3 (round 6; the closed form
 * [R^T | -R^T t] before: equal for a rotation up to the last ulps, but not what the reference does with a matrix that
 * is no rotation -- a singular one yields the ZERO inverse and a zeroed frame_pose).  Returns 1 if integrated. */
int orc_integrate_odometry_stereo(double *pose, const double *R, const double *t)
{
    double scale = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    if (!(scale > 0.05 && scale < 10))
        return 0;
    /* rigid_body_transformation.inv() (utils.cpp:78): cv::invert(DECOMP_LU) of a 4 x 4 CV_64F = hal::LU64f on [A | I]
     * (LUImpl: partial pivoting, pivot < DBL_EPSILON * 100 -> singular -> the inverse is set to 0), then back substitution */
    double A[16] = {R[0], R[1], R[2], t[0], R[3], R[4], R[5], t[1], R[6], R[7], R[8], t[2], 0, 0, 0, 1};
    double inv[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    int singular = 0;
    for (int i = 0; i < 4; i++) {
        int k = i;
        for (int j = i + 1; j < 4; j++)
            if (fabs(A[j * 4 + i]) > fabs(A[k * 4 + i]))
                k = j;
        if (fabs(A[k * 4 + i]) < DBL_EPSILON * 100) {
            singular = 1;
            break;
        }
        if (k != i) {
            for (int j = i; j < 4; j++) {
                double tmp = A[i * 4 + j];
                A[i * 4 + j] = A[k * 4 + j];
                A[k * 4 + j] = tmp;
            }
            for (int j = 0; j < 4; j++) {
                double tmp = inv[i * 4 + j];
                inv[i * 4 + j] = inv[k * 4 + j];
                inv[k * 4 + j] = tmp;
            }
        }
        double d = -1 / A[i * 4 + i];
        for (int j = i + 1; j < 4; j++) {
            double alpha = A[j * 4 + i] * d;
            for (int c = i + 1; c < 4; c++)
                A[j * 4 + c] += alpha * A[i * 4 + c];
            for (int c = 0; c < 4; c++)
                inv[j * 4 + c] += alpha * inv[i * 4 + c];
        }
    }
    if (singular)
        memset(inv, 0, sizeof(inv));
    else
        for (int i = 3; i >= 0; i--)
            for (int j = 0; j < 4; j++) {
                double sum = inv[i * 4 + j];
                for (int c = i + 1; c < 4; c++)
                    sum -= A[i * 4 + c] * inv[c * 4 + j];
                inv[i * 4 + j] = sum / A[i * 4 + i];
            }
    double out[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++)
                s += pose[i * 4 + k] * inv[k * 4 + j];
            out[i * 4 + j] = s;
        }
    memcpy(pose, out, sizeof(out));
    return 1;
}
