/*
 * vo_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's hot path
 *   circularMatching() -> triangulatePoints() -> solvePnPRansac()
 * (ZhenghaoFei/visual_odom: src/feature.cpp:118-148, src/main.cpp:169-171,
 *  src/visualOdometry.cpp:132-193) including the OpenCV 4.5.x CPU algorithms
 * those call sites dispatch to (calcOpticalFlowPyrLK, triangulatePoints,
 * convertPointsFromHomogeneous, solvePnPRansac/EPnP/LevMarq, Rodrigues).
 *
 * PARITY UNPINNED: the reference ships no tests / golden vectors and OpenCV is
 * installed neither in the authoring container nor on the GPU boxes (probed by
 * every route in round 5: tools/opencv_probe.sh, profiles/r05_opencv_probe.txt),
 * so this restatement could not be diffed against the real library.  It is pinned instead by analytic ground
 * truth and independent numpy re-derivations (tests/test_oracle_*.py).  The
 * reference's OWN logic around those algorithms is pinned for real: `make ref`
 * compiles /root/reference/src/{feature,bucket,visualOdometry,utils}.cpp where
 * they lie against a stand-in for the OpenCV declarations (ref_shim/) into
 * oracle/_ref/, with the OpenCV algorithms forwarded to this oracle, and
 * tests/test_reference_glue.py holds orc_glue.c and the oracle call chain to it.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call anything declared here.  The product (visual_odom_amd/, libvo_hip)
 * never does.
 */
#ifndef VO_ORACLE_H
#define VO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- image pyramid / derivative (OpenCV imgproc pyrDown, video lkpyramid) ---- */
/* dst must hold ((w+1)/2)*((h+1)/2) bytes. 5-tap [1 4 6 4 1]^2, REFLECT_101, (v+128)>>8 */
void orc_pyr_down(const uint8_t *src, int w, int h, uint8_t *dst);
/* dst: h*w*2 int16, interleaved (Ix,Iy); Scharr 3x3 unnormalised, REFLECT_101 */
void orc_scharr(const uint8_t *src, int w, int h, int16_t *dst);

/* ---------- cv::calcOpticalFlowPyrLK (feature.cpp:136-139 call sites) ---------------- */
/* accum_mode: how the 441 products of A11 / A12 / A22 and of b1 / b2 are summed (everything else is identical):
 *   0 = exact int64 accumulators, rounded to f32 once (determinism recipe, default: what the HIP kernel computes; OpenCV's
 *       NEON builds also accumulate in integers)
 *   1 = f32 accumulators in scalar pixel order (an x86 build without SIMD, e.g. -DCV_ENABLE_INTRINSICS=OFF)
 *   2 = OpenCV 4.5.x's universal-intrinsics path on x86 (lkpyramid.cpp `#if CV_SIMD128 && !CV_NEON`, v_int16x8 /
 *       v_float32x4 -- 128 bits wide in EVERY x86 build, the file has no AVX2 dispatch) [upstream-memory]: per window row the
 *       first 16 columns go through two 8-pixel blocks -- A: lane (x mod 4) of three v_float32x4 accumulators gets
 *       v_muladd(f, f, q) with f = (float)int16; b: the int16-saturated residuals of pixels (k, k + 4) of a block are paired
 *       by v_dotprod (exact int32), converted to f32 and added to lanes of qb0 (k = 0, 1) / qb1 (k = 2, 3) -- and columns
 *       16 .. 20 through the scalar tail into an f32 scalar; at the end scalar += v_reduce_sum(q) = (q0 + q2) + (q1 + q3).
 *       v_muladd WITHOUT fused multiply-add: the default x86-64 baseline (SSE3) of the distro packages the reference's CI
 *       installs (.github/workflows/cmake.yml:20,25)
 *   3 = the same with v_muladd = _mm_fmadd_ps: a build whose CPU_BASELINE includes FMA3 / AVX2 (only the A sums differ)
 * tools/opencv_crosscheck.py reports which mode a real cv2 matches bit for bit. */
int orc_calc_optical_flow_pyr_lk(const uint8_t *prev, const uint8_t *next, int w, int h,
                                 const float *prev_pts, int n, float *next_pts,
                                 uint8_t *status, float *err,
                                 int win, int max_level, int max_count, double eps,
                                 double min_eig_threshold, int accum_mode, int nthreads);
/* number of LK inner iterations executed by the last call, summed over points/levels (bench) */
long long orc_lk_last_iteration_count(void);
/* hist101[k] = (point, level) solves that ran k inner iterations since the last reset */
void orc_lk_iteration_histogram(long long *hist101, int reset);

/* accum_mode of the four LK calls inside orc_circular_matching* (default 0) */
void orc_set_circular_matching_accum_mode(int mode);
/* ---------- feature.cpp:76-148 : circularMatching + deleteUnmatchFeaturesCircle ------- */
/* pts_l0 [n*2] in; outputs sized n*2 floats each; ages [n_ages] in/out (ages += 1, then
 * compacted together with the points, feature.cpp:83-86,111).  Returns survivors M.
 * status4 (optional, 4*n) receives the raw per-hop LK status before compaction.
 * keep_idx (optional, n) receives original indices of the survivors. */
int orc_circular_matching(const uint8_t *l0, const uint8_t *r0, const uint8_t *l1,
                          const uint8_t *r1, int w, int h, float *pts_l0, int n,
                          float *pts_r0, float *pts_r1, float *pts_l1, float *pts_l0_ret,
                          int *ages, int *n_ages, uint8_t *status4, int *keep_idx,
                          int nthreads);
/* same, maxLevel of the four calcOpticalFlowPyrLK calls as a parameter (reference: 3) */
int orc_circular_matching_lvl(const uint8_t *l0, const uint8_t *r0, const uint8_t *l1,
                              const uint8_t *r1, int w, int h, float *pts_l0, int n,
                              float *pts_r0, float *pts_r1, float *pts_l1, float *pts_l0_ret,
                              int *ages, int *n_ages, uint8_t *status4, int *keep_idx,
                              int nthreads, int max_level);

/* visualOdometry.cpp:44-77,119-125 : checkValidMatch(thr) + removeInvalidPoints x4.
 * Compacts the four arrays in place, returns K.  valid (optional, m) gets the mask. */
int orc_check_valid_and_remove(float *pts_l0, float *pts_r0, float *pts_l1, float *pts_r1,
                               const float *pts_l0_ret, int m, int threshold, uint8_t *valid);

/* ---------- main.cpp:169-171 : triangulatePoints + convertPointsFromHomogeneous ------ */
void orc_triangulate_points(const float *P_l, const float *P_r, const float *pts_l,
                            const float *pts_r, int n, float *points4d /* 4 x n row-major */);
void orc_convert_points_from_homogeneous(const float *points4d_t /* n x 4 */, int n,
                                         float *points3d /* n x 3 */);
/* both in one call: xyz [n*3] */
void orc_triangulate(const float *P_l, const float *P_r, const float *pts_l,
                     const float *pts_r, int n, float *xyz);

/* ---------- visualOdometry.cpp:161-189 : solvePnPRansac + Rodrigues ------------------ */
/* xyz [n*3] f32, uv [n*2] f32, K [9] f32 row-major, rvec/tvec in/out (f64, used as shared
 * buffers exactly like OpenCV does with useExtrinsicGuess=true).  Returns 1 on success,
 * 0 on RANSAC failure, <0 on bad input.  inliers (optional, n) / n_inliers out.
 * dbg (optional, 8 doubles): [0]=niters executed, [1]=best hypothesis index,
 * [2]=maxGoodCount, [3]=LM iterations. */
int orc_solve_pnp_ransac(const float *xyz, const float *uv, int n, const float *K,
                         double *rvec, double *tvec, int iterations_count,
                         float reprojection_error, double confidence, int32_t *inliers,
                         int *n_inliers, double *dbg);
/* solvePnPRansac with exactly 4 points = solvePnP(SOLVEPNP_P3P) (orc_p3p.c): number of solutions (0: rvec / tvec
 * untouched); rvecs / tvecs (optional, [4][3]): all solutions in solveP3P's final order */
int orc_solve_p3p(const float *xyz, const float *uv, const float *K, double *rvec, double *tvec, double *rvecs,
                  double *tvecs);
/* p3p::solve: image points in pixels (f64), K4 = fx fy cx cy; up to 4 poses R [4][9], t [4][3] */
int orc_p3p_solve(const double *K4, const double *uv, const double *xyz, int p4p, double *R_out, double *t_out);
/* polynom_solver.cpp solve_deg4: real roots of a x^4 + b x^3 + c x^2 + d x + e */
int orc_solve_deg4(double a, double b, double c, double d, double e, double *x);
void orc_rodrigues_vec2mat(const double *rvec, double *R /*9*/, double *dRdr /*27 or NULL*/);
void orc_rodrigues_mat2vec(const double *R, double *rvec);
/* EPnP on n>=4 correspondences; K as f32 3x3; uv already in pixels (f32) */
void orc_epnp(const float *xyz, const float *uv, int n, const float *K, double *R, double *t);
/* cv::projectPoints, zero distortion; uv_out f32 [n*2] */
void orc_project_points(const float *xyz, int n, const double *rvec, const double *tvec,
                        const float *K, float *uv_out);
/* the cv::RNG(0xffffffffffffffff) 5-subset stream RANSAC consumes: idx[iters*5] */
void orc_ransac_subsets(int count, int iters, int32_t *idx);

/* ---------- visualOdometry.cpp:146-157 : findEssentialMat(RANSAC) + recoverPose ("next" row f4) ---- */
/* EMEstimatorCallback::runKernel: 5 normalised correspondences q1/q2 [5*2] f64 -> up to 10 row-major
 * essential matrices in Es [10*9]; returns their number */
int orc_five_point(const double *q1, const double *q2, double *Es);
float orc_sampson_error(const double *E, double x1x, double x1y, double x2x, double x2y);
/* cv::findEssentialMat(points1, points2, focal, pp, RANSAC, prob, threshold, mask): pixel points f32
 * [n*2]; E [9] row-major; mask (optional, n) 0/1.  Returns 1 when a model was found.
 * dbg (optional, 3): iterations executed, 10 * iteration + model index of the winner, best inlier count */
int orc_find_essential_mat(const float *pts1, const float *pts2, int n, double focal, double ppx, double ppy,
                           double prob, double threshold, double *E, uint8_t *mask, double *dbg);
void orc_decompose_essential_mat(const double *E, double *R1, double *R2, double *t);
/* cv::recoverPose(E, points1, points2, R, t, focal, pp, mask); mask in/out (NULL: none). Returns #good */
int orc_recover_pose(const double *E, const float *pts1, const float *pts2, int n, double focal, double ppx,
                     double ppy, double *R, double *t, uint8_t *mask);

/* generic one-sided Jacobi SVD (core/lapack.cpp JacobiSVDImpl_<double>), exposed for tests.
 * A is m x n row-major (m >= n).  w[n], u[m*n] (columns = left vectors), vt[n*n]. */
void orc_svd(const double *A, int m, int n, double *w, double *u, double *vt);

/* ---------- host glue restated for full-sequence replay ("next" rows f1/f3) ---------- */
/* feature.cpp:206-253 + bucket.cpp:14-51 (quirks B1-B3 reproduced).  points/ages in/out,
 * capacity cap entries each; n_points/n_ages in/out. */
int orc_bucketing_features(int rows, int cols, float *points, int *ages, int *n_points,
                           int *n_ages, int cap, int bucket_size, int features_per_bucket);
/* cv::FAST(img, thr, nonmax=true) TYPE_9_16; pts [cap*2] f32 in row-major scan order */
int orc_fast_detect(const uint8_t *img, int w, int h, int threshold, int nonmax, float *pts,
                    int cap);
/* utils.cpp:57-131 */
void orc_rotation_matrix_to_euler(const double *R, float *euler3);
int orc_integrate_odometry_stereo(double *frame_pose /*4x4*/, const double *R,
                                  const double *t);

#ifdef __cplusplus
}
#endif
#endif
