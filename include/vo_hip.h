/*
 * vo_hip.h -- C ABI of libvo_hip.so: the MI355X (gfx950) stereo visual-odometry front end.
 *
 * The reference (ZhenghaoFei/visual_odom) has no plugin / FFI layer: its hot path is reached
 * through plain C++ free functions.  Every entry point below names the reference interface it
 * replaces (file:line into the reference tree); INTEGRATION.md shows the C++ adapter a maintainer
 * adds so that circularMatching() / trackingFrame2Frame() call these instead of OpenCV.
 *
 * Conventions: plain pointers and sizes, no exceptions, no torch types.  Return value 0 = VO_OK or
 * a negative VO_ERR_* code (vo_last_error() gives text).  Caller allocates every output; the
 * library owns all device memory and its HIP stream inside vo_ctx.  One vo_ctx per host thread per
 * GPU; a ctx is not thread-safe.  Points are interleaved float32 (x, y) like cv::Point2f, images
 * are 8-bit gray row-major with a byte stride, like the continuous CV_8UC1 cv::Mat the reference
 * passes around (utils.cpp:179).
 */
#ifndef VO_HIP_H
#define VO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VO_OK 0
#define VO_ERR_ARG (-1)      /* bad argument / size beyond the capacity given to vo_create */
#define VO_ERR_HIP (-2)      /* a HIP runtime call failed */
#define VO_ERR_STATE (-3)    /* call order violated (e.g. run before configure) */
#define VO_ERR_TOO_FEW (-4)  /* fewer than 4 correspondences reached solvePnPRansac (OpenCV would CV_Assert), fewer
                                than 5 reached findEssentialMat */
#define VO_ERR_OVERFLOW (-5) /* detection / bucketing produced more features than the capacity given to vo_create:
                                the stored result is truncated and differs from the reference's -- never silent */
/* positive return codes of the pose calls: the call worked, the reference's algorithm reported a failure */
#define VO_NO_MODEL 1        /* solvePnPRansac returned false (no consensus); rvec / tvec hold the last hypothesis */
#define VO_NO_ESSENTIAL 2    /* mono_rotation: findEssentialMat found no model (the reference's recoverPose throws);
                                R_out is left untouched.  Outranks VO_NO_MODEL: the reference throws at
                                visualOdometry.cpp:152-153, before it reaches solvePnPRansac -- rvec / tvec / inliers are
                                filled as usual, n_inliers == 0 tells a frame whose PnP found nothing either */

typedef struct vo_ctx vo_ctx;

/* LK / RANSAC parameters; vo_default_params() fills the reference's literals
 * (feature.cpp:127-128,136: win 21 (fixed), maxLevel 3, COUNT+EPS 30 / 0.01, minEig 1e-3;
 *  visualOdometry.cpp:168-172: 500 iterations, 0.5 px, confidence 0.999f;
 *  visualOdometry.cpp:120: circular-consistency threshold 0). */
typedef struct vo_params {
    int lk_max_level;
    int lk_max_count;
    double lk_epsilon;
    double lk_min_eig_threshold;
    /* 0 (default): a feature stops at the first hop whose result deleteUnmatchFeaturesCircle() would
     * reject (status 0 or a negative coordinate, feature.cpp:96-104) -- its later hops are reported as
     * status 0 and never reach any output of the reference's interface, so the compacted results are
     * unchanged.  1: every feature runs all four hops like the reference's four independent
     * calcOpticalFlowPyrLK calls, so the raw per-hop tracks / status4 are reproduced too. */
    int lk_full_chain;
    int consistency_threshold;
    int ransac_iterations;
    float ransac_reproj_error;
    double ransac_confidence;
    /* trackingFrame2Frame(..., bool mono_rotation) (visualOdometry.h:42, .cpp:146-157).  0 (what main.cpp:181
     * passes): rotation = Rodrigues(rvec) of the PnP solve.  1: rotation comes from
     * findEssentialMat(points_t0, points_t1, focal, pp, RANSAC, em_prob, em_threshold) + recoverPose on the
     * left-image tracks; the PnP solve still provides the translation. */
    int mono_rotation;
    double em_prob;      /* 0.999 */
    double em_threshold; /* 1.0 px */
} vo_params;

void vo_default_params(vo_params *p);

/* Detection / bucketing parameters; vo_default_detect_params() fills the reference's literals
 * (feature.cpp:43-45: FAST threshold 20, nonmaxSuppression true; visualOdometry.cpp:95: re-detect when
 *  fewer than 2000 features are carried over; :106-107: bucket_size = rows / 10 (0 here = that rule),
 *  features_per_bucket = 1).
 * LIMITS of the device bucketing (VO_ERR_ARG beyond them, nothing is computed): 1 <= features_per_bucket <= 8; buckets counted
 * as the reference allocates them, n = (rows / bucket_size + 1) * (cols / bucket_size + 1) -- KITTI at rows / 10: 11 x 34 =
 * 374, 1080p: 11 x 18 = 198 -- n <= 1024, or (fine grids) n <= 4096 with n * features_per_bucket <= 8192; images up to 4096
 * pixels wide. */
typedef struct vo_detect_params {
    int fast_threshold;
    int fast_nonmax;
    int redetect_below;
    int bucket_size;
    int features_per_bucket;
} vo_detect_params;

void vo_default_detect_params(vo_detect_params *p);

/* device: HIP device ordinal.  max_w/max_h: largest image.  max_pts: per-frame point capacity.
 * max_frames: largest batch for the vo_batch_* API (1 is enough for the drop-in calls).
 * Returns NULL on failure (no HIP device, out of memory). */
vo_ctx *vo_create(int device, int max_w, int max_h, int max_pts, int max_frames);
void vo_destroy(vo_ctx *ctx);
const char *vo_last_error(const vo_ctx *ctx);
int vo_set_params(vo_ctx *ctx, const vo_params *p);

/* How the pose chain (PnP / RANSAC, f64) is scheduled next to the tracking stages (pyramids, FAST, LK).  None of the
 * three knobs changes a result.  By default (all "probe") the first run of a new (mode, image size, frames per run,
 * point-load) key times the candidates on the caller's own data -- a batch run is idempotent; a lock-step step is
 * repeated without the two kernels that advance its state -- keeps the fastest and remembers it for the process.  That
 * first run therefore takes about a dozen runs' time.  Pin a knob to skip its probe (all three pinned: no probe at
 * all).  The reference has no counterpart: its calls are synchronous CPU code.
 *   pose_waves    register budget of the pose kernels in waves per SIMD: 1 = 512 registers, 2 = 256; 0 = probe
 *   pose_streams  1 or 2 pose streams (2: the chains of consecutive runs overlap); 0 = probe
 *   prepare       lock-step loop: pyramids + FAST of the new pairs one step ahead on a prepare stream; -1 = probe
 *   epnp_wide_frames  see the field below; 0 = probe */
typedef struct vo_schedule {
    int pose_waves;
    int pose_streams;
    int prepare;
    /* round 4: the four-kernel form of the EPnP hypotheses (12 x 12 SVD by two wavefronts per hypothesis) for launches of up
     * to this many frames: 4 or 16; 0 = probe.  Launches of <= 4 frames always take it, launches of > 16 never (the knob only
     * acts for 5 .. 16 frames / sequences per run: at 640 x 480 the wide form wins by 10-30 % there, at 1241 x 376 it loses
     * 3-10 % at 16 -- measured, profiles/r04_experiments.md -- so it is part of the probed schedule, not a constant). */
    int epnp_wide_frames;
} vo_schedule;
/* s == NULL: probe everything (the default) */
int vo_set_schedule(vo_ctx *ctx, const vo_schedule *s);
/* the schedule the next run will use; *probed (optional) = 1 when it came out of a probe of this key, 2 while the
 * lock-step loop is still comparing candidates over real steps (see below), 0 for defaults / pins.
 * Lock-step loop: the probe's repeated runs of a step cannot include the two kernels that advance the sequences'
 * state, and what they hide shifts the ranking by 5-25 %; so the probe only nominates -- one candidate per
 * (pose_streams, prepare), each with the pose_waves the probe prefers for it, runs for 34 .. 58 REAL steps (10 of them an
 * untimed ramp), then -- round 6, from 32 sequences on: the dry runs prefer the wrong budget at some sizes -- each of them
 * once more with the OTHER pose_waves, i.e. all eight schedules; end-of-step GPU timestamps decide (a pipeline drain
 * wherever two consecutive candidates differ in `prepare`, all within the first ~250-450 steps of a loop; results
 * never depend on any of it).  The synchronous drop-in call vo_track_frame compares its candidates by the latency of
 * a run, everything else by steady-state throughput. */
int vo_get_schedule(const vo_ctx *ctx, vo_schedule *current, int *probed);
/* what the last probe run by this context measured: *n (<= VO_PROBE_LOG_MAX) candidates and their steady-state
 * milliseconds per run (the arrays hold VO_PROBE_LOG_MAX entries); real[i] (optional) = 1 where the figure was re-measured
 * over real steps of the lock-step loop */
#define VO_PROBE_LOG_MAX 16
int vo_get_probe_log(const vo_ctx *ctx, vo_schedule *cands, float *ms, int *real, int *n);
/* The process-wide table of settled schedules, out and in: a service exports it once (after a warm-up run of every shape it
 * uses) and imports it at start-up, so that no context of the new process probes -- the first run of a probed key otherwise
 * costs 25-100 runs' time (at most 2 x 2 x 2 + 1 candidates of 7 + K runs, K = 6 .. 24; 0.03 s for one 1241 x 376 frame, ~0.6 s for a 256-frame batch at 2000 points, ~1.2 s for 256
 * sequences; an upper bound: ~100 runs of the caller's own shape) and, in the lock-step loop, up to three pipeline drains
 * within its first ~200 steps.  key = (device, mode, width, height, pyramid levels, frames per run, point-load bucket, flags);
 * records are valid for the same library build and device model.  Neither call needs a context.
 *   vo_export_schedule: writes min(*n, cap) records, *n = records in the table (recs may be NULL with cap = 0 to ask); a key
 *                       whose comparison over real steps is still running (vo_get_schedule: probed = 2) is not in the table yet
 *   vo_import_schedule: all records are validated first (VO_ERR_ARG leaves the table untouched); existing keys are replaced */
typedef struct vo_schedule_record {
    int64_t key[8];
    vo_schedule schedule;
} vo_schedule_record;
int vo_export_schedule(vo_schedule_record *recs, int cap, int *n);
int vo_import_schedule(const vo_schedule_record *recs, int n);
int vo_get_params(const vo_ctx *ctx, vo_params *p);

/* ------------------------------------------------------------------------------------------
 * Drop-in calls (host buffers in, host buffers out, synchronous).
 * ------------------------------------------------------------------------------------------ */

/* Replaces circularMatching() -- feature.h:61-65 / feature.cpp:118-148 -- i.e. the four
 * cv::calcOpticalFlowPyrLK calls plus deleteUnmatchFeaturesCircle() (feature.cpp:76-116).
 * pts_l0_xy [2n].  out_* [2n] each, compacted to *n_out survivors in input order.
 * status4 (optional, [4][n]): raw LK status of the 4 hops before compaction.
 * keep_idx (optional, [n]): input index of each survivor (the adapter compacts `ages` with it).
 * apply_consistency != 0 additionally applies checkValidMatch(thr) + removeInvalidPoints
 * (visualOdometry.cpp:44-77,119-125) so the outputs are the K points that reach triangulation.
 * img_l0 == img_r0 == NULL: the t0 pair is the previous call's t1 pair (see vo_track_frame, THE KEPT PAIR).
 * VALUES (round 6; tests/adversarial.py, tests/test_gpu_round6.py): a start point with a NaN coordinate, +-inf or a value
 * beyond int32 fails its first hop with status 0 exactly as in calcOpticalFlowPyrLK (x86's cvFloor(NaN) is INT_MIN: "left of
 * the window"), its reported positions are the propagated (NaN / huge) values, and deleteUnmatchFeaturesCircle drops it;
 * denormal, negative and out-of-image coordinates take the reference's path (+-winSize admissibility window, then the sign
 * tests of feature.cpp:96-104); constant / saturated / 1-pixel-checkerboard images fail the min-eigenvalue test as there. */
int vo_circular_match(vo_ctx *ctx, const uint8_t *img_l0, const uint8_t *img_r0, const uint8_t *img_l1,
                      const uint8_t *img_r1, int w, int h, int stride, const float *pts_l0_xy, int n,
                      float *out_l0, float *out_r0, float *out_r1, float *out_l1, float *out_l0_ret,
                      uint8_t *status4, int32_t *keep_idx, int *n_out, int apply_consistency);

/* Replaces cv::triangulatePoints + cv::convertPointsFromHomogeneous -- main.cpp:169-171.
 * P_l / P_r: 3x4 float32 row-major (main.cpp:73-74).  xyz_out [3n] float32 (N x 1 CV_32FC3). */
int vo_triangulate(vo_ctx *ctx, const float *P_l, const float *P_r, const float *pts_l_xy,
                   const float *pts_r_xy, int n, float *xyz_out);

/* Replaces cv::solvePnPRansac(..., useExtrinsicGuess=true, SOLVEPNP_ITERATIVE) + cv::Rodrigues --
 * visualOdometry.cpp:161-189.  K: 3x3 float32 row-major.  rvec_io / tvec_io: f64[3]; on success
 * they receive the refined pose (on VO_ERR_TOO_FEW they are left untouched).  R_out (optional,
 * f64[9] row-major) = Rodrigues(rvec).  inliers (optional, int32[n]) / n_inliers as cv::Mat
 * inliers.  Iterations / threshold / confidence come from vo_params.
 * R_out is Rodrigues(rvec) whatever vo_params.mono_rotation says (that flag belongs to vo_track_frame).
 * Returns VO_OK when a model was found, VO_NO_MODEL when RANSAC found none (OpenCV returns false).
 * Point counts: n >= 6 RANSAC over 5-point EPnP + refinement; n == 5 EPnP on all five; n == 4 OpenCV's P3P switch
 * (`npoints == 4 -> SOLVEPNP_P3P`, solvePnP's answer as is, all four points inliers; without a P3P solution
 * VO_NO_MODEL with rvec_io / tvec_io UNTOUCHED, as solvePnP leaves them); n < 4 VO_ERR_TOO_FEW. */
int vo_pnp_ransac(vo_ctx *ctx, const float *xyz, const float *uv, int n, const float *K, double *rvec_io,
                  double *tvec_io, double *R_out, int32_t *inliers, int *n_inliers);

/* replaces the pair  E = cv::findEssentialMat(pts0, pts1, focal, pp, cv::RANSAC, prob, threshold, mask);
 *                     cv::recoverPose(E, pts0, pts1, R, t, focal, pp, mask);
 * of reference src/visualOdometry.cpp:152-153 (pixel coordinates, f32 xy pairs).  E, R: 3x3 row-major f64;
 * t: unit translation; mask (optional, n bytes): 1 = RANSAC inlier that passes the cheirality check;
 * *n_good = recoverPose's return value.  Returns VO_OK, 1 when RANSAC found no model (outputs untouched),
 * VO_ERR_TOO_FEW below 5 points. */
int vo_essential_pose(vo_ctx *ctx, const float *pts0_xy, const float *pts1_xy, int n, double focal, double ppx,
                      double ppy, double prob, double threshold, double *E, double *R, double *t, uint8_t *mask,
                      int *n_good);
/* Replaces cv::FAST as called by featureDetectionFast() -- feature.cpp:39-47: TYPE_9_16 corners of an
 * 8-bit image in row-major order.  pts_out [2 * cap]; *n_out = corners found (may exceed cap, in which
 * case only the first cap are written).  VO_ERR_OVERFLOW when the corners exceed the context's own corner-list
 * capacity max(4 * max_pts, 16384, max_w * max_h / 16) although the caller's cap would have held them. */
int vo_fast_detect(vo_ctx *ctx, const uint8_t *img, int w, int h, int stride, int threshold, int nonmax,
                   float *pts_out, int cap, int *n_out);

/* Replaces the head of matchingFeatures() -- visualOdometry.cpp:95-108: appendNewFeatures(image, set)
 * when the set has fewer than redetect_below points (feature.cpp:255-262), then bucketingFeatures()
 * (feature.cpp:206-253, bucket.cpp:14-51, quirks of SURVEY.md App. B1-B3 reproduced).
 * pts_io [2 * cap] / ages_io [cap]: in: *n_pts points and *n_ages ages (n_ages >= n_pts allowed, as in
 * the reference after a consistency filter); out: the bucketed set (*n_pts == *n_ages).
 * img == NULL (also vo_fast_detect): the left image of the pair vo_track_frame kept (see there).
 * A carried point whose bucket index the reference would read outside its bucket vector (NaN / infinite / huge / far
 * negative coordinates: undefined behaviour in feature.cpp:233-236) is IGNORED here (quotients beyond +-32768 or an index
 * outside the (rows/bs + 1) x (cols/bs + 1) buckets); every in-image point takes the reference's path. */
int vo_detect_bucket(vo_ctx *ctx, const uint8_t *img, int w, int h, int stride, const vo_detect_params *dp,
                     float *pts_io, int *n_pts, int32_t *ages_io, int *n_ages, int cap);

/* Replaces the tail of the frame loop -- main.cpp:196-208: rotationMatrixToEulerAngles (utils.cpp:107-131),
 * the |euler| < 0.1 rad gate, and integrateOdometryStereo (utils.cpp:57-91: frame_pose <- frame_pose *
 * inv([R|t; 0 0 0 1]) iff 0.05 < |t| < 10).  Pure host arithmetic (4x4 f64), no ctx, no device.
 * pose16: 4x4 row-major f64 in/out.  euler_out (optional, float[3]).  Returns 1 when the motion was
 * integrated, 0 when a gate rejected it (pose unchanged, as in the reference). */
int vo_integrate_odometry(double *pose16, const double *R9, const double *t3, float *euler_out);

/* The whole per-frame hot path in one call: circularMatching + consistency filter + triangulation
 * + PnP/RANSAC (matchingFeatures' tail visualOdometry.cpp:116-127, main.cpp:169-181), one upload,
 * one download.  out_l0/out_r0/out_l1/out_r1 [2n] and xyz_out [3n] are compacted to *n_out (= K).
 * keep_idx (optional, [n]) -> input index of each of the K points; keep_idx_circ/n_circ (optional)
 * -> survivors of deleteUnmatchFeaturesCircle alone (what `ages` is compacted with, quirk B3).
 * Images and points are pageable host memory; the caller's buffers are free on return.  (Inside: each image is
 * repacked into a page-locked slot and read by the GPU over PCIe while the host repacks the next one, nothing
 * synchronises before the results -- 0.65-0.68 ms per KITTI frame at ~2000 points on an MI355X)
 *
 * THE KEPT PAIR (the reference's `imageLeft_t0 = imageLeft_t1; imageRight_t0 = imageRight_t1`, main.cpp:157-158):
 * img_l0 == NULL and img_r0 == NULL name the stereo pair the previous vo_track_frame / vo_circular_match of this context
 * received as (img_l1, img_r1) -- it is still on the device with its pyramids, so only the new pair crosses the link and
 * only its pyramids are built (and the chain's first hop, l0 -> r0, runs meanwhile: 0.61-0.63 ms per call); results are
 * those of the call with all four images.  vo_detect_bucket / vo_fast_detect
 * with img == NULL read the kept pair's LEFT image (what matchingFeatures detects on next, visualOdometry.cpp:95-108);
 * with an image of their own they leave the kept pair alone.  VO_ERR_STATE when there is no kept pair of this size:
 * first call, another w x h, or a vo_batch_* upload / configure of another shape / vo_seq_configure since (they own the
 * image table from then on).  One NULL and one non-NULL t0 image is VO_ERR_ARG.
 * ONE CALLER PER KEPT PAIR: the pair is the context's, not the caller's -- two users of one context (two frame loops, or a
 * loop plus direct calls) with images of the same size would read each other's t1 pair as t0 without any error.  A user who
 * shares a context remembers vo_kept_pair_id() after its call and passes NULL only while the id is unchanged (what
 * visual_odom_amd.odometry.StereoOdometry does); a context with a single frame loop needs none of this. */
int vo_track_frame(vo_ctx *ctx, const uint8_t *img_l0, const uint8_t *img_r0, const uint8_t *img_l1,
                   const uint8_t *img_r1, int w, int h, int stride, const float *pts_l0_xy, int n,
                   const float *P_l, const float *P_r, float *out_l0, float *out_r0, float *out_l1,
                   float *out_r1, float *xyz_out, int32_t *keep_idx, int *n_out, int32_t *keep_idx_circ,
                   int *n_circ, double *rvec_io, double *tvec_io, double *R_out, int32_t *inliers,
                   int *n_inliers);

/* identity of the kept pair: > 0 and different after every vo_track_frame / vo_circular_match that left a new t1 pair on the
 * device; 0 = there is none (no call yet, or the batch / sequence API owns the image table) */
int64_t vo_kept_pair_id(const vo_ctx *ctx);

/* ------------------------------------------------------------------------------------------
 * Batched, device-resident API (throughput mode: many independent frames per launch so that a
 * 256-CU part is filled; inputs stay in HBM between calls).
 *   image table: n_images pyramids; frame f = quad of image indices (l0, r0, l1, r1).
 * ------------------------------------------------------------------------------------------ */
#define VO_STAGE_PYRAMID 1
#define VO_STAGE_LK 2
#define VO_STAGE_FILTER 4
#define VO_STAGE_TRIANGULATE 8
#define VO_STAGE_PNP 16
#define VO_STAGE_ALL 31      /* the path with the LK input points given (vo_batch_set_points) */
#define VO_STAGE_DETECT 32   /* + FAST / bucketing of every frame's left t0 image produce those points */
#define VO_NUM_STAGES 6      /* timing order: PYRAMID, DETECT, LK, FILTER, TRIANGULATE, PNP */

int vo_batch_configure(vo_ctx *ctx, int n_images, int w, int h, int n_frames);
/* host -> device copy of one level-0 image */
int vo_batch_upload_image(vo_ctx *ctx, int image_idx, const uint8_t *host_pixels, int stride);
/* device -> device copy (e.g. from a torch uint8 tensor's data_ptr()) */
int vo_batch_upload_image_dev(vo_ctx *ctx, int image_idx, const void *dev_pixels, int stride);
/* quads4 [n_frames][4] = (l0, r0, l1, r1) image indices.  A synchronous drop-in call (vo_track_frame, vo_circular_match,
 * vo_fast_detect, vo_detect_bucket) runs on frame 0 with a quadruple of its own: set the quads again before the next
 * vo_batch_run that follows one (a run without it reads what vo_batch_set_quads last uploaded, all zero after a new shape) */
int vo_batch_set_quads(vo_ctx *ctx, const int32_t *quads4, int n_frames);
/* Which images VO_STAGE_PYRAMID (re)builds: [first_image, first_image + n_images).  Default after
 * vo_batch_configure: all of them.  A streaming caller keeps the image table as a ring, uploads only the
 * new stereo pair and builds only its two pyramids: the t1 pyramids of one frame are the t0 pyramids of the
 * next (the reference rebuilds all four pyramids twice per frame inside calcOpticalFlowPyrLK,
 * feature.cpp:136-139; main.cpp:157-158 only swaps the cv::Mat headers). */
int vo_batch_set_pyramid_range(vo_ctx *ctx, int first_image, int n_images);
int vo_batch_set_points(vo_ctx *ctx, int frame, const float *pts_l0_xy, int n);
int vo_batch_set_projection(vo_ctx *ctx, const float *P_l, const float *P_r);
/* VO_STAGE_DETECT inputs: the features carried into `frame` from the previous frame (n_pts may be 0;
 * n_ages >= n_pts), and the detection parameters (NULL = reference defaults).  The stage leaves the
 * bucketed set as the frame's LK input points, exactly like vo_batch_set_points would. */
int vo_batch_set_features(vo_ctx *ctx, int frame, const float *pts_xy, int n_pts, const int32_t *ages,
                          int n_ages);
int vo_batch_set_detect_params(vo_ctx *ctx, const vo_detect_params *dp);
/* the bucketed set of one frame after VO_STAGE_DETECT (after vo_batch_sync); pts [2 * cap], ages [cap].
 * VO_ERR_OVERFLOW when the frame's carried + detected features or its bucketed set exceeded the capacity
 * (*n and the arrays are still filled with the truncated set). */
int vo_batch_get_features(vo_ctx *ctx, int frame, float *pts_xy, int32_t *ages, int *n);
/* enqueue the selected stages for all frames on the ctx stream (asynchronous) */
int vo_batch_run(vo_ctx *ctx, int stages);
/* same, bracketed per stage by HIP events on the ctx stream; blocks; ms_per_stage[VO_NUM_STAGES]
 * in the order PYRAMID, DETECT, LK, FILTER, TRIANGULATE, PNP */
int vo_batch_run_timed(vo_ctx *ctx, int stages, float *ms_per_stage);
/* asynchronous variant: like vo_batch_run, with the per-stage HIP events of ring slot `slot`
 * (0 <= slot < VO_EVENT_SLOTS) recorded on the ctx stream; after vo_batch_sync,
 * vo_batch_slot_times returns the stage durations of that slot.  Lets a benchmark time K
 * back-to-back steps without a host round trip per step. */
#define VO_EVENT_SLOTS 256
int vo_batch_run_slot(vo_ctx *ctx, int stages, int slot);
int vo_batch_slot_times(vo_ctx *ctx, int slot, float *ms_per_stage);
int vo_batch_sync(vo_ctx *ctx);
/* results of one frame (after vo_batch_sync); any pointer may be NULL */
int vo_batch_get_tracks(vo_ctx *ctx, int frame, float *r0, float *r1, float *l1, float *l0_ret,
                        uint8_t *status4, int n);
int vo_batch_get_filtered(vo_ctx *ctx, int frame, float *l0, float *r0, float *l1, float *r1, float *xyz,
                          int32_t *keep_idx, int *n_out, int32_t *keep_idx_circ, int *n_circ);
/* rvec / tvec / R are pure outputs here (a batch frame has no caller-side pose): a four-point frame whose P3P has no
 * solution (status 0, lm_iters -1) returns rvec = tvec = 0 and R = identity -- the reference's zeroed rvec,
 * visualOdometry.cpp:162 -- whereas the drop-in calls vo_pnp_ransac / vo_track_frame leave their rvec_io / tvec_io untouched */
int vo_batch_get_pose(vo_ctx *ctx, int frame, double *rvec, double *tvec, double *R, int32_t *inliers,
                      int *n_inliers, int *status, int32_t *dbg4 /* niters, best, max_good, lm_iters */);
/* with vo_params.mono_rotation the R of vo_batch_get_pose / vo_track_frame is recoverPose's rotation (left
 * untouched when no essential matrix was found, where OpenCV throws).  The essential-matrix side of a frame:
 * E, R (3x3 row-major), t (unit translation), mask (n bytes, 1 = RANSAC inlier passing the cheirality check),
 * status 1 ok / 0 no model / -1 fewer than 5 points; dbg2 = samples drawn, 10 * sample + model of the winner */
int vo_batch_get_essential(vo_ctx *ctx, int frame, double *E, double *R, double *t, uint8_t *mask, int n,
                           int *n_inliers, int *n_good, int *status, int32_t *dbg2);
/* ------------------------------------------------------------------------------------------
 * Lock-step sequence loop: the reference's frame loop (main.cpp:123-224) for S independent sequences at
 * once, one frame of every sequence per step, with the state that chains frame k to frame k + 1 --
 * currentVOFeatures (points = pointsLeft_t1, visualOdometry.cpp:127; ages, feature.cpp:83-86,111, quirk B3),
 * the previous stereo pair (main.cpp:157-158) and frame_pose (utils.cpp:84) -- resident in HBM.  Per step the
 * host only hands over the new stereo pair of every sequence; nothing comes back until it asks.
 * Within a sequence frames are serial, so S sequences x 1 frame is the exact-replay way to fill the GPU
 * (BASELINE config 5: sequences 00-07, one vo_ctx per GPU, several sequences per ctx).
 *
 *   vo_seq_configure(ctx, S, w, h, ring, max_steps)   ring = stereo pairs resident per sequence: 2, or 3 so that
 *                                                     the upload of pair k + 1 overlaps the step on pairs (k - 1, k)
 *   per step:  vo_seq_push_pair(ctx, s, left, right, stride, pinned)  for every sequence that has a new pair
 *              vo_seq_step(ctx)                                        asynchronous
 *   A sequence processes a frame in a step iff it received a pair for this step AND for the previous one (its
 *   first pair only builds pyramids, main.cpp:110-113); a sequence that receives nothing pauses, and when it resumes
 *   the first pair again only builds pyramids: the transition across the pause is dropped and the next processed frame
 *   is flagged VO_SEQ_F_GAP.
 *   Detection / LK / RANSAC parameters: vo_set_params and vo_batch_set_detect_params before vo_seq_configure (or at
 *   least before the first vo_seq_step: with few sequences FAST runs on a pair's left image as soon as the pair is on
 *   the device, one step before its corners are needed), vo_batch_set_projection before the first step.
 *   S == 1 runs on a second set of the context's streams whose kernels are confined to disjoint halves of the
 *   device's compute units (tracking / copies on one, everything behind LK on the other): one sequence is two
 *   latency-bound chains that otherwise slow each other.  Same results; any other configuration call returns the
 *   context to its ordinary streams.
 * ------------------------------------------------------------------------------------------ */
#define VO_SEQ_ROW 27          /* doubles per trajectory row: frame_pose 3x4 (12), rvec (3), tvec (3), rotation (9) */
#define VO_SEQ_INFO 8          /* ints per row: n_bucketed, n_circ, n_tracked, n_inliers, pnp_status, flags,
                                  ransac_iters, overflow */
#define VO_SEQ_F_ACTIVE 1      /* flags */
#define VO_SEQ_F_INTEGRATED 2  /* the motion passed the gates of main.cpp:201 / utils.cpp:80 and was integrated */
#define VO_SEQ_F_TOO_FEW 4     /* fewer than 4 points reached solvePnPRansac (the reference asserts) */
#define VO_SEQ_F_NO_ESSENTIAL 8
#define VO_SEQ_F_GAP 16        /* first frame processed after the sequence had paused (no pair for >= 1 step): the motion
                                  between the last pair before the pause and the first pair after it was never
                                  estimated -- the carried features (tracked in the last pair before the pause) and
                                  frame_pose are kept, the image pair restarts.  The reference's loop has no such
                                  case (it reads consecutive files, main.cpp:123-158); a caller that wants a clean
                                  restart calls vo_seq_reset(seq) instead of resuming. */
int vo_seq_configure(vo_ctx *ctx, int n_seq, int w, int h, int ring, int max_steps);
/* empty feature set, identity pose, no trajectory rows, no resident pair -- for sequence `seq`, or all if seq < 0.
 * seq < 0 also rewinds the loop's step counter (all max_steps rows of every sequence are available again) and clears
 * the "a step failed half-way" state after which every vo_seq_push_pair / vo_seq_step returns VO_ERR_STATE.
 * vo_seq_step refuses (VO_ERR_STATE, the pairs pushed for that step are dropped) when an active sequence has used up
 * its max_steps rows; vo_seq_reset(seq) gives them back.  The refusal is a PAUSE for every sequence whose pair it dropped,
 * whatever the caller pushes next -- the same pairs again or later ones (the library cannot tell which): the reset sequence
 * starts over with its next pair; each OTHER sequence of the dropped step restarts its image pair with its next pair (that
 * pair builds pyramids only: one frame is not processed) and the frame after it carries VO_SEQ_F_GAP -- a pair is never
 * matched against the pair from two pushes ago.  To lose no frame, size max_steps for the run or reset between runs. */
int vo_seq_reset(vo_ctx *ctx, int seq);
/* the next stereo pair of sequence `seq` (8-bit gray, byte stride).  host_pinned = 0: pageable memory, staged
 * through the library's pinned buffers (the call returns when the images have been copied out of the caller's
 * memory; vo_seq_push_pairs spreads a step's copies over up to 8 host threads; from 32 sequences on, a step whose pairs
 * are ALL pageable crosses the link as one contiguous copy-engine transfer -- the fastest way in: 256 KITTI sequences at
 * 2 000 points per frame run at 24.9 k frames/s from pageable memory, 23.3 k from page-locked, 25.6 k resident).  host_pinned = 1: page-locked memory (hipHostMalloc / hipHostRegister / torch pin_memory) read by the
 * copy engine directly; it must stay unchanged until the step that consumes it has finished.  Either way the
 * transfer runs on a copy stream next to the previous step's kernels. */
int vo_seq_push_pair(vo_ctx *ctx, int seq, const uint8_t *left, const uint8_t *right, int stride, int host_pinned);
/* same from device memory (e.g. torch uint8 tensors) */
int vo_seq_push_pair_dev(vo_ctx *ctx, int seq, const void *left, const void *right, int stride);
/* n pairs in one call: sequence seq_ids[i] gets (left[i], right[i]); kind 0 = pageable host, 1 = page-locked host,
 * 2 = device memory.  (A push only records where a pair is -- pageable images are copied to pinned staging -- and
 * vo_seq_step moves all pairs of the step with one kernel on the copy stream.) */
int vo_seq_push_pairs(vo_ctx *ctx, int n, const int32_t *seq_ids, const void *const *left, const void *const *right,
                      int stride, int kind);
/* enqueue one step over all sequences (asynchronous; the host runs at most 3 steps ahead of the device, 4 when the pairs
 * come from host memory -- a deeper queue measured 0-25 % slower, profiles/r06_experiments.md section 5) */
int vo_seq_step(vo_ctx *ctx);
int vo_seq_sync(vo_ctx *ctx);
/* state of one sequence after vo_seq_sync: currentVOFeatures (points [2 * n_pts], ages [n_ages], n_ages >= n_pts)
 * and frame_pose (4x4 row-major f64); any pointer may be NULL; caller arrays hold max_pts entries */
int vo_seq_get_state(vo_ctx *ctx, int seq, float *pts_xy, int *n_pts, int32_t *ages, int *n_ages, double *pose16);
/* rows [first, first + count) of the sequence's trajectory: one row per processed frame; rows27 [count][VO_SEQ_ROW],
 * info8 [count][VO_SEQ_INFO]; *n_rows = rows available.  VO_ERR_OVERFLOW if a returned frame overflowed. */
int vo_seq_get_trajectory(vo_ctx *ctx, int seq, int first, int count, double *rows27, int32_t *info8, int *n_rows);
/* HIP-event duration of the last `n` steps' stages is available through vo_batch_slot_times: step k records
 * ring slot k % VO_EVENT_SLOTS */

/* one pyramid level of one image back to the host (tests): out must hold w_l*h_l bytes (NULL: only the size is returned) */
int vo_batch_get_pyramid_level(vo_ctx *ctx, int image_idx, int level, uint8_t *out, int *w_l, int *h_l);

/* algorithmic HBM bytes of one frame at the current configuration:
 * bytes[0] = pyramid, [1] = LK, [2] = post (filter + triangulation + PnP), for n points */
int vo_model_bytes(const vo_ctx *ctx, int w, int h, int n_points, double *bytes3);

#ifdef __cplusplus
}
#endif
#endif
