/*
 * dropin_glue.cpp -- TEST ONLY: the drop-in, demonstrated with the reference's own code.
 *
 * Builds (in the authoring container, where /root/reference exists) a library in which the reference's unmodified
 * sources -- src/bucket.cpp, feature.cpp, visualOdometry.cpp, utils.cpp, compiled where they lie against the
 * OpenCV stand-in oracle/ref_shim/vo_cv_shim.h -- run on top of libvo_hip.so instead of OpenCV:
 *   * visualOdometry.cpp is compiled with -DcircularMatching=circularMatching_hip, i.e. the one call site
 *     (visualOdometry.cpp:112-118) reaches the SHIPPED adapter adapters/feature_hip.cpp (same signature as
 *     feature.h:61-65) exactly as a USE_HIP switch next to the existing USE_CUDA one would;
 *   * ref_frame_step_adapter additionally replaces main.cpp:169-171,181 by the adapter's triangulate_hip /
 *     trackingFrame2Frame_hip, i.e. the frame loop a maintainer gets after the edits INTEGRATION.md lists;
 *   * the OpenCV entry points the other reference functions call are defined here over the C ABI:
 *     cv::FAST -> vo_fast_detect, cv::triangulatePoints (+ convertPointsFromHomogeneous) -> vo_triangulate,
 *     cv::solvePnPRansac + cv::Rodrigues -> vo_pnp_ransac, cv::findEssentialMat + cv::recoverPose -> vo_essential_pose.
 * tests/test_gpu_parity.py runs the reference's matchingFeatures() / trackingFrame2Frame() / integrateOdometryStereo()
 * through this library on the MI355X and compares, frame after frame, with the same reference code running over the
 * CPU oracle (oracle/_ref).  No reference source is copied or modified.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <iostream>
#include <sstream>
#include <stdexcept>

#include "feature.h"
#include "utils.h"
#include "visualOdometry.h"
#include "vo_hip.h"

/* ---- the adapter: adapters/feature_hip.cpp, compiled by the Makefile next to this file (ONE source of truth; round 5 kept a
 * copy here).  The OpenCV stand-ins below share its context. ------------------------------------------------------------- */
#include "feature_hip.h"

static vo_ctx *ctx_for(int w, int h, int n) { return vo_adapter_context_for(w, h, n); }
static void check(vo_ctx *c, int rc)
{
    if (rc < 0)
        throw std::runtime_error(vo_last_error(c));
}

/* ---- OpenCV entry points of the remaining reference code, over the C ABI ---------------------------------------- */
namespace cv {

void KeyPoint::convert(const std::vector<KeyPoint> &keypoints, std::vector<Point2f> &points2f, const std::vector<int> &)
{
    points2f.resize(keypoints.size());
    for (size_t i = 0; i < keypoints.size(); i++)
        points2f[i] = keypoints[i].pt;
}

void FAST(Mat image, std::vector<KeyPoint> &keypoints, int threshold, bool nonmaxSuppression)
{
    vo_ctx *c = ctx_for(image.cols, image.rows, 0);
    int cap = 1 << 16, n = 0;
    std::vector<float> pts;
    for (;;) {
        pts.resize((size_t)2 * cap);
        check(c, vo_fast_detect(c, image.data, image.cols, image.rows, (int)image.step, threshold, nonmaxSuppression ? 1 : 0,
                                pts.data(), cap, &n));
        if (n <= cap)
            break;
        cap = n;
    }
    keypoints.resize((size_t)n);
    for (int i = 0; i < n; i++)
        keypoints[i].pt = Point2f(pts[2 * i], pts[2 * i + 1]);
}

void goodFeaturesToTrack(Mat, std::vector<Point2f> &, int, double, double, Mat, int, bool, double) { abort(); }

void calcOpticalFlowPyrLK(Mat, Mat, std::vector<Point2f> &, std::vector<Point2f> &, std::vector<uchar> &,
                          std::vector<float> &, Size, int, TermCriteria, int, double)
{
    fprintf(stderr, "dropin_glue: calcOpticalFlowPyrLK reached -- circularMatching() was not replaced\n");
    abort();
}

/* vo_triangulate is triangulatePoints + convertPointsFromHomogeneous in one call; the homogeneous array handed
 * back carries w = 1 so that the reference's own convertPointsFromHomogeneous call returns those points unchanged */
void triangulatePoints(const Mat &P1, const Mat &P2, const std::vector<Point2f> &p1, const std::vector<Point2f> &p2,
                       Mat &points4D)
{
    const int n = (int)p1.size();
    float Pl[12], Pr[12];
    for (int i = 0; i < 12; i++) {
        Pl[i] = P1.at<float>(i / 4, i % 4);
        Pr[i] = P2.at<float>(i / 4, i % 4);
    }
    vo_ctx *c = ctx_for(32, 32, n);
    std::vector<float> xyz((size_t)3 * (n > 0 ? n : 1));
    check(c, vo_triangulate(c, Pl, Pr, n ? &p1[0].x : nullptr, n ? &p2[0].x : nullptr, n, xyz.data()));
    points4D = Mat(4, n, CV_32FC1);
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++)
            points4D.at<float>(k, i) = xyz[(size_t)3 * i + k];
        points4D.at<float>(3, i) = 1.f;
    }
}

void convertPointsFromHomogeneous(const Mat &src, Mat &dst)
{
    const int n = src.rows;
    dst = Mat(n, 1, CV_32FC3);
    for (int i = 0; i < n; i++) {
        const float w = src.at<float>(i, 3), scale = w != 0.f ? 1.f / w : 1.f;
        float *o = (float *)(dst.data + (size_t)i * dst.step);
        for (int k = 0; k < 3; k++)
            o[k] = src.at<float>(i, k) * scale;
    }
}

static double g_last_R[9]; /* vo_pnp_ransac returns Rodrigues(rvec) with the pose; cv::Rodrigues hands it out */

bool solvePnPRansac(const Mat &objectPoints, const std::vector<Point2f> &imagePoints, const Mat &cameraMatrix,
                    const Mat &, Mat &rvec, Mat &tvec, bool, int, float, double, Mat &inliers, int)
{
    const int n = (int)imagePoints.size();
    vo_ctx *c = ctx_for(32, 32, n);
    std::vector<float> xyz((size_t)3 * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++)
        memcpy(&xyz[(size_t)3 * i], objectPoints.data + (size_t)i * objectPoints.step, 3 * sizeof(float));
    float K[9];
    for (int i = 0; i < 9; i++)
        K[i] = cameraMatrix.at<float>(i / 3, i % 3);
    std::vector<int32_t> inl((size_t)(n > 0 ? n : 1));
    int n_inl = 0;
    const int rc = vo_pnp_ransac(c, xyz.data(), n ? &imagePoints[0].x : nullptr, n, K, rvec.ptr<double>(), tvec.ptr<double>(),
                                 g_last_R, inl.data(), &n_inl);
    if (rc == VO_ERR_TOO_FEW)
        throw std::runtime_error("solvePnPRansac: npoints >= 4 (OpenCV asserts here)");
    check(c, rc);
    inliers = Mat(n_inl, n_inl > 0 ? 1 : 0, CV_32FC1);
    return rc == VO_OK;
}

void Rodrigues(const Mat &, Mat &dst)
{
    dst = Mat(3, 3, CV_64F);
    for (int i = 0; i < 9; i++)
        dst.at<double>(i / 3, i % 3) = g_last_R[i];
}

static double g_em_R[9], g_em_t[3];
static std::vector<uint8_t> g_em_mask;
static int g_em_good;

Mat findEssentialMat(const std::vector<Point2f> &points1, const std::vector<Point2f> &points2, double focal, Point2d pp,
                     int, double prob, double threshold, Mat &mask)
{
    const int n = (int)points1.size();
    vo_ctx *c = ctx_for(32, 32, n);
    double E[9];
    g_em_mask.assign((size_t)(n > 0 ? n : 1), 0);
    const int rc = vo_essential_pose(c, n ? &points1[0].x : nullptr, n ? &points2[0].x : nullptr, n, focal, pp.x, pp.y,
                                     prob, threshold, E, g_em_R, g_em_t, g_em_mask.data(), &g_em_good);
    if (rc != VO_OK)
        return Mat(); /* OpenCV returns an empty E; recoverPose then throws */
    mask = Mat(n, 1, CV_8UC1);
    Mat Em(3, 3, CV_64F);
    for (int i = 0; i < 9; i++)
        Em.at<double>(i / 3, i % 3) = E[i];
    return Em;
}

int recoverPose(const Mat &E, const std::vector<Point2f> &points1, const std::vector<Point2f> &, Mat &R, Mat &t, double,
                Point2d, Mat &mask)
{
    if (E.rows != 3)
        throw std::runtime_error("recoverPose: E is not 3x3");
    R = Mat(3, 3, CV_64F);
    t = Mat(3, 1, CV_64F);
    for (int i = 0; i < 9; i++)
        R.at<double>(i / 3, i % 3) = g_em_R[i];
    for (int k = 0; k < 3; k++)
        t.at<double>(k) = g_em_t[k];
    for (size_t i = 0; i < points1.size(); i++)
        mask.at<uchar>((int)i, 0) = g_em_mask[i];
    return g_em_good;
}

} // namespace cv

struct Quiet {
    std::streambuf *o, *e;
    std::ostringstream sink;
    Quiet() : o(std::cout.rdbuf(sink.rdbuf())), e(std::cerr.rdbuf(sink.rdbuf())) {}
    ~Quiet()
    {
        std::cout.rdbuf(o);
        std::cerr.rdbuf(e);
    }
};

static std::vector<cv::Point2f> to_points(const float *p, int n)
{
    std::vector<cv::Point2f> v((size_t)n);
    for (int i = 0; i < n; i++)
        v[i] = cv::Point2f(p[2 * i], p[2 * i + 1]);
    return v;
}
static int from_points(const std::vector<cv::Point2f> &v, float *p, int cap)
{
    const int n = (int)v.size() < cap ? (int)v.size() : cap;
    for (int i = 0; i < n; i++) {
        p[2 * i] = v[i].x;
        p[2 * i + 1] = v[i].y;
    }
    return (int)v.size();
}

/* same interface and same body as oracle/ref_shim/ref_glue.cpp::ref_frame_step (main.cpp:144-208); adapter_tail: the
 * triangulation and trackingFrame2Frame call sites edited as INTEGRATION.md says (triangulate_hip, trackingFrame2Frame_hip) */
static int frame_step(const uint8_t *l0, const uint8_t *r0, const uint8_t *l1, const uint8_t *r1, int w, int h,
                              float fx, float cx, float cy, float bf, float *feat_pts, int *feat_ages, int *n_pts,
                              int *n_ages, int cap, double *translation, double *rotation, double *frame_pose,
                              int mono_rotation, float *out_l0, float *out_r0, float *out_l1, float *out_r1, int *n_out,
                              int *integrated, bool adapter_tail)
{
    Quiet q;
    try {
        cv::Mat L0(h, w, CV_8UC1, (void *)l0, (size_t)w), R0(h, w, CV_8UC1, (void *)r0, (size_t)w),
            L1(h, w, CV_8UC1, (void *)l1, (size_t)w), R1(h, w, CV_8UC1, (void *)r1, (size_t)w);
        cv::Mat projMatrl = (cv::Mat_<float>(3, 4) << fx, 0., cx, 0., 0., fx, cy, 0., 0, 0., 1., 0.);
        cv::Mat projMatrr = (cv::Mat_<float>(3, 4) << fx, 0., cx, bf, 0., fx, cy, 0., 0, 0., 1., 0.);
        FeatureSet fs;
        fs.points = to_points(feat_pts, *n_pts);
        fs.ages.assign(feat_ages, feat_ages + *n_ages);
        cv::Mat rot(3, 3, CV_64F), trans(3, 1, CV_64F), pose(4, 4, CV_64F);
        for (int i = 0; i < 9; i++)
            rot.at<double>(i / 3, i % 3) = rotation[i];
        for (int i = 0; i < 3; i++)
            trans.at<double>(i) = translation[i];
        for (int i = 0; i < 16; i++)
            pose.at<double>(i / 4, i % 4) = frame_pose[i];

        std::vector<cv::Point2f> pl0, pr0, pl1, pr1;
        matchingFeatures(L0, R0, L1, R1, fs, pl0, pr0, pl1, pr1);
        cv::Mat points3D_t0, points4D_t0;
        if (adapter_tail) {
            triangulate_hip(projMatrl, projMatrr, pl0, pr0, points3D_t0);
            trackingFrame2Frame_hip(projMatrl, projMatrr, pl0, pl1, points3D_t0, rot, trans, mono_rotation != 0);
        } else {
            cv::triangulatePoints(projMatrl, projMatrr, pl0, pr0, points4D_t0);
            cv::convertPointsFromHomogeneous(points4D_t0.t(), points3D_t0);
            trackingFrame2Frame(projMatrl, projMatrr, pl0, pl1, points3D_t0, rot, trans, mono_rotation != 0);
        }
        cv::Vec3f e = rotationMatrixToEulerAngles(rot);
        cv::Mat rigid_body_transformation;
        *integrated = 0;
        if (abs(e[1]) < 0.1 && abs(e[0]) < 0.1 && abs(e[2]) < 0.1) {
            cv::Mat before = pose.clone();
            integrateOdometryStereo(0, rigid_body_transformation, pose, rot, trans);
            *integrated = cv::norm(before, pose) != 0.0;
        }
        if ((int)fs.points.size() > cap || (int)fs.ages.size() > cap || (int)pl0.size() > cap)
            return -1;
        *n_pts = from_points(fs.points, feat_pts, cap);
        for (size_t i = 0; i < fs.ages.size(); i++)
            feat_ages[i] = fs.ages[i];
        *n_ages = (int)fs.ages.size();
        *n_out = from_points(pl0, out_l0, cap);
        from_points(pr0, out_r0, cap);
        from_points(pl1, out_l1, cap);
        from_points(pr1, out_r1, cap);
        for (int i = 0; i < 9; i++)
            rotation[i] = rot.at<double>(i / 3, i % 3);
        for (int i = 0; i < 3; i++)
            translation[i] = trans.at<double>(i);
        for (int i = 0; i < 16; i++)
            frame_pose[i] = pose.at<double>(i / 4, i % 4);
        return 0;
    } catch (const std::exception &ex) {
        fprintf(stderr, "dropin_glue: %s\n", ex.what());
        return -2;
    }
}

/* the adapter's detectAndBucket_hip (head of matchingFeatures, visualOdometry.cpp:95-108) on a caller's feature set */
extern "C" int adapter_detect_bucket(const uint8_t *img, int w, int h, float *pts, int *ages, int *n_pts, int *n_ages, int cap)
{
    try {
        cv::Mat I(h, w, CV_8UC1, (void *)img, (size_t)w);
        FeatureSet fs;
        fs.points = to_points(pts, *n_pts);
        fs.ages.assign(ages, ages + *n_ages);
        detectAndBucket_hip(I, fs);
        if ((int)fs.points.size() > cap || (int)fs.ages.size() > cap)
            return -1;
        *n_pts = from_points(fs.points, pts, cap);
        for (size_t i = 0; i < fs.ages.size(); i++)
            ages[i] = fs.ages[i];
        *n_ages = (int)fs.ages.size();
        return 0;
    } catch (const std::exception &ex) {
        fprintf(stderr, "dropin_glue: %s\n", ex.what());
        return -2;
    }
}

#define VO_STEP_ARGS                                                                                                        \
    const uint8_t *l0, const uint8_t *r0, const uint8_t *l1, const uint8_t *r1, int w, int h, float fx, float cx, float cy,  \
        float bf, float *feat_pts, int *feat_ages, int *n_pts, int *n_ages, int cap, double *translation, double *rotation,  \
        double *frame_pose, int mono_rotation, float *out_l0, float *out_r0, float *out_l1, float *out_r1, int *n_out,        \
        int *integrated
#define VO_STEP_PASS                                                                                                        \
    l0, r0, l1, r1, w, h, fx, cx, cy, bf, feat_pts, feat_ages, n_pts, n_ages, cap, translation, rotation, frame_pose,        \
        mono_rotation, out_l0, out_r0, out_l1, out_r1, n_out, integrated
/* the adapter's opt-in "t0 pair = the previous call's t1 pair" (adapters/feature_hip.h) and its count of such calls */
extern "C" void adapter_keep_pair(int on) { vo_adapter_keep_pair(on != 0); }
extern "C" long adapter_kept_calls(void) { return vo_adapter_kept_calls(); }
extern "C" int ref_frame_step(VO_STEP_ARGS) { return frame_step(VO_STEP_PASS, false); }
extern "C" int ref_frame_step_adapter(VO_STEP_ARGS) { return frame_step(VO_STEP_PASS, true); }
