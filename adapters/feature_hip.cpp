// feature_hip.cpp -- see feature_hip.h.  Plain C++11 (the reference's CMAKE_CXX_STANDARD) over the C ABI of libvo_hip.so.
#include "feature_hip.h"

#include <string.h>

#include <algorithm>
#include <iostream>
#include <stdexcept>

vo_ctx* vo_adapter_context_for(int w, int h, int n) // grows with the largest image / point count seen
{
    static vo_ctx* g_ctx = nullptr;
    static int cw = 0, ch = 0, cn = 0;
    if (!g_ctx || w > cw || h > ch || n > cn) {
        if (g_ctx)
            vo_destroy(g_ctx);
        cw = std::max(w, cw);
        ch = std::max(h, ch);
        cn = std::max(n, std::max(cn, 16384)); // >= the corners FAST returns on a KITTI frame
        g_ctx = vo_create(/*device*/ 0, cw, ch, cn, /*max_frames*/ 1);
        if (!g_ctx)
            throw std::runtime_error("vo_create failed (no HIP device?)"); // no CPU fallback
    }
    return g_ctx;
}
static vo_ctx* ctx_for(int w, int h, int n) { return vo_adapter_context_for(w, h, n); }
static void check(vo_ctx* c, int rc)
{
    if (rc < 0)
        throw std::runtime_error(vo_last_error(c));
}

// vo_adapter_keep_pair: what the previous circularMatching_hip call left on the device as its t1 pair
static bool g_keep_pair = false;
static long g_kept_calls = 0;
static struct {
    const unsigned char *l1, *r1;
    int cols, rows;
    size_t step;
    int64_t id;
} g_last = {nullptr, nullptr, 0, 0, 0, 0};
void vo_adapter_keep_pair(bool on)
{
    g_keep_pair = on;
    g_last.id = 0;
}
long vo_adapter_kept_calls() { return g_kept_calls; }

void circularMatching_hip(cv::Mat l0, cv::Mat r0, cv::Mat l1, cv::Mat r1, std::vector<cv::Point2f>& p_l0,
                          std::vector<cv::Point2f>& p_r0, std::vector<cv::Point2f>& p_l1, std::vector<cv::Point2f>& p_r1,
                          std::vector<cv::Point2f>& p_l0_ret, FeatureSet& feats)
{
    if (l0.type() != CV_8UC1 || l0.cols != r0.cols || l0.cols != l1.cols || l0.cols != r1.cols || l0.rows != r0.rows ||
        l0.rows != l1.rows || l0.rows != r1.rows || l0.step != r0.step || l0.step != l1.step || l0.step != r1.step)
        throw std::runtime_error("circularMatching_hip: four 8-bit gray images of one size and one row step");
    const int n = (int)p_l0.size();
    vo_ctx* c = ctx_for(l0.cols, l0.rows, n);
    std::vector<cv::Point2f> o_l0(n), o_r0(n), o_r1(n), o_l1(n), o_ret(n);
    std::vector<int32_t> keep(n > 0 ? n : 1);
    int m = 0;
    // the t0 pair is the pair the previous call sent as t1 (opt-in, see feature_hip.h): name the kept pair
    const bool kept = g_keep_pair && g_last.id != 0 && vo_kept_pair_id(c) == g_last.id && l0.data == g_last.l1 &&
                      r0.data == g_last.r1 && l0.cols == g_last.cols && l0.rows == g_last.rows && l0.step == g_last.step;
    g_last.id = 0;
    // cv::Point2f is two packed floats -> reinterpret as float*
    check(c, vo_circular_match(c, kept ? nullptr : l0.data, kept ? nullptr : r0.data, l1.data, r1.data, l0.cols, l0.rows,
                               (int)l0.step, (const float*)p_l0.data(), n, (float*)o_l0.data(), (float*)o_r0.data(),
                               (float*)o_r1.data(), (float*)o_l1.data(), (float*)o_ret.data(), /*status4*/ nullptr,
                               keep.data(), &m, /*apply_consistency*/ 0));
    g_kept_calls += kept ? 1 : 0;
    g_last.l1 = l1.data;
    g_last.r1 = r1.data;
    g_last.cols = l1.cols;
    g_last.rows = l1.rows;
    g_last.step = l1.step;
    g_last.id = vo_kept_pair_id(c);
    // deleteUnmatchFeaturesCircle's side effects on ages (feature.cpp:83-86,111)
    for (size_t i = 0; i < feats.ages.size(); i++)
        feats.ages[i] += 1;
    std::vector<int> ages(m);
    for (int i = 0; i < m; i++)
        ages[i] = feats.ages[keep[i]];
    // quirk B3: ages may be longer than points; the tail beyond points.size() survives the compaction
    ages.insert(ages.end(), feats.ages.begin() + std::min<size_t>(n, feats.ages.size()), feats.ages.end());
    feats.ages.swap(ages);
    o_l0.resize(m);
    o_r0.resize(m);
    o_r1.resize(m);
    o_l1.resize(m);
    o_ret.resize(m);
    p_l0.swap(o_l0);
    p_r0.swap(o_r0);
    p_r1.swap(o_r1);
    p_l1.swap(o_l1);
    p_l0_ret.swap(o_ret);
}

void triangulate_hip(cv::Mat& Pl, cv::Mat& Pr, std::vector<cv::Point2f>& pl, std::vector<cv::Point2f>& pr, cv::Mat& points3D_t0)
{
    if (Pl.type() != CV_32F || Pr.type() != CV_32F || Pl.rows != 3 || Pl.cols != 4 || Pr.rows != 3 || Pr.cols != 4)
        throw std::runtime_error("triangulate_hip: projection matrices are 3 x 4 CV_32F (main.cpp:73-74)");
    const int n = (int)pl.size();
    float P[2][12];
    for (int i = 0; i < 12; i++) {
        P[0][i] = Pl.at<float>(i / 4, i % 4);
        P[1][i] = Pr.at<float>(i / 4, i % 4);
    }
    vo_ctx* c = ctx_for(32, 32, n);
    points3D_t0.create(n, 1, CV_32FC3);
    std::vector<float> xyz((size_t)3 * (n > 0 ? n : 1));
    check(c, vo_triangulate(c, P[0], P[1], n ? &pl[0].x : nullptr, n ? &pr[0].x : nullptr, n, xyz.data()));
    for (int i = 0; i < n; i++)
        memcpy(points3D_t0.data + (size_t)i * points3D_t0.step, &xyz[(size_t)3 * i], 3 * sizeof(float));
}

void trackingFrame2Frame_hip(cv::Mat& Pl, cv::Mat& /*Pr*/, std::vector<cv::Point2f>& pointsLeft_t0,
                             std::vector<cv::Point2f>& pointsLeft_t1, cv::Mat& points3D_t0, cv::Mat& rotation,
                             cv::Mat& translation, bool mono_rotation)
{
    const int n = (int)pointsLeft_t1.size();
    vo_ctx* c = ctx_for(32, 32, n);
    double R_mono[9];
    if (mono_rotation) { // visualOdometry.cpp:146-157
        double E[9], t_mono[3];
        int good = 0;
        const int rc = vo_essential_pose(c, n ? &pointsLeft_t0[0].x : nullptr, n ? &pointsLeft_t1[0].x : nullptr, n,
                                         /*focal*/ Pl.at<float>(0, 0), /*pp*/ Pl.at<float>(0, 2), Pl.at<float>(1, 2), 0.999,
                                         1.0, E, R_mono, t_mono, /*mask*/ nullptr, &good);
        if (rc != VO_OK)
            throw std::runtime_error("recoverPose: E is not 3x3"); // OpenCV throws on the empty E
    }
    float K[9]; // projMatrl(0:3, 0:3), visualOdometry.cpp:163-165
    for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++)
            K[3 * r + k] = Pl.at<float>(r, k);
    double rvec[3] = {0, 0, 0}; // rvec zeroed each call, visualOdometry.cpp:162
    if (translation.type() != CV_64F || translation.total() != 3)
        throw std::runtime_error("trackingFrame2Frame_hip: translation is 3 x 1 CV_64F (main.cpp:82)");
    std::vector<float> xyz((size_t)3 * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) // N x 1 CV_32FC3, rows possibly padded
        memcpy(&xyz[(size_t)3 * i], points3D_t0.data + (size_t)i * points3D_t0.step, 3 * sizeof(float));
    double tv[3] = {translation.at<double>(0), translation.at<double>(1), translation.at<double>(2)}, R[9];
    std::vector<int32_t> inliers(n > 0 ? n : 1);
    int n_inl = 0;
    const int rc = vo_pnp_ransac(c, xyz.data(), n ? &pointsLeft_t1[0].x : nullptr, n, K, rvec, tv, R, inliers.data(), &n_inl);
    if (rc == VO_ERR_TOO_FEW)
        throw std::runtime_error("solvePnPRansac: npoints >= 4"); // OpenCV asserts here
    check(c, rc);                                                  // rc == VO_NO_MODEL: the reference ignores it too
    for (int k = 0; k < 3; k++)
        translation.at<double>(k) = tv[k];
    rotation.create(3, 3, CV_64F);
    for (int k = 0; k < 9; k++) // Rodrigues only `if (!mono_rotation)`, visualOdometry.cpp:186-189
        rotation.at<double>(k / 3, k % 3) = mono_rotation ? R_mono[k] : R[k];
    std::cout << "inliers size: " << n_inl << std::endl; // visualOdometry.cpp:191
}

void detectAndBucket_hip(cv::Mat& image, FeatureSet& feats)
{
    vo_ctx* c = ctx_for(image.cols, image.rows, 0);
    vo_detect_params dp;
    vo_default_detect_params(&dp); // FAST 20 / nonmax / re-detect below 2000 / bucket = rows / 10 / 1 per bucket
    const int cap = 1 << 16;
    std::vector<float> pts((size_t)2 * cap);
    std::vector<int32_t> ages(cap);
    int n_pts = (int)feats.points.size(), n_ages = (int)feats.ages.size();
    if (n_pts > cap || n_ages > cap)
        throw std::runtime_error("detectAndBucket_hip: feature set beyond the adapter's capacity");
    for (int i = 0; i < n_pts; i++) {
        pts[2 * i] = feats.points[i].x;
        pts[2 * i + 1] = feats.points[i].y;
    }
    for (int i = 0; i < n_ages; i++)
        ages[i] = feats.ages[i];
    check(c, vo_detect_bucket(c, image.data, image.cols, image.rows, (int)image.step, &dp, pts.data(), &n_pts, ages.data(),
                              &n_ages, cap));
    feats.points.resize(n_pts);
    for (int i = 0; i < n_pts; i++)
        feats.points[i] = cv::Point2f(pts[2 * i], pts[2 * i + 1]);
    feats.ages.assign(ages.begin(), ages.begin() + n_ages);
}
