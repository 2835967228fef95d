/*
 * ref_glue.cpp -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Links the reference's own feature.cpp + bucket.cpp (compiled from /root/reference/src against vo_cv_shim.h)
 * with the oracle's restatement of the two OpenCV algorithms they call, and exports their entry points with a
 * C interface so that tests/test_reference_glue.py can compare the oracle's restated glue (orc_glue.c) with the
 * real thing on identical inputs.  Built by `make -C oracle ref` into oracle/_ref/ (git-ignored, travels to the
 * GPU box with the snapshot); only possible where /root/reference exists.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <iostream>
#include <sstream>

#include "feature.h" /* the reference's header, found through -I/root/reference/src */
#include "vo_oracle.h"

namespace cv {

void KeyPoint::convert(const std::vector<KeyPoint> &keypoints, std::vector<Point2f> &points2f, const std::vector<int> &)
{
    points2f.resize(keypoints.size());
    for (size_t i = 0; i < keypoints.size(); i++)
        points2f[i] = keypoints[i].pt;
}

static std::vector<uchar> continuous(const Mat &m)
{
    std::vector<uchar> buf((size_t)m.rows * m.cols);
    for (int y = 0; y < m.rows; y++)
        memcpy(buf.data() + (size_t)y * m.cols, m.data + (size_t)y * m.step, (size_t)m.cols);
    return buf;
}

void FAST(Mat image, std::vector<KeyPoint> &keypoints, int threshold, bool nonmaxSuppression)
{
    const std::vector<uchar> img = continuous(image);
    int cap = 1 << 16;
    std::vector<float> pts;
    int n;
    for (;;) {
        pts.resize((size_t)2 * cap);
        n = orc_fast_detect(img.data(), image.cols, image.rows, threshold, nonmaxSuppression ? 1 : 0, pts.data(), cap);
        if (n <= cap)
            break;
        cap = n;
    }
    keypoints.resize((size_t)n);
    for (int i = 0; i < n; i++)
        keypoints[i].pt = Point2f(pts[2 * i], pts[2 * i + 1]);
}

void goodFeaturesToTrack(Mat, std::vector<Point2f> &, int, double, double, Mat, int, bool, double)
{
    fprintf(stderr, "vo_cv_shim: goodFeaturesToTrack is not on the path and not provided\n");
    abort();
}

void calcOpticalFlowPyrLK(Mat prevImg, Mat nextImg, std::vector<Point2f> &prevPts, std::vector<Point2f> &nextPts,
                          std::vector<uchar> &status, std::vector<float> &err, Size winSize, int maxLevel,
                          TermCriteria criteria, int flags, double minEigThreshold)
{
    if (flags != 0 || winSize.width != winSize.height || prevImg.rows != nextImg.rows || prevImg.cols != nextImg.cols) {
        fprintf(stderr, "vo_cv_shim: calcOpticalFlowPyrLK called in a way the reference never does\n");
        abort();
    }
    const std::vector<uchar> a = continuous(prevImg), b = continuous(nextImg);
    const int n = (int)prevPts.size();
    nextPts.resize((size_t)n);
    status.resize((size_t)n);
    err.resize((size_t)n);
    const int max_count = (criteria.type & TermCriteria::COUNT) ? criteria.maxCount : 30;
    const double eps = (criteria.type & TermCriteria::EPS) ? criteria.epsilon : 0.01;
    orc_calc_optical_flow_pyr_lk(a.data(), b.data(), prevImg.cols, prevImg.rows, n ? &prevPts[0].x : 0, n,
                                 n ? &nextPts[0].x : 0, status.data(), err.data(), winSize.width, maxLevel, max_count,
                                 eps, minEigThreshold, 0, 1);
}

} // namespace cv

/* the reference prints progress lines to cout / cerr: keep the test output clean */
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

extern "C" {

/* circularMatching(img_l_0, img_r_0, img_l_1, img_r_1, points_l_0, ..., current_features) (feature.cpp:118-148).
 * pts_l0 [n*2] in/out (compacted), outputs sized n*2; ages [*n_ages] in/out.  Returns the survivor count. */
int ref_circular_matching(const uint8_t *l0, const uint8_t *r0, const uint8_t *l1, const uint8_t *r1, int w, int h,
                          float *pts_l0, int n, float *pts_r0, float *pts_r1, float *pts_l1, float *pts_l0_ret,
                          int *ages, int *n_ages)
{
    Quiet q;
    cv::Mat L0(h, w, (uchar *)l0, (size_t)w), R0(h, w, (uchar *)r0, (size_t)w), L1(h, w, (uchar *)l1, (size_t)w),
        R1(h, w, (uchar *)r1, (size_t)w);
    std::vector<cv::Point2f> p_l0 = to_points(pts_l0, n), p_r0, p_l1, p_r1, p_ret;
    FeatureSet fs;
    fs.ages.assign(ages, ages + *n_ages);
    circularMatching(L0, R0, L1, R1, p_l0, p_r0, p_l1, p_r1, p_ret, fs);
    const int m = (int)p_l0.size();
    from_points(p_l0, pts_l0, n);
    from_points(p_r0, pts_r0, n);
    from_points(p_r1, pts_r1, n);
    from_points(p_l1, pts_l1, n);
    from_points(p_ret, pts_l0_ret, n);
    for (size_t i = 0; i < fs.ages.size(); i++)
        ages[i] = fs.ages[i];
    *n_ages = (int)fs.ages.size();
    return m;
}

/* bucketingFeatures(image, current_features, bucket_size, features_per_bucket) (feature.cpp:206-253, bucket.cpp).
 * points / ages hold *n_points / *n_ages entries on entry, capacity cap each; returns 0, or -1 when the result does
 * not fit. */
int ref_bucketing_features(int rows, int cols, float *points, int *ages, int *n_points, int *n_ages, int cap,
                           int bucket_size, int features_per_bucket)
{
    Quiet q;
    cv::Mat image(rows, cols, 0, (size_t)cols);
    FeatureSet fs;
    fs.points = to_points(points, *n_points);
    fs.ages.assign(ages, ages + *n_ages);
    bucketingFeatures(image, fs, bucket_size, features_per_bucket);
    if ((int)fs.points.size() > cap || (int)fs.ages.size() > cap)
        return -1;
    from_points(fs.points, points, cap);
    for (size_t i = 0; i < fs.ages.size(); i++)
        ages[i] = fs.ages[i];
    *n_points = (int)fs.points.size();
    *n_ages = (int)fs.ages.size();
    return 0;
}

/* appendNewFeatures(image, current_features) (feature.cpp:255-262): FAST corners appended with age 0 */
int ref_append_new_features(const uint8_t *img, int w, int h, float *points, int *ages, int *n_points, int *n_ages,
                            int cap)
{
    Quiet q;
    cv::Mat image(h, w, (uchar *)img, (size_t)w);
    FeatureSet fs;
    fs.points = to_points(points, *n_points);
    fs.ages.assign(ages, ages + *n_ages);
    appendNewFeatures(image, fs);
    if ((int)fs.points.size() > cap || (int)fs.ages.size() > cap)
        return -1;
    from_points(fs.points, points, cap);
    for (size_t i = 0; i < fs.ages.size(); i++)
        ages[i] = fs.ages[i];
    *n_points = (int)fs.points.size();
    *n_ages = (int)fs.ages.size();
    return 0;
}

} /* extern "C" */
