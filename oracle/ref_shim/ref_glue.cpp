/*
 * ref_glue.cpp -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Links the reference's own feature.cpp, bucket.cpp, visualOdometry.cpp and utils.cpp (compiled from
 * /root/reference/src against vo_cv_shim.h) with the oracle's restatement of the OpenCV algorithms they call, and
 * exports their entry points with a C interface so that tests/test_reference_glue.py can compare the oracle's
 * restated glue (orc_glue.c) -- and the product -- with the real thing on identical inputs.  Built by `make -C oracle ref` into oracle/_ref/ (git-ignored, travels to the
 * GPU box with the snapshot); only possible where /root/reference exists.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <iostream>
#include <sstream>

#include "feature.h" /* the reference's headers, found through -I/root/reference/src */
#include "utils.h"
#include "visualOdometry.h"
#include "evaluate_odometry.h" /* src/evaluate: calcSequenceErrors and friends (no OpenCV in there) */
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

void triangulatePoints(const Mat &P1, const Mat &P2, const std::vector<Point2f> &p1, const std::vector<Point2f> &p2,
                       Mat &points4D)
{
    const int n = (int)p1.size();
    float Pl[12], Pr[12];
    for (int i = 0; i < 12; i++) {
        Pl[i] = (float)P1.get(i / 4, i % 4);
        Pr[i] = (float)P2.get(i / 4, i % 4);
    }
    points4D = Mat(4, n, CV_32FC1); /* output depth follows the Point2f inputs */
    std::vector<float> out((size_t)4 * (n > 0 ? n : 1));
    if (n > 0)
        orc_triangulate_points(Pl, Pr, &p1[0].x, &p2[0].x, n, out.data());
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < n; i++)
            points4D.at<float>(k, i) = out[(size_t)k * n + i];
}

void convertPointsFromHomogeneous(const Mat &src, Mat &dst)
{
    /* src: N x 4 CV_32F (the reference passes points4D.t()); dst: N x 1 CV_32FC3 */
    const int n = src.rows;
    std::vector<float> in((size_t)4 * (n > 0 ? n : 1)), out((size_t)3 * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 4; k++)
            in[(size_t)4 * i + k] = src.at<float>(i, k);
    orc_convert_points_from_homogeneous(in.data(), n, out.data());
    dst = Mat(n, 1, CV_32FC3);
    for (int i = 0; i < n; i++)
        memcpy(dst.data + (size_t)i * dst.step, &out[(size_t)3 * i], 3 * sizeof(float));
}

bool solvePnPRansac(const Mat &objectPoints, const std::vector<Point2f> &imagePoints, const Mat &cameraMatrix,
                    const Mat &distCoeffs, Mat &rvec, Mat &tvec, bool useExtrinsicGuess, int iterationsCount,
                    float reprojectionError, double confidence, Mat &inliers, int flags)
{
    if (!useExtrinsicGuess || flags != SOLVEPNP_ITERATIVE || objectPoints.type() != CV_32FC3) {
        fprintf(stderr, "vo_cv_shim: solvePnPRansac called in a way the reference never does\n");
        abort();
    }
    for (int i = 0; i < distCoeffs.rows; i++)
        if (distCoeffs.get(i, 0) != 0.0)
            abort(); /* the reference passes zeros */
    const int n = (int)imagePoints.size();
    std::vector<float> xyz((size_t)3 * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++)
        memcpy(&xyz[(size_t)3 * i], objectPoints.data + (size_t)i * objectPoints.step, 3 * sizeof(float));
    float K[9];
    for (int i = 0; i < 9; i++)
        K[i] = (float)cameraMatrix.get(i / 3, i % 3);
    double rv[3], tv[3];
    for (int k = 0; k < 3; k++) {
        rv[k] = rvec.at<double>(k);
        tv[k] = tvec.at<double>(k);
    }
    std::vector<int32_t> inl((size_t)(n > 0 ? n : 1));
    int n_inl = 0;
    const int rc = orc_solve_pnp_ransac(xyz.data(), n ? &imagePoints[0].x : 0, n, K, rv, tv, iterationsCount,
                                        reprojectionError, confidence, inl.data(), &n_inl, 0);
    if (rc < 0) {
        fprintf(stderr, "vo_cv_shim: solvePnPRansac needs >= 5 points (OpenCV asserts)\n");
        abort();
    }
    for (int k = 0; k < 3; k++) {
        rvec.at<double>(k) = rv[k];
        tvec.at<double>(k) = tv[k];
    }
    inliers = Mat(n_inl, n_inl > 0 ? 1 : 0, CV_32FC1); /* only its size is looked at */
    return rc == 1;
}

void Rodrigues(const Mat &src, Mat &dst)
{
    double r[3] = {src.at<double>(0), src.at<double>(1), src.at<double>(2)}, R[9];
    orc_rodrigues_vec2mat(r, R, 0);
    dst = Mat(3, 3, CV_64F);
    for (int i = 0; i < 9; i++)
        dst.at<double>(i / 3, i % 3) = R[i];
}

Mat findEssentialMat(const std::vector<Point2f> &points1, const std::vector<Point2f> &points2, double focal, Point2d pp,
                     int method, double prob, double threshold, Mat &mask)
{
    if (method != RANSAC)
        abort();
    const int n = (int)points1.size();
    double E[9];
    std::vector<uint8_t> m((size_t)(n > 0 ? n : 1));
    const int ok = orc_find_essential_mat(n ? &points1[0].x : 0, n ? &points2[0].x : 0, n, focal, pp.x, pp.y, prob,
                                          threshold, E, m.data(), 0);
    if (!ok)
        return Mat();
    mask = Mat(n, 1, CV_8UC1);
    for (int i = 0; i < n; i++)
        mask.at<uchar>(i, 0) = m[(size_t)i];
    Mat Em(3, 3, CV_64F);
    for (int i = 0; i < 9; i++)
        Em.at<double>(i / 3, i % 3) = E[i];
    return Em;
}

int recoverPose(const Mat &E, const std::vector<Point2f> &points1, const std::vector<Point2f> &points2, Mat &R, Mat &t,
                double focal, Point2d pp, Mat &mask)
{
    if (E.rows != 3 || E.cols != 3) {
        fprintf(stderr, "vo_cv_shim: recoverPose on an empty E (OpenCV throws)\n");
        abort();
    }
    const int n = (int)points1.size();
    double Ed[9], Rd[9], td[3];
    for (int i = 0; i < 9; i++)
        Ed[i] = E.at<double>(i / 3, i % 3);
    std::vector<uint8_t> m((size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++)
        m[(size_t)i] = mask.at<uchar>(i, 0);
    const int good = orc_recover_pose(Ed, n ? &points1[0].x : 0, n ? &points2[0].x : 0, n, focal, pp.x, pp.y, Rd, td,
                                      m.data());
    R = Mat(3, 3, CV_64F);
    t = Mat(3, 1, CV_64F);
    for (int i = 0; i < 9; i++)
        R.at<double>(i / 3, i % 3) = Rd[i];
    for (int k = 0; k < 3; k++)
        t.at<double>(k) = td[k];
    for (int i = 0; i < n; i++)
        mask.at<uchar>(i, 0) = m[(size_t)i];
    return good;
}

} // namespace cv

/* the reference prints progress lines to cout / cerr: keep the test output clean */
struct Quiet {
    std::ostringstream sink; // (first: members are constructed in declaration order, and o / e use it -- UBSan, round 5)
    std::streambuf *o, *e;
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
    cv::Mat L0(h, w, CV_8UC1, (void *)l0, (size_t)w), R0(h, w, CV_8UC1, (void *)r0, (size_t)w),
        L1(h, w, CV_8UC1, (void *)l1, (size_t)w), R1(h, w, CV_8UC1, (void *)r1, (size_t)w);
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
    static uchar dummy;
    cv::Mat image(rows, cols, CV_8UC1, &dummy, (size_t)cols); /* only rows / cols are read */
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
    cv::Mat image(h, w, CV_8UC1, (void *)img, (size_t)w);
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

/* One pass of the body of main()'s frame loop (main.cpp:144-208) through the reference's own functions:
 * matchingFeatures() -> cv::triangulatePoints + convertPointsFromHomogeneous -> trackingFrame2Frame(mono_rotation)
 * -> rotationMatrixToEulerAngles -> the three 0.1 rad gates -> integrateOdometryStereo().
 * State in / out: features (points / ages, capacity cap), translation[3], rotation[9], frame_pose[16].
 * Outputs: the four filtered point sets (capacity cap each, *n_out entries), n_bucketed, integrated flag.
 * Returns 0, or -1 when a capacity is too small. */
int ref_frame_step(const uint8_t *l0, const uint8_t *r0, const uint8_t *l1, const uint8_t *r1, int w, int h, float fx,
                   float cx, float cy, float bf, float *feat_pts, int *feat_ages, int *n_pts, int *n_ages, int cap,
                   double *translation, double *rotation, double *frame_pose, int mono_rotation, float *out_l0,
                   float *out_r0, float *out_l1, float *out_r1, int *n_out, int *integrated)
{
    Quiet q;
    cv::Mat L0(h, w, CV_8UC1, (void *)l0, (size_t)w), R0(h, w, CV_8UC1, (void *)r0, (size_t)w),
        L1(h, w, CV_8UC1, (void *)l1, (size_t)w), R1(h, w, CV_8UC1, (void *)r1, (size_t)w);
    /* main.cpp:73-74 */
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
    cv::triangulatePoints(projMatrl, projMatrr, pl0, pr0, points4D_t0);
    cv::convertPointsFromHomogeneous(points4D_t0.t(), points3D_t0);
    trackingFrame2Frame(projMatrl, projMatrr, pl0, pl1, points3D_t0, rot, trans, mono_rotation != 0);
    cv::Vec3f e = rotationMatrixToEulerAngles(rot);
    cv::Mat rigid_body_transformation;
    *integrated = 0;
    if (abs(e[1]) < 0.1 && abs(e[0]) < 0.1 && abs(e[2]) < 0.1) {
        /* integrateOdometryStereo applies (or skips) the pose itself by its scale test: report what happened */
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
}

/* utils.cpp:107-131 and :57-91 on their own */
void ref_rotation_matrix_to_euler(const double *R, float *euler3)
{
    cv::Mat r(3, 3, CV_64F);
    for (int i = 0; i < 9; i++)
        r.at<double>(i / 3, i % 3) = R[i];
    cv::Vec3f e = rotationMatrixToEulerAngles(r);
    for (int k = 0; k < 3; k++)
        euler3[k] = e[k];
}

void ref_integrate_odometry_stereo(double *frame_pose, const double *R, const double *t)
{
    Quiet q;
    cv::Mat rot(3, 3, CV_64F), trans(3, 1, CV_64F), pose(4, 4, CV_64F), rbt;
    for (int i = 0; i < 9; i++)
        rot.at<double>(i / 3, i % 3) = R[i];
    for (int i = 0; i < 3; i++)
        trans.at<double>(i) = t[i];
    for (int i = 0; i < 16; i++)
        pose.at<double>(i / 4, i % 4) = frame_pose[i];
    integrateOdometryStereo(0, rbt, pose, rot, trans);
    for (int i = 0; i < 16; i++)
        frame_pose[i] = pose.at<double>(i / 4, i % 4);
}


/* calcSequenceErrors(poses_gt, poses_result) (evaluate/evaluate_odometry.cpp:71-116, compiled where it lies): poses are
 * n rows of 12 doubles (3x4 row-major, the KITTI pose-file layout loadPoses reads).  out5 [cap][5] =
 * (first_frame, r_err, t_err, len, speed) per segment; returns the number of segments. */
int ref_calc_sequence_errors(const double *gt12, const double *res12, int n, float *out5, int cap)
{
    std::vector<Matrix> G, Rr;
    for (int i = 0; i < n; i++) {
        Matrix a = Matrix::eye(4), b = Matrix::eye(4);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) {
                a.val[r][c] = gt12[12 * i + 4 * r + c];
                b.val[r][c] = res12[12 * i + 4 * r + c];
            }
        G.push_back(a);
        Rr.push_back(b);
    }
    std::vector<errors> e = calcSequenceErrors(G, Rr);
    int k = 0;
    for (const errors &x : e) {
        if (k >= cap)
            break;
        out5[5 * k + 0] = (float)x.first_frame;
        out5[5 * k + 1] = x.r_err;
        out5[5 * k + 2] = x.t_err;
        out5[5 * k + 3] = x.len;
        out5[5 * k + 4] = x.speed;
        k++;
    }
    return (int)e.size();
}

} /* extern "C" */
