/*
 * vo_cv_shim.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A type-only stand-in for the handful of OpenCV declarations that the reference's own glue sources
 * (/root/reference/src/feature.cpp, bucket.cpp) need in order to compile WHERE THEY LIE, without OpenCV:
 * cv::Mat (header only: rows, cols, data, step), cv::Point2f, cv::KeyPoint, cv::Size, cv::TermCriteria and the
 * declarations of cv::FAST / cv::calcOpticalFlowPyrLK / cv::goodFeaturesToTrack.  No OpenCV algorithm lives
 * here: ref_glue.cpp defines FAST and calcOpticalFlowPyrLK by forwarding to the oracle's restatement
 * (orc_fast_detect, orc_calc_optical_flow_pyr_lk), so what oracle/_ref/libvo_refglue.so pins is the REFERENCE'S OWN
 * LOGIC -- the order of the four LK calls, deleteUnmatchFeaturesCircle's erase / age semantics, the bucket
 * class and bucketingFeatures' indexing quirks, appendNewFeatures -- not OpenCV's arithmetic (still unpinned).
 */
#ifndef VO_CV_SHIM_H
#define VO_CV_SHIM_H

#include <stddef.h>
#include <vector>

typedef unsigned char uchar;

namespace cv {

template <typename T>
struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<float> Point2f;

struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
};

struct TermCriteria {
    enum { COUNT = 1, MAX_ITER = COUNT, EPS = 2 };
    int type, maxCount;
    double epsilon;
    TermCriteria() : type(0), maxCount(0), epsilon(0) {}
    TermCriteria(int t, int c, double e) : type(t), maxCount(c), epsilon(e) {}
};

/* 8-bit single-channel image header (the reference passes cv::Mat by value = a header copy) */
struct Mat {
    int rows, cols;
    uchar *data;
    size_t step;
    Mat() : rows(0), cols(0), data(0), step(0) {}
    Mat(int r, int c, uchar *d, size_t s) : rows(r), cols(c), data(d), step(s) {}
};

struct KeyPoint {
    Point2f pt;
    static void convert(const std::vector<KeyPoint> &keypoints, std::vector<Point2f> &points2f,
                        const std::vector<int> &keypointIndexes = std::vector<int>());
};

void FAST(Mat image, std::vector<KeyPoint> &keypoints, int threshold, bool nonmaxSuppression = true);
void goodFeaturesToTrack(Mat image, std::vector<Point2f> &corners, int maxCorners, double qualityLevel,
                         double minDistance, Mat mask, int blockSize, bool useHarrisDetector, double k);
void calcOpticalFlowPyrLK(Mat prevImg, Mat nextImg, std::vector<Point2f> &prevPts, std::vector<Point2f> &nextPts,
                          std::vector<uchar> &status, std::vector<float> &err, Size winSize, int maxLevel,
                          TermCriteria criteria, int flags, double minEigThreshold);

} // namespace cv

#endif
