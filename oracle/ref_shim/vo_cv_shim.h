/*
 * vo_cv_shim.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A stand-in for the OpenCV declarations that the reference's own sources (/root/reference/src/feature.cpp,
 * bucket.cpp, visualOdometry.cpp, utils.cpp) need in order to compile WHERE THEY LIE, without OpenCV:
 *   * value types: cv::Point_, Size, TermCriteria, KeyPoint, Vec3f, Scalar;
 *   * a small dense cv::Mat (2-D, 8U / 32F / 64F, up to 3 channels, reference-counted storage like the real one)
 *     with exactly the members those sources use: at<>, zeros / eye, Mat_<T>(r, c) << a, b, ..., t(), inv()
 *     (cv::invert(DECOMP_LU) restated: LU elimination with partial pivoting, zero matrix when singular), operator*, hconcat / vconcat / transpose / norm, clone, col, type, size;
 *   * drawing / window / file functions as no-ops;
 *   * the algorithm entry points (FAST, calcOpticalFlowPyrLK, triangulatePoints, convertPointsFromHomogeneous,
 *     solvePnPRansac, Rodrigues, findEssentialMat, recoverPose) as declarations only: ref_glue.cpp defines them by
 *     forwarding to the oracle's restatement (oracle/orc_*.c).
 * So what oracle/_ref/libvo_refglue.so pins is the REFERENCE'S OWN LOGIC -- matchingFeatures(), circularMatching(),
 * deleteUnmatchFeaturesCircle(), the bucket class and bucketingFeatures' indexing quirks, appendNewFeatures(),
 * checkValidMatch / removeInvalidPoints, trackingFrame2Frame()'s argument plumbing, rotationMatrixToEulerAngles(),
 * integrateOdometryStereo() -- not OpenCV's arithmetic, which stays unpinned (DESIGN.md "oracle").
 */
#ifndef VO_CV_SHIM_H
#define VO_CV_SHIM_H

#include <assert.h> /* the real headers pull it in; utils.cpp relies on that */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include <iostream>
#include <memory>
#include <string>
#include <vector>

typedef unsigned char uchar;

#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC2 CV_MAKETYPE(CV_32F, 2)
#define CV_32FC3 CV_MAKETYPE(CV_32F, 3)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)
#define CV_RGB(r, g, b) cv::Scalar((b), (g), (r), 0)

namespace cv {

template <typename T>
struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
    template <typename U>
    Point_(const Point_<U> &p) : x((T)p.x), y((T)p.y) {}
};
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;
typedef Point_<int> Point;

struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
};
inline std::ostream &operator<<(std::ostream &os, const Size &s) { return os << "[" << s.width << " x " << s.height << "]"; }

struct Scalar {
    double val[4];
    Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : val{a, b, c, d} {}
};

struct Vec3f {
    float val[3];
    Vec3f() : val{0, 0, 0} {}
    Vec3f(float a, float b, float c) : val{a, b, c} {}
    float &operator[](int i) { return val[i]; }
    const float &operator[](int i) const { return val[i]; }
};

struct TermCriteria {
    enum { COUNT = 1, MAX_ITER = COUNT, EPS = 2 };
    int type, maxCount;
    double epsilon;
    TermCriteria() : type(0), maxCount(0), epsilon(0) {}
    TermCriteria(int t, int c, double e) : type(t), maxCount(c), epsilon(e) {}
};

enum { RANSAC = 8, SOLVEPNP_ITERATIVE = 0, COLOR_BGR2GRAY = 6, COLOR_GRAY2BGR = 8, IMREAD_COLOR = 1 };

class Mat {
public:
    int rows, cols;
    uchar *data;
    size_t step; /* bytes per row */

    Mat() : rows(0), cols(0), data(0), step(0), type_(0) {}
    Mat(int r, int c, int type) { create(r, c, type); }
    /* header over caller-owned memory (the reference's images) */
    Mat(int r, int c, int type, void *d, size_t s = 0) : rows(r), cols(c), data((uchar *)d), type_(type)
    {
        step = s ? s : (size_t)c * elem_size();
    }
    void create(int r, int c, int type)
    {
        rows = r;
        cols = c;
        type_ = type;
        step = (size_t)c * elem_size();
        store_.reset(new std::vector<uchar>((size_t)r * step + 8, 0));
        data = store_->data();
    }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    static Mat eye(int r, int c, int type)
    {
        Mat m(r, c, type);
        for (int i = 0; i < r && i < c; i++)
            m.set(i, i, 1.0);
        return m;
    }
    int type() const { return type_; }
    int depth() const { return type_ & 7; }
    int channels() const { return (type_ >> 3) + 1; }
    size_t elem_size() const { return (size_t)channels() * (depth() == CV_8U ? 1 : depth() == CV_32F ? 4 : 8); }
    bool empty() const { return data == 0 || rows * cols == 0; }
    size_t total() const { return (size_t)rows * cols; }
    Size size() const { return Size(cols, rows); }

    template <typename T>
    T &at(int i, int j) { return *(T *)(data + (size_t)i * step + (size_t)j * sizeof(T)); }
    template <typename T>
    const T &at(int i, int j) const { return *(const T *)(data + (size_t)i * step + (size_t)j * sizeof(T)); }
    template <typename T>
    T &at(int i) { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    template <typename T>
    const T &at(int i) const { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    template <typename T>
    T *ptr(int r = 0) { return (T *)(data + (size_t)r * step); }
    template <typename T>
    const T *ptr(int r = 0) const { return (const T *)(data + (size_t)r * step); }

    /* single-channel element access as double, whatever the depth */
    double get(int i, int j) const
    {
        return depth() == CV_64F ? at<double>(i, j) : depth() == CV_32F ? (double)at<float>(i, j) : (double)at<uchar>(i, j);
    }
    void set(int i, int j, double v)
    {
        if (depth() == CV_64F)
            at<double>(i, j) = v;
        else if (depth() == CV_32F)
            at<float>(i, j) = (float)v;
        else
            at<uchar>(i, j) = (uchar)v;
    }
    Mat clone() const
    {
        Mat m(rows, cols, type_);
        for (int i = 0; i < rows; i++)
            memcpy(m.data + (size_t)i * m.step, data + (size_t)i * step, (size_t)cols * elem_size());
        return m;
    }
    Mat col(int j) const
    {
        Mat m(rows, 1, type_);
        for (int i = 0; i < rows; i++)
            memcpy(m.data + (size_t)i * m.step, data + (size_t)i * step + (size_t)j * elem_size(), elem_size());
        return m;
    }
    Mat t() const
    {
        Mat m(cols, rows, type_);
        const size_t es = elem_size();
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols; j++)
                memcpy(m.data + (size_t)j * m.step + (size_t)i * es, data + (size_t)i * step + (size_t)j * es, es);
        return m;
    }
    /* square inverse as cv::invert(DECOMP_LU) forms it for n > 3: LU elimination with partial pivoting on [A | I]
     * (hal::LU64f / LUImpl: pivot < DBL_EPSILON * 100 -> singular -> zero matrix), back substitution */
    Mat inv() const
    {
        const int n = rows;
        std::vector<double> A((size_t)n * n), b((size_t)n * n, 0.0);
        for (int i = 0; i < n; i++) {
            for (int j = 0; j < n; j++)
                A[(size_t)i * n + j] = get(i, j);
            b[(size_t)i * n + i] = 1.0;
        }
        for (int i = 0; i < n; i++) {
            int k = i;
            for (int j = i + 1; j < n; j++)
                if (fabs(A[(size_t)j * n + i]) > fabs(A[(size_t)k * n + i]))
                    k = j;
            if (fabs(A[(size_t)k * n + i]) < 2.220446049250313e-16 * 100)
                return Mat::zeros(n, n, type_);
            if (k != i) {
                for (int j = i; j < n; j++)
                    std::swap(A[(size_t)i * n + j], A[(size_t)k * n + j]);
                for (int j = 0; j < n; j++)
                    std::swap(b[(size_t)i * n + j], b[(size_t)k * n + j]);
            }
            const double d = -1 / A[(size_t)i * n + i];
            for (int j = i + 1; j < n; j++) {
                const double alpha = A[(size_t)j * n + i] * d;
                for (int c = i + 1; c < n; c++)
                    A[(size_t)j * n + c] += alpha * A[(size_t)i * n + c];
                for (int c = 0; c < n; c++)
                    b[(size_t)j * n + c] += alpha * b[(size_t)i * n + c];
            }
        }
        for (int i = n - 1; i >= 0; i--)
            for (int j = 0; j < n; j++) {
                double s = b[(size_t)i * n + j];
                for (int c = i + 1; c < n; c++)
                    s -= A[(size_t)i * n + c] * b[(size_t)c * n + j];
                b[(size_t)i * n + j] = s / A[(size_t)i * n + i];
            }
        Mat m(n, n, type_);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++)
                m.set(i, j, b[(size_t)i * n + j]);
        return m;
    }

protected:
    int type_;
    std::shared_ptr<std::vector<uchar> > store_;
};

inline Mat operator*(const Mat &a, const Mat &b)
{
    Mat m(a.rows, b.cols, a.type());
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < b.cols; j++) {
            double s = 0;
            for (int k = 0; k < a.cols; k++)
                s += a.get(i, k) * b.get(k, j);
            m.set(i, j, s);
        }
    return m;
}

inline std::ostream &operator<<(std::ostream &os, const Mat &m)
{
    os << "[";
    for (int i = 0; i < m.rows; i++) {
        for (int j = 0; j < m.cols; j++)
            os << (j ? ", " : "") << m.get(i, j);
        os << (i + 1 < m.rows ? ";\n " : "");
    }
    return os << "]";
}

template <typename T>
struct DepthOf;
template <>
struct DepthOf<float> {
    enum { value = CV_32F };
};
template <>
struct DepthOf<double> {
    enum { value = CV_64F };
};
template <>
struct DepthOf<uchar> {
    enum { value = CV_8U };
};

template <typename T>
class Mat_;

/* (cv::Mat_<T>(r, c) << a, b, c, ...) */
template <typename T>
class MatCommaInitializer_ {
public:
    MatCommaInitializer_(Mat_<T> *m) : m_(m), k_(0) {}
    template <typename U>
    MatCommaInitializer_<T> &operator,(U v)
    {
        put((T)v);
        return *this;
    }
    void put(T v);
    operator Mat() const;

private:
    Mat_<T> *m_;
    int k_;
};

template <typename T>
class Mat_ : public Mat {
public:
    Mat_(int r, int c) : Mat(r, c, DepthOf<T>::value) {}
    template <typename U>
    MatCommaInitializer_<T> operator<<(U v)
    {
        MatCommaInitializer_<T> ci(this);
        ci.put((T)v);
        return ci;
    }
};

template <typename T>
void MatCommaInitializer_<T>::put(T v)
{
    m_->template at<T>(k_ / m_->cols, k_ % m_->cols) = v;
    k_++;
}
template <typename T>
MatCommaInitializer_<T>::operator Mat() const
{
    return *m_;
}

inline void hconcat(const Mat &a, const Mat &b, Mat &dst)
{
    Mat m(a.rows, a.cols + b.cols, a.type());
    for (int i = 0; i < a.rows; i++) {
        for (int j = 0; j < a.cols; j++)
            m.set(i, j, a.get(i, j));
        for (int j = 0; j < b.cols; j++)
            m.set(i, a.cols + j, b.get(i, j));
    }
    dst = m;
}
inline void vconcat(const Mat &a, const Mat &b, Mat &dst)
{
    Mat m(a.rows + b.rows, a.cols, a.type());
    for (int j = 0; j < a.cols; j++) {
        for (int i = 0; i < a.rows; i++)
            m.set(i, j, a.get(i, j));
        for (int i = 0; i < b.rows; i++)
            m.set(a.rows + i, j, b.get(i, j));
    }
    dst = m;
}
inline void transpose(const Mat &src, Mat &dst) { dst = src.t(); }
inline double norm(const Mat &a, const Mat &b) /* NORM_L2 of the difference */
{
    double s = 0;
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) {
            const double d = a.get(i, j) - b.get(i, j);
            s += d * d;
        }
    return sqrt(s);
}

struct KeyPoint {
    Point2f pt;
    static void convert(const std::vector<KeyPoint> &keypoints, std::vector<Point2f> &points2f,
                        const std::vector<int> &keypointIndexes = std::vector<int>());
};

/* ---- display / file I/O: not on the path, no-ops ---- */
inline void circle(Mat, Point, int, Scalar, int = 1) {}
inline void line(Mat, Point2f, Point2f, Scalar, int = 1) {}
inline void imshow(const std::string &, const Mat &) {}
inline int waitKey(int = 0) { return -1; }
inline void cvtColor(const Mat &src, Mat &dst, int, int = 0) { dst = src; }
inline Mat imread(const std::string &, int = 1) { return Mat(); }

/* ---- the OpenCV algorithms the reference calls: defined in ref_glue.cpp over the oracle's restatement ---- */
void FAST(Mat image, std::vector<KeyPoint> &keypoints, int threshold, bool nonmaxSuppression = true);
void goodFeaturesToTrack(Mat image, std::vector<Point2f> &corners, int maxCorners, double qualityLevel,
                         double minDistance, Mat mask, int blockSize, bool useHarrisDetector, double k);
void calcOpticalFlowPyrLK(Mat prevImg, Mat nextImg, std::vector<Point2f> &prevPts, std::vector<Point2f> &nextPts,
                          std::vector<uchar> &status, std::vector<float> &err, Size winSize, int maxLevel,
                          TermCriteria criteria, int flags, double minEigThreshold);
void triangulatePoints(const Mat &projMatr1, const Mat &projMatr2, const std::vector<Point2f> &projPoints1,
                       const std::vector<Point2f> &projPoints2, Mat &points4D);
void convertPointsFromHomogeneous(const Mat &src, Mat &dst);
bool solvePnPRansac(const Mat &objectPoints, const std::vector<Point2f> &imagePoints, const Mat &cameraMatrix,
                    const Mat &distCoeffs, Mat &rvec, Mat &tvec, bool useExtrinsicGuess, int iterationsCount,
                    float reprojectionError, double confidence, Mat &inliers, int flags);
void Rodrigues(const Mat &src, Mat &dst);
Mat findEssentialMat(const std::vector<Point2f> &points1, const std::vector<Point2f> &points2, double focal, Point2d pp,
                     int method, double prob, double threshold, Mat &mask);
int recoverPose(const Mat &E, const std::vector<Point2f> &points1, const std::vector<Point2f> &points2, Mat &R, Mat &t,
                double focal, Point2d pp, Mat &mask);

} // namespace cv

#endif
