// cv_compat.h -- the handful of OpenCV value types that cross the adjustBundle() boundary.
//
// The reference's data model (SfMToyLib/SfMCommon.h:55-99) is built on cv::Matx34f, cv::Point3f,
// cv::Point2f and a CV_32F 3x3 cv::Mat.  OpenCV is not installable in this environment, so this header
// provides layout- and API-compatible stand-ins for exactly the members adjustBundle() touches
// (BA.cpp:111-221).  Building with -DSFMBA_HAVE_OPENCV uses the real headers instead; the shim source
// is identical in both cases.
#pragma once
#ifdef SFMBA_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#else
#include <cstring>
#include <vector>

namespace cv {

template <typename T, int M, int N>
struct Matx {
    T val[M * N];
    Matx() { for (int i = 0; i < M * N; ++i) val[i] = T(0); }
    T& operator()(int r, int c) { return val[r * N + c]; }
    const T& operator()(int r, int c) const { return val[r * N + c]; }
    template <int M1, int N1>
    Matx<T, M1, N1> get_minor(int r0, int c0) const {
        Matx<T, M1, N1> m;
        for (int r = 0; r < M1; ++r) for (int c = 0; c < N1; ++c) m(r, c) = (*this)(r0 + r, c0 + c);
        return m;
    }
    Matx<T, N, M> t() const {
        Matx<T, N, M> m;
        for (int r = 0; r < M; ++r) for (int c = 0; c < N; ++c) m(c, r) = (*this)(r, c);
        return m;
    }
};
typedef Matx<float, 3, 4> Matx34f;
typedef Matx<float, 3, 3> Matx33f;

template <typename T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T x_, T y_) : x(x_), y(y_) {} };
template <typename T> struct Point3_ { T x, y, z; Point3_() : x(0), y(0), z(0) {} Point3_(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {} };
typedef Point_<float> Point2f;
typedef Point3_<float> Point3f;

struct KeyPoint { Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1; };
struct DMatch { int queryIdx = -1, trainIdx = -1, imgIdx = -1; float distance = 0; DMatch() {} DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), distance(d) {} };

// Dense row-major float matrix: only what Intrinsics::K needs (K.at<float>(r, c), BA.cpp:138,151-153,188-189).
class Mat {
public:
    Mat() : rows(0), cols(0) {}
    Mat(int r, int c) : rows(r), cols(c), data_((size_t)r * c, 0.0f) {}
    template <typename T> T& at(int r, int c) { return reinterpret_cast<T&>(data_[(size_t)r * cols + c]); }
    template <typename T> const T& at(int r, int c) const { return reinterpret_cast<const T&>(data_[(size_t)r * cols + c]); }
    bool empty() const { return data_.empty(); }
    int rows, cols;
private:
    std::vector<float> data_;
};

}  // namespace cv
#endif
