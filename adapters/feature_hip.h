// feature_hip.h -- the drop-in adapter a maintainer adds to ZhenghaoFei/visual_odom (src/) so that circularMatching(),
// the triangulation pair of main.cpp and trackingFrame2Frame() run on libvo_hip.so (MI355X) instead of OpenCV.
//
//   circularMatching_hip      same signature as circularMatching        feature.h:61-65   (body feature.cpp:118-148)
//   trackingFrame2Frame_hip   same signature as trackingFrame2Frame     visualOdometry.h:36-42 (body visualOdometry.cpp:132-193)
//   triangulate_hip           replaces cv::triangulatePoints + cv::convertPointsFromHomogeneous   main.cpp:169-171
//   detectAndBucket_hip       replaces the head of matchingFeatures     visualOdometry.cpp:95-108
//
// Switch: -DUSE_HIP next to the reference's own USE_CUDA (CMakeLists.txt:5-9, visualOdometry.cpp:112-118); see
// adapters/USE_HIP.cmake and INTEGRATION.md.  OpenCV is needed for the TYPES of the existing signatures only (cv::Mat,
// cv::Point2f): no OpenCV algorithm is called.  Where the OpenCV headers are not installed (this repository's test build)
// the same file compiles against a type-only stand-in that the include path provides (tests/ref_dropin/Makefile:
// -I oracle/ref_shim); tests/ref_dropin compiles THIS file with the reference's unmodified sources and
// tests/test_gpu_parity.py::test_reference_sources_run_on_libvo_hip runs the result on the GPU.
#pragma once

#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>   // a machine that has OpenCV: the real cv::Mat / cv::Point2f
#else
#include "vo_cv_shim.h"       // no OpenCV installed: type-only stand-in (test builds of this repository)
#endif

#include <vector>

#include "feature.h"          // FeatureSet (feature.h:33-43), from the reference tree
#include "vo_hip.h"           // C ABI of libvo_hip.so

// same signature as circularMatching (feature.h:61-65)
void circularMatching_hip(cv::Mat img_l_0, cv::Mat img_r_0, cv::Mat img_l_1, cv::Mat img_r_1,
                          std::vector<cv::Point2f>& points_l_0, std::vector<cv::Point2f>& points_r_0,
                          std::vector<cv::Point2f>& points_l_1, std::vector<cv::Point2f>& points_r_1,
                          std::vector<cv::Point2f>& points_l_0_return, FeatureSet& current_features);

// same signature as trackingFrame2Frame (visualOdometry.h:36-42)
void trackingFrame2Frame_hip(cv::Mat& projMatrl, cv::Mat& projMatrr, std::vector<cv::Point2f>& pointsLeft_t0,
                             std::vector<cv::Point2f>& pointsLeft_t1, cv::Mat& points3D_t0, cv::Mat& rotation,
                             cv::Mat& translation, bool mono_rotation = true);

// replaces main.cpp:169-171 (points3D_t0: N x 1 CV_32FC3, what convertPointsFromHomogeneous returns)
void triangulate_hip(cv::Mat& projMatrl, cv::Mat& projMatrr, std::vector<cv::Point2f>& pointsLeft_t0,
                     std::vector<cv::Point2f>& pointsRight_t0, cv::Mat& points3D_t0);

// replaces visualOdometry.cpp:95-108: appendNewFeatures when fewer than 2000 features are carried + bucketingFeatures
// (bucket_size = rows / 10, one feature per bucket: the reference's literals)
void detectAndBucket_hip(cv::Mat& image, FeatureSet& current_features);

// OPT-IN, default off: the caller promises that the t0 pair of every circularMatching_hip call IS the t1 pair of the call
// before it -- the reference's loop hands its frames over exactly so (main.cpp:157-158: imageLeft_t0 = imageLeft_t1 shares
// the pixel buffer).  A call whose t0 images have the size, row step and DATA POINTERS of the previous call's t1 images then
// names the pair libvo_hip.so kept on the device with its pyramids (vo_hip.h, THE KEPT PAIR) instead of sending it again:
// two images cross PCIe instead of four and the first hop of the chain starts before they have arrived.  Same results bit
// for bit as long as the promise holds; the pointers only identify the buffer, they cannot see a caller who rewrites it.
// Anything else (first call, other buffers, another user of the context in between: vo_kept_pair_id) sends four images.
void vo_adapter_keep_pair(bool on);
long vo_adapter_kept_calls(); // circularMatching_hip calls that went without their t0 pair so far

// the context the adapter functions share (one per process: the reference is single-threaded), grown to hold a w x h image
// and n points; throws std::runtime_error when there is no HIP device (there is no CPU fallback)
vo_ctx* vo_adapter_context_for(int w, int h, int n);
