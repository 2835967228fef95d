// vo_run.cpp -- the reference's frame loop (ZhenghaoFei/visual_odom src/main.cpp:28-227) as a C++ host
// program over the C ABI of libvo_hip.so: same state (FeatureSet points / ages, frame_pose, translation),
// same order of operations, every arithmetic step inside the library.  No OpenCV: images are read as
// binary PGM (P5), calibration is given as numbers (calibration/kitti00.yaml: fx cx cy bf), GUI calls are
// dropped and the trajectory the reference only draws (utils.cpp:19-48) is written in the KITTI pose
// format (12 doubles per line, what loadPoses reads, evaluate_odometry.cpp:24-27).
//
//   vo_run <sequence_dir> <fx> <cx> <cy> <bf> <n_frames> <poses_out.txt> [features_per_bucket]
//   images: <sequence_dir>/image_0/%06d.pgm (left), image_1/%06d.pgm (right)   (utils.cpp:172-190)
//
// build: g++ -O2 -std=c++17 vo_run.cpp -I../include -L../visual_odom_amd -lvo_hip -Wl,-rpath,... -o vo_run
#include "vo_hip.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> px;
};

static bool read_pgm(const std::string &path, Image &im)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f)
        return false;
    char magic[3] = {0, 0, 0};
    int maxv = 0;
    bool ok = fscanf(f, "%2s %d %d %d", magic, &im.w, &im.h, &maxv) == 4 && !strcmp(magic, "P5") && maxv == 255;
    if (ok) {
        fgetc(f); // the single whitespace byte after the header
        im.px.resize((size_t)im.w * im.h);
        ok = fread(im.px.data(), 1, im.px.size(), f) == im.px.size();
    }
    fclose(f);
    return ok;
}

static std::string frame_path(const std::string &dir, int cam, int id)
{
    char buf[64];
    snprintf(buf, sizeof(buf), "/image_%d/%06d.pgm", cam, id);
    return dir + buf;
}

#define CHECK(call)                                                                       \
    do {                                                                                  \
        int rc_ = (call);                                                                 \
        if (rc_ < 0) {                                                                    \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, vo_last_error(ctx));      \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 8) {
        fprintf(stderr, "usage: %s <sequence_dir> <fx> <cx> <cy> <bf> <n_frames> <poses_out> [features_per_bucket [mono_rotation]]\n", argv[0]);
        return 1;
    }
    const std::string dir = argv[1];
    const float fx = (float)atof(argv[2]), cx = (float)atof(argv[3]), cy = (float)atof(argv[4]), bf = (float)atof(argv[5]);
    const int n_frames = atoi(argv[6]);
    const char *out_path = argv[7];
    // projMatrl / projMatrr (main.cpp:73-74)
    const float P_l[12] = {fx, 0, cx, 0, 0, fx, cy, 0, 0, 0, 1, 0};
    const float P_r[12] = {fx, 0, cx, bf, 0, fx, cy, 0, 0, 0, 1, 0};

    Image l0, r0, l1, r1;
    if (!read_pgm(frame_path(dir, 0, 0), l0) || !read_pgm(frame_path(dir, 1, 0), r0)) {
        fprintf(stderr, "cannot read frame 0 under %s\n", dir.c_str());
        return 1;
    }
    const int cap = 32768;
    vo_ctx *ctx = vo_create(0, l0.w, l0.h, 8192, 1);
    if (!ctx) {
        fprintf(stderr, "vo_create failed: no HIP device (there is no CPU fallback)\n");
        return 2;
    }
    vo_detect_params dp;
    vo_default_detect_params(&dp);
    if (argc > 8)
        dp.features_per_bucket = atoi(argv[8]);
    if (argc > 9 && atoi(argv[9]) != 0) { // trackingFrame2Frame(..., mono_rotation = true): rotation from recoverPose
        vo_params prm;
        CHECK(vo_get_params(ctx, &prm));
        prm.mono_rotation = 1;
        CHECK(vo_set_params(ctx, &prm));
    }

    // main.cpp:81-94
    std::vector<float> points((size_t)2 * cap);  // currentVOFeatures.points
    std::vector<int32_t> ages(cap);              // currentVOFeatures.ages
    int n_pts = 0, n_ages = 0;
    double frame_pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double rotation[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, translation[3] = {0, 0, 0};
    std::vector<float> pl0((size_t)2 * cap), pr0((size_t)2 * cap), pl1((size_t)2 * cap), pr1((size_t)2 * cap), xyz((size_t)3 * cap);
    std::vector<int32_t> keep(cap), keep_circ(cap), inliers(cap), ages_next(cap);

    FILE *out = fopen(out_path, "w");
    if (!out)
        return 1;
    auto dump_pose = [&] {
        for (int k = 0; k < 12; k++)
            fprintf(out, "%.9e%c", frame_pose[k], k == 11 ? '\n' : ' ');
    };
    dump_pose();

    for (int id = 1; id < n_frames; id++) {
        if (!read_pgm(frame_path(dir, 0, id), l1) || !read_pgm(frame_path(dir, 1, id), r1))
            break; // the reference runs until imread fails (main.cpp:123)
        // matchingFeatures head: appendNewFeatures + bucketingFeatures (visualOdometry.cpp:95-108)
        // From the second iteration on the t0 pair is the pair the previous vo_track_frame received as t1 (main.cpp:157-158)
        // and the library still holds it, pyramids and all: NULL instead of the t0 images (vo_hip.h, THE KEPT PAIR)
        const bool kept = id > 1;
        CHECK(vo_detect_bucket(ctx, kept ? nullptr : l0.px.data(), l0.w, l0.h, l0.w, &dp, points.data(), &n_pts, ages.data(), &n_ages,
                               cap));
        const int n_in = n_pts;
        // circularMatching + consistency filter + triangulation + PnP (visualOdometry.cpp:110-127, main.cpp:169-181)
        int k_out = 0, m_circ = 0, n_inl = 0;
        double rvec[3] = {0, 0, 0}; // visualOdometry.cpp:162
        int rc = vo_track_frame(ctx, kept ? nullptr : l0.px.data(), kept ? nullptr : r0.px.data(), l1.px.data(), r1.px.data(), l0.w,
                                l0.h, l0.w, points.data(),
                                n_in, P_l, P_r, pl0.data(), pr0.data(), pl1.data(), pr1.data(), xyz.data(), keep.data(), &k_out,
                                keep_circ.data(), &m_circ, rvec, translation, rotation, inliers.data(), &n_inl);
        if (rc == VO_ERR_TOO_FEW) {
            fprintf(stderr, "frame %d: fewer than 5 correspondences (the reference asserts here)\n", id);
            return 3;
        }
        CHECK(rc);
        if (rc == VO_NO_ESSENTIAL) { // mono_rotation and findEssentialMat found nothing: the reference's recoverPose throws
            fprintf(stderr, "frame %d: no essential matrix (the reference throws here)\n", id);
            return 4;
        }
        // deleteUnmatchFeaturesCircle: ages += 1, compacted with the circular-matching survivors only
        // (feature.cpp:83-86,111); the consistency filter leaves ages alone (quirk B3)
        for (int i = 0; i < m_circ; i++)
            ages_next[i] = ages[keep_circ[i]] + 1;
        memcpy(ages.data(), ages_next.data(), sizeof(int32_t) * m_circ);
        n_ages = m_circ;
        memcpy(points.data(), pl1.data(), sizeof(float) * 2 * k_out); // currentVOFeatures.points = pointsLeft_t1
        n_pts = k_out;
        std::swap(l0, l1); // main.cpp:157-158
        std::swap(r0, r1);
        // main.cpp:196-208
        float euler[3];
        const int applied = vo_integrate_odometry(frame_pose, rotation, translation, euler);
        fprintf(stderr, "frame %d: %d bucketed -> %d tracked -> %d inliers, t = (%.4f %.4f %.4f)%s\n", id, n_in, k_out, n_inl,
                translation[0], translation[1], translation[2], applied ? "" : "  [rejected by the motion gates]");
        dump_pose();
    }
    fclose(out);
    vo_destroy(ctx);
    return 0;
}
