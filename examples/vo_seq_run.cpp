// vo_seq_run.cpp -- the reference's frame loop (ZhenghaoFei/visual_odom src/main.cpp:123-224) for SEVERAL sequences at
// once, as a C++ host program over the lock-step sequence API of libvo_hip.so (vo_seq_*): one step = one new stereo
// pair of every sequence; the state that chains a sequence's frames (currentVOFeatures, previous pair, frame_pose) stays
// on the GPU, the host only reads images and, at the end, the trajectories.  This is the BASELINE "config 5" harness:
// one process per GPU (sequence s -> GPU s % n_gpus, visual_odom_amd/replicas.py), several sequences per process,
// no collective.  Sequences may have different lengths: a sequence whose images run out simply stops.
//
//   vo_seq_run <fx> <cx> <cy> <bf> <max_frames> <features_per_bucket> <out_prefix> <sequence_dir> [<sequence_dir> ...]
//   images:  <sequence_dir>/image_0/%06d.pgm (left), image_1/%06d.pgm (right)       (utils.cpp:172-190)
//   output:  <out_prefix>_<s>.txt, KITTI pose format (12 doubles per line, evaluate_odometry.cpp:24-27)
//
// build: g++ -O2 -std=c++17 vo_seq_run.cpp -I../include -L../visual_odom_amd -lvo_hip -Wl,-rpath,... -o vo_seq_run
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
        fgetc(f);
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
    if (argc < 9) {
        fprintf(stderr, "usage: %s <fx> <cx> <cy> <bf> <max_frames> <features_per_bucket> <out_prefix> <sequence_dir> ...\n", argv[0]);
        return 1;
    }
    const float fx = (float)atof(argv[1]), cx = (float)atof(argv[2]), cy = (float)atof(argv[3]), bf = (float)atof(argv[4]);
    const int max_frames = atoi(argv[5]), per_bucket = atoi(argv[6]);
    const std::string prefix = argv[7];
    std::vector<std::string> dirs(argv + 8, argv + argc);
    const int S = (int)dirs.size();
    const float P_l[12] = {fx, 0, cx, 0, 0, fx, cy, 0, 0, 0, 1, 0}; // projMatrl / projMatrr (main.cpp:73-74)
    const float P_r[12] = {fx, 0, cx, bf, 0, fx, cy, 0, 0, 0, 1, 0};

    Image l, r;
    if (!read_pgm(frame_path(dirs[0], 0, 0), l)) {
        fprintf(stderr, "cannot read frame 0 under %s\n", dirs[0].c_str());
        return 1;
    }
    const int w = l.w, h = l.h;
    vo_ctx *ctx = vo_create(0, w, h, 4096, S);
    if (!ctx) {
        fprintf(stderr, "vo_create failed: no HIP device (there is no CPU fallback)\n");
        return 2;
    }
    vo_detect_params dp;
    vo_default_detect_params(&dp);
    dp.features_per_bucket = per_bucket;
    CHECK(vo_batch_set_detect_params(ctx, &dp)); // before vo_seq_configure: they decide how a step is scheduled
    CHECK(vo_seq_configure(ctx, S, w, h, /*ring*/ 3, max_frames + 1));
    CHECK(vo_batch_set_projection(ctx, P_l, P_r));

    std::vector<char> live(S, 1);
    for (int id = 0; id < max_frames; id++) {
        int pushed = 0;
        for (int s = 0; s < S; s++) {
            if (!live[s])
                continue;
            if (!read_pgm(frame_path(dirs[s], 0, id), l) || !read_pgm(frame_path(dirs[s], 1, id), r) || l.w != w || l.h != h) {
                live[s] = 0; // the reference runs until imread fails (main.cpp:123)
                continue;
            }
            // pageable host memory: copied to the library's pinned staging now, moved to the GPU when the step runs
            CHECK(vo_seq_push_pair(ctx, s, l.px.data(), r.px.data(), w, /*host_pinned*/ 0));
            pushed++;
        }
        if (!pushed)
            break;
        CHECK(vo_seq_step(ctx)); // asynchronous: the next pairs are read from disk while this step runs
    }
    CHECK(vo_seq_sync(ctx));
    for (int s = 0; s < S; s++) {
        int n = 0;
        CHECK(vo_seq_get_trajectory(ctx, s, 0, 0, nullptr, nullptr, &n));
        std::vector<double> rows((size_t)(n > 0 ? n : 1) * VO_SEQ_ROW);
        std::vector<int32_t> info((size_t)(n > 0 ? n : 1) * VO_SEQ_INFO);
        CHECK(vo_seq_get_trajectory(ctx, s, 0, n, rows.data(), info.data(), &n));
        const std::string path = prefix + "_" + std::to_string(s) + ".txt";
        FILE *out = fopen(path.c_str(), "w");
        if (!out)
            return 1;
        fprintf(out, "1.000000000e+00 0 0 0 0 1.000000000e+00 0 0 0 0 1.000000000e+00 0\n"); // frame_pose of the first pair
        int integrated = 0;
        for (int k = 0; k < n; k++) {
            for (int j = 0; j < 12; j++)
                fprintf(out, "%.9e%c", rows[(size_t)k * VO_SEQ_ROW + j], j == 11 ? '\n' : ' ');
            integrated += (info[(size_t)k * VO_SEQ_INFO + 5] & VO_SEQ_F_INTEGRATED) != 0;
        }
        fclose(out);
        fprintf(stderr, "sequence %d (%s): %d frames, %d integrated -> %s\n", s, dirs[s].c_str(), n, integrated, path.c_str());
    }
    vo_destroy(ctx);
    return 0;
}
