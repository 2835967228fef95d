// vo_seq_run.cpp -- the reference's frame loop (ZhenghaoFei/visual_odom src/main.cpp:123-224) for SEVERAL sequences at
// once, as a C++ host program over the lock-step sequence API of libvo_hip.so (vo_seq_*): one step = one new stereo
// pair of every sequence; the state that chains a sequence's frames (currentVOFeatures, previous pair, frame_pose) stays
// on the GPU, the host only reads images and, at the end, the trajectories.  One worker = one host thread + one vo_ctx
// on one GPU (examples/vo_seq_host.h); examples/vo_multi_gpu.cpp runs one such worker per GPU (BASELINE "config 5").
// Sequences may have different lengths: a sequence whose images run out simply stops.
//
//   vo_seq_run [--device D] [--decode-threads T] <fx> <cx> <cy> <bf> <max_frames> <features_per_bucket> <out_prefix>
//              <sequence_dir> [<sequence_dir> ...]
//   images:  <sequence_dir>/image_0/%06d.png (left), image_1/%06d.png (right), or .pgm         (utils.cpp:172-190)
//   output:  <out_prefix>_<s>.txt, KITTI pose format (12 doubles per line, evaluate_odometry.cpp:24-27);
//            stderr: end-to-end frames/s from the files (read + decode + upload + compute) and the time the loop spent
//            waiting for the decoders
//
// build: g++ -O2 -std=c++17 vo_seq_run.cpp -I../include -L../visual_odom_amd -lvo_hip -lz -lpthread -Wl,-rpath,... -o vo_seq_run
#include "vo_seq_host.h"

#include <cstdlib>
#include <cstring>

int main(int argc, char **argv)
{
    int device = 0, decode_threads = 4, a = 1;
    while (a + 1 < argc && !strncmp(argv[a], "--", 2)) {
        if (!strcmp(argv[a], "--device"))
            device = atoi(argv[a + 1]);
        else if (!strcmp(argv[a], "--decode-threads"))
            decode_threads = atoi(argv[a + 1]);
        else
            break;
        a += 2;
    }
    if (argc - a < 8) {
        fprintf(stderr, "usage: %s [--device D] [--decode-threads T] <fx> <cx> <cy> <bf> <max_frames> <features_per_bucket> "
                        "<out_prefix> <sequence_dir> ...\n", argv[0]);
        return 1;
    }
    vohost::Calib cal;
    cal.fx = (float)atof(argv[a]);
    cal.cx = (float)atof(argv[a + 1]);
    cal.cy = (float)atof(argv[a + 2]);
    cal.bf = (float)atof(argv[a + 3]);
    const int max_frames = atoi(argv[a + 4]), per_bucket = atoi(argv[a + 5]);
    const std::string prefix = argv[a + 6];
    std::vector<std::string> dirs(argv + a + 7, argv + argc);
    vohost::WorkerResult r = vohost::run_worker(device, dirs, cal, max_frames, per_bucket, decode_threads);
    if (r.rc) {
        fprintf(stderr, "%s\n", r.error.c_str());
        return r.rc;
    }
    for (size_t s = 0; s < dirs.size(); s++) {
        const std::string path = prefix + "_" + std::to_string(s) + ".txt";
        int integrated = 0;
        if (!vohost::write_trajectory(path, r.rows[s], r.info[s], &integrated))
            return 1;
        fprintf(stderr, "sequence %zu (%s): %zu frames, %d integrated -> %s\n", s, dirs[s].c_str(),
                r.rows[s].size() / VO_SEQ_ROW, integrated, path.c_str());
    }
    fprintf(stderr, "device %d: %ld frames of %zu sequences in %.3f s = %.1f frames/s end to end from the image files "
                    "(%d decoder threads; %.3f s spent waiting for them)\n",
            device, r.frames, dirs.size(), r.seconds, r.seconds > 0 ? r.frames / r.seconds : 0.0, decode_threads,
            r.decode_seconds);
    return 0;
}
