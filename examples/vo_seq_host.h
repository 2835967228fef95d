// vo_seq_host.h -- one host worker of the lock-step sequence loop: ONE host thread + ONE vo_ctx on ONE GPU running the
// reference's frame loop (src/main.cpp:123-224) for the sequences handed to it.  examples/vo_seq_run.cpp runs one worker,
// examples/vo_multi_gpu.cpp one worker per GPU (sequence s -> GPU s % n, one host
// thread + one vo_ctx + hipSetDevice per GPU -- the library sets the device itself in every call -- no collective).
//
// Per step the worker pushes the next stereo pair of every live sequence and calls vo_seq_step (asynchronous); the pairs
// of step k + 1 are read and decoded by a thread pool while step k runs (vo_io.h).  A sequence whose images run out
// simply stops, like the reference's loop ends when imread fails.
#pragma once

#include "vo_hip.h"
#include "vo_io.h"

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

namespace vohost {

struct Calib {
    float fx = 0, cx = 0, cy = 0, bf = 0;
};

struct WorkerResult {
    int rc = 0;
    std::string error;
    int device = 0;
    long frames = 0;          // frames processed (trajectory rows) over all sequences of the worker
    double seconds = 0;       // wall clock from the first push to the last result, decode + upload + compute
    double decode_seconds = 0; // time the loop spent WAITING for the decoder pool (0 = decode fully hidden)
    std::vector<std::vector<double>> rows;   // per sequence: n x VO_SEQ_ROW
    std::vector<std::vector<int32_t>> info;  // per sequence: n x VO_SEQ_INFO
};

inline WorkerResult run_worker(int device, const std::vector<std::string> &dirs, const Calib &cal, int max_frames,
                               int features_per_bucket, int decode_threads)
{
    WorkerResult out;
    out.device = device;
    const int S = (int)dirs.size();
    out.rows.resize(S);
    out.info.resize(S);
    if (S == 0)
        return out;
    const float P_l[12] = {cal.fx, 0, cal.cx, 0, 0, cal.fx, cal.cy, 0, 0, 0, 1, 0}; // projMatrl / projMatrr (main.cpp:73-74)
    const float P_r[12] = {cal.fx, 0, cal.cx, cal.bf, 0, cal.fx, cal.cy, 0, 0, 0, 1, 0};
    voio::Image first;
    if (!voio::read_frame(dirs[0], 0, 0, first)) {
        out.rc = 1;
        out.error = "cannot read frame 0 under " + dirs[0];
        return out;
    }
    const int w = first.w, h = first.h;
    vo_ctx *ctx = vo_create(device, w, h, 4096, S);
    if (!ctx) {
        out.rc = 2;
        out.error = "vo_create failed on device " + std::to_string(device) + " (there is no CPU fallback)";
        return out;
    }
    auto fail = [&](const char *what, int rc) {
        out.rc = 2;
        out.error = std::string(what) + " failed (" + std::to_string(rc) + "): " + vo_last_error(ctx);
        vo_destroy(ctx);
        return out;
    };
    int rc;
    vo_detect_params dp;
    vo_default_detect_params(&dp);
    dp.features_per_bucket = features_per_bucket;
    if ((rc = vo_batch_set_detect_params(ctx, &dp)) < 0) // before vo_seq_configure: they decide how a step is scheduled
        return fail("vo_batch_set_detect_params", rc);
    if ((rc = vo_seq_configure(ctx, S, w, h, /*ring*/ 3, max_frames + 1)) < 0)
        return fail("vo_seq_configure", rc);
    if ((rc = vo_batch_set_projection(ctx, P_l, P_r)) < 0)
        return fail("vo_batch_set_projection", rc);

    // decoder pool with look-ahead: frames id + 1 .. id + depth are being read and decoded while frame id is pushed and
    // runs; depth = as many frames as keep every decoder thread busy (one sequence: a frame is only two files)
    const int depth = std::max(1, std::min(16, (decode_threads + 2 * S - 1) / (2 * S)));
    std::vector<voio::FrameSet> sets((size_t)depth + 1);
    voio::ThreadPool pool(decode_threads); // declared after the sets: on an early return it drains its jobs before they go
    std::vector<char> live(S, 1);
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k <= depth && k < max_frames; k++)
        sets[k % (depth + 1)].decode(pool, dirs, live, k, w, h);
    for (int id = 0; id < max_frames; id++) {
        voio::FrameSet &cur = sets[id % (depth + 1)];
        const auto tw = std::chrono::steady_clock::now();
        cur.wait(); // frame `id` of every live sequence is decoded
        out.decode_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
        int pushed = 0;
        for (int s = 0; s < S; s++) {
            if (!live[s])
                continue;
            if (!cur.ok[s]) {
                live[s] = 0; // the reference runs until imread fails (main.cpp:123)
                continue;
            }
            pushed++;
        }
        if (!pushed)
            break;
        // pageable host memory, all pairs of the step in ONE call: the library copies them to its pinned staging area on several
        // threads before it returns, and -- from 32 sequences on, when every sequence has a pair -- moves the step's block to
        // the GPU in one copy-engine transfer when the step runs (vo_hip.h, vo_seq_push_pair)
        std::vector<int32_t> ids;
        std::vector<const void *> lp, rp;
        for (int s = 0; s < S; s++)
            if (live[s]) {
                ids.push_back(s);
                lp.push_back(cur.left[s].px.data());
                rp.push_back(cur.right[s].px.data());
            }
        if ((rc = vo_seq_push_pairs(ctx, (int)ids.size(), ids.data(), lp.data(), rp.data(), w, /*kind: pageable*/ 0)) < 0)
            return fail("vo_seq_push_pairs", rc);
        if ((rc = vo_seq_step(ctx)) < 0)
            return fail("vo_seq_step", rc);
        if (id + depth + 1 < max_frames) // this slot is free again: the frame `depth + 1` ahead goes into it
            cur.decode(pool, dirs, live, id + depth + 1, w, h);
    }
    pool.wait();
    if ((rc = vo_seq_sync(ctx)) < 0)
        return fail("vo_seq_sync", rc);
    out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int s = 0; s < S; s++) {
        int n = 0;
        if ((rc = vo_seq_get_trajectory(ctx, s, 0, 0, nullptr, nullptr, &n)) < 0)
            return fail("vo_seq_get_trajectory", rc);
        out.rows[s].resize((size_t)(n > 0 ? n : 1) * VO_SEQ_ROW);
        out.info[s].resize((size_t)(n > 0 ? n : 1) * VO_SEQ_INFO);
        if ((rc = vo_seq_get_trajectory(ctx, s, 0, n, out.rows[s].data(), out.info[s].data(), &n)) < 0)
            return fail("vo_seq_get_trajectory", rc);
        out.rows[s].resize((size_t)n * VO_SEQ_ROW);
        out.info[s].resize((size_t)n * VO_SEQ_INFO);
        out.frames += n;
    }
    vo_destroy(ctx);
    return out;
}

// KITTI pose format (12 doubles per line, evaluate_odometry.cpp:24-27), first line = frame_pose of the first pair
inline bool write_trajectory(const std::string &path, const std::vector<double> &rows, const std::vector<int32_t> &info,
                             int *integrated)
{
    FILE *out = fopen(path.c_str(), "w");
    if (!out)
        return false;
    fprintf(out, "1.000000000e+00 0 0 0 0 1.000000000e+00 0 0 0 0 1.000000000e+00 0\n");
    const size_t n = rows.size() / VO_SEQ_ROW;
    int integ = 0;
    for (size_t k = 0; k < n; k++) {
        for (int j = 0; j < 12; j++)
            fprintf(out, "%.9e%c", rows[k * VO_SEQ_ROW + j], j == 11 ? '\n' : ' ');
        integ += (info[k * VO_SEQ_INFO + 5] & VO_SEQ_F_INTEGRATED) != 0;
    }
    fclose(out);
    if (integrated)
        *integrated = integ;
    return true;
}

} // namespace vohost
