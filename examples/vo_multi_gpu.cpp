// vo_multi_gpu.cpp -- BASELINE config 5 as a C++ host: independent stereo sequences sharded over the GPUs of one node,
// sequence s -> worker s % n, ONE host thread + ONE vo_ctx per worker, no collective of any kind (SURVEY.md section
// 8e: "replicas only").  Each worker is examples/vo_seq_host.h's loop -- the reference's main() frame loop
// (src/main.cpp:123-224) for its share of the sequences in lock step on its GPU; the process only adds up frame counts.
//
//   vo_multi_gpu --devices 0,1,2,3 [--decode-threads T] <fx> <cx> <cy> <bf> <max_frames> <features_per_bucket>
//                <out_prefix> <sequence_dir> [<sequence_dir> ...]
//   --devices  one entry per worker; an ordinal may repeat (two workers = two host threads + two contexts on the same
//              GPU -- how the single-GPU test exercises the threading: tests/test_gpu_round3.py)
//   output:    <out_prefix>_<s>.txt per sequence (KITTI pose format), and ONE JSON line on stdout:
//              {"workers": [{"device", "sequences", "frames", "seconds", "fps"}...], "frames", "seconds", "fps"}
//              seconds = the slowest worker's wall clock (read + decode + upload + compute), fps = all frames / that.
//
// build: g++ -O2 -std=c++17 vo_multi_gpu.cpp -I../include -L../visual_odom_amd -lvo_hip -lz -lpthread -Wl,-rpath,... -o vo_multi_gpu
#include "vo_seq_host.h"

#include <cstdlib>
#include <cstring>
#include <thread>

#include <pthread.h>
#include <sched.h>

int main(int argc, char **argv)
{
    std::vector<int> devices;
    int decode_threads = 4, a = 1;
    while (a + 1 < argc && !strncmp(argv[a], "--", 2)) {
        if (!strcmp(argv[a], "--devices")) {
            for (char *tok = strtok(argv[a + 1], ","); tok; tok = strtok(nullptr, ","))
                devices.push_back(atoi(tok));
        } else if (!strcmp(argv[a], "--decode-threads")) {
            decode_threads = atoi(argv[a + 1]);
        } else {
            break;
        }
        a += 2;
    }
    if (devices.empty() || argc - a < 8) {
        fprintf(stderr, "usage: %s --devices 0,1,... [--decode-threads T] <fx> <cx> <cy> <bf> <max_frames> "
                        "<features_per_bucket> <out_prefix> <sequence_dir> ...\n", argv[0]);
        return 1;
    }
    vohost::Calib cal;
    cal.fx = (float)atof(argv[a]);
    cal.cx = (float)atof(argv[a + 1]);
    cal.cy = (float)atof(argv[a + 2]);
    cal.bf = (float)atof(argv[a + 3]);
    const int max_frames = atoi(argv[a + 4]), per_bucket = atoi(argv[a + 5]);
    const std::string prefix = argv[a + 6];
    const std::vector<std::string> dirs(argv + a + 7, argv + argc);
    const int W = (int)devices.size(), S = (int)dirs.size();

    // sequence s -> worker s % W
    std::vector<std::vector<std::string>> share(W);
    std::vector<std::vector<int>> ids(W);
    for (int s = 0; s < S; s++) {
        share[s % W].push_back(dirs[s]);
        ids[s % W].push_back(s);
    }
    // One host per GPU: worker k (its submitting thread and the decoder threads it starts, which inherit the mask) keeps the k-th
    // contiguous slice of the cores this process may use -- what bench.py does per rank (visual_odom_amd/replicas.py,
    // plan_affinity); with more workers than cores the workers share cores round-robin.
    std::vector<int> avail;
    {
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0)
            for (int c = 0; c < CPU_SETSIZE; c++)
                if (CPU_ISSET(c, &set))
                    avail.push_back(c);
    }
    std::vector<int> slice_first(W, -1), slice_n(W, 0);
    std::vector<vohost::WorkerResult> res(W);
    std::vector<std::thread> threads;
    for (int k = 0; k < W; k++)
        threads.emplace_back([&, k] {
            if (W > 1 && !avail.empty()) {
                cpu_set_t mine;
                CPU_ZERO(&mine);
                const int per = (int)avail.size() / W;
                if (per >= 1) {
                    for (int c = k * per; c < (k + 1) * per; c++)
                        CPU_SET(avail[c], &mine);
                    slice_first[k] = avail[k * per];
                    slice_n[k] = per;
                } else {
                    CPU_SET(avail[k % avail.size()], &mine);
                    slice_first[k] = avail[k % avail.size()];
                    slice_n[k] = 1;
                }
                (void)pthread_setaffinity_np(pthread_self(), sizeof(mine), &mine);
            }
            res[k] = vohost::run_worker(devices[k], share[k], cal, max_frames, per_bucket, decode_threads);
        });
    for (auto &t : threads)
        t.join();

    long frames = 0;
    double slowest = 0;
    for (int k = 0; k < W; k++) {
        if (res[k].rc) {
            fprintf(stderr, "worker %d (device %d): %s\n", k, devices[k], res[k].error.c_str());
            return res[k].rc;
        }
        for (size_t j = 0; j < ids[k].size(); j++) {
            const std::string path = prefix + "_" + std::to_string(ids[k][j]) + ".txt";
            if (!vohost::write_trajectory(path, res[k].rows[j], res[k].info[j], nullptr))
                return 1;
        }
        frames += res[k].frames;
        slowest = res[k].seconds > slowest ? res[k].seconds : slowest;
    }
    printf("{\"workers\": [");
    for (int k = 0; k < W; k++)
        printf("%s{\"device\": %d, \"sequences\": %zu, \"frames\": %ld, \"seconds\": %.6f, \"fps\": %.3f, \"decode_wait_s\": %.6f, "
               "\"host_cores\": %d, \"first_core\": %d}",
               k ? ", " : "", devices[k], share[k].size(), res[k].frames, res[k].seconds,
               res[k].seconds > 0 ? res[k].frames / res[k].seconds : 0.0, res[k].decode_seconds, slice_n[k], slice_first[k]);
    printf("], \"frames\": %ld, \"seconds\": %.6f, \"fps\": %.3f, \"parallelism\": \"replicas x%d (one host thread + one vo_ctx "
           "per worker, no collective)\"}\n", frames, slowest, slowest > 0 ? frames / slowest : 0.0, W);
    return 0;
}
