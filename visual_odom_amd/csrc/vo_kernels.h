// vo_kernels.h -- parameter / result records and launch wrappers shared by the kernel
// translation units (pyramid.hip, lk.hip, post.hip, pnp.hip) and the C-ABI host code (capi.hip).
#pragma once

#include "vo_dev.h"

namespace vo {

struct LkParams {
    int max_level;   // 3 in the reference (feature.cpp:136) -> 4 pyramid levels
    int max_count;   // 30
    double epsilon;  // (0.01)^2 after OpenCV's sanitising
    float min_eig;   // 1e-3
    int full_chain;  // 1: run all four hops even after a hop the circular filter will reject
};

struct PnpParams {
    int iters;          // 500   (visualOdometry.cpp:168)
    float reproj;       // 0.5f  (visualOdometry.cpp:169)
    double confidence;  // (double)0.999f (visualOdometry.cpp:170)
    float K[9];         // intrinsic matrix, f32 row-major = projMatrl(0:3, 0:3)
};

// per-frame result record (all f64 like OpenCV's rvec / tvec / rotation)
struct PnpResult {
    double rvec[3], tvec[3], R[9];
    int n_inliers;
    int status;   // 1 ok, 0 RANSAC found no model, <0 bad input (fewer than 5 points)
    int niters;   // RANSAC iterations OpenCV would have executed
    int best_iter;
    int max_good;
    int lm_iters;
};

// progress of the replayed RANSAC loop of one frame (RANSACPointSetRegistrator::run)
struct RansacState {
    int it;        // hypotheses consumed so far (= iterations OpenCV would have executed at the end)
    int niters;    // current adaptive iteration bound
    int max_good;  // best inlier count
    int best;      // index of the hypothesis that achieved it (-1: none)
    uint64_t rng;  // POSITION in the raw cv::RNG(-1) stream after the subsets drawn so far (ransac_subsets_kernel)
    int arrive;    // ransac_rest_kernel: workgroups of this frame that have finished their hypotheses (0 between launches)
    int pad_;
};

// findEssentialMat(points0, points1, focal, pp, RANSAC, prob, threshold) + recoverPose (visualOdometry.cpp:152-153)
struct EmParams {
    double focal, ppx, ppy; // projMatrl(0,0), (0,2), (1,2) as doubles (visualOdometry.cpp:146-147)
    double prob;            // 0.999
    double threshold;       // 1.0 pixel
    int max_iters;          // 1000 (OpenCV 4.5 default of this overload)
};

struct EmResult {
    double E[9], R[9], t[3]; // essential matrix, recoverPose rotation / unit translation
    int status;              // 1 ok, 0 RANSAC found no model, -1 fewer than 5 points
    int n_inliers;           // RANSAC inliers of E
    int n_good;              // points passing the cheirality check (recoverPose's return value)
    int niters;              // samples OpenCV would have drawn
    int best;                // 10 * sample + model index of the winner
};

// one processed frame of one sequence of the lock-step loop (vo_seq_*, seq.hip)
struct SeqFrameInfo {
    int n_bucketed;   // points that entered circularMatching
    int n_circ;       // survivors of deleteUnmatchFeaturesCircle (length of `ages` afterwards)
    int n_tracked;    // survivors of the consistency filter (= currentVOFeatures.points.size() afterwards)
    int n_inliers;    // solvePnPRansac inliers
    int pnp_status;   // 1 ok, 0 no model, < 0 fewer than 5 points
    int flags;        // VO_SEQ_F_* (include/vo_hip.h)
    int ransac_iters; // RANSAC iterations OpenCV would have executed
    int overflow;     // detection / bucketing capacity exceeded in this frame (results truncated)
};
// what thread 0 of a sequence's select_refine_kernel workgroup needs to finish the frame in the lock-step loop
// (vo_seqtail.h); active == nullptr: no tail (batch mode, stand-alone calls, schedule-probe dry runs)
struct SeqTail {
    const int *active = nullptr;   // [S] this step's activity (bit 1: first frame after a pause)
    const EmResult *em = nullptr;  // mono_rotation: recoverPose results
    double *pose = nullptr;        // [S][16] frame_pose
    double *traj = nullptr;        // [S][max_steps][VO_SEQ_ROW]
    SeqFrameInfo *info = nullptr;  // [S][max_steps]
    int *n_rows = nullptr;         // [S]
    int max_steps = 0;
};
// one pushed stereo pair of a step (seq_ingest_kernel): source images + first image-table index of its ring slot
struct SeqIngest {
    const uint8_t *left, *right;
    int stride, image0;
};
#ifndef VO_SEQ_ROW // also in include/vo_hip.h (public)
#define VO_SEQ_ROW 27 // doubles per trajectory row: frame_pose 3x4, rvec, tvec, rotation 3x3
#define VO_SEQ_F_ACTIVE 1
#define VO_SEQ_F_INTEGRATED 2
#define VO_SEQ_F_TOO_FEW 4
#define VO_SEQ_F_NO_ESSENTIAL 8
#define VO_SEQ_F_GAP 16
#endif

// workspace of the four-kernel EPnP used for small launches (pnp.hip): VO_EPNP_WS_DOUBLES doubles per hypothesis of the first
// RANSAC chunk (VO_EPNP_WS_HYPS hypotheses per frame), allocated for up to VO_EPNP_WS_MAX_FRAMES frames
// slim pose chain (waves = 4): the 12 x 12 M^T M (+ 12 column norms) of every hypothesis of the FIRST RANSAC chunk in a global
// workspace, lane-interleaved per 64-hypothesis block: [frames][VO_EPNP_GWS_BLOCKS][VO_EPNP_UT_DOUBLES][64] doubles
constexpr int VO_EPNP_UT_DOUBLES = 144 + 12, VO_EPNP_GWS_BLOCKS = 2;
constexpr int VO_EPNP_WS_DOUBLES = 288, VO_EPNP_WS_HYPS = 128, VO_EPNP_WS_MAX_FRAMES = 16, VO_EPNP_SPLIT_DEFAULT_FRAMES = 4;
// workgroups per frame of ransac_rest_kernel (64 hypotheses each) and the doubles of its global workspace per frame
inline int pnp_rest_groups(int iters) { return iters > VO_EPNP_WS_HYPS ? (iters - VO_EPNP_WS_HYPS + 63) / 64 : 1; }
inline size_t pnp_rest_ws_doubles(int iters) { return (size_t)pnp_rest_groups(iters) * VO_EPNP_UT_DOUBLES * 64; }

#ifndef VO_HOST_EMUL
struct EmBufs {
    double2 *q0 = nullptr, *q1 = nullptr; // [B][cap] normalised points
    int32_t *subsets = nullptr;           // [B][max_iters][5]
    RansacState *rstate = nullptr;        // [B]
    double *models = nullptr;             // [B][128][10][9]
    int *nmodels = nullptr;               // [B][128]
    int *counts = nullptr;                // [B][128][10]
    double *bestE = nullptr;              // [B][9]
    uint8_t *mask = nullptr;              // [B][cap]
};

#ifdef VO_DEV_VARIANTS // the round-3 three-kernel pyramid chain (A/B partner of launch_pyramid_fused, VO_PYR_FUSED=0)
void launch_border_fill(const PyrImage *d_imgs, int n_images, int first_level, int n_levels, const int *lstride,
                        const int *lh, hipStream_t stream);
void launch_pyr_down(const PyrImage *d_imgs, int n_images, int level, int dw, int dh, hipStream_t stream);
void launch_scharr(const PyrImage *d_imgs, int n_images, int first_level, int n_levels, const int *lw, const int *lh,
                   hipStream_t stream);
#endif
void launch_pyramid_fused(const PyrImage *d_imgs, int n_images, int n_levels, const int *lw, const int *lh, const int *lstride,
                          hipStream_t stream);
// pyramid.hip: a staged host image over PCIe by a kernel (+ optionally n_pts float2 and their count from pinned memory)
void launch_pull_image(const void *src_pinned_dev, void *dst, size_t bytes, hipStream_t stream, const void *pts_pinned_dev = nullptr,
                       void *pts_dst = nullptr, int n_pts = 0, int *count_dst = nullptr);
void launch_lk_circular(const PyrImage *d_imgs, const Quad *d_quads, const float2 *d_pts, const int *d_npts,
                        int cap, int max_pts, int n_frames, float2 *d_trk, uint8_t *d_status,
                        const LkParams &prm, hipStream_t stream);
// hops hop_begin .. hop_end - 1 of the chain (lk_hops_kernel; [0, 1) + [1, 4) == launch_lk_circular bit for bit)
void launch_lk_hops(const PyrImage *d_imgs, const Quad *d_quads, const float2 *d_pts, const int *d_npts, int cap, int max_pts,
                    int n_frames, float2 *d_trk, uint8_t *d_status, const LkParams &prm, int hop_begin, int hop_end,
                    hipStream_t stream);
#ifdef VO_DEV_VARIANTS
void launch_lk_circular_pair(const PyrImage *d_imgs, const Quad *d_quads, const float2 *d_pts, const int *d_npts,
                        int cap, int max_pts, int n_frames, float2 *d_trk, uint8_t *d_status,
                        const LkParams &prm, hipStream_t stream);
#endif
void launch_detect_bucket(const PyrImage *d_imgs, const Quad *d_quads, const int *d_detect, int n_frames, int w,
                          int h, int threshold, int nonmax, unsigned long long *d_nmsmask,
                          int *d_rowcnt, int *d_rowoff,
                          const int *d_ntracked, int *d_nnew, int cap, float2 *d_feat, const int *d_ages,
                          int bucket_size, int fpb, float2 *d_out_pts, int *d_out_ages, int *d_out_n, int out_cap,
                          const int *d_active, int *d_overflow, hipStream_t stream);
void launch_fast_corners(const PyrImage *d_imgs, const Quad *d_quads, const int *d_detect, int n_frames, int w, int h,
                         int threshold, int nonmax, unsigned long long *d_nmsmask, int *d_rowcnt, int *d_rowoff,
                         const int *d_ntracked, int *d_nnew, int cap, float2 *d_out, hipStream_t stream);
// grids the device bucketing takes: <= 8 per bucket; <= 1 024 buckets, or <= 4 096 with buckets x per-bucket <= 8 192 (fast.hip)
bool bucket_grid_ok(int w, int h, int bucket_size, int fpb);
void launch_bucket(const float2 *d_feat, const float2 *d_corners, const int *d_ages, const int *d_ntracked,
                   const int *d_nnew, int cap, int w, int h, int bucket_size, int fpb, float2 *d_out_pts, int *d_out_ages,
                   int *d_out_n, int out_cap, const int *d_active, int *d_overflow, int n_frames, hipStream_t stream);
void launch_seq_ingest(const SeqIngest *tab, int n_pairs, int w, int h, int pitch, uint8_t *pix0, size_t img_bytes,
                       bool over_pcie, hipStream_t stream);
void launch_seq_prepare(const int *active, const int *n_tracked, int redetect_below, int *detect,
                        const int *n_corners, int *n_new, int n_seq, hipStream_t stream);
void launch_seq_carry(const int *active, const float2 *outB, const int *nB, const int *idxA, const int *nA,
                      const int *ages, const int *n_bucketed, int cap, int fcap, float2 *feat, int *fages,
                      int *n_tracked, const int *overflow, int *n_rows_carry, int *n_ages, SeqFrameInfo *info,
                      int max_steps, int n_seq, hipStream_t stream);
// Everything vo_track_frame returns for its one frame, gathered by one kernel into one host-visible buffer (layout: a
// 512-byte header -- nA, nB, has_em at bytes 0 / 4 / 8, the PnpResult at byte 16, the EmResult at byte 256 -- then fixed
// capacity arrays l0, r0, l1, r1 [cap] float2, xyz [cap][3] float, keep_idx, keep_idx_circ, inliers [cap] int32).
struct FrameGather {
    const int *nA, *nB;
    const float2 *outB; // [4][cap]
    const float *xyz;
    const int32_t *idxB, *idxA, *inliers;
    const PnpResult *result;
    const EmResult *em; // null unless mono_rotation
    int cap;
    int pose_only = 0; // no point arrays, only the PnpResult and its inliers (vo_pnp_ransac)
};
constexpr size_t VO_GATHER_HEADER = 512;
inline size_t frame_gather_bytes(int cap) { return VO_GATHER_HEADER + (size_t)cap * (4 * 8 + 12 + 3 * 4); }
void launch_frame_gather(const FrameGather &g, uint8_t *out, hipStream_t stream);
// vo_circular_match's results in one host-visible buffer (layout: count at byte 0; from byte 16 five rows l0, r0, r1, l1, l0_ret
// of [cap] float2, keep_idx [cap] int32, the raw LK status [4][cap] bytes): stage A (deleteUnmatchFeaturesCircle) or, with
// `consistency`, stage B (+ checkValidMatch / removeInvalidPoints) with l0_ret picked out of the raw tracks by index
struct CircGather {
    const float2 *outA; // [5][cap]
    const int32_t *idxA;
    const int *nA;
    const float2 *outB; // [4][cap] l0, r0, l1, r1
    const int32_t *idxB;
    const int *nB;
    const float2 *trk;     // [4][cap] raw tracks (row 3 = l0_ret)
    const uint8_t *status; // [4][cap]
    int n, cap, consistency;
};
inline size_t circ_gather_bytes(int cap) { return 16 + (size_t)cap * (5 * 8 + 4 + 4); }
void launch_circ_gather(const CircGather &g, uint8_t *out, hipStream_t stream);
void launch_words_in(const void *src, int n0, void *dst0, int n1, void *dst1, int *count_dst, int count, hipStream_t stream);
void launch_words_out(const void *src, int n, void *dst, hipStream_t stream);
// vo_detect_bucket's feature set in and out through page-locked host memory (post.hip)
inline size_t features_stage_bytes(int fcap) { return 16 + (size_t)fcap * (sizeof(float2) + sizeof(int)); } // (in: fcap points + fcap ages; out: vo_fast_detect's corners, features_out_kernel's layout)
inline size_t features_out_bytes(int cap) { return 16 + (size_t)cap * (sizeof(float2) + sizeof(int)); }
void launch_features_in(const uint8_t *src, size_t ages_off, int n_pts, int n_ages, int detect, float2 *feat, int *fages, int fcap,
                        int *n_tracked, int *detect_flag, hipStream_t stream);
void launch_features_out(const float2 *pts, const int *ages, const int *n, const int *overflow, int cap, uint8_t *out,
                         hipStream_t stream);
void launch_compact(const float2 *pts_in, const float2 *trk, const uint8_t *status, const int *n_pts, int cap,
                    int threshold, float2 *outA, int *idxA, int *nA, float2 *outB, int *idxB, int *nB,
                    int n_frames, hipStream_t stream);
void launch_triangulate(const float *Pl, const float *Pr, const float2 *pl, const float2 *pr,
                        size_t frame_stride, const int *n_pts, int cap, int max_pts, int n_frames, float *xyz,
                        hipStream_t stream);
void launch_pnp(const float *xyz, const float2 *uv, size_t uv_stride, const int *n_pts, int cap, int n_frames,
                const PnpParams &prm, int32_t *subsets, double *models, int *counts, RansacState *state,
                int32_t *inliers, PnpResult *results, int waves, hipStream_t stream, double *epnp_ws = nullptr,
                int ws_frames = 0, double *gws = nullptr, double *rest_ws = nullptr);
// epnp_ws: workspace of the four-kernel EPnP used for small launches (pnp.hip; constants above); null = always the one-kernel form
void launch_pnp_ransac(const float *xyz, const float2 *uv, size_t uv_stride, const int *n_pts, int cap, int n_frames,
                       const PnpParams &prm, int32_t *subsets, double *models, int *counts, RansacState *state, int waves,
                       hipStream_t stream, double *epnp_ws, int ws_frames, double *gws, int wide_frames, double *rest_ws);
void launch_pnp_refine(const float *xyz, const float2 *uv, size_t uv_stride, const int *n_pts, int cap, int n_frames,
                       const PnpParams &prm, const double *models, const RansacState *state, int32_t *inliers,
                       PnpResult *results, int waves, const SeqTail &tail, hipStream_t stream);
// subsets of hypotheses [h0, h0 + hn) of every frame (cv::RNG(-1) stream, continued per frame)
void launch_ransac_subsets(const int *n_pts, int n_frames, int iters, int h0, int hn, int32_t *subsets,
                           RansacState *rstate, hipStream_t stream);
// per-device tables of the pose solve (the raw RNG stream); called by vo_create on the context's device
int pnp_init_device(hipStream_t stream);
void launch_essential(const float2 *p0, const float2 *p1, size_t stride, const int *n_pts, int cap, int n_frames,
                      const EmParams &prm, const EmBufs &eb, EmResult *results, bool crowded, hipStream_t stream);

#endif // VO_HOST_EMUL

} // namespace vo
