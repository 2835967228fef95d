// capi_dropin.hip -- the synchronous drop-in calls (host buffers in, host buffers out): vo_circular_match, vo_triangulate,
// vo_pnp_ransac, vo_fast_detect, vo_detect_bucket, vo_integrate_odometry, vo_track_frame.
#include "capi_internal.h"
#ifdef VO_DEV_VARIANTS
#include <chrono>
#endif

namespace vo_capi {

#ifdef VO_DEV_VARIANTS
// developer build: host-side time stamps of the last vo_track_frame (ns, steady clock): [0] entry, [1] configure + sync_all
// done, [2..5] image k staged and its copy enqueued, [6] points enqueued, [7] run_stages returned (everything enqueued),
// [8] the final stream synchronisation returned, [9] results copied out (tools/host_gap_probe.py)
long long g_host_stamp[16];
#define VO_HOST_STAMP(k) (g_host_stamp[k] = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count())
#else
#define VO_HOST_STAMP(k) ((void)0)
#endif

/* ---------------------------------- drop-in calls ---------------------------------------- */

// The inputs of a single-frame drop-in call on their way to the device WITHOUT a synchronisation in between (round 5): one
// sync_all up front (idle streams: a few microseconds; after it every staging buffer is free and nothing reads the points),
// then four kernels that pull the images out of the pinned staging slots over PCIe (pyramid.hip, launch_pull_image), the last of
// them the points and their count too, all queued on the tracking stream -- the caller's run_stages queues its kernels behind
// them while the pixels are still crossing PCIe.  Round 4 went through vo_batch_set_points here: sync_all + a pageable copy +
// a stream synchronisation, i.e. the host waited for the four uploads (~80 us for KITTI) before it launched anything (108 +
// 21 us of the call, profiles/r04_experiments.md section 7).
int single_frame_setup(vo_ctx *c, const uint8_t *l0, const uint8_t *r0, const uint8_t *l1,
                              const uint8_t *r1, int w, int h, int stride, const float *pts, int n, bool defer_t1)
{
    // no t0 images: the t0 pair is the pair the previous call received as t1 (the reference's loop keeps it the same way,
    // main.cpp:157-158) -- it is on the device with its pyramids, two images cross the link instead of four
    const bool keep = !l0 && !r0;
    if ((!keep && (!l0 || !r0)) || !l1 || !r1 || n < 0 || (n > 0 && !pts))
        return fail(c, VO_ERR_ARG, "null image / points");
    if (n > c->cap)
        return fail(c, VO_ERR_ARG, "more points than max_pts given to vo_create");
    if (stride < w) // (before anything changes: a refused call leaves the kept pair as it is)
        return fail(c, VO_ERR_ARG, "stride smaller than the width");
    if (keep && (c->tf_base < 0 || c->seq.on || c->n_images != 4 || c->n_frames != 1 || c->w != w || c->h != h ||
                 c->img_stale[c->tf_base] || c->img_stale[c->tf_base + 1])) // (stale: the call that uploaded the pair failed before its pyramids were built)
        return fail(c, VO_ERR_STATE, "no t0 images given, and the context does not hold the t1 pair of a previous call of this "
                                     "size (first call, another size, or the batch / sequence API used the images since)");
    int rc = vo_batch_configure(c, 4, w, h, 1);
    if (rc != VO_OK)
        return rc;
    rc = sync_all(c); // a queued run of the batch API may still read the points / write the staging slots
    if (rc != VO_OK)
        return rc;
    static_assert(VO_STAGE_SLOTS >= 4, "one staging slot per image of the call");
    VO_HOST_STAMP(1);
    c->stage_next = 0;
    c->defer.n = 0;
    const int t0 = keep ? c->tf_base : 0, t1 = t0 ^ 2; // image slots of the two pairs
    const uint8_t *imgs[4] = {l0, r0, l1, r1};
    use_const_quad(c, t0 == 0 ? 0 : 1);
    c->tf_base = -1; // (until every image of the call is on its way)
    // one frame: the call's points replace the whole current set, also the bucketed set a VO_STAGE_DETECT run left
    // current (nothing reads that one any more: sync_all above)
    c->pts_sel = -1;
    // Round 6, on the kept pair: hop 0 of the chain (l0 -> r0) reads the t0 pair only, which is on the device with its
    // pyramids -- so only the points go now (a pull of their own) and the t1 pair is DEFERRED to run_stages, which launches
    // hop 0 first and sends the pair on the filter stream beside it (vo_ctx::defer): 0.64 -> 0.61 ms per call at 2 039 points,
    // 0.60 -> 0.58 at 340 (gpurun_out/r6_split1).  Not with four images: two chain launches end on their slowest feature twice
    // (hop 0 alone 74 - 82 us, hops 1 .. 3 172 - 190, the whole chain 200 - 236) and the t0 pyramids become a launch set of their
    // own (27 us) -- measured 0.68 -> 0.70 ms, so four images go the way they always went.
    bool split = defer_t1 && keep;
#ifdef VO_DEV_VARIANTS
    static const int split_env = [] { const char *e = getenv("VO_SYNC_SPLIT"); return e ? atoi(e) : 1; }();
    split = split && split_env != 0 && !c->lk_pair;
#endif
    for (int i = keep ? 2 : 0; i < (split ? 2 : 4); i++) { // the points travel with the last image
        rc = upload_image(c, (i < 2 ? t0 : t1) + (i & 1), imgs[i], stride, hipMemcpyHostToDevice, /*idle*/ true, pts, i == 3 ? n : -1);
        if (rc != VO_OK)
            return rc;
        VO_HOST_STAMP(2 + i);
    }
    if (split) {
        if (n > 0)
            memcpy(c->h_pts_stage, pts, sizeof(float2) * (size_t)n);
        launch_pull_image(nullptr, nullptr, 0, c->stream, c->d_pts_stage, c->d_pts, n, c->d_npts);
        VO_HIP_TRY(c, hipGetLastError());
        c->defer.img[0] = l1;
        c->defer.img[1] = r1;
        c->defer.first = t1;
        c->defer.stride = stride;
        c->defer.n = 2;
        c->img_stale[t1] = c->img_stale[t1 + 1] = 1; // (what is in those slots is not this call's pair until run_stages has sent it)
    }
    if (keep) {
        c->pyr_first = t1; // (vo_batch_configure above restored "every pyramid")
        c->pyr_count = 2;
    }
    c->tf_base = t1;
    c->tf_gen++; // a new kept pair (vo_kept_pair_id)
    c->h_npts[0] = n;
    c->pts_on_device = false;
    c->max_pts_set = n;
    VO_HOST_STAMP(6);
    return VO_OK;
}

} // namespace vo_capi

extern "C" {

int vo_circular_match(vo_ctx *c, const uint8_t *l0, const uint8_t *r0, const uint8_t *l1, const uint8_t *r1,
                      int w, int h, int stride, const float *pts, int n, float *out_l0, float *out_r0,
                      float *out_r1, float *out_l1, float *out_l0_ret, uint8_t *status4, int32_t *keep_idx,
                      int *n_out, int apply_consistency)
{
    if (!c || !n_out)
        return VO_ERR_ARG;
    DeferGuard guard{c};
    int rc = single_frame_setup(c, l0, r0, l1, r1, w, h, stride, pts, n, /*defer_t1*/ true);
    if (rc != VO_OK)
        return rc;
    // a synchronous call: every stage on the tracking stream (run_stages, `serial`), then ONE kernel that gathers whatever
    // the caller asked for into the host-visible result buffer and one synchronisation (round 5: the batch getters cost this
    // call ten copies and five synchronisations, ~0.17 of its 0.57 ms)
    rc = run_stages_auto(c, VO_STAGE_PYRAMID | VO_STAGE_LK | VO_STAGE_FILTER, false, nullptr, /*sync_call*/ true);
    if (rc != VO_OK)
        return rc;
    const vo_ctx::PoseBufs &pb = c->pb[c->last];
    CircGather g;
    g.outA = c->d_outA;
    g.idxA = c->d_idxA;
    g.nA = c->d_nA;
    g.outB = pb.outB;
    g.idxB = pb.idxB;
    g.nB = pb.nB;
    g.trk = c->d_trk2[c->trk_last];
    g.status = c->d_status2[c->trk_last];
    g.n = status4 ? n : 0;
    g.cap = c->cap;
    g.consistency = apply_consistency ? 1 : 0;
    if (!c->last_run_serial) { // (developer build, VO_SYNC_SERIAL=0: the filter ran on its own stream and this call has no
                               // pose stage whose event would order the gather behind it -- ADVICE r05)
        rc = sync_all(c);
        if (rc != VO_OK)
            return rc;
    }
    launch_circ_gather(g, c->d_gather, c->stream); // (circ_gather_bytes(cap) <= frame_gather_bytes(cap))
    VO_HIP_TRY(c, hipGetLastError());
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    const uint8_t *hb = c->h_gather;
    int count = 0;
    memcpy(&count, hb, sizeof(int));
    const size_t cap = (size_t)c->cap;
    float *outs[5] = {out_l0, out_r0, out_r1, out_l1, out_l0_ret};
    for (int k = 0; k < 5; k++)
        if (outs[k] && count > 0)
            memcpy(outs[k], hb + 16 + (size_t)k * cap * 8, (size_t)count * 8);
    if (keep_idx && count > 0)
        memcpy(keep_idx, hb + 16 + 5 * cap * 8, (size_t)count * 4);
    if (status4 && n > 0)
        for (int hop = 0; hop < 4; hop++) // [4][n] for the caller
            memcpy(status4 + (size_t)hop * n, hb + 16 + 5 * cap * 8 + cap * 4 + (size_t)hop * cap, (size_t)n);
    *n_out = count;
    return VO_OK;
}

int vo_triangulate(vo_ctx *c, const float *P_l, const float *P_r, const float *pl, const float *pr, int n,
                   float *xyz_out)
{
    if (!c || !P_l || !P_r || n < 0 || (n > 0 && (!pl || !pr || !xyz_out)))
        return VO_ERR_ARG;
    if (n > c->cap)
        return fail(c, VO_ERR_ARG, "more points than max_pts given to vo_create");
    if (n == 0)
        return VO_OK;
    int rc = vo_batch_set_projection(c, P_l, P_r);
    if (rc != VO_OK)
        return rc;
    rc = sync_all(c);
    if (rc != VO_OK)
        return rc;
    // frame 0, stage-B rows 0 (left) and 1 (right); the points in and the result out through page-locked memory the kernels
    // address (words_in_kernel / words_out_kernel, post.hip): one synchronisation, no copy call
    vo_ctx::PoseBufs &pb = c->pb[c->last];
    memcpy(c->h_feat_stage, pl, sizeof(float2) * (size_t)n);
    memcpy(c->h_feat_stage + sizeof(float2) * (size_t)n, pr, sizeof(float2) * (size_t)n);
    launch_words_in(c->d_feat_stage, 2 * n, pb.outB, 2 * n, pb.outB + c->cap, pb.nB, n, c->stream);
    launch_triangulate(c->d_P, c->d_P + 12, pb.outB, pb.outB + c->cap, (size_t)4 * c->cap, pb.nB, c->cap, n, 1,
                       pb.xyz, c->stream);
    launch_words_out(pb.xyz, 3 * n, c->d_gather, c->stream);
    VO_HIP_TRY(c, hipGetLastError());
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    memcpy(xyz_out, c->h_gather, sizeof(float) * 3 * (size_t)n);
    return VO_OK;
}

int vo_pnp_ransac(vo_ctx *c, const float *xyz, const float *uv, int n, const float *K, double *rvec_io,
                  double *tvec_io, double *R_out, int32_t *inliers, int *n_inliers)
{
    if (!c || !K || n < 0 || (n > 0 && (!xyz || !uv)))
        return VO_ERR_ARG;
    if (n > c->cap)
        return fail(c, VO_ERR_ARG, "more points than max_pts given to vo_create");
    int rcs = sync_all(c);
    if (rcs != VO_OK)
        return rcs;
    vo_ctx::PoseBufs &pb = c->pb[c->last];
    PnpParams pp;
    pp.iters = c->prm.ransac_iterations;
    pp.reproj = c->prm.ransac_reproj_error;
    pp.confidence = c->prm.ransac_confidence;
    memcpy(pp.K, K, sizeof(pp.K));
    // the correspondences in and the pose out through page-locked memory the kernels address (the staging buffer holds
    // 12 bytes x 4 max_pts or more, a correspondence is 20): one synchronisation per call, no copy call
    if (n > 0) {
        memcpy(c->h_feat_stage, xyz, sizeof(float) * 3 * (size_t)n);
        memcpy(c->h_feat_stage + sizeof(float) * 3 * (size_t)n, uv, sizeof(float2) * (size_t)n);
    }
    launch_words_in(c->d_feat_stage, 3 * n, pb.xyz, 2 * n, pb.outB + 2 * (size_t)c->cap, pb.nB, n, c->stream);
    if (c->n_frames < 1)
        c->n_frames = 1;
    launch_pnp(pb.xyz, pb.outB + 2 * (size_t)c->cap, (size_t)4 * c->cap, pb.nB, c->cap, 1, pp, pb.subsets,
               pb.models, pb.counts, pb.rstate, pb.inliers, pb.results, standalone_waves(c), c->stream, pb.epnp_ws, 1, pb.epnp_gws, pb.rest_ws);
    FrameGather g = {};
    g.nA = g.nB = pb.nB;
    g.inliers = pb.inliers;
    g.result = pb.results;
    g.cap = c->cap;
    g.pose_only = 1;
    launch_frame_gather(g, c->d_gather, c->stream);
    VO_HIP_TRY(c, hipGetLastError());
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    PnpResult r;
    memcpy(&r, c->h_gather + 16, sizeof(r));
    // by the rules of get_pose_impl with pnp_rotation: R = Rodrigues(rvec) whatever vo_params.mono_rotation says
    if (r.status == 0 && r.lm_iters < 0) { // four points, P3P without a solution: rvec / tvec stay the caller's
        if (R_out && rvec_io)
            rodrigues_v2m(rvec_io, R_out, nullptr);
    } else if (r.status >= 0) {
        if (rvec_io)
            memcpy(rvec_io, r.rvec, sizeof(r.rvec));
        if (tvec_io)
            memcpy(tvec_io, r.tvec, sizeof(r.tvec));
        if (R_out)
            memcpy(R_out, r.R, sizeof(r.R));
    }
    const int ninl = r.n_inliers < c->cap ? r.n_inliers : c->cap;
    if (inliers && ninl > 0)
        memcpy(inliers, c->h_gather + VO_GATHER_HEADER + (size_t)c->cap * (4 * 8 + 12 + 2 * 4), sizeof(int32_t) * (size_t)ninl);
    if (n_inliers)
        *n_inliers = r.n_inliers;
    if (r.status < 0)
        return fail(c, VO_ERR_TOO_FEW, "fewer than 4 correspondences reached solvePnPRansac (CV_Assert(npoints >= 4))");
    return r.status == 1 ? VO_OK : VO_NO_MODEL;
}

} // extern "C"

namespace vo_capi {

// one image as a 1-frame batch whose quad points at image 0 four times
// img == nullptr: the LEFT image of the t1 pair of the previous vo_track_frame / vo_circular_match (what the reference's
// loop detects on next, visualOdometry.cpp:95-108 on imageLeft_t0 = the previous imageLeft_t1) -- nothing is uploaded.
// An image that is given goes to the slot pair that does not hold that t1 pair, which stays valid.
int single_image_setup(vo_ctx *c, const uint8_t *img, int w, int h, int stride)
{
    if (img && stride < w)
        return fail(c, VO_ERR_ARG, "stride smaller than the width");
    if (!img && (c->tf_base < 0 || c->seq.on || c->n_images != 4 || c->n_frames != 1 || c->w != w || c->h != h ||
                 c->img_stale[c->tf_base] || c->img_stale[c->tf_base + 1]))
        return fail(c, VO_ERR_STATE, "no image given, and the context does not hold the t1 pair of a previous vo_track_frame "
                                     "of this size");
    int rc = vo_batch_configure(c, 4, w, h, 1);
    if (rc != VO_OK)
        return rc;
    rc = sync_all(c); // a queued run of the batch API may still read the feature lists / the staging slots
    if (rc != VO_OK)
        return rc;
    if (!img) {
        use_const_quad(c, c->tf_base == 0 ? 2 : 3);
        return VO_OK;
    }
    const int slot = c->tf_base == 0 ? 2 : 0;
    const int keep = c->tf_base; // (upload_image itself does not touch it; the batch API's wrappers do)
    c->stage_next = 0;           // (drained above: every staging slot is free, the GPU pulls the image itself)
    rc = upload_image(c, slot, img, stride, hipMemcpyHostToDevice, /*idle*/ true);
    c->tf_base = rc == VO_OK ? keep : -1;
    if (rc != VO_OK)
        return rc;
    use_const_quad(c, slot == 0 ? 2 : 3);
    return VO_OK;
}

} // namespace vo_capi

extern "C" {

int vo_fast_detect(vo_ctx *c, const uint8_t *img, int w, int h, int stride, int threshold, int nonmax,
                   float *pts_out, int cap, int *n_out)
{
    if (!c || !n_out || cap < 0 || (cap > 0 && !pts_out))
        return VO_ERR_ARG;
    if (w > 4096)
        return fail(c, VO_ERR_ARG, "vo_fast_detect: images up to 4096 pixels wide");
    int rc = single_image_setup(c, img, w, h, stride);
    if (rc != VO_OK)
        return rc;
    // (no copy call: the flags by features_in_kernel, the corners and their count back through page-locked memory, one
    // synchronisation -- see vo_detect_bucket)
    launch_features_in(c->d_feat_stage, 0, 0, 0, /*detect*/ 1, c->d_feat, c->d_fages, c->fcap, c->d_ntracked, c->d_detect, c->stream);
    c->h_ntracked[0] = 0;
    c->h_detect[0] = 1;
    c->detect_uploaded = true;
    threshold = threshold < 0 ? 0 : threshold > 255 ? 255 : threshold;
    launch_detect_bucket(c->d_imgs, c->quads_cur, c->d_detect, 1, w, h, threshold, nonmax, c->d_nmsmask, c->d_rowcnt, c->d_rowoff,
                         c->d_ntracked, c->d_nnew, c->fcap, c->d_feat, c->d_fages, /*bucket_size*/ 0, 1, nullptr,
                         nullptr, nullptr, 0, nullptr, nullptr, c->stream);
    launch_features_out(c->d_feat, c->d_fages, c->d_nnew, c->d_overflow, c->fcap, c->d_feat_stage, c->stream);
    VO_HIP_TRY(c, hipGetLastError());
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int n = reinterpret_cast<const int *>(c->h_feat_stage)[0];
    int k = n < cap ? n : cap;
    k = k < c->fcap ? k : c->fcap;
    if (k > 0)
        memcpy(pts_out, c->h_feat_stage + 16, sizeof(float2) * (size_t)k);
    *n_out = n;
    if (n > c->fcap && cap > c->fcap) // the caller's buffer would have held them, the context's corner list does not
        return fail(c, VO_ERR_OVERFLOW, "vo_fast_detect: more corners than the context's corner-list capacity "
                                        "(max(4 x max_pts, 16384, max_w x max_h / 16)): only that many were written");
    return VO_OK;
}

int vo_detect_bucket(vo_ctx *c, const uint8_t *img, int w, int h, int stride, const vo_detect_params *dp,
                     float *pts_io, int *n_pts, int32_t *ages_io, int *n_ages, int cap)
{
    if (!c || !n_pts || !n_ages || !pts_io || !ages_io || cap < 1 || *n_pts > cap || *n_ages > cap) // (the arrays hold cap entries)
        return VO_ERR_ARG;
    const int np = *n_pts, na = *n_ages;
    if (np < 0 || na < np || na > c->fcap)
        return fail(c, VO_ERR_ARG, "vo_detect_bucket: bad counts (need 0 <= n_pts <= n_ages <= capacity)");
    int rc = single_image_setup(c, img, w, h, stride);
    if (rc != VO_OK)
        return rc;
    const vo_detect_params saved = c->dprm;
    rc = vo_batch_set_detect_params(c, dp);
    if (rc != VO_OK)
        return rc;
    // The carried set in and the bucketed set out through page-locked memory the kernels address themselves, ONE
    // synchronisation per call (round 5; the batch API's setters and getters cost this call seven copies and five
    // synchronisations, 0.15 of its 0.18 ms on an idle GPU).  single_image_setup has drained the context.
    if (np > 0)
        memcpy(c->h_feat_stage, pts_io, sizeof(float2) * (size_t)np);
    const size_t ages_off = sizeof(float2) * (size_t)c->fcap;
    if (na > 0)
        memcpy(c->h_feat_stage + ages_off, ages_io, sizeof(int32_t) * (size_t)na);
    const int detect = np < c->dprm.redetect_below ? 1 : 0; // appendNewFeatures only then (visualOdometry.cpp:95)
    launch_features_in(c->d_feat_stage, ages_off, np, na, detect, c->d_feat, c->d_fages, c->fcap, c->d_ntracked, c->d_detect,
                       c->stream);
    c->h_ntracked[0] = np;
    c->h_detect[0] = detect; // (what run_stages would upload: it finds the flag on the device already)
    c->detect_uploaded = true;
    rc = run_stages(c, VO_STAGE_DETECT, false);
    c->dprm = saved;
    if (rc != VO_OK)
        return rc;
    launch_features_out(cur_pts(c), cur_ages(c), cur_npts(c), c->d_overflow, c->cap, c->d_gather, c->stream);
    VO_HIP_TRY(c, hipGetLastError());
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int k = reinterpret_cast<const int *>(c->h_gather)[0], ovf = reinterpret_cast<const int *>(c->h_gather)[1];
    if (k > cap)
        return fail(c, VO_ERR_ARG, "vo_detect_bucket: bucketed set larger than the caller's capacity");
    const int kc = k < c->cap ? k : c->cap;
    memcpy(pts_io, c->h_gather + 16, sizeof(float2) * (size_t)kc);
    memcpy(ages_io, c->h_gather + 16 + sizeof(float2) * (size_t)c->cap, sizeof(int32_t) * (size_t)kc);
    *n_pts = k;
    *n_ages = k;
    if (ovf)
        return fail(c, VO_ERR_OVERFLOW, ovf & 1 ? "VO_STAGE_DETECT: carried + detected features exceed the feature-list "
                                                  "capacity (4 x max_pts, >= 16384, >= w * h / 16): bucketed set truncated"
                                                : "VO_STAGE_DETECT: the bucketed set exceeds max_pts");
    return VO_OK;
}

int vo_integrate_odometry(double *pose, const double *R, const double *t, float *euler_out)
{
    if (!pose || !R || !t)
        return VO_ERR_ARG;
    return integrate_odometry(pose, R, t, euler_out); // vo_integrate.h: the code the sequence loop runs on the device
}

int vo_track_frame(vo_ctx *c, const uint8_t *l0, const uint8_t *r0, const uint8_t *l1, const uint8_t *r1, int w,
                   int h, int stride, const float *pts, int n, const float *P_l, const float *P_r,
                   float *out_l0, float *out_r0, float *out_l1, float *out_r1, float *xyz_out,
                   int32_t *keep_idx, int *n_out, int32_t *keep_idx_circ, int *n_circ, double *rvec_io,
                   double *tvec_io, double *R_out, int32_t *inliers, int *n_inliers)
{
    if (!c || !P_l || !P_r)
        return VO_ERR_ARG;
    VO_HOST_STAMP(0);
    DeferGuard guard{c};
    int rc = single_frame_setup(c, l0, r0, l1, r1, w, h, stride, pts, n, /*defer_t1*/ true);
    if (rc != VO_OK)
        return rc;
    rc = vo_batch_set_projection(c, P_l, P_r);
    if (rc != VO_OK)
        return rc;
    rc = run_stages_auto(c, VO_STAGE_ALL, false, nullptr, /*sync_call*/ true);
    if (rc != VO_OK)
        return rc;
    VO_HOST_STAMP(7);
    // Results: one kernel behind the pose solve gathers the counts, the PnpResult and every output array into one
    // host-visible buffer, one synchronisation, host copies from there -- instead of eleven device-to-host copies and
    // four rounds of stream synchronisation through vo_batch_get_filtered + vo_batch_get_pose (0.25 of the call's 1.45 ms).
    vo_ctx::PoseBufs &pb = c->pb[c->last];
    const bool mono = c->prm.mono_rotation && c->em_ready;
    FrameGather g;
    g.nA = c->d_nA;
    g.nB = pb.nB;
    g.outB = pb.outB;
    g.xyz = pb.xyz;
    g.idxB = pb.idxB;
    g.idxA = c->d_idxA;
    g.inliers = pb.inliers;
    g.result = pb.results;
    g.em = mono ? pb.em_results : nullptr;
    g.cap = c->cap;
    if (pb.pending) { // (the chain ran on other streams: `done` covers the filter and both pose chains)
        VO_HIP_TRY(c, hipStreamWaitEvent(c->stream, pb.done, 0));
        pb.pending = false;
    }
    launch_frame_gather(g, c->d_gather, c->stream);
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    VO_HOST_STAMP(8);
    const uint8_t *hb = c->h_gather;
    int hdr[3];
    memcpy(hdr, hb, sizeof(hdr));
    const int M = hdr[0], K = hdr[1];
    PnpResult r;
    memcpy(&r, hb + 16, sizeof(r));
    const size_t cap = (size_t)c->cap;
    const uint8_t *arr = hb + VO_GATHER_HEADER;
    float *outs[4] = {out_l0, out_r0, out_l1, out_r1};
    for (int k = 0; k < 4; k++)
        if (outs[k] && K > 0)
            memcpy(outs[k], arr + (size_t)k * cap * 8, (size_t)K * 8);
    const uint8_t *ax = arr + 4 * cap * 8, *ak = ax + cap * 12, *ac = ak + cap * 4, *ai = ac + cap * 4;
    if (xyz_out && K > 0)
        memcpy(xyz_out, ax, (size_t)K * 12);
    if (keep_idx && K > 0)
        memcpy(keep_idx, ak, (size_t)K * 4);
    if (keep_idx_circ && M > 0)
        memcpy(keep_idx_circ, ac, (size_t)M * 4);
    if (n_out)
        *n_out = K;
    if (n_circ)
        *n_circ = M;
    // the pose, by the rules of vo_batch_get_pose / fetch_pose
    if (r.status == 0 && r.lm_iters < 0) { // P3P without a solution: rvec / tvec untouched (see get_pose_impl)
        if (R_out && rvec_io && !c->prm.mono_rotation)
            rodrigues_v2m(rvec_io, R_out, nullptr);
    } else if (r.status >= 0) {
        if (rvec_io)
            memcpy(rvec_io, r.rvec, sizeof(r.rvec));
        if (tvec_io)
            memcpy(tvec_io, r.tvec, sizeof(r.tvec));
        if (R_out && !c->prm.mono_rotation)
            memcpy(R_out, r.R, sizeof(r.R)); // `if (!mono_rotation) Rodrigues(rvec, rotation)` (visualOdometry.cpp:186-189)
    }
    int em_status = 1;
    if (mono) {
        EmResult e;
        memcpy(&e, hb + 256, sizeof(e));
        if (e.status == 1 && R_out)
            memcpy(R_out, e.R, sizeof(e.R));
        em_status = e.status;
    }
    if (inliers && r.n_inliers > 0)
        memcpy(inliers, ai, (size_t)r.n_inliers * 4);
    if (n_inliers)
        *n_inliers = r.n_inliers;
    VO_HOST_STAMP(9);
    if (r.status < 0)
        return fail(c, VO_ERR_TOO_FEW, "fewer than 4 correspondences reached solvePnPRansac (CV_Assert(npoints >= 4))");
    if (em_status != 1) // mono_rotation and findEssentialMat found nothing: R_out was left untouched
        return VO_NO_ESSENTIAL;
    return r.status == 1 ? VO_OK : VO_NO_MODEL;
}

#ifdef VO_DEV_VARIANTS
// developer build only: the 100 MHz stamps the pose kernels left for frame 0 / hypothesis 0 (pnp.hip, tools/pose_phases.py)
int vo_dev_host_stamps(long long *out16)
{
    if (!out16)
        return VO_ERR_ARG;
    memcpy(out16, g_host_stamp, sizeof(g_host_stamp));
    return VO_OK;
}

int vo_dev_pose_prof(vo_ctx *c, long long *out64)
{
    if (!c || !out64 || sync_all(c) != VO_OK)
        return VO_ERR_ARG;
    return vo::pose_prof_read(out64) == 0 ? VO_OK : VO_ERR_HIP;
}

} // extern "C"

namespace vo_capi {

#endif

} // namespace vo_capi
