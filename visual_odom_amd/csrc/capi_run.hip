// capi_run.hip -- run_stages: the one place that enqueues the stages of a run (pyramids, detection, LK, filter, triangulation,
// pose solve) on the context's streams, for the batch API, the lock-step loop and the drop-in calls alike; sync_all.
#include "capi_internal.h"

namespace vo_capi {

// The t1 pair a synchronous drop-in call deferred (vo_ctx::defer): staged and pulled over PCIe on stream `on`.
int flush_deferred(vo_ctx *c, hipStream_t on)
{
    const int n = c->defer.n;
    c->defer.n = 0;
    for (int k = 0; k < n; k++) {
        int rc = upload_image(c, c->defer.first + k, c->defer.img[k], c->defer.stride, hipMemcpyHostToDevice, /*idle*/ true, nullptr, -1, on);
        if (rc != VO_OK)
            return rc;
    }
    return VO_OK;
}

// dry (lock-step loop, schedule probe): everything but the two kernels that advance a sequence's state (seq_carry,
// seq_integrate) -- the step can then be repeated any number of times
int run_stages(vo_ctx *c, int stages, bool timed, hipEvent_t *evs, bool dry)
{
    if (!evs)
        evs = c->ev;
    if (c->n_images == 0)
        return fail(c, VO_ERR_STATE, "vo_batch_run before vo_batch_configure");
    if ((stages & (VO_STAGE_TRIANGULATE | VO_STAGE_PNP)) && !c->have_P)
        return fail(c, VO_ERR_STATE, "vo_batch_run: projection matrices not set");
    VO_HIP_TRY(c, hipSetDevice(c->device));
    (void)hipGetLastError(); // the launch check at the end must report THIS call's launches, not a stale error of the thread
    const int B = c->n_frames, cap = c->cap;
    const bool touches_pose = (stages & (VO_STAGE_FILTER | VO_STAGE_TRIANGULATE | VO_STAGE_PNP)) != 0;
    vo_ctx::PoseBufs &pb = c->pb[c->cur];
    const bool crowded = c->sched.waves >= 2; // essential-matrix kernels: their reduced-register variant goes with the PnP one
    vo_ctx::Seq &sq = c->seq;
    const bool prep = sq.on && c->sched.prep; // lock-step loop: pyramids (and, from vo_seq_step, FAST) on the prepare stream
    hipStream_t pyrs = prep ? sq.copy : c->stream;
    int e = 0;
    // A synchronous drop-in call on the kept pair that left its t1 pair in host memory (single_frame_setup): hop 0 of the LK
    // chain reads the t0 pair only, so it starts before the t1 pair has crossed PCIe -- lk_hops_kernel [0, 1) on the tracking
    // stream, the two pulls + the t1 pyramids on the idle filter stream beside it, lk_hops_kernel [1, 4) behind ev_t1_ready.
    // Same bits as the one-launch chain (tests/test_kernel_emulation.py, the batch-against-call fuzz); the call gets shorter by
    // what now hides under hop 0.  Any other run that finds a deferred pair sends it first, the old way.
    const bool split = c->defer.n == 2 && !sq.on && B == 1 && (stages & VO_STAGE_PYRAMID) && (stages & VO_STAGE_LK) &&
                       !(stages & VO_STAGE_DETECT) && !c->tuning;
    if (c->defer.n && !split) {
        int rcd = flush_deferred(c, c->stream);
        if (rcd != VO_OK)
            return rcd;
    }
    if (timed)
        VO_HIP_TRY(c, hipEventRecord(evs[e], pyrs));
    e++;
    if (!split && (stages & VO_STAGE_PYRAMID)) { // (split: the t1 pyramids follow their pixels, in the LK stage below)
        const PyrImage *tab = c->d_imgs + c->pyr_first;
        const int ni = c->pyr_count;
        if (ni > 0) {
            // One launch per level, no LDS (round 4): a level is read once and gives its Scharr image, the next level and its
            // own border (pyramid.hip).  (Round 3: eight launches of three kernels that each fetched the level again.)
#ifdef VO_DEV_VARIANTS
            static const bool fused = [] { const char *e = getenv("VO_PYR_FUSED"); return !(e && e[0] == '0'); }();
            if (!fused) {
                launch_border_fill(tab, ni, 0, 1, c->lstride, c->lh, pyrs);
                launch_scharr(tab, ni, 0, 1, c->lw, c->lh, pyrs);
                for (int l = 0; l + 1 < c->levels; l++)
                    launch_pyr_down(tab, ni, l, c->lw[l + 1], c->lh[l + 1], pyrs);
                launch_border_fill(tab, ni, 1, c->levels, c->lstride, c->lh, pyrs);
                launch_scharr(tab, ni, 1, c->levels, c->lw, c->lh, pyrs);
            } else
#endif
                launch_pyramid_fused(tab, ni, c->levels, c->lw, c->lh, c->lstride, pyrs);
            std::fill(c->img_stale.begin() + c->pyr_first, c->img_stale.begin() + c->pyr_first + ni, (uint8_t)0);
        }
    }
    if (prep)
        VO_HIP_TRY(c, hipEventRecord(sq.ev_pyr, pyrs));
    const int *seq_active = sq.on ? sq.d_active + (size_t)(sq.step % VO_SEQ_INFLIGHT) * sq.S : nullptr;
    if (!sq.on && (stages & VO_STAGE_LK) && !split) { // (split: checked behind the deferred pyramids, below)
        for (int f = 0; f < B; f++) {
            const Quad &q = c->h_quads[f];
            if (c->img_stale[q.l0] | c->img_stale[q.r0] | c->img_stale[q.l1] | c->img_stale[q.r1])
                return fail(c, VO_ERR_STATE, "vo_batch_run: VO_STAGE_LK on an image uploaded after its pyramid was last "
                                             "built (run VO_STAGE_PYRAMID over it first)");
        }
    }
    if (timed)
        VO_HIP_TRY(c, hipEventRecord(evs[e], pyrs));
    e++;
    // DETECT and LK write the set of buffers (bucketed features / tracks + status) that the filter of two runs
    // ago read; the filter of the previous run reads the other set
    const int wset = (stages & (VO_STAGE_DETECT | VO_STAGE_LK)) ? c->trk_next : c->trk_last;
    if ((stages & (VO_STAGE_DETECT | VO_STAGE_LK)) && c->trk_busy[wset]) {
        VO_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_trk_free[wset], 0));
        c->trk_busy[wset] = false;
    }
    if (timed && !(stages & VO_STAGE_DETECT))
        VO_HIP_TRY(c, hipEventRecord(evs[VO_NUM_STAGES + 2], c->stream));
    if (stages & VO_STAGE_DETECT) {
        const int bs = c->dprm.bucket_size > 0 ? c->dprm.bucket_size : c->h / 10;
        const int fpb = c->dprm.features_per_bucket;
        const int cells = (c->h / bs + 1) * (c->w / bs + 1);
        if (!bucket_grid_ok(c->w, c->h, bs, fpb))
            return fail(c, VO_ERR_ARG, "vo_batch_run: bucket grid beyond the limits of the device bucketing (vo_hip.h, vo_detect_params)");
        if (c->w > 4096)
            return fail(c, VO_ERR_ARG, "vo_batch_run: VO_STAGE_DETECT handles images up to 4096 pixels wide");
        // appendNewFeatures only when fewer than redetect_below features were carried in (visualOdometry.cpp:95)
        if (timed)
            VO_HIP_TRY(c, hipEventRecord(evs[VO_NUM_STAGES + 2], c->stream));
        if (sq.on) {
            // the carried set lives on the device (seq_carry_kernel of the previous step wrote it on the filter stream)
            if (sq.carry_pending) {
                VO_HIP_TRY(c, hipStreamWaitEvent(c->stream, sq.ev_carry, 0));
                sq.carry_pending = false;
            }
            const int rp = (int)((sq.step - 1) % sq.ring); // ring slot of this step's t0 pair
            const bool ahead = prep && sq.have_corners[rp]; // its corners were detected one step ago on the prepare stream
            for (int r2 = 0; r2 < sq.ring; r2++)
                if (sq.fast_pending[r2] && (r2 == rp || !ahead)) {
                    // (inline detection shares the FAST scratch buffers with a look-ahead pass that may still run)
                    VO_HIP_TRY(c, hipStreamWaitEvent(c->stream, sq.ev_fast[r2], 0));
                    sq.fast_pending[r2] = false;
                }
            launch_seq_prepare(seq_active, c->d_ntracked, c->dprm.redetect_below, c->d_detect,
                               ahead ? sq.d_ncorn + (size_t)rp * sq.S : nullptr, c->d_nnew, B, c->stream);
            c->detect_uploaded = false;
        } else {
            bool changed = false;
            for (int f = 0; f < B; f++) {
                const int d = c->h_ntracked[f] < c->dprm.redetect_below ? 1 : 0;
                changed |= d != c->h_detect[f];
                c->h_detect[f] = d;
            }
            if (changed || !c->detect_uploaded) {
                VO_HIP_TRY(c, hipMemcpyAsync(c->d_detect, c->h_detect.data(), sizeof(int) * B, hipMemcpyHostToDevice,
                                             c->stream));
                VO_HIP_TRY(c, hipStreamSynchronize(c->stream)); // h_detect is reused by the next call
                c->detect_uploaded = true;
            }
        }
        int t = c->dprm.fast_threshold;
        t = t < 0 ? 0 : t > 255 ? 255 : t;
        if (prep && sq.have_corners[(sq.step - 1) % sq.ring]) {
            const int rp = (int)((sq.step - 1) % sq.ring);
            launch_bucket(c->d_feat, sq.d_corners + (size_t)rp * sq.S * c->fcap, c->d_fages, c->d_ntracked, c->d_nnew, c->fcap,
                          c->w, c->h, bs, fpb, c->d_pts_det[wset], c->d_ages_det[wset], c->d_npts_det[wset], cap, seq_active,
                          c->d_overflow, B, c->stream);
        } else {
            launch_detect_bucket(c->d_imgs, c->quads_cur, c->d_detect, B, c->w, c->h, t, c->dprm.fast_nonmax,
                                 c->d_nmsmask, c->d_rowcnt, c->d_rowoff, c->d_ntracked, c->d_nnew, c->fcap, c->d_feat, c->d_fages, bs, fpb,
                                 c->d_pts_det[wset], c->d_ages_det[wset], c->d_npts_det[wset], cap, seq_active,
                                 c->d_overflow, c->stream);
        }
        if (sq.on && !prep) { // (seq_enqueue_inputs: the NEXT step's PCIe ingest waits for this)
            VO_HIP_TRY(c, hipEventRecord(sq.ev_detect, c->stream));
            sq.detect_pending = true;
        }
        c->pts_sel = wset;
        // the bucketed count is only known on the device; every later grid is sized by its bound
        const int bound = cells * fpb < cap ? cells * fpb : cap;
        c->max_pts_set = bound;
        c->pts_on_device = true;
    }
    if (timed)
        VO_HIP_TRY(c, hipEventRecord(evs[e], c->stream));
    e++;
    if (stages & VO_STAGE_LK) {
        if (prep) // the t1 pyramids of this step were built on the prepare stream
            VO_HIP_TRY(c, hipStreamWaitEvent(c->stream, sq.ev_pyr, 0));
        LkParams lp;
        lp.max_level = c->levels - 1;
        int mc = c->prm.lk_max_count;
        lp.max_count = mc < 0 ? 0 : mc > 100 ? 100 : mc;
        double eps = c->prm.lk_epsilon;
        eps = eps < 0. ? 0. : eps > 10. ? 10. : eps;
        lp.epsilon = eps * eps;
        lp.min_eig = (float)c->prm.lk_min_eig_threshold;
        lp.full_chain = c->prm.lk_full_chain;
        if (split) {
            const Quad &q = c->h_quads[0];
            if (c->img_stale[q.l0] | c->img_stale[q.r0])
                return fail(c, VO_ERR_STATE, "synchronous call: the t0 pair has no pyramids");
            launch_lk_hops(c->d_imgs, c->quads_cur, cur_pts(c), cur_npts(c), cap, c->max_pts_set, B, c->d_trk2[wset],
                           c->d_status2[wset], lp, 0, 1, c->stream);
            hipStream_t side = c->stream_filter; // idle: the chain of a synchronous call stays on the tracking stream
            const int t1 = c->defer.first;
            int rcd = flush_deferred(c, side);
            if (rcd != VO_OK)
                return rcd;
            launch_pyramid_fused(c->d_imgs + t1, 2, c->levels, c->lw, c->lh, c->lstride, side);
            c->img_stale[t1] = c->img_stale[t1 + 1] = 0;
            VO_HIP_TRY(c, hipEventRecord(c->ev_t1_ready, side));
            VO_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_t1_ready, 0));
            if (c->img_stale[q.l1] | c->img_stale[q.r1])
                return fail(c, VO_ERR_STATE, "synchronous call: the t1 pair has no pyramids");
            launch_lk_hops(c->d_imgs, c->quads_cur, cur_pts(c), cur_npts(c), cap, c->max_pts_set, B, c->d_trk2[wset],
                           c->d_status2[wset], lp, 1, 4, c->stream);
        } else
#ifdef VO_DEV_VARIANTS
        if (c->lk_pair)
            launch_lk_circular_pair(c->d_imgs, c->quads_cur, cur_pts(c), cur_npts(c), cap, c->max_pts_set, B, c->d_trk2[wset],
                                    c->d_status2[wset], lp, c->stream);
        else
#endif
            launch_lk_circular(c->d_imgs, c->quads_cur, cur_pts(c), cur_npts(c), cap, c->max_pts_set, B, c->d_trk2[wset],
                               c->d_status2[wset], lp, c->stream);
        c->trk_last = wset;
        c->trk_next = wset ^ 1;
        if (sq.on) { // the ring slots holding this step's pairs may be overwritten once this LK has finished
            const int r0 = (int)((sq.step - 1) % sq.ring), r1 = (int)(sq.step % sq.ring);
            VO_HIP_TRY(c, hipEventRecord(sq.ev_slot_free[r0], c->stream));
            VO_HIP_TRY(c, hipEventRecord(sq.ev_slot_free[r1], c->stream));
            sq.slot_busy[r0] = sq.slot_busy[r1] = true;
        }
    }
    if (timed)
        VO_HIP_TRY(c, hipEventRecord(evs[e], c->stream)); // evs[3]: end of LK on the tracking stream
    e++;
    // Everything after LK is small, latency-bound work and leaves the tracking stream so that the next
    // run's pyramid / LK launches overlap it:
    //   filter stream: filter + triangulation of run k start as soon as LK(k) is done (they must not
    //                  queue behind the pose solve of run k - 1, which is still running next to LK(k));
    //   pose stream:   the PnP / RANSAC chain of run k.
    // Run k writes buffer set k % 2; its filter first waits for the pose solve of run k - 2 (same set).
    // The tracking stream only waits -- before its next DETECT / LK, i.e. after a whole pyramid stage --
    // for the filter to have consumed the points / tracks / status it is about to overwrite.
    // A synchronous drop-in call (vo_track_frame) has nothing to overlap with: everything on the tracking stream saves the
    // three cross-stream hand-offs of the chain (~12 us each in the kernel timeline of one call).
    bool serial = c->serial_pose || (c->sync_call && !sq.on);
#ifdef VO_DEV_VARIANTS
    static const int sync_serial_env = [] { const char *e = getenv("VO_SYNC_SERIAL"); return e ? atoi(e) : -1; }();
    if (sync_serial_env == 0 && !c->serial_pose)
        serial = false; // A/B: the synchronous call on the batch mode's streams
#endif
    c->last_run_serial = serial;
    hipStream_t fs = serial ? c->stream : c->stream_filter;
    const bool two_pose_streams = !serial && !c->prm.mono_rotation && c->sched.streams == 2;
    hipStream_t ps = serial ? c->stream : (two_pose_streams && (c->cur & 1)) ? c->stream_pnp2 : c->stream_pnp;
    // (serial: filter, triangulation and pose chain follow LK on the tracking stream itself -- stream order is the dependency,
    // and none of the events that hand work from one stream to the next is recorded: each cost ~6 us of idle GPU between two
    // kernels of the synchronous call, three of them per call, profiles/r04_track_frame_timeline.txt)
    if (touches_pose) {
        if (!serial) {
            VO_HIP_TRY(c, hipEventRecord(pb.ready, c->stream));
            VO_HIP_TRY(c, hipStreamWaitEvent(fs, pb.ready, 0));
        }
        if (pb.pending) {
            VO_HIP_TRY(c, hipStreamWaitEvent(fs, pb.done, 0));
            pb.pending = false;
        }
    }
    hipStream_t ts = touches_pose ? fs : c->stream;
    if (timed)
        VO_HIP_TRY(c, hipEventRecord(evs[e], ts)); // evs[4]
    e++;
    if (stages & VO_STAGE_FILTER) {
        launch_compact(cur_pts(c), c->d_trk2[c->trk_last], c->d_status2[c->trk_last], cur_npts(c), cap,
                       c->prm.consistency_threshold, c->d_outA, c->d_idxA, c->d_nA, pb.outB, pb.idxB, pb.nB, B, fs);
        if (sq.on) { // currentVOFeatures of every sequence after this frame (seq.hip)
            if (!dry)
                launch_seq_carry(seq_active, pb.outB, pb.nB, c->d_idxA, c->d_nA, cur_ages(c), cur_npts(c), cap, c->fcap,
                                 c->d_feat, c->d_fages, c->d_ntracked, c->d_overflow, sq.d_rows_carry, sq.d_nages, sq.d_info,
                                 sq.max_steps, B, fs);
            // (a dry run keeps the DEPENDENCY -- the next run's detection waits for this run's filter like it waits for
            // the carried features in a real step -- without the kernel that would advance the state)
            VO_HIP_TRY(c, hipEventRecord(sq.ev_carry, fs));
            sq.carry_pending = true;
        }
        if (!serial) {
            VO_HIP_TRY(c, hipEventRecord(c->ev_trk_free[c->trk_last], fs));
            c->trk_busy[c->trk_last] = true;
        }
        if (!serial && c->pts_sel >= 0 && c->pts_sel != c->trk_last) {
            // the points / ages this filter read belong to the OTHER set (a run without DETECT after a run with it):
            // the next DETECT into that set must wait for this filter too
            VO_HIP_TRY(c, hipEventRecord(c->ev_trk_free[c->pts_sel], fs));
            c->trk_busy[c->pts_sel] = true;
        }
    }
    if (timed)
        VO_HIP_TRY(c, hipEventRecord(evs[e], ts)); // evs[5]
    e++;
    if (stages & VO_STAGE_TRIANGULATE) // stage-B rows: 0 = l0, 1 = r0, 2 = l1, 3 = r1
        launch_triangulate(c->d_P, c->d_P + 12, pb.outB, pb.outB + cap, (size_t)4 * cap, pb.nB, cap,
                           c->max_pts_set, B, pb.xyz, fs);
    if (timed)
        VO_HIP_TRY(c, hipEventRecord(evs[e], ts)); // evs[6]: end of triangulation
    e++;
    if (stages & VO_STAGE_PNP) {
        if (!serial) {
            VO_HIP_TRY(c, hipEventRecord(pb.tri_done, fs));
            VO_HIP_TRY(c, hipStreamWaitEvent(ps, pb.tri_done, 0));
        }
        PnpParams pp;
        pp.iters = c->prm.ransac_iterations;
        pp.reproj = c->prm.ransac_reproj_error;
        pp.confidence = c->prm.ransac_confidence;
        // intrinsic_matrix = projMatrl(0:3, 0:3) (visualOdometry.cpp:163-165)
        for (int r = 0; r < 3; r++)
            for (int k = 0; k < 3; k++)
                pp.K[r * 3 + k] = c->h_P[r * 4 + k];
        if (c->prm.mono_rotation) {
            // rotation from the essential matrix of (pointsLeft_t0, pointsLeft_t1) = stage-B rows 0 and 2
            // (visualOdometry.cpp:146-157); the PnP solve below still provides the translation
            int rce = ensure_em(c);
            if (rce != VO_OK)
                return rce;
            EmParams ep;
            ep.focal = (double)c->h_P[0];
            ep.ppx = (double)c->h_P[2];
            ep.ppy = (double)c->h_P[6];
            ep.prob = c->prm.em_prob;
            ep.threshold = c->prm.em_threshold;
            ep.max_iters = EM_MAX_ITERS;
            // its own stream: the two chains only share their inputs, and together they would outlast the LK
            // launch they hide behind
            hipStream_t es = serial ? c->stream : c->stream_em;
            if (!serial)
                VO_HIP_TRY(c, hipStreamWaitEvent(es, pb.tri_done, 0));
            launch_essential(pb.outB, pb.outB + 2 * cap, (size_t)4 * cap, pb.nB, cap, B, ep, c->em, pb.em_results,
                             /*crowded*/ crowded, es);
            if (!serial)
                VO_HIP_TRY(c, hipEventRecord(pb.em_done, es));
        }
        launch_pnp_ransac(pb.xyz, pb.outB + 2 * cap, (size_t)4 * cap, pb.nB, cap, B, pp, pb.subsets, pb.models, pb.counts,
                          pb.rstate, c->sched.waves, ps, pb.epnp_ws,
                          c->max_frames < VO_EPNP_WS_MAX_FRAMES ? c->max_frames : VO_EPNP_WS_MAX_FRAMES, pb.epnp_gws, c->sched.wide,
                          pb.rest_ws);
        if (c->prm.mono_rotation && !serial)
            VO_HIP_TRY(c, hipStreamWaitEvent(ps, pb.em_done, 0)); // `done` covers both chains; the tail below reads E's rotation
        SeqTail tail;
        // frame_pose is chained: step k integrates after step k - 1, whichever stream ran it -- only the refinement kernels of
        // consecutive chains are ordered, their RANSAC parts overlap.  (A dry run of the schedule probe keeps the ORDER without
        // the integration: with two pose streams its refinements otherwise overlap as no real step's can, and the probe saw
        // 0.34 ms per step where the loop then ran at 0.49 -- one sequence, profiles/r03_schedule_sweep.jsonl of r3_30.)
        if (sq.on && sq.integ_pending)
            VO_HIP_TRY(c, hipStreamWaitEvent(ps, sq.ev_integ, 0));
        if (sq.on && !dry) { // euler gates + integrateOdometryStereo of every sequence, one trajectory row each: inside
                             // select_refine_kernel (vo_seqtail.h)
            tail.active = seq_active;
            tail.em = c->prm.mono_rotation ? pb.em_results : nullptr;
            tail.pose = sq.d_pose;
            tail.traj = sq.d_traj;
            tail.info = sq.d_info;
            tail.n_rows = sq.d_rows;
            tail.max_steps = sq.max_steps;
        }
        launch_pnp_refine(pb.xyz, pb.outB + 2 * cap, (size_t)4 * cap, pb.nB, cap, B, pp, pb.models, pb.rstate, pb.inliers,
                          pb.results, c->sched.waves, tail, ps);
        if (sq.on) {
            VO_HIP_TRY(c, hipEventRecord(sq.ev_integ, ps));
            sq.integ_pending = true;
        }
        c->last_pose_stream = ps;
        if (timed)
            VO_HIP_TRY(c, hipEventRecord(evs[e], ps)); // evs[7]: pose solve timed from the end of triangulation
        if (!serial) { // (serial: whoever needs the results waits for the tracking stream)
            VO_HIP_TRY(c, hipEventRecord(pb.done, ps));
            pb.pending = true;
        }
    } else if (timed) {
        VO_HIP_TRY(c, hipEventRecord(evs[e], ts));
    }
    VO_HIP_TRY(c, hipGetLastError());
    if (touches_pose) {
        c->last = c->cur;
        c->cur ^= 1;
    }
    return VO_OK;
}

// both streams idle (every getter and every synchronous entry point ends with this)
int sync_all(vo_ctx *c)
{
    VO_HIP_TRY(c, hipSetDevice(c->device));
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream_filter));
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream_pnp));
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream_pnp2));
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream_em));
    if (c->streams.copy)
        VO_HIP_TRY(c, hipStreamSynchronize(c->streams.copy));
    if (c->streams.prep)
        VO_HIP_TRY(c, hipStreamSynchronize(c->streams.prep));
    if (c->partitioned) // (the copy / prepare streams of the partitioned twin; its other streams are the ones above)
        for (int k = 5; k < 7; k++)
            VO_HIP_TRY(c, hipStreamSynchronize(c->streams.part[k]));
    return VO_OK;
}

} // namespace vo_capi
