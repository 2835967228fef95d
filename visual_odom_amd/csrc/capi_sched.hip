// capi_sched.hip -- the schedule of the pose chain next to the tracking stages: probe on the caller's data, pins, the
// per-process table of settled schedules and its export / import (vo_set_schedule, vo_get_schedule, vo_get_probe_log,
// vo_export_schedule, vo_import_schedule).
#include "capi_internal.h"

namespace vo_capi {

std::mutex g_tune_mu;
std::map<TuneKey, vo_ctx::Schedule> g_tuned; // per process: a second context of the same shape starts tuned

TuneKey tune_key(const vo_ctx *c, int stages)
{
    long long pts = c->max_pts_set;
    if (stages & VO_STAGE_DETECT) { // the bucketed count is only known on the device: its bound, like the launches
        const int bs = c->dprm.bucket_size > 0 ? c->dprm.bucket_size : c->h / 10;
        const long long cells = bs > 0 ? (long long)(c->h / bs + 1) * (c->w / bs + 1) : 1;
        pts = cells * c->dprm.features_per_bucket < c->cap ? cells * c->dprm.features_per_bucket : c->cap;
    }
    TuneKey key;
    key.k[0] = c->device;
    key.k[1] = c->seq.on ? 1 : 0;
    key.k[2] = c->w;
    key.k[3] = c->h;
    key.k[4] = c->levels;
    key.k[5] = c->n_frames;
    // (the synchronous drop-in call is keyed on the image shape only: a live sequence whose feature count drifts across a
    // bucket boundary must not pay a probe -- ~20 frame times -- in the middle of real-time use)
    key.k[6] = (c->sync_call && !c->seq.on) ? 0 : pts_bucket(pts);
    key.k[7] = (c->prm.mono_rotation ? 1 : 0) | ((stages & VO_STAGE_DETECT) ? 2 : 0) | (c->sync_call && !c->seq.on ? 16 : 0);
    return key;
}

// the four-kernel EPnP is a choice only for 5 .. VO_EPNP_WS_MAX_FRAMES frames per launch (pnp.hip, launch_pnp_ransac)
bool wide_knob_live(const vo_ctx *c)
{
    return c->n_frames > VO_EPNP_SPLIT_DEFAULT_FRAMES && c->n_frames <= VO_EPNP_WS_MAX_FRAMES && c->max_frames >= c->n_frames;
}

void apply_pins(const vo_ctx *c, vo_ctx::Schedule *s)
{
    if (c->pin.pose_waves)
        s->waves = c->pin.pose_waves;
    if (c->pin.pose_streams)
        s->streams = c->pin.pose_streams;
    if (c->pin.prepare >= 0)
        s->prep = c->pin.prepare;
    if (c->pin.epnp_wide_frames)
        s->wide = c->pin.epnp_wide_frames;
    if (!wide_knob_live(c))
        s->wide = 4; // (launches of <= 4 frames take the wide form anyway, launches of > 16 never: one value, no candidates)
    if (c->prm.mono_rotation)
        s->streams = 1; // the essential-matrix chain already runs next to the PnP chain on its own stream
    if (!c->seq.on)
        s->prep = 0;
}

bool all_pinned(const vo_ctx *c)
{
    return c->pin.pose_waves && (c->pin.pose_streams || c->prm.mono_rotation) && (!c->seq.on || c->pin.prepare >= 0) &&
           (!wide_knob_live(c) || c->pin.epnp_wide_frames);
}

// make `s` the schedule the next run uses.  Moving the lock-step loop's ingest between the plain copy stream and the
// prepare stream is only done with every stream idle (the ring slots, the FAST scratch buffers and the staging area are
// ordered per stream).
int set_sched(vo_ctx *c, const vo_ctx::Schedule &s)
{
    if (c->seq.on && (s.prep != c->sched.prep || !c->seq.copy)) {
        int rc = sync_all(c);
        if (rc != VO_OK)
            return rc;
        c->seq.copy = ensure_copy_stream(&c->streams, s.prep != 0, c->partitioned);
        if (!c->seq.copy)
            return fail(c, VO_ERR_HIP, "could not create the copy stream");
        for (auto &b : c->seq.fast_pending)
            b = false;
        for (auto &b : c->seq.slot_busy)
            b = false;
        c->seq.stage_busy[0] = c->seq.stage_busy[1] = false;
    }
    c->sched = s;
    return VO_OK;
}

// Resolve the schedule for the run that is about to be enqueued.  Returns 1 when this key has to be probed first
// (nothing cached, not everything pinned), 0 when c->sched is settled, < 0 on error.
int sched_resolve(vo_ctx *c, int stages)
{
    const TuneKey key = tune_key(c, stages);
    if (memcmp(key.k, c->sched_key, sizeof(key.k)) == 0)
        return 0;
    vo_ctx::Schedule s;
    bool found = false;
    {
        std::lock_guard<std::mutex> lk(g_tune_mu);
        auto it = g_tuned.find(key);
        if (it != g_tuned.end()) {
            s = it->second;
            found = true;
        }
    }
    if (!found && !all_pinned(c) && !c->serial_pose)
        return 1;
    apply_pins(c, &s);
    int rc = set_sched(c, s);
    if (rc != VO_OK)
        return rc;
    memcpy(c->sched_key, key.k, sizeof(key.k));
    c->sched_probed = found && !all_pinned(c);
    c->probe_n = 0; // this key was settled without a probe of this context: vo_get_probe_log must not show another key's
    return 0;
}

// What the pending step's kernels read that comes from the host: the pushed pairs -> ring slot step % ring with ONE
// kernel on the copy stream -- after the LK that still reads the slot's previous occupant (ring 2: the previous step's;
// ring 3: the one before, long finished), next to the previous step's kernels -- and the step's activity flags.
// dry (schedule probe): the same transfers again (same bytes to the same places), without the slot / staging bookkeeping.
int seq_enqueue_inputs(vo_ctx *c, bool dry)
{
    vo_ctx::Seq &q = c->seq;
    const int slot = (int)(q.step % VO_SEQ_INFLIGHT), r = (int)(q.step % q.ring);
    if (q.n_ing > 0) {
        if (!dry && q.slot_busy[r]) {
            VO_HIP_TRY(c, hipStreamWaitEvent(q.copy, q.ev_slot_free[r], 0));
            q.slot_busy[r] = false;
        }
        SeqIngest *d_tab = q.d_ing + (size_t)slot * q.S;
        SeqIngest *h_tab = q.h_ing + (size_t)slot * q.S;
        // Every sequence pushed a PAGEABLE pair: the step's half of the staging area is one contiguous block -- the copy engine
        // takes it across the link in one transfer (57 GB/s against the kernel's 48, and no shader wave, no L2 queue entry beside
        // LK: tools/ubench/ingest_under_load.hip), and the ingest kernel re-pitches it from its device twin, HBM to HBM.
        bool over_pcie = q.ing_pcie;
        // (from 32 sequences on: a transfer of a few megabytes is over before the copy's fixed cost is paid -- 8 sequences 0.70 ms
        // per step through the kernel, 0.80 through the engine; 64: 1.67 / 1.63; 256: 5.20 / 4.77 and, at 2 000 points, 11.34 / 10.28)
        if (q.d_stage && q.S >= 32 && q.n_pageable == q.S && q.n_ing == q.S) {
            const size_t half = q.stage_img * 2 * (size_t)q.S, off = (size_t)(q.step & 1) * half;
            VO_HIP_TRY(c, hipMemcpyAsync(q.d_stage + off, q.h_stage + off, half, hipMemcpyHostToDevice, q.copy));
            if (!dry) // (a dry re-run finds the entries already pointing at the device twin)
                for (int i = 0; i < q.n_ing; i++) {
                    h_tab[i].left = q.d_stage + (h_tab[i].left - q.h_stage);
                    h_tab[i].right = q.d_stage + (h_tab[i].right - q.h_stage);
                }
            over_pcie = false;
        }
        // A kernel that reads over PCIe starts behind the running step's DETECTION -- when the step can afford it.  Its host reads
        // park in the L2's queues, and the memory-bound kernels at the head of a step (pyramid pass, FAST, bucketing) pay for that
        // far more than LK does (256 sequences, page-locked pairs, 2 000 points: detection 0.99 ms beside the ingest against
        // 0.42 alone).  Measured with pinned schedules (gpurun_out/r6_ingwait): +1.5 .. +8 % where LK is long against the
        // transfer (2 000 points: 8.8 ms against 5), -3 .. -18 % where the step is the link's (340 points: 2.1 ms against 5 --
        // there every microsecond of the window counts).  So the wait is taken iff the LK of the newest step that has certainly
        // finished (VO_SEQ_INFLIGHT + 1 steps back, its stage events sit in the ring) lasted >= 1.6 x the transfer at 48 GB/s.
        // A heuristic on a schedule, never on a result.  (Not with the prepare stream: there the copy stream carries the step's
        // own pyramids.)
        bool wait_detect = false;
        if (over_pcie && !dry && q.step > VO_SEQ_INFLIGHT + 1) {
            hipEvent_t *old = &c->ring[(size_t)((q.step - VO_SEQ_INFLIGHT - 1) % VO_EVENT_SLOTS) * (VO_EV_PER_RUN)];
            float lk_ms = 0.f;
            if (hipEventQuery(old[3]) == hipSuccess && hipEventElapsedTime(&lk_ms, old[2], old[3]) == hipSuccess)
                wait_detect = (double)lk_ms >= 1.6 * ((double)q.n_ing * 2.0 * c->w * c->h / 48e6);
            else
                (void)hipGetLastError(); // (a step without an LK stage left no such events)
        }
#ifdef VO_DEV_VARIANTS
        static const int wait_env = [] { const char *e = getenv("VO_INGEST_WAIT"); return e ? atoi(e) : -1; }(); // A/B: 0 / 1 force
        if (wait_env >= 0)
            wait_detect = wait_env != 0;
#endif
        if (wait_detect && over_pcie && !dry && !c->sched.prep && q.detect_pending) {
            VO_HIP_TRY(c, hipStreamWaitEvent(q.copy, q.ev_detect, 0));
            q.detect_pending = false;
        }
        VO_HIP_TRY(c, hipMemcpyAsync(d_tab, h_tab, sizeof(SeqIngest) * q.n_ing, hipMemcpyHostToDevice, q.copy));
        launch_seq_ingest(d_tab, q.n_ing, c->w, c->h, c->lstride[0],
                          c->d_pix + c->loff[0] + (size_t)VO_BY * c->lstride[0] + VO_BX, c->img_bytes, over_pcie, q.copy);
        if (!dry && q.staged) {
            const int g = (int)(q.step & 1);
            VO_HIP_TRY(c, hipEventRecord(q.ev_stage[g], q.copy));
            q.stage_busy[g] = true;
            q.staged = false;
        }
    }
    if (!c->sched.prep) {
        VO_HIP_TRY(c, hipEventRecord(q.ev_upload, q.copy));
        VO_HIP_TRY(c, hipStreamWaitEvent(c->stream, q.ev_upload, 0));
    }
    if (q.n_active > 0)
        VO_HIP_TRY(c, hipMemcpyAsync(q.d_active + (size_t)slot * q.S, q.h_active + (size_t)slot * q.S, sizeof(int) * q.S,
                                     hipMemcpyHostToDevice, c->stream));
    return VO_OK;
}

// FAST + non-maximum suppression of the pairs in ring slot r (this step's new pairs), for the NEXT step's
// appendNewFeatures: on the prepare stream behind their pyramids (fast_score reads level 0 only), while this step's LK runs
int seq_lookahead(vo_ctx *c, int r)
{
    vo_ctx::Seq &q = c->seq;
    int t = c->dprm.fast_threshold;
    t = t < 0 ? 0 : t > 255 ? 255 : t;
    launch_fast_corners(c->d_imgs, q.d_quads + (size_t)r * q.S, nullptr, q.S, c->w, c->h, t, c->dprm.fast_nonmax,
                        c->d_nmsmask, c->d_rowcnt, c->d_rowoff, nullptr, q.d_ncorn + (size_t)r * q.S, c->fcap,
                        q.d_corners + (size_t)r * q.S * c->fcap, q.copy);
    VO_HIP_TRY(c, hipEventRecord(q.ev_fast[r], q.copy));
    q.fast_pending[r] = true;
    q.have_corners[r] = true;
    VO_HIP_TRY(c, hipGetLastError());
    return VO_OK;
}

// one run of the probe: the stages, plus -- lock-step loop with the prepare stream -- the look-ahead detection a real
// step launches behind them (it recomputes the corners the real step will compute: idempotent)
int probe_run(vo_ctx *c, int stages, bool timed, hipEvent_t *evs, bool dry)
{
    int rc = dry && c->seq.on ? seq_enqueue_inputs(c, true) : VO_OK;
    if (rc == VO_OK)
        rc = run_stages(c, stages, timed, evs, dry);
    if (rc == VO_OK && dry && c->seq.on && c->sched.prep)
        rc = seq_lookahead(c, (int)(c->seq.step % c->seq.ring));
    return rc;
}

// STEADY-STATE milliseconds per run: n and n + K back-to-back runs are timed and the difference is divided by K, so that
// what every measurement has once -- the ramp-up and the last run's pose chain, which nothing overlaps -- cancels (timing
// one short burst instead favours the schedule with the shortest lone chain: the first version of this probe picked the
// 512-register kernels for 256 sequences, 10 % below the 256-register ones in the real loop).  K >= 20 ms of work, 6 .. 24.
// latency (the synchronous drop-in calls: one run, then the caller waits for it): the mean of K runs each followed by a
// synchronisation -- what such a caller sees; the steady-state figure hides exactly the chain latency it is waiting for.
int probe_candidate(vo_ctx *c, int stages, bool timed, hipEvent_t *evs, bool dry, bool latency, double *ms_per_run)
{
    using clk = std::chrono::steady_clock;
    if (latency) {
        int rc = sync_all(c);
        double total = 0;
        const int K = 8;
        for (int i = 0; i < K + 2 && rc == VO_OK; i++) {
            const auto t0 = clk::now();
            rc = probe_run(c, stages, timed, evs, dry);
            if (rc == VO_OK)
                rc = sync_all(c);
            if (i >= 2)
                total += std::chrono::duration<double, std::milli>(clk::now() - t0).count();
        }
        *ms_per_run = total / K;
        return rc;
    }
    auto burst = [&](int n, double *ms) {
        int rc = sync_all(c);
        const auto t0 = clk::now();
        for (int i = 0; i < n && rc == VO_OK; i++)
            rc = probe_run(c, stages, timed, evs, dry);
        if (rc == VO_OK)
            rc = sync_all(c);
        *ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
        return rc;
    };
    double warm = 0, ta = 0, tb = 0;
    int rc = burst(1, &warm);
    if (rc != VO_OK)
        return rc;
    int K = warm > 0 ? (int)ceil(20.0 / warm) : 24;
    K = K < 6 ? 6 : K > 24 ? 24 : K;
    rc = burst(3, &ta);
    if (rc == VO_OK)
        rc = burst(3 + K, &tb);
    if (rc != VO_OK)
        return rc;
    *ms_per_run = (tb - ta) / K;
    return VO_OK;
}

// Probe every candidate the pins leave open on the data the caller is about to process, keep the fastest.
// Batch mode: plain runs (a batch run is idempotent).  Lock-step loop: dry runs of the pending step.
int tune_schedule(vo_ctx *c, int stages, bool timed, hipEvent_t *evs, bool dry, bool latency, bool publish)
{
    const TuneKey key = tune_key(c, stages);
    // Candidates: every (waves, streams, prepare) the pins leave open with the four-kernel EPnP at its default reach -- at most
    // 2 x 2 x 2 -- and then, where that knob is live (5 .. 16 frames per run) and not pinned, the WINNER once more with the wide
    // reach: 8 + 1 candidates, not the 16 of the full product (that doubled the probe's cost beyond what vo_hip.h
    // states and filled the probe log to its last entry).
    static_assert(2 * 2 * 2 + 1 <= VO_PROBE_LOG_MAX, "the probe log holds every candidate");
    std::vector<vo_ctx::Schedule> cands;
    for (int waves = 1; waves <= 2; waves++)
        for (int streams = 1; streams <= (c->sync_call && !c->seq.on ? 1 : 2); streams++) // (a synchronous call runs on one stream)
            for (int prep = 1; prep >= 0; prep--) {
                vo_ctx::Schedule s, t;
                s.waves = waves;
                s.streams = streams;
                s.prep = prep;
                s.wide = c->pin.epnp_wide_frames && wide_knob_live(c) ? c->pin.epnp_wide_frames : VO_EPNP_SPLIT_DEFAULT_FRAMES;
                t = s;
                apply_pins(c, &t);
                if (t.waves != s.waves || t.streams != s.streams || t.prep != s.prep || t.wide != s.wide)
                    continue; // pinned away / not applicable
                if (prep && !c->seq.have_corners[c->seq.on ? (c->seq.step - 1) % c->seq.ring : 0])
                    continue; // no look-ahead corners for this step's t0 pair: the prepare variant cannot be shown
                cands.push_back(s);
            }
    if (cands.empty()) {
        vo_ctx::Schedule s;
        apply_pins(c, &s);
        cands.push_back(s);
    }
    bool probe_wide = wide_knob_live(c) && !c->pin.epnp_wide_frames;
    const bool measure = cands.size() > 1 || probe_wide;
    c->tuning = true;
    int rc = VO_OK, best = 0;
    double best_ms = 0;
    for (size_t i = 0; i < cands.size() && rc == VO_OK; i++) {
        rc = set_sched(c, cands[i]);
        double ms = 0;
        if (rc == VO_OK)
            rc = measure ? probe_candidate(c, stages, timed, evs, dry, latency, &ms) : VO_OK;
        if (rc == VO_OK && (i == 0 || ms < best_ms)) {
            best = (int)i;
            best_ms = ms;
        }
        if (i < VO_PROBE_LOG_MAX) {
            c->probe_cand[i] = vo_schedule{cands[i].waves, cands[i].streams, cands[i].prep, cands[i].wide};
            c->probe_ms[i] = (float)ms;
            c->probe_real[i] = 0;
        }
        if (i + 1 == cands.size() && probe_wide && rc == VO_OK) {
            vo_ctx::Schedule wd = cands[best]; // the winner with the wide reach, once (appended: the loop runs it next)
            wd.wide = VO_EPNP_WS_MAX_FRAMES;
            cands.push_back(wd);
            probe_wide = false;
        }
    }
    c->probe_n = (int)(cands.size() < VO_PROBE_LOG_MAX ? cands.size() : VO_PROBE_LOG_MAX);
    c->tuning = false;
    if (rc != VO_OK)
        return rc;
    rc = set_sched(c, cands[best]);
    if (rc != VO_OK)
        return rc;
    // The per-process table (what a second context resolves from, what vo_export_schedule writes) only gets SETTLED schedules:
    // the lock-step loop's dry probe merely nominates -- its caller publishes after the comparison over real steps, or at once
    // when there is nothing to compare (the nominee used to be exported as if it were settled).
    if (publish) {
        std::lock_guard<std::mutex> lk(g_tune_mu);
        g_tuned[key] = cands[best];
    }
    memcpy(c->sched_key, key.k, sizeof(key.k));
    c->sched_probed = true;
    return VO_OK;
}

// run_stages for the batch entry points: settles the schedule first (cached, pinned or probed) when the run has a pose chain
// sync_call: a drop-in call that returns results -- the caller waits for every run, so candidates are compared by latency
int run_stages_auto(vo_ctx *c, int stages, bool timed, hipEvent_t *evs, bool sync_call)
{
    if (!c->tuning)
        c->sync_call = sync_call;
    if ((stages & VO_STAGE_PNP) && !c->tuning && c->n_images > 0 && c->have_P) {
        int need = sched_resolve(c, stages);
        if (need < 0)
            return need;
        if (need) {
            int rc = tune_schedule(c, stages, timed, evs, false, sync_call);
            if (rc != VO_OK)
                return rc;
        }
    }
    return run_stages(c, stages, timed, evs);
}

} // namespace vo_capi

extern "C" {

int vo_set_schedule(vo_ctx *c, const vo_schedule *s)
{
    if (!c)
        return VO_ERR_ARG;
    vo_schedule p = {0, 0, -1, 0};
    if (s)
        p = *s;
    if (p.epnp_wide_frames != 0 && p.epnp_wide_frames != VO_EPNP_SPLIT_DEFAULT_FRAMES && p.epnp_wide_frames != VO_EPNP_WS_MAX_FRAMES)
        return fail(c, VO_ERR_ARG, "vo_set_schedule: epnp_wide_frames 0 / 4 / 16");
    const int max_waves =
#ifdef VO_DEV_VARIANTS
        4; // the slim pose chain (pnp.hip): measured slower everywhere, kept for the record in the developer build
#else
        2;
#endif
    if (p.pose_waves < 0 || p.pose_waves == 3 || p.pose_waves > max_waves || p.pose_streams < 0 || p.pose_streams > 2 ||
        p.prepare < -1 || p.prepare > 1)
        return fail(c, VO_ERR_ARG, "vo_set_schedule: pose_waves 0 / 1 / 2, pose_streams 0 / 1 / 2, prepare -1 / 0 / 1");
    int rc = sync_all(c);
    if (rc != VO_OK)
        return rc;
    c->pin = p;
    c->sched_key[0] = -1; // resolved again at the next run
    if (c->seq.ab_running())
        c->seq.ab_phase = 0; // a comparison over real steps in progress is abandoned: the caller has just said what they want
    if (c->seq.on) {      // the lock-step loop reads sched between steps: apply what is pinned now
        vo_ctx::Schedule sc = c->sched;
        apply_pins(c, &sc);
        rc = set_sched(c, sc);
    }
    return rc;
}

int vo_get_schedule(const vo_ctx *c, vo_schedule *cur, int *probed)
{
    if (!c || !cur)
        return VO_ERR_ARG;
    cur->pose_waves = c->sched.waves;
    cur->pose_streams = c->sched.streams;
    cur->prepare = c->seq.on ? c->sched.prep : 0;
    cur->epnp_wide_frames = c->sched.wide;
    if (probed)
        *probed = (c->seq.on && c->seq.ab_running()) ? 2 : c->sched_probed ? 1 : 0;
    return VO_OK;
}

int vo_get_probe_log(const vo_ctx *c, vo_schedule *cands, float *ms, int *real, int *n)
{
    if (!c || !n)
        return VO_ERR_ARG;
    *n = c->probe_n;
    for (int i = 0; i < c->probe_n; i++) {
        if (cands)
            cands[i] = c->probe_cand[i];
        if (ms)
            ms[i] = c->probe_ms[i];
        if (real)
            real[i] = c->probe_real[i];
    }
    return VO_OK;
}

// The per-process table of settled schedules, out and in (a service must not pay up to 1.2 s of probing and
// three pipeline drains in every process).  A record = the key a schedule was settled for (device, mode, image size, pyramid
// levels, frames per run, point-load bucket, flags -- tune_key above) + the schedule.  Export after a warm-up run of each shape
// the service uses; import the records at start-up, before the first run: every context of the process then resolves those
// keys from the table (vo_get_schedule reports probed = 1) without running a single probe.  Records are plain data and stay
// valid for the same library build and device model; a key the table does not hold is probed as before.
int vo_export_schedule(vo_schedule_record *recs, int cap, int *n)
{
    if (!n || cap < 0 || (cap > 0 && !recs))
        return VO_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_tune_mu);
    int k = 0;
    for (const auto &kv : g_tuned) {
        if (k < cap) {
            for (int i = 0; i < 8; i++)
                recs[k].key[i] = kv.first.k[i];
            recs[k].schedule = vo_schedule{kv.second.waves, kv.second.streams, kv.second.prep, kv.second.wide};
        }
        k++;
    }
    *n = k; // records in the table (may exceed cap: call again with a bigger array)
    return VO_OK;
}

int vo_import_schedule(const vo_schedule_record *recs, int n)
{
    if (n < 0 || (n > 0 && !recs))
        return VO_ERR_ARG;
    const int max_waves =
#ifdef VO_DEV_VARIANTS
        4;
#else
        2;
#endif
    for (int k = 0; k < n; k++) { // validate everything before anything is taken over
        const vo_schedule &sc = recs[k].schedule;
        if ((sc.pose_waves != 1 && sc.pose_waves != 2 && !(sc.pose_waves == 4 && max_waves == 4)) ||
            (sc.pose_streams != 1 && sc.pose_streams != 2) || (sc.prepare != 0 && sc.prepare != 1) ||
            (sc.epnp_wide_frames != VO_EPNP_SPLIT_DEFAULT_FRAMES && sc.epnp_wide_frames != VO_EPNP_WS_MAX_FRAMES) || recs[k].key[0] < 0 ||
            recs[k].key[2] < 32 || recs[k].key[3] < 32 || recs[k].key[4] < 1 || recs[k].key[4] > VO_MAX_LEVELS || recs[k].key[5] < 1)
            return VO_ERR_ARG;
    }
    std::lock_guard<std::mutex> lk(g_tune_mu);
    for (int k = 0; k < n; k++) {
        TuneKey key;
        for (int i = 0; i < 8; i++)
            key.k[i] = recs[k].key[i];
        vo_ctx::Schedule sc;
        sc.waves = recs[k].schedule.pose_waves;
        sc.streams = recs[k].schedule.pose_streams;
        sc.prep = recs[k].schedule.prepare;
        sc.wide = recs[k].schedule.epnp_wide_frames;
        g_tuned[key] = sc;
    }
    return VO_OK;
}

} // extern "C"
