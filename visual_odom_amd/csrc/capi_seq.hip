// capi_seq.hip -- the lock-step sequence loop (vo_seq_*): the reference's frame loop (main.cpp:123-224) for S sequences at
// once, state carried on the device.
#include "capi_internal.h"

#include <sched.h>

#include <atomic>
#include <thread>

#include <algorithm>

namespace vo_capi {

void seq_free(vo_ctx *c)
{
    vo_ctx::Seq &q = c->seq;
    void *ptrs[] = {q.d_quads, q.d_active, q.d_pose, q.d_traj, q.d_info, q.d_rows, q.d_rows_carry, q.d_nages, q.d_ing,
                    q.d_corners, q.d_ncorn};
    for (void *p : ptrs)
        if (p)
            (void)hipFree(p);
    if (q.h_active)
        (void)hipHostFree(q.h_active);
    if (q.h_ing)
        (void)hipHostFree(q.h_ing);
    if (q.h_stage)
        (void)hipHostFree(q.h_stage);
    if (q.d_stage)
        (void)hipFree(q.d_stage);
    hipEvent_t evs[] = {q.ev_upload, q.ev_carry, q.ev_integ, q.ev_pyr, q.ev_stage[0], q.ev_stage[1], q.ev_detect};
    for (auto &e : q.ev_fast)
        if (e)
            (void)hipEventDestroy(e);
    for (auto &e : q.ev_ab)
        if (e)
            (void)hipEventDestroy(e);
    for (hipEvent_t e : evs)
        if (e)
            (void)hipEventDestroy(e);
    for (auto &e : q.ev_slot_free)
        if (e)
            (void)hipEventDestroy(e);
    for (auto &e : q.ev_step)
        if (e)
            (void)hipEventDestroy(e);
    if (q.copy)
        (void)hipStreamSynchronize(q.copy); // belongs to the context's stream set, not to the loop
    q = vo_ctx::Seq();
}

} // namespace vo_capi

extern "C" {

/* ---------------------------------- lock-step sequence loop -------------------------------- */

int vo_seq_configure(vo_ctx *c, int n_seq, int w, int h, int ring, int max_steps)
{
    if (!c)
        return VO_ERR_ARG;
    if (n_seq < 1 || n_seq > c->max_frames || ring < 2 || ring > VO_SEQ_MAX_RING || max_steps < 1 ||
        2 * ring * n_seq > c->max_images)
        return fail(c, VO_ERR_ARG, "vo_seq_configure: need 1 <= n_seq <= max_frames, ring 2 or 3, "
                                   "2 * ring * n_seq <= 6 * max_frames images");
    // image table: ring slot r holds the pairs [r][s] = images (r * S + s) * 2 + {0 left, 1 right}, so that the
    // pairs a step receives are one contiguous range for the pyramid stage
    int rc = vo_batch_configure(c, 2 * ring * n_seq, w, h, n_seq);
    if (rc != VO_OK)
        return rc;
    rc = sync_all(c);
    if (rc != VO_OK)
        return rc;
    seq_free(c);
    c->tf_base = -1; // the ring owns the image table (with ring 2 and one sequence the configure above changed nothing)
    // ONE sequence: the loop's two chains of small kernels on disjoint halves of the compute units (capi.hip,
    // ensure_partitioned_streams); more sequences fill the chip and want all of it
    rc = select_streams(c, n_seq == 1);
    if (rc != VO_OK)
        return rc;
    vo_ctx::Seq &q = c->seq;
    const size_t S = (size_t)n_seq;
    q.S = n_seq;
    q.ring = ring;
    q.max_steps = max_steps;
    // Prepare stream or plain copy stream?  Moving the pyramids and FAST off the tracking stream shortens a step's
    // critical path, which is what a SMALL number of sequences is bound by (round 2: 1 sequence 1.39 k -> 1.58 k frames/s,
    // 8 sequences 9.3 k -> 12.3 k); with many sequences the GPU is saturated, the step costs the sum of its kernels
    // either way and the extra concurrency only disturbs them (64 sequences 41.3 k -> 35.8 k, 256: 49.2 k -> 46.7 k).
    // Where the crossover lies depends on the image size and the point load, so it is part of the probed schedule: until
    // the first full step has been probed the loop runs WITH the prepare stream (so that the look-ahead corners the
    // prepare variant needs exist when the probe compares the two), unless this shape was probed before or is pinned.
    bool ok = true;
    {
        c->seq.on = true; // (for the key; seq_free below has cleared it)
        const TuneKey key = tune_key(c, VO_STAGE_ALL | VO_STAGE_DETECT);
        c->seq.on = false;
        vo_ctx::Schedule sc;
        bool found = false;
        {
            std::lock_guard<std::mutex> lk(g_tune_mu);
            auto it = g_tuned.find(key);
            if (it != g_tuned.end()) {
                sc = it->second;
                found = true;
            }
        }
        q.on = true;
        apply_pins(c, &sc);
        q.on = false;
        c->sched = sc;
        c->sched_probed = found && !all_pinned(c);
        if (found || all_pinned(c))
            memcpy(c->sched_key, key.k, sizeof(key.k));
        else
            c->sched_key[0] = -1; // the first full step probes
        q.copy = ensure_copy_stream(&c->streams, sc.prep != 0, c->partitioned);
        ok = q.copy != nullptr;
    }
    ok = ok && dmalloc(&q.d_corners, (size_t)ring * S * c->fcap) == hipSuccess;
    ok = ok && dmalloc(&q.d_ncorn, (size_t)ring * S) == hipSuccess;
    ok = ok && hipMemset(q.d_ncorn, 0, sizeof(int) * ring * S) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&q.ev_pyr, hipEventDisableTiming) == hipSuccess;
    for (auto &e : q.ev_fast)
        ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    ok = ok && dmalloc(&q.d_quads, (size_t)ring * S) == hipSuccess;
    ok = ok && dmalloc(&q.d_active, (size_t)VO_SEQ_INFLIGHT * S) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&q.h_active, sizeof(int) * VO_SEQ_INFLIGHT * S, hipHostMallocDefault) == hipSuccess;
    ok = ok && dmalloc(&q.d_ing, (size_t)VO_SEQ_INFLIGHT * S) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&q.h_ing, sizeof(SeqIngest) * VO_SEQ_INFLIGHT * S, hipHostMallocDefault) == hipSuccess;
    ok = ok && dmalloc(&q.d_pose, S * 16) == hipSuccess;
    ok = ok && dmalloc(&q.d_traj, S * (size_t)max_steps * VO_SEQ_ROW) == hipSuccess;
    ok = ok && dmalloc(&q.d_info, S * (size_t)max_steps) == hipSuccess;
    ok = ok && dmalloc(&q.d_rows, S) == hipSuccess;
    ok = ok && dmalloc(&q.d_rows_carry, S) == hipSuccess;
    ok = ok && dmalloc(&q.d_nages, S) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&q.ev_upload, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&q.ev_detect, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&q.ev_carry, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&q.ev_integ, hipEventDisableTiming) == hipSuccess;
    for (auto &e : q.ev_slot_free)
        ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    for (auto &e : q.ev_step)
        ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    for (auto &e : q.ev_stage)
        ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    for (auto &e : q.ev_ab)
        ok = ok && hipEventCreate(&e) == hipSuccess;
    if (!ok) {
        seq_free(c);
        (void)select_streams(c, false); // (n_seq == 1 had moved the context onto the CU-partitioned twin: not without a loop)
        return fail(c, VO_ERR_HIP, "vo_seq_configure: allocation failed");
    }
    std::vector<Quad> tab((size_t)ring * S);
    for (int r = 0; r < ring; r++)
        for (int s = 0; s < n_seq; s++) {
            const int a = (r * n_seq + s) * 2, b = (((r + 1) % ring) * n_seq + s) * 2;
            tab[(size_t)r * S + s] = Quad{a, a + 1, b, b + 1};
        }
    VO_HIP_TRY(c, hipMemcpy(q.d_quads, tab.data(), sizeof(Quad) * tab.size(), hipMemcpyHostToDevice));
    VO_HIP_TRY(c, hipMemset(q.d_info, 0, sizeof(SeqFrameInfo) * S * (size_t)max_steps));
    q.pushed.assign(S, 0);
    q.had_prev.assign(S, 0);
    q.ever.assign(S, 0);
    q.gap.assign(S, 0);
    q.h_rows.assign(S, 0);
    q.step = 0;
    q.on = true;
    return vo_seq_reset(c, -1);
}

int vo_seq_reset(vo_ctx *c, int seq)
{
    if (!c)
        return VO_ERR_ARG;
    vo_ctx::Seq &q = c->seq;
    if (!q.on)
        return fail(c, VO_ERR_STATE, "vo_seq_reset before vo_seq_configure");
    if (seq >= q.S)
        return fail(c, VO_ERR_ARG, "vo_seq_reset: bad sequence");
    if (q.begun && q.n_ing > 0 && !(q.broken && seq < 0))
        return fail(c, VO_ERR_STATE, "vo_seq_reset between vo_seq_push_pair and vo_seq_step");
    int rc = sync_all(c);
    if (rc != VO_OK)
        return rc;
    const int s0 = seq < 0 ? 0 : seq, s1 = seq < 0 ? q.S : seq + 1;
    double eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int s = s0; s < s1; s++) {
        VO_HIP_TRY(c, hipMemcpy(q.d_pose + (size_t)s * 16, eye, sizeof(eye), hipMemcpyHostToDevice));
        q.pushed[s] = q.had_prev[s] = q.ever[s] = q.gap[s] = 0;
        q.h_rows[s] = 0;
    }
    if (seq < 0) {
        // everything is idle (sync_all above) and no sequence has a resident pair any more: the loop starts over -- ring
        // slot 0, event slot 0, all max_steps trajectory rows available again (a long-lived context that recycles its
        // sequences never runs out of steps)
        q.step = 0;
        if (q.ab_running())
            q.ab_phase = 0; // an unfinished comparison is abandoned: the dry probe's pick stays
        q.begun = q.staged = q.broken = false;
        q.n_ing = 0;
        q.carry_pending = q.integ_pending = false;
        for (auto &b : q.slot_busy)
            b = false;
        for (auto &b : q.fast_pending)
            b = false;
        for (auto &b : q.have_corners)
            b = false;
        for (auto &b : q.step_pending)
            b = false;
        q.stage_busy[0] = q.stage_busy[1] = false;
    }
    const size_t n = (size_t)(s1 - s0);
    VO_HIP_TRY(c, hipMemset(q.d_rows + s0, 0, sizeof(int) * n));
    VO_HIP_TRY(c, hipMemset(q.d_rows_carry + s0, 0, sizeof(int) * n));
    VO_HIP_TRY(c, hipMemset(q.d_nages + s0, 0, sizeof(int) * n));
    VO_HIP_TRY(c, hipMemset(c->d_ntracked + s0, 0, sizeof(int) * n));
    VO_HIP_TRY(c, hipMemset(c->d_fages + (size_t)s0 * c->fcap, 0, sizeof(int) * n * c->fcap));
    return VO_OK;
}

} // extern "C"

#define VO_SEQ_RUNAHEAD 3 // steps the host runs ahead of the device in the lock-step loop (seq_begin_step)

namespace vo_capi {

// First touch of the pending step (a push or the step call itself): its slot of the pinned per-step tables must
// have been consumed (step - VO_SEQ_INFLIGHT has finished), which also bounds the host's run-ahead.
int seq_begin_step(vo_ctx *c)
{
    vo_ctx::Seq &q = c->seq;
    if (q.begun)
        return VO_OK;
    const int slot = (int)(q.step % VO_SEQ_INFLIGHT);
    if (q.step_pending[slot]) {
        VO_HIP_TRY(c, hipEventSynchronize(q.ev_step[slot]));
        q.step_pending[slot] = false;
    }
    // The host stays VO_SEQ_RUNAHEAD = 3 steps ahead of the device, not the VO_SEQ_INFLIGHT = 8 the per-step tables would allow
    // (late in round 6).  Found through the schedule comparison: a schedule whose pose chains cannot keep up with the tracking
    // stages fills a deep run-ahead with their backlog first and for ~50 steps reads as fast as its steps are issued (256
    // resident sequences: 1,1,1 measured 3.36 ms per step in its window, best of eight, and sustained 3.91).  And the deep
    // run-ahead itself costs throughput -- kernels of steps k + 2 ... k + 7 queued on every stream beside step k's: with the
    // schedule pinned (gpurun_out/r6_ra, 340 points, frames/s at 8 / 4 / 3 / 2 steps) 16 sequences 20.4 / 24.4 / 24.7 / 20.9 k,
    // 32: 31.8 / 35.5 / 36.2 / 29.5 k, 64: 50.0 / 49.9 / 56.9 / 41.5 k, 128: 71.5 / 71.0 / 73.6 / 55.2 k, 256: 76.0 / 76.6 / 76.3 /
    // 71.2 k, 256 from page-locked memory 50.9 / 51.1 / 51.4 / 45.8 k, 256 at 2 000 points 25.7 / 25.7 / 25.9 / 25.5 k; 8
    // sequences 12.7 k at 8 steps, 15.9 k at 3.  Pairs that come from HOST memory get one step more (the transfer of step k + 1
    // has to be under way while step k runs): 64 sequences from page-locked memory 41.8 / 42.0 / 38.7 k at 8 / 4 / 3 steps,
    // 8 sequences 12.4 / 14.6 / 14.4 k (gpurun_out/r6_ra2, r6_ra3).  (q.ing_pcie still describes the previous step here.)
    int ahead = q.ing_pcie ? VO_SEQ_RUNAHEAD + 1 : VO_SEQ_RUNAHEAD;
#ifdef VO_DEV_VARIANTS
    static const int ahead_env = [] { const char *e = getenv("VO_SEQ_RUNAHEAD"); return e ? atoi(e) : 0; }(); // A/B: 1 .. 8
    if (ahead_env > 0)
        ahead = ahead_env;
#endif
    if (ahead < VO_SEQ_INFLIGHT && q.step >= ahead) {
        const int s3 = (int)((q.step - ahead) % VO_SEQ_INFLIGHT);
        if (q.step_pending[s3]) {
            VO_HIP_TRY(c, hipEventSynchronize(q.ev_step[s3]));
            q.step_pending[s3] = false;
        }
    }
    q.n_ing = 0;
    q.begun = true;
    return VO_OK;
}

// A push only records where the pair is; vo_seq_step moves all pairs of the step with ONE kernel on the copy stream
// (seq_ingest_kernel).  mode 0: pageable host memory, copied into the pinned staging area now so the caller's buffer
// is free on return; 1: page-locked host memory, read by the GPU over PCIe when the step runs; 2: device memory.
// a pageable image on its way into the pinned staging area (vo_seq_push_pairs copies a step's images with several threads)
struct StageCopy {
    uint8_t *dst;
    const uint8_t *src;
    int stride;
};
static void stage_copy(const StageCopy &k, int w, int h)
{
    if (k.stride == w)
        memcpy(k.dst, k.src, (size_t)w * h);
    else
        for (int y = 0; y < h; y++)
            memcpy(k.dst + (size_t)y * w, k.src + (size_t)y * k.stride, (size_t)w);
}

static int seq_push_impl(vo_ctx *c, int seq, const void *left, const void *right, int stride, int mode, std::vector<StageCopy> *defer)
{
    if (!c)
        return VO_ERR_ARG;
    vo_ctx::Seq &q = c->seq;
    if (!q.on)
        return fail(c, VO_ERR_STATE, "vo_seq_push_pair before vo_seq_configure");
    if (q.broken)
        return fail(c, VO_ERR_STATE, "vo_seq_push_pair: a previous vo_seq_step failed half-way; vo_seq_reset(ctx, -1) first");
    if (seq < 0 || seq >= q.S || !left || !right || stride < c->w)
        return fail(c, VO_ERR_ARG, "vo_seq_push_pair: bad sequence / image / stride");
    if (q.pushed[seq])
        return fail(c, VO_ERR_STATE, "vo_seq_push_pair: this sequence already has a pair for the pending step");
    VO_HIP_TRY(c, hipSetDevice(c->device));
    int rc = seq_begin_step(c);
    if (rc != VO_OK)
        return rc;
    const int r = (int)(q.step % q.ring);
    SeqIngest e;
    e.stride = stride;
    e.image0 = (r * q.S + seq) * 2;
    if (mode == 0) {
        const int g = (int)(q.step & 1);
        const size_t img = (size_t)c->w * c->h;
        if (!q.h_stage || q.stage_img != img) {
            if (q.h_stage) {
                VO_HIP_TRY(c, hipStreamSynchronize(q.copy));
                VO_HIP_TRY(c, hipHostFree(q.h_stage));
                q.h_stage = nullptr;
                if (q.d_stage)
                    VO_HIP_TRY(c, hipFree(q.d_stage));
                q.d_stage = nullptr;
            }
            q.stage_img = img;
            VO_HIP_TRY(c, hipHostMalloc((void **)&q.h_stage, img * 2 * 2 * (size_t)q.S, hipHostMallocDefault));
            if (hipMalloc((void **)&q.d_stage, img * 2 * 2 * (size_t)q.S) != hipSuccess) { // (optional: without it the kernel reads h_stage over PCIe)
                (void)hipGetLastError();
                q.d_stage = nullptr;
            }
        }
        if (q.stage_busy[g]) { // the ingest kernel of step - 2 still reads this half of the staging area
            VO_HIP_TRY(c, hipEventSynchronize(q.ev_stage[g]));
            q.stage_busy[g] = false;
        }
        uint8_t *sl = q.h_stage + (((size_t)g * q.S + seq) * 2) * img, *sr = sl + img;
        const uint8_t *srcs[2] = {(const uint8_t *)left, (const uint8_t *)right};
        uint8_t *dsts[2] = {sl, sr};
        for (int side = 0; side < 2; side++) {
            const StageCopy k{dsts[side], srcs[side], stride};
            if (defer)
                defer->push_back(k); // (the caller copies the whole step's images at once, in parallel)
            else
                stage_copy(k, c->w, c->h);
        }
        e.left = sl;
        e.right = sr;
        e.stride = c->w;
        q.staged = true;
    } else if (mode == 1) {
        void *dl = nullptr, *dr = nullptr;
        if (hipHostGetDevicePointer(&dl, const_cast<void *>(left), 0) != hipSuccess ||
            hipHostGetDevicePointer(&dr, const_cast<void *>(right), 0) != hipSuccess) {
            (void)hipGetLastError();
            return fail(c, VO_ERR_ARG, "vo_seq_push_pair: host_pinned = 1 but the memory is not page-locked / mapped "
                                       "(hipHostMalloc, hipHostRegister, torch pin_memory)");
        }
        e.left = (const uint8_t *)dl;
        e.right = (const uint8_t *)dr;
    } else {
        e.left = (const uint8_t *)left;
        e.right = (const uint8_t *)right;
    }
    if (q.n_ing == 0) {
        q.ing_pcie = false;
        q.n_pageable = 0;
    }
    q.ing_pcie = q.ing_pcie || mode != 2;
    q.n_pageable += mode == 0;
    q.h_ing[(size_t)(q.step % VO_SEQ_INFLIGHT) * q.S + q.n_ing++] = e;
    q.pushed[seq] = 1;
    return VO_OK;
}

int seq_push(vo_ctx *c, int seq, const void *left, const void *right, int stride, int mode)
{
    return seq_push_impl(c, seq, left, right, stride, mode, nullptr);
}

} // namespace vo_capi

extern "C" {

int vo_seq_push_pair(vo_ctx *c, int seq, const uint8_t *left, const uint8_t *right, int stride, int host_pinned)
{
    return seq_push(c, seq, left, right, stride, host_pinned ? 1 : 0);
}

int vo_seq_push_pair_dev(vo_ctx *c, int seq, const void *left, const void *right, int stride)
{
    return seq_push(c, seq, left, right, stride, 2);
}

int vo_seq_push_pairs(vo_ctx *c, int n, const int32_t *seq_ids, const void *const *left, const void *const *right,
                      int stride, int kind)
{
    if (!c || n < 0 || (n > 0 && (!seq_ids || !left || !right)) || kind < 0 || kind > 2)
        return VO_ERR_ARG;
    // Pageable images are copied into the page-locked staging area before the call returns.  One thread moves ~30 GB/s: the
    // 239 MB of a 256-sequence KITTI step took 8 ms -- more than the step at the reference-default load (3.5 ms), and what
    // bounded that configuration at 31 k frames/s (gpurun_out/r6_ingab3).  From 16 images on the copies of a call are
    // spread over up to 8 threads (never more than the caller's affinity mask holds, one image at a time per thread).
    std::vector<StageCopy> copies;
    int rc = VO_OK;
    for (int i = 0; i < n && rc == VO_OK; i++)
        rc = seq_push_impl(c, seq_ids[i], left[i], right[i], stride, kind, kind == 0 ? &copies : nullptr);
    // (copies queued before a failing pair belong to pairs that were accepted: they are carried out all the same)
    if (!copies.empty()) {
        int cpus = (int)std::thread::hardware_concurrency();
#ifdef __linux__
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0)
            cpus = CPU_COUNT(&set);
#endif
        const int w = c->w, h = c->h;
        int nt = (int)copies.size() / 16;
        nt = nt > 8 ? 8 : nt;
        nt = nt > cpus ? cpus : nt;
        if (nt <= 1) {
            for (const StageCopy &k : copies)
                stage_copy(k, w, h);
        } else {
            std::atomic<size_t> next{0};
            auto work = [&]() {
                for (size_t i = next.fetch_add(1); i < copies.size(); i = next.fetch_add(1))
                    stage_copy(copies[i], w, h);
            };
            std::vector<std::thread> pool;
            for (int t = 1; t < nt; t++)
                pool.emplace_back(work);
            work();
            for (std::thread &t : pool)
                t.join();
        }
    }
    return rc;
}

// The comparison of schedules over real steps: untimed ramp steps behind a switch, and the shortest timed window.  3 and 12 until
// late in round 6 -- with 256 sequences fed from page-locked memory (a step that is the PCIe link's) such a window measured
// 5.14 ms per step for 2,2,0, which then sustains 5.7, against 5.17 for 1,1,0, which sustains 5.15; behind 10 ramp steps a
// 24-step window reads 5.3-5.5 against 5.05-5.2 (gpurun_out/r6_abwin): the chains of two pose streams take that long to
// overlap as they do sustained.
#define VO_AB_RAMP 10
#define VO_AB_MIN 24

int vo_seq_step(vo_ctx *c)
{
    if (!c)
        return VO_ERR_ARG;
    vo_ctx::Seq &q = c->seq;
    if (!q.on)
        return fail(c, VO_ERR_STATE, "vo_seq_step before vo_seq_configure");
    if (!c->have_P)
        return fail(c, VO_ERR_STATE, "vo_seq_step: projection matrices not set");
    if (q.broken)
        return fail(c, VO_ERR_STATE, "vo_seq_step: a previous step failed half-way; vo_seq_reset(ctx, -1) first");
    // everything that can be refused is refused BEFORE the step consumes its pairs
    {
        const int bs = c->dprm.bucket_size > 0 ? c->dprm.bucket_size : c->h / 10;
        const int fpb = c->dprm.features_per_bucket;
        if (!bucket_grid_ok(c->w, c->h, bs, fpb))
            return fail(c, VO_ERR_ARG, "vo_seq_step: bucket grid beyond the limits of the device bucketing (vo_hip.h, vo_detect_params)");
        if (c->w > 4096)
            return fail(c, VO_ERR_ARG, "vo_seq_step: detection handles images up to 4096 pixels wide");
    }
    for (int s = 0; s < q.S; s++)
        if (q.pushed[s] && q.had_prev[s] && q.h_rows[s] >= q.max_steps) {
            // refuse the step and drop its pending pairs: the loop stays usable (trajectories can be read,
            // vo_seq_reset(s) gives the sequence its rows back)
            // Dropping a pair is a PAUSE of its sequence: if the caller moves on instead of re-pushing the same
            // pairs after vo_seq_reset(s), the next pair of such a sequence restarts its image pair (it is NOT matched against
            // the pair from two pushes ago) and the frame after that carries VO_SEQ_F_GAP, exactly like a resumed sequence.
            for (int k = 0; k < q.S; k++) {
                if (q.pushed[k])
                    q.had_prev[k] = 0;
                q.pushed[k] = 0;
            }
            q.begun = false;
            q.n_ing = 0;
            q.staged = false; // (the staging area holds only the dropped pairs: nothing was enqueued that reads it)
            return fail(c, VO_ERR_STATE, "vo_seq_step: a sequence's trajectory capacity (max_steps of vo_seq_configure) is "
                                         "exhausted; the pairs pushed for this step were dropped (re-push them after "
                                         "vo_seq_reset(seq), or go on: the affected sequences resume as after a pause)");
        }
    VO_HIP_TRY(c, hipSetDevice(c->device));
    int rc = seq_begin_step(c);
    if (rc != VO_OK)
        return rc;
    const int slot = (int)(q.step % VO_SEQ_INFLIGHT);
    const int r = (int)(q.step % q.ring);
    // a sequence processes a frame iff it has a pair for this step and had one for the previous step.  A sequence that
    // RESUMES after steps without a pair restarts its image pair (this pair only builds pyramids) but keeps its carried
    // features and pose -- a case the reference's loop does not have; its next processed frame carries VO_SEQ_F_GAP
    // (active value 3) so that the missing transition is on record.
    int *act = q.h_active + (size_t)slot * q.S;
    int n_active = 0;
    for (int s = 0; s < q.S; s++) {
        const bool on = q.pushed[s] && q.had_prev[s];
        if (q.pushed[s] && !q.had_prev[s] && q.ever[s])
            q.gap[s] = 1;
        act[s] = on ? (q.gap[s] ? 3 : 1) : 0;
        if (on) {
            q.gap[s] = 0;
            q.h_rows[s]++;
        }
        n_active += on;
        q.ever[s] |= q.pushed[s];
        q.had_prev[s] = q.pushed[s];
        q.pushed[s] = 0;
    }
    q.n_active = n_active;
    rc = seq_enqueue_inputs(c, /*dry*/ false);
    if (rc != VO_OK) { // (the step's bookkeeping is already consumed: same treatment as a failure further down)
        const std::string why = c->err;
        (void)sync_all(c);
        c->err = why;
        q.broken = true;
        return rc;
    }
    q.begun = false;
    c->pyr_first = r * q.S * 2;
    c->pyr_count = q.S * 2;
    int stages = VO_STAGE_PYRAMID;
    if (n_active > 0) {
        c->quads_cur = q.d_quads + (size_t)((q.step - 1) % q.ring) * q.S;
        stages |= VO_STAGE_DETECT | VO_STAGE_LK | VO_STAGE_FILTER | VO_STAGE_TRIANGULATE | VO_STAGE_PNP;
    }
    hipEvent_t *step_evs = &c->ring[(size_t)(q.step % VO_EVENT_SLOTS) * (VO_EV_PER_RUN)];
    rc = VO_OK;
    if (n_active > 0 && 2 * n_active >= q.S && !c->tuning) {
        // a step that shows the loop's real load: settle the schedule (cached / pinned / probed with dry runs of THIS
        // step -- everything but seq_carry and seq_integrate, so the step can be repeated)
        int need = sched_resolve(c, stages);
        if (need < 0)
            rc = need;
        else if (need) {
            rc = tune_schedule(c, stages, true, step_evs, /*dry*/ true, /*latency*/ false, /*publish*/ false);
            bool ab_started = false;
            if (rc == VO_OK && c->probe_n > 1) {
                // The dry runs leave out the two kernels that advance the state, and with them some of what the streams
                // hide: measured against every pinned schedule (tools/schedule_sweep.py) their verdict on the prepare knob was
                // wrong by 8-25 % at 1-32 sequences, and once the pose chain got shorter (round 3) they ranked the other two
                // knobs wrongly by 5-8 % in five of sixteen loops (two pose streams look better dry than real with one
                // sequence, one stream with 128).  So the dry probe only NOMINATES; up to four candidates then run for a while
                // each over REAL steps and end-of-step GPU timestamps decide.
                auto dry_ms = [&](const vo_ctx::Schedule &x) {
                    for (int i = 0; i < c->probe_n; i++)
                        if (c->probe_cand[i].pose_waves == x.waves && c->probe_cand[i].pose_streams == x.streams &&
                            c->probe_cand[i].prepare == x.prep && c->probe_cand[i].epnp_wide_frames == x.wide)
                            return (double)c->probe_ms[i];
                    return -1.0;
                };
                // The dry pick first (it stays if the loop ends before the comparison does).  Where the reach of the
                // four-kernel EPnP is a choice (5 .. 16 sequences) the dry pick with the OTHER reach comes second: dry, the two
                // differ by a per cent or two and the probe's choice between them is a coin flip, real steps differ by 17 %
                // at 8 sequences (0.54 against 0.65 ms per step, gpurun r5 `S=8` runs: 14.3 k or 12.0 k frames/s by that
                // flip).  Then one candidate per other (pose_streams, prepare) pair -- the two knobs the dry runs misjudge
                // -- each with the register budget the dry runs prefer for it, best dry time first, until four are named.
                int n = 0;
                c->ab_list[n++] = c->sched;
                if (wide_knob_live(c) && !c->pin.epnp_wide_frames) {
                    c->ab_list[n] = c->sched;
                    c->ab_list[n].wide = c->sched.wide == VO_EPNP_WS_MAX_FRAMES ? VO_EPNP_SPLIT_DEFAULT_FRAMES : VO_EPNP_WS_MAX_FRAMES;
                    n++;
                }
                struct Pair {
                    int st, pr, bi;
                } pairs[4];
                int np = 0;
                for (int st = 1; st <= 2; st++)
                    for (int pr = 1; pr >= 0; pr--) {
                        if (st == c->sched.streams && pr == c->sched.prep)
                            continue;
                        int bi = -1;
                        for (int i = 0; i < c->probe_n; i++)
                            if (c->probe_cand[i].pose_streams == st && c->probe_cand[i].prepare == pr &&
                                (bi < 0 || c->probe_ms[i] < c->probe_ms[bi]))
                                bi = i;
                        if (bi >= 0)
                            pairs[np++] = Pair{st, pr, bi};
                    }
                std::sort(pairs, pairs + np, [&](const Pair &x, const Pair &y) { return c->probe_ms[x.bi] < c->probe_ms[y.bi]; });
                for (int k = 0; k < np && n < 4; k++) {
                    c->ab_list[n].waves = c->probe_cand[pairs[k].bi].pose_waves;
                    // the reach of the four-kernel EPnP the dry probe preferred at ITS winner goes to every other nominee
                    c->ab_list[n].wide = c->sched.wide;
                    c->ab_list[n].streams = pairs[k].st;
                    c->ab_list[n].prep = pairs[k].pr;
                    n++;
                }
                if (n > 1) {
                    double ms = dry_ms(c->sched);
                    ms = ms > 0.02 ? ms : 0.02;
                    q.ab_n = (int)ceil(25.0 / ms);
                    q.ab_n = q.ab_n < VO_AB_MIN ? VO_AB_MIN : q.ab_n > 48 ? 48 : q.ab_n;
                    q.ab_cnt = n;
                    q.ab_extra = false;
                    q.ab_phase = 1;
                    q.ab_left = VO_AB_RAMP + q.ab_n;
                    memcpy(c->ab_key, c->sched_key, sizeof(c->ab_key));
                    c->sched_probed = false; // "in progress" (vo_get_schedule reports 2)
                    ab_started = true;
                }
            }
            if (rc == VO_OK && !ab_started) { // nothing to compare over real steps: the probe's pick is the settled schedule
                TuneKey key;
                memcpy(key.k, c->sched_key, sizeof(key.k));
                std::lock_guard<std::mutex> lk(g_tune_mu);
                g_tuned[key] = c->sched;
            }
        }
    }
    if (rc == VO_OK)
        rc = run_stages(c, stages, true, step_evs);
    if (rc == VO_OK && !c->sched.prep)
        q.have_corners[r] = false; // the pair now in slot r has no look-ahead corners
    if (rc == VO_OK && c->sched.prep)
        rc = seq_lookahead(c, r);
    if (rc != VO_OK) {
        // the step has consumed its pairs and part of it may be running: wait for the device, then refuse everything
        // until the caller starts over -- the ring / staging slots of this step must not be rewritten under it
        const std::string why = c->err;
        (void)sync_all(c);
        c->err = why;
        q.broken = true;
        return rc;
    }
    // end of the step = end of its last stream: the pose stream when a frame was processed; without a processed frame
    // the step's work is the ingest + pyramids (+ FAST) -- on the prepare stream when there is one
    hipStream_t end_stream = n_active > 0 && c->last_pose_stream ? c->last_pose_stream : c->sched.prep ? q.copy : c->stream;
    VO_HIP_TRY(c, hipEventRecord(q.ev_step[slot], end_stream));
    q.step_pending[slot] = true;
    q.step++;
    if (q.ab_running() && n_active > 0 && 2 * n_active >= q.S) {
        const int ph = q.ab_phase - 1;
        q.ab_left--;
        if (q.ab_left == q.ab_n) { // ramp over: the clock starts at the end of this step
            VO_HIP_TRY(c, hipEventRecord(q.ev_ab[2 * ph], end_stream));
        } else if (q.ab_left == 0) {
            VO_HIP_TRY(c, hipEventRecord(q.ev_ab[2 * ph + 1], end_stream));
            if (q.ab_phase < q.ab_cnt) {
                rc = set_sched(c, c->ab_list[q.ab_phase]); // (drains every stream first when the prepare knob changes)
                if (rc != VO_OK)
                    return rc;
                q.ab_phase++;
                q.ab_left = VO_AB_RAMP + q.ab_n;
            } else {
                VO_HIP_TRY(c, hipEventSynchronize(q.ev_ab[2 * ph + 1]));
                int best = 0;
                float t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int i = 0; i < q.ab_cnt; i++) {
                    VO_HIP_TRY(c, hipEventElapsedTime(&t[i], q.ev_ab[2 * i], q.ev_ab[2 * i + 1]));
                    if (t[i] < t[best])
                        best = i;
                }
                // Round 6: every nominee ran with the register budget the DRY runs prefer for its (pose_streams, prepare) pair --
                // and at 64 sequences of 1241 x 376 they prefer the wrong one: 2,2,1 wins the comparison at 1.42 ms per step where
                // 1,2,1 runs 1.29 (45.2 k against 49.6 k frames/s, gpurun_out/r6_s64; the loop was bimodal by that pick).  So the
                // winner runs once more with the other budget before anything is settled -- from 32 sequences on: below, a window
                // of ~30 steps flatters the 256-register kernels (16 sequences: 0.76 ms per step in the window, 0.85 once the
                // pose stream's backlog has built up; 18.7 k frames/s where the untouched pick runs 20.9 k, gpurun_out/r6_twin).
                // ... and (later in round 6) so does the nominee of every other (pose_streams, prepare) pair, fastest pair first: with
                // 256 sequences of 340 points fed from page-locked memory the dry runs name 2,1,0 for the pair (1, 0) as often as
                // 1,1,0 -- it sustains 41 k frames/s where 1,1,0 sustains 49.5 k and everything with two pose streams 45 k
                // (gpurun_out/r6_pinsched), so the pair lost the comparison without its better half having run, and the loop was
                // bimodal by that.  (Every pair, not the best two or three: 2,1,0, 2,1,1 and 1,2,1 all measure 6.3 ms there, 0.01-0.03
                // apart -- gpurun_out/r6_ab2, r6_ab3.  From 32 sequences on the comparison is therefore exhaustive: all eight
                // schedules over real steps, ~34 steps each.)
                if (!q.ab_extra && !c->pin.pose_waves && q.S >= 32) {
                    q.ab_extra = true;
                    int order[8], no = 0; // best nominee of every (pose_streams, prepare) pair, fastest pair first
                    for (int i = 0; i < q.ab_cnt; i++) {
                        int k = 0;
                        for (; k < no; k++)
                            if (c->ab_list[order[k]].streams == c->ab_list[i].streams && c->ab_list[order[k]].prep == c->ab_list[i].prep)
                                break;
                        if (k == no)
                            order[no++] = i;
                        else if (t[i] < t[order[k]])
                            order[k] = i;
                    }
                    std::sort(order, order + no, [&](int a, int b) { return t[a] < t[b]; });
                    const int first_new = q.ab_cnt;
                    for (int k = 0; k < no && q.ab_cnt < 8; k++) {
                        vo_ctx::Schedule twin = c->ab_list[order[k]];
                        twin.waves = twin.waves == 1 ? 2 : 1;
                        bool have = false;
                        for (int i = 0; i < q.ab_cnt; i++)
                            have = have || (c->ab_list[i].waves == twin.waves && c->ab_list[i].streams == twin.streams &&
                                            c->ab_list[i].prep == twin.prep && c->ab_list[i].wide == twin.wide);
                        if (!have)
                            c->ab_list[q.ab_cnt++] = twin;
                    }
                    if (q.ab_cnt > first_new) {
                        rc = set_sched(c, c->ab_list[first_new]);
                        if (rc != VO_OK)
                            return rc;
                        q.ab_phase++;
                        q.ab_left = VO_AB_RAMP + q.ab_n;
                        return VO_OK;
                    }
                }
                rc = set_sched(c, c->ab_list[best]);
                if (rc != VO_OK)
                    return rc;
                TuneKey key;
                memcpy(key.k, c->ab_key, sizeof(key.k));
                {
                    std::lock_guard<std::mutex> lk(g_tune_mu);
                    g_tuned[key] = c->ab_list[best];
                }
                for (int k = 0; k < q.ab_cnt; k++) { // the log shows what was measured over real steps
                    bool logged = false;
                    for (int i = 0; i < c->probe_n; i++)
                        if (c->probe_cand[i].pose_waves == c->ab_list[k].waves && c->probe_cand[i].pose_streams == c->ab_list[k].streams &&
                            c->probe_cand[i].prepare == c->ab_list[k].prep && c->probe_cand[i].epnp_wide_frames == c->ab_list[k].wide) {
                            c->probe_ms[i] = t[k] / q.ab_n;
                            c->probe_real[i] = 1;
                            logged = true;
                        }
                    if (!logged && c->probe_n < VO_PROBE_LOG_MAX) { // a nominee the dry probe did not run in this form (the wide reach)
                        const int i = c->probe_n++;
                        c->probe_cand[i] = vo_schedule{c->ab_list[k].waves, c->ab_list[k].streams, c->ab_list[k].prep, c->ab_list[k].wide};
                        c->probe_ms[i] = t[k] / q.ab_n;
                        c->probe_real[i] = 1;
                    }
                }
                q.ab_phase = q.ab_cnt + 1;
                c->sched_probed = true;
            }
        }
    }
    return VO_OK;
}

int vo_seq_sync(vo_ctx *c)
{
    if (!c)
        return VO_ERR_ARG;
    int rc = sync_all(c);
    if (rc == VO_OK)
        for (auto &p : c->seq.step_pending)
            p = false;
    return rc;
}

int vo_seq_get_state(vo_ctx *c, int seq, float *pts, int *n_pts, int32_t *ages, int *n_ages, double *pose16)
{
    if (!c)
        return VO_ERR_ARG;
    vo_ctx::Seq &q = c->seq;
    if (!q.on)
        return fail(c, VO_ERR_STATE, "vo_seq_get_state before vo_seq_configure");
    if (seq < 0 || seq >= q.S)
        return fail(c, VO_ERR_ARG, "vo_seq_get_state: bad sequence");
    int rc = vo_seq_sync(c);
    if (rc != VO_OK)
        return rc;
    int np = 0, na = 0;
    VO_HIP_TRY(c, hipMemcpy(&np, c->d_ntracked + seq, sizeof(int), hipMemcpyDeviceToHost));
    VO_HIP_TRY(c, hipMemcpy(&na, q.d_nages + seq, sizeof(int), hipMemcpyDeviceToHost));
    if (pts && np > 0)
        VO_HIP_TRY(c, hipMemcpy(pts, c->d_feat + (size_t)seq * c->fcap, sizeof(float2) * np, hipMemcpyDeviceToHost));
    if (ages && na > 0)
        VO_HIP_TRY(c, hipMemcpy(ages, c->d_fages + (size_t)seq * c->fcap, sizeof(int) * na, hipMemcpyDeviceToHost));
    if (pose16)
        VO_HIP_TRY(c, hipMemcpy(pose16, q.d_pose + (size_t)seq * 16, sizeof(double) * 16, hipMemcpyDeviceToHost));
    if (n_pts)
        *n_pts = np;
    if (n_ages)
        *n_ages = na;
    return VO_OK;
}

int vo_seq_get_trajectory(vo_ctx *c, int seq, int first, int count, double *rows, int32_t *info, int *n_rows)
{
    if (!c)
        return VO_ERR_ARG;
    vo_ctx::Seq &q = c->seq;
    if (!q.on)
        return fail(c, VO_ERR_STATE, "vo_seq_get_trajectory before vo_seq_configure");
    if (seq < 0 || seq >= q.S || first < 0 || count < 0)
        return fail(c, VO_ERR_ARG, "vo_seq_get_trajectory: bad sequence / range");
    int rc = vo_seq_sync(c);
    if (rc != VO_OK)
        return rc;
    int n = 0;
    VO_HIP_TRY(c, hipMemcpy(&n, q.d_rows + seq, sizeof(int), hipMemcpyDeviceToHost));
    n = n < q.max_steps ? n : q.max_steps;
    if (n_rows)
        *n_rows = n;
    const int k = first + count <= n ? count : (first < n ? n - first : 0);
    static_assert(sizeof(SeqFrameInfo) == VO_SEQ_INFO * sizeof(int32_t), "SeqFrameInfo layout is the public info8 row");
    if (k > 0 && rows)
        VO_HIP_TRY(c, hipMemcpy(rows, q.d_traj + ((size_t)seq * q.max_steps + first) * VO_SEQ_ROW,
                                sizeof(double) * VO_SEQ_ROW * k, hipMemcpyDeviceToHost));
    bool ovf = false;
    if (k > 0) {
        std::vector<SeqFrameInfo> tmp((size_t)k);
        VO_HIP_TRY(c, hipMemcpy(tmp.data(), q.d_info + (size_t)seq * q.max_steps + first, sizeof(SeqFrameInfo) * k,
                                hipMemcpyDeviceToHost));
        for (const SeqFrameInfo &f : tmp)
            ovf |= f.overflow != 0;
        if (info)
            memcpy(info, tmp.data(), sizeof(SeqFrameInfo) * k);
    }
    if (ovf)
        return fail(c, VO_ERR_OVERFLOW, "vo_seq_get_trajectory: a frame's detection / bucketing exceeded the capacity "
                                        "given to vo_create (its result is truncated)");
    return VO_OK;
}

} // extern "C"
