// capi.hip -- host side of libvo_hip.so: context, device memory, streams, the batch API and its getters (C ABI declared in
// include/vo_hip.h).  No exceptions cross the ABI; every HIP failure becomes VO_ERR_HIP with a message retrievable through
// vo_last_error().  The other units of the ABI: capi_run.hip, capi_sched.hip, capi_seq.hip, capi_dropin.hip (capi_internal.h).
#include "capi_internal.h"

namespace vo_capi {

// pyramid geometry exactly as buildOpticalFlowPyramid: stop when the next level would not be
// larger than the 21 x 21 window
int plan_levels(vo_ctx *c, int w, int h)
{
    int cw = w, ch = h, l = 0;
    size_t off = 0;
    for (;; l++) {
        c->lw[l] = cw;
        c->lh[l] = ch;
        c->lstride[l] = level_stride(cw);
        c->loff[l] = off;
        off += (size_t)c->lstride[l] * (ch + 2 * VO_BY);
        off = (off + 255) / 256 * 256;
        int nw = (cw + 1) / 2, nh = (ch + 1) / 2;
        if (l == c->prm.lk_max_level || l + 1 >= VO_MAX_LEVELS || nw <= 21 || nh <= 21)
            break;
        cw = nw;
        ch = nh;
    }
    c->levels = l + 1;
    c->img_bytes = off;
    return 0;
}

// HIP streams of a context come from a per-device pool and go back to it in vo_destroy (they are never destroyed).
// Why: the runtime multiplexes a process's streams onto a few hardware queues in creation order, and which of a
// context's streams share a queue moves the latency-bound modes by 25 % (round-2 measurement: the SECOND vo_ctx of a
// process ran a one-sequence step in 0.80 instead of 0.63 ms).  A context created after another one was destroyed
// (bench.py's legs, a test session, a service that reconfigures) now gets the very same streams -- same mapping, same
// speed; contexts that are alive at the same time (one per host thread, examples/vo_multi_gpu.cpp) get a set each.
constexpr int VO_MAX_DEVICES = 64;
std::mutex g_pool_mu;
std::vector<StreamSet> g_pool[VO_MAX_DEVICES];
int g_pool_created[VO_MAX_DEVICES] = {};

bool acquire_streams(int device, StreamSet *out)
{
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (device < VO_MAX_DEVICES && !g_pool[device].empty()) {
            // the set that was created first: a process that only ever has one context alive always runs on the same streams
            size_t best = 0;
            for (size_t i = 1; i < g_pool[device].size(); i++)
                if (g_pool[device][i].id < g_pool[device][best].id)
                    best = i;
            *out = g_pool[device][best];
            g_pool[device].erase(g_pool[device].begin() + best);
            return true;
        }
    }
    StreamSet s;
    if (device < VO_MAX_DEVICES) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        s.id = g_pool_created[device]++;
    }
    bool ok = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess;
    // the post-LK streams carry a few hundred waves next to LK's hundred thousand: at equal priority
    // their kernels trickle in behind LK's workgroups (pose chain 1.1 ms alone, 5-10 ms next to LK) and
    // the next run ends up waiting for them, so they get the highest stream priority
    int least = 0, greatest = 0;
    ok = ok && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess;
    ok = ok && hipStreamCreateWithPriority(&s.pnp, hipStreamNonBlocking, greatest) == hipSuccess;
    ok = ok && hipStreamCreateWithPriority(&s.pnp2, hipStreamNonBlocking, greatest) == hipSuccess;
    ok = ok && hipStreamCreateWithPriority(&s.filter, hipStreamNonBlocking, greatest) == hipSuccess;
    ok = ok && hipStreamCreateWithPriority(&s.em, hipStreamNonBlocking, greatest) == hipSuccess;
    *out = s;
    return ok;
}

// The lock-step loop's copy stream, created the first time a context of this set needs it.  Two flavours: the PREPARE
// stream has the highest priority (it carries the new pairs' pyramids and FAST besides the ingest kernel, short
// memory-bound kernels that have to find SIMD slots between the running step's LK waves); the plain COPY stream only
// carries the ingest kernel and keeps the default priority -- at the highest priority that kernel runs into the current
// step's pyramid / detection / LK kernels and costs each ~0.2 ms at 256 sequences (measured, round 3: 22.4 k -> 21.1 k
// frames/s).
hipStream_t ensure_copy_stream(StreamSet *s, bool prepare, bool partitioned)
{
    if (partitioned) // (created with the rest of the partitioned set)
        return prepare ? s->part[6] : s->part[5];
    hipStream_t &st = prepare ? s->prep : s->copy;
    if (st)
        return st;
    int least = 0, greatest = 0;
    bool ok = hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess;
    ok = ok && (prepare ? hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest)
                        : hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) == hipSuccess;
    return ok ? st : nullptr;
}

// The PARTITIONED twin of a stream set (round 5): the post-LK streams get half of the GPU's compute units for themselves
// and the tracking / copy / prepare streams the other half (hipExtStreamCreateWithCUMask).  For ONE sequence in the
// lock-step loop: its step is two latency-bound chains of small kernels -- detection + LK of frame k + 1, the pose solve of
// frame k -- that otherwise land on the same CUs and slow each other (kernel trace of the loop: the refinement 152-162 us
// against 115 alone, bucketing 48 against 10, LK 230 against 200).  Measured (tools/cu_mask_ab.sh, pairs resident): 2 763 ->
// 3 115 frames/s with 128 + 128 CUs and two pose streams; 64 / 96 CUs for the pose side 3 014 / 2 978; with ONE pose stream
// the partition loses (1 445), and at 8 sequences it loses with any split (13.7 k -> 12.9 k): vo_seq_configure uses it for
// n_seq == 1 only and the schedule probe settles the rest.  part[]: stream, pnp, pnp2, filter, em, copy, prep.
bool ensure_partitioned_streams(StreamSet *s, int device)
{
    if (s->part_tried)
        return s->part[0] != nullptr;
    s->part_tried = true;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess)
        return false;
    int n_cu = prop.multiProcessorCount, pose_cus = n_cu / 2;
#ifdef VO_DEV_VARIANTS
    if (const char *e = getenv("VO_POSE_CUS")) // developer build: 0 = no partition, n = CUs of the pose side
        pose_cus = atoi(e);
#endif
    if (n_cu < 16 || n_cu > 1024 || pose_cus <= 0 || pose_cus >= n_cu)
        return false;
    std::vector<uint32_t> pose_mask((size_t)(n_cu + 31) / 32, 0u), trk_mask(pose_mask.size(), 0u);
    for (int cu = 0; cu < n_cu; cu++)
        (cu < pose_cus ? pose_mask : trk_mask)[(size_t)cu >> 5] |= 1u << (cu & 31);
    const uint32_t words = (uint32_t)pose_mask.size();
    hipStream_t t[7] = {};
    bool ok = true;
    for (int k = 0; k < 7 && ok; k++) // stream, pnp, pnp2, filter, em, copy, prep
        ok = hipExtStreamCreateWithCUMask(&t[k], words, (k == 0 || k >= 5 ? trk_mask : pose_mask).data()) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        for (hipStream_t st : t)
            if (st)
                (void)hipStreamDestroy(st);
        return false;
    }
    for (int k = 0; k < 7; k++)
        s->part[k] = t[k];
    return true;
}

// which twin of its stream set a context enqueues on; the caller has drained every stream (sync_all)
int select_streams(vo_ctx *c, bool partitioned)
{
    if (partitioned && !ensure_partitioned_streams(&c->streams, c->device))
        partitioned = false; // (no CU masks on this device / runtime: the ordinary set)
    const StreamSet &s = c->streams;
    c->partitioned = partitioned;
    c->stream = partitioned ? s.part[0] : s.stream;
    c->stream_pnp = partitioned ? s.part[1] : s.pnp;
    c->stream_pnp2 = partitioned ? s.part[2] : s.pnp2;
    c->stream_filter = partitioned ? s.part[3] : s.filter;
    c->stream_em = partitioned ? s.part[4] : s.em;
    c->last_pose_stream = nullptr;
    return VO_OK;
}

void release_streams(int device, const StreamSet &s)
{
    hipStream_t all[] = {s.stream, s.pnp, s.pnp2, s.filter, s.em, s.copy, s.prep, s.part[0], s.part[1], s.part[2], s.part[3],
                         s.part[4], s.part[5], s.part[6]};
    for (hipStream_t st : all)
        if (st)
            (void)hipStreamSynchronize(st);
    if (!s.stream || device >= VO_MAX_DEVICES)
        return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool[device].push_back(s);
}

} // namespace vo_capi

extern "C" {

void vo_default_params(vo_params *p)
{
    p->lk_max_level = 3;
    p->lk_max_count = 30;
    p->lk_epsilon = 0.01;
    p->lk_min_eig_threshold = 0.001;
    p->lk_full_chain = 0;
    p->consistency_threshold = 0;
    p->ransac_iterations = 500;
    p->ransac_reproj_error = 0.5f;
    p->ransac_confidence = (double)0.999f; // `float confidence = 0.999` in the reference
    p->mono_rotation = 0;                  // main.cpp:181 passes false
    p->em_prob = 0.999;                    // visualOdometry.cpp:152
    p->em_threshold = 1.0;                 // visualOdometry.cpp:152
}

void vo_default_detect_params(vo_detect_params *p)
{
    p->fast_threshold = 20;     // feature.cpp:43
    p->fast_nonmax = 1;         // feature.cpp:44
    p->redetect_below = 2000;   // visualOdometry.cpp:95
    p->bucket_size = 0;         // 0 = rows / 10, visualOdometry.cpp:106
    p->features_per_bucket = 1; // visualOdometry.cpp:107
}

const char *vo_last_error(const vo_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

void vo_destroy(vo_ctx *c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    if (c->stream)
        (void)sync_all(c); // nothing of this context may still run on streams that go back to the pool
    void *ptrs[] = {c->d_der, c->d_pix, c->d_imgs, c->d_quads, c->d_pts, c->d_trk2[0], c->d_trk2[1], c->d_outA,
                    c->d_status2[0], c->d_status2[1], c->d_npts, c->d_nA, c->d_idxA, c->d_P, c->d_rowoff, c->d_nmsmask, c->d_rowcnt, c->d_detect,
                    c->d_ntracked, c->d_nnew, c->d_feat, c->d_fages, c->d_ages, c->d_pts_det[0], c->d_pts_det[1],
                    c->d_npts_det[0], c->d_npts_det[1], c->d_ages_det[0], c->d_ages_det[1], c->d_overflow};
    for (void *p : ptrs)
        if (p)
            (void)hipFree(p);
    seq_free(c);
    for (auto &b : c->pb) {
        void *q[] = {b.outB, b.idxB, b.nB, b.xyz, b.subsets, b.inliers, b.models, b.counts, b.rstate, b.results,
                     b.em_results, b.epnp_ws, b.epnp_gws, b.rest_ws};
        for (void *p : q)
            if (p)
                (void)hipFree(p);
        if (b.ready)
            (void)hipEventDestroy(b.ready);
        if (b.done)
            (void)hipEventDestroy(b.done);
        if (b.tri_done)
            (void)hipEventDestroy(b.tri_done);
        if (b.em_done)
            (void)hipEventDestroy(b.em_done);
    }
    {
        void *q[] = {c->em.q0, c->em.q1, c->em.subsets, c->em.rstate, c->em.models, c->em.nmodels, c->em.counts,
                     c->em.bestE, c->em.mask};
        for (void *p : q)
            if (p)
                (void)hipFree(p);
    }
    for (auto &ev : c->ev_trk_free)
        if (ev)
            (void)hipEventDestroy(ev);
    if (c->ev_t1_ready)
        (void)hipEventDestroy(c->ev_t1_ready);
    if (c->h_stage)
        (void)hipHostFree(c->h_stage);
    if (c->h_gather)
        (void)hipHostFree(c->h_gather);
    if (c->h_pts_stage)
        (void)hipHostFree(c->h_pts_stage);
    if (c->h_feat_stage)
        (void)hipHostFree(c->h_feat_stage);
    for (auto &e : c->ev)
        if (e)
            (void)hipEventDestroy(e);
    for (auto &e : c->ring)
        if (e)
            (void)hipEventDestroy(e);
    release_streams(c->device, c->streams); // synchronises them; back to the per-device pool
    delete c;
}

vo_ctx *vo_create(int device, int max_w, int max_h, int max_pts, int max_frames)
{
    if (max_w < 32 || max_h < 32 || max_pts < 1 || max_frames < 1)
        return nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return nullptr;
    if (hipSetDevice(device) != hipSuccess)
        return nullptr;
    vo_ctx *c = new vo_ctx();
    c->device = device;
    c->max_w = max_w;
    c->max_h = max_h;
    c->cap = max_pts;
    c->max_frames = max_frames;
    c->max_images = 6 * max_frames; // 4 per frame for independent quads; 2 x ring (<= 3) per sequence of the lock-step loop
    vo_default_params(&c->prm);
    c->ransac_cap = 1000;
    const size_t B = (size_t)max_frames, cap = (size_t)max_pts;
    bool ok = acquire_streams(device, &c->streams);
    ok = ok && pnp_init_device(c->streams.stream) == 0;
    c->stream = c->streams.stream;
    c->stream_pnp = c->streams.pnp;
    c->stream_pnp2 = c->streams.pnp2;
    c->stream_filter = c->streams.filter;
    c->stream_em = c->streams.em;
    for (auto &ev : c->ev_trk_free)
        ok = ok && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_t1_ready, hipEventDisableTiming) == hipSuccess;
#ifdef VO_DEV_VARIANTS
    // developer build only (python -m visual_odom_amd.build --dev -> libvo_hip_dev.so): VO_SERIAL_POSE=1 enqueues the pose
    // solve on the tracking stream (no overlap), so that a kernel trace shows every kernel's stand-alone duration;
    // VO_LK_PAIR=1 selects the measured-slower two-features-per-wavefront LK kernel
    {
        const char *e = getenv("VO_SERIAL_POSE");
        c->serial_pose = e && e[0] == '1';
        const char *elp = getenv("VO_LK_PAIR");
        c->lk_pair = elp && elp[0] == '1';
    }
#endif
    for (auto &e : c->ev)
        ok = ok && hipEventCreate(&e) == hipSuccess;
    c->ring.assign((size_t)VO_EVENT_SLOTS * (VO_EV_PER_RUN), nullptr);
    for (auto &e : c->ring)
        ok = ok && hipEventCreate(&e) == hipSuccess;
    // worst case pyramid bytes per image (5 levels, padded strides)
    {
        size_t per = 0;
        int cw = max_w, ch = max_h;
        for (int l = 0; l < VO_MAX_LEVELS; l++) {
            per += (size_t)level_stride(cw) * (ch + 2 * VO_BY) + 256;
            cw = (cw + 1) / 2;
            ch = (ch + 1) / 2;
        }
        c->pix_capacity = per * (size_t)c->max_images;
    }
    c->stage_slot = (size_t)level_stride(max_w) * max_h;
    c->stage_slot = (c->stage_slot + 255) / 256 * 256; // (slots stay 16-byte aligned for launch_pull_image)
    ok = ok && hipHostMalloc((void **)&c->h_stage, c->stage_slot * VO_STAGE_SLOTS, hipHostMallocMapped) == hipSuccess;
    ok = ok && hipHostGetDevicePointer((void **)&c->d_stage, c->h_stage, 0) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&c->h_gather, frame_gather_bytes(c->cap), hipHostMallocMapped) == hipSuccess;
    ok = ok && hipHostGetDevicePointer((void **)&c->d_gather, c->h_gather, 0) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&c->h_pts_stage, sizeof(float2) * (size_t)c->cap + 16, hipHostMallocMapped) == hipSuccess;
    ok = ok && hipHostGetDevicePointer((void **)&c->d_pts_stage, c->h_pts_stage, 0) == hipSuccess;
    ok = ok && dmalloc(&c->d_pix, c->pix_capacity) == hipSuccess;
    ok = ok && dmalloc(&c->d_der, c->pix_capacity) == hipSuccess;
    ok = ok && dmalloc(&c->d_imgs, (size_t)c->max_images) == hipSuccess;
    ok = ok && dmalloc(&c->d_quads, B + VO_CONST_QUADS) == hipSuccess; // (the table + const_quad()'s four)
    ok = ok && dmalloc(&c->d_pts, B * cap) == hipSuccess;
    for (int k = 0; k < 2; k++) {
        ok = ok && dmalloc(&c->d_trk2[k], B * 4 * cap) == hipSuccess;
        ok = ok && dmalloc(&c->d_status2[k], B * 4 * cap) == hipSuccess;
    }
    ok = ok && dmalloc(&c->d_outA, B * 5 * cap) == hipSuccess;
    ok = ok && dmalloc(&c->d_npts, B) == hipSuccess;
    ok = ok && dmalloc(&c->d_nA, B) == hipSuccess;
    ok = ok && dmalloc(&c->d_idxA, B * cap) == hipSuccess;
    ok = ok && dmalloc(&c->d_P, (size_t)24) == hipSuccess;
    for (auto &b : c->pb) {
        ok = ok && dmalloc(&b.outB, B * 4 * cap) == hipSuccess;
        ok = ok && dmalloc(&b.nB, B) == hipSuccess;
        ok = ok && dmalloc(&b.idxB, B * cap) == hipSuccess;
        ok = ok && dmalloc(&b.xyz, B * cap * 3) == hipSuccess;
        ok = ok && dmalloc(&b.subsets, B * c->ransac_cap * 5) == hipSuccess;
        ok = ok && dmalloc(&b.inliers, B * cap) == hipSuccess;
        ok = ok && dmalloc(&b.models, B * c->ransac_cap * 6) == hipSuccess;
        ok = ok && dmalloc(&b.counts, B * c->ransac_cap) == hipSuccess;
        ok = ok && dmalloc(&b.results, B) == hipSuccess;
        ok = ok && dmalloc(&b.rstate, B) == hipSuccess;
        ok = ok && dmalloc(&b.epnp_ws, (size_t)(c->max_frames < VO_EPNP_WS_MAX_FRAMES ? c->max_frames : VO_EPNP_WS_MAX_FRAMES) *
                                           VO_EPNP_WS_HYPS * VO_EPNP_WS_DOUBLES) == hipSuccess;
        ok = ok && dmalloc(&b.rest_ws, (size_t)(c->max_frames < VO_EPNP_WS_MAX_FRAMES ? c->max_frames : VO_EPNP_WS_MAX_FRAMES) *
                                           pnp_rest_ws_doubles(c->ransac_cap)) == hipSuccess;
#ifdef VO_DEV_VARIANTS
        ok = ok && dmalloc(&b.epnp_gws, B * VO_EPNP_GWS_BLOCKS * VO_EPNP_UT_DOUBLES * 64) == hipSuccess;
#endif
        ok = ok && hipEventCreateWithFlags(&b.ready, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&b.done, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&b.tri_done, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&b.em_done, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipMemset(b.nB, 0, B * sizeof(int)) == hipSuccess;
    }
    vo_default_detect_params(&c->dprm);
    // carried + detected features of one frame before bucketing: cv::FAST on a textured image returns one corner
    // per ~20-40 pixels at most (non-maximum suppression leaves no two adjacent corners); beyond this capacity the
    // DETECT stage reports VO_ERR_OVERFLOW instead of silently bucketing a truncated list
    c->fcap = max_pts * 4 > 16384 ? max_pts * 4 : 16384;
    if ((long long)max_w * max_h / 16 > c->fcap)
        c->fcap = (int)((long long)max_w * max_h / 16);
    ok = ok && dmalloc(&c->d_nmsmask, B * (size_t)max_h * ((max_w + 63) / 64)) == hipSuccess;
    ok = ok && dmalloc(&c->d_rowcnt, B * (size_t)max_h) == hipSuccess;
    ok = ok && dmalloc(&c->d_rowoff, B * (size_t)max_h) == hipSuccess;
    ok = ok && hipMemset(c->d_rowcnt, 0, B * (size_t)max_h * sizeof(int)) == hipSuccess;
    ok = ok && dmalloc(&c->d_detect, B) == hipSuccess;
    ok = ok && dmalloc(&c->d_ntracked, B) == hipSuccess;
    ok = ok && dmalloc(&c->d_nnew, B) == hipSuccess;
    ok = ok && dmalloc(&c->d_feat, B * (size_t)c->fcap) == hipSuccess;
    ok = ok && dmalloc(&c->d_fages, B * (size_t)c->fcap) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&c->h_feat_stage, features_stage_bytes(c->fcap), hipHostMallocMapped) == hipSuccess;
    ok = ok && hipHostGetDevicePointer((void **)&c->d_feat_stage, c->h_feat_stage, 0) == hipSuccess;
    ok = ok && dmalloc(&c->d_ages, B * cap) == hipSuccess;
    ok = ok && dmalloc(&c->d_overflow, B) == hipSuccess;
    ok = ok && hipMemset(c->d_overflow, 0, B * sizeof(int)) == hipSuccess;
    for (int k = 0; k < 2; k++) {
        ok = ok && dmalloc(&c->d_pts_det[k], B * cap) == hipSuccess;
        ok = ok && dmalloc(&c->d_npts_det[k], B) == hipSuccess;
        ok = ok && dmalloc(&c->d_ages_det[k], B * cap) == hipSuccess;
        ok = ok && hipMemset(c->d_npts_det[k], 0, B * sizeof(int)) == hipSuccess;
    }
    if (ok) {
        ok = hipMemset(c->d_npts, 0, B * sizeof(int)) == hipSuccess &&
             hipMemset(c->d_nA, 0, B * sizeof(int)) == hipSuccess &&
             hipMemset(c->d_ntracked, 0, B * sizeof(int)) == hipSuccess &&
             hipMemset(c->d_nnew, 0, B * sizeof(int)) == hipSuccess &&
             hipMemset(c->d_detect, 0, B * sizeof(int)) == hipSuccess &&
             hipMemset(c->d_fages, 0, B * (size_t)c->fcap * sizeof(int)) == hipSuccess &&
             hipMemset(c->d_ages, 0, B * cap * sizeof(int)) == hipSuccess;
    }
    c->h_ntracked.assign(B, 0);
    c->h_detect.assign(B, 1);
    if (!ok) {
        vo_destroy(c);
        return nullptr;
    }
    c->h_npts.assign(B, 0);
    c->h_quads.assign(B, Quad{0, 0, 0, 0});
    c->img_stale.assign((size_t)c->max_images, 0);
    c->quads_cur = c->d_quads;
    if (hipMemcpy(c->d_quads + B, VO_CONST_QUAD_TABLE, sizeof(Quad) * VO_CONST_QUADS, hipMemcpyHostToDevice) != hipSuccess) {
        vo_destroy(c);
        return nullptr;
    }
    return c;
}

int vo_set_params(vo_ctx *c, const vo_params *p)
{
    if (!c || !p)
        return VO_ERR_ARG;
    if (p->lk_max_level < 0 || p->lk_max_level >= VO_MAX_LEVELS || p->ransac_iterations < 1 ||
        p->ransac_iterations > c->ransac_cap || !(p->ransac_confidence > 0 && p->ransac_confidence < 1) ||
        (p->mono_rotation && (!(p->em_prob > 0 && p->em_prob < 1) || !(p->em_threshold > 0))))
        return fail(c, VO_ERR_ARG, "vo_set_params: parameter out of range");
    c->prm = *p;
    c->n_images = 0; // pyramid plan depends on lk_max_level: force re-configure
    return VO_OK;
}

int64_t vo_kept_pair_id(const vo_ctx *c) { return c && c->tf_base >= 0 ? c->tf_gen : 0; }

int vo_get_params(const vo_ctx *c, vo_params *p)
{
    if (!c || !p)
        return VO_ERR_ARG;
    *p = c->prm;
    return VO_OK;
}

int vo_batch_configure(vo_ctx *c, int n_images, int w, int h, int n_frames)
{
    if (!c)
        return VO_ERR_ARG;
    if (n_images < 1 || n_images > c->max_images || n_frames < 1 || n_frames > c->max_frames || w < 32 ||
        h < 32 || w > c->max_w || h > c->max_h)
        return fail(c, VO_ERR_ARG, "vo_batch_configure: size beyond the capacity given to vo_create");
    VO_HIP_TRY(c, hipSetDevice(c->device));
    if (c->seq.on) { // leaving the lock-step sequence loop: its steps may still be in flight
        int rcs = sync_all(c);
        if (rcs != VO_OK)
            return rcs;
        c->seq.on = false;
        (void)select_streams(c, false); // (the one-sequence loop runs on the partitioned streams)
        c->quads_cur = c->d_quads;
        c->n_images = 0; // force the full re-plan below
    }
    if (c->n_images == n_images && c->w == w && c->h == h && c->n_frames == n_frames) {
        c->pyr_first = 0; // a (re)configure always restores "build every pyramid"
        c->pyr_count = n_images;
        c->quads_cur = c->d_quads; // ... and "the quads are what vo_batch_set_quads last uploaded": a synchronous drop-in call
                                   // may have left the launches reading one of its constant quadruples
        return VO_OK;
    }
    c->sched_key[0] = -1; // a new shape: the schedule is resolved again at its first run
    c->tf_base = -1;      // (and the image table is laid out again)
    c->quads_cur = c->d_quads;
    plan_levels(c, w, h);
    if (c->img_bytes * (size_t)n_images > c->pix_capacity)
        return fail(c, VO_ERR_ARG, "vo_batch_configure: pyramid storage exceeds capacity");
    std::vector<PyrImage> tab((size_t)n_images);
    for (int i = 0; i < n_images; i++) {
        memset(&tab[i], 0, sizeof(PyrImage));
        for (int l = 0; l < c->levels; l++) {
            // pointers address pixel (0, 0); VO_BY rows and VO_BX columns of border precede it
            const size_t org = (size_t)i * c->img_bytes + c->loff[l] + (size_t)VO_BY * c->lstride[l] + VO_BX;
            tab[i].lvl[l] = c->d_pix + org;
            tab[i].der[l] = c->d_der + org;
            tab[i].w[l] = c->lw[l];
            tab[i].h[l] = c->lh[l];
            tab[i].stride[l] = c->lstride[l];
        }
    }
    VO_HIP_TRY(c, hipMemcpyAsync(c->d_imgs, tab.data(), sizeof(PyrImage) * n_images, hipMemcpyHostToDevice,
                                 c->stream));
    // the Scharr border is BORDER_CONSTANT 0 (and stays 0: scharr_kernel writes interiors only)
    VO_HIP_TRY(c, hipMemsetAsync(c->d_der, 0, sizeof(uint32_t) * c->img_bytes * (size_t)n_images, c->stream));
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->n_images = n_images;
    c->pyr_first = 0;
    c->pyr_count = n_images;
    c->n_frames = n_frames;
    c->w = w;
    c->h = h;
    c->max_pts_set = 0;
    std::fill(c->h_npts.begin(), c->h_npts.end(), 0);
    std::fill(c->h_ntracked.begin(), c->h_ntracked.end(), 0);
    std::fill(c->img_stale.begin(), c->img_stale.end(), 0);
    std::fill(c->h_quads.begin(), c->h_quads.end(), Quad{0, 0, 0, 0});
    c->quads_set = false;
    VO_HIP_TRY(c, hipMemsetAsync(c->d_quads, 0, sizeof(Quad) * c->max_frames, c->stream));
    VO_HIP_TRY(c, hipMemsetAsync(c->d_overflow, 0, sizeof(int) * c->max_frames, c->stream));
    c->detect_uploaded = false; // the per-frame detect flags on the device belong to the previous batch shape
    VO_HIP_TRY(c, hipMemsetAsync(c->d_npts, 0, sizeof(int) * c->max_frames, c->stream));
    c->pts_sel = -1;
    VO_HIP_TRY(c, hipMemsetAsync(c->d_ntracked, 0, sizeof(int) * c->max_frames, c->stream));
    VO_HIP_TRY(c, hipMemsetAsync(c->d_fages, 0, sizeof(int) * (size_t)c->max_frames * c->fcap, c->stream));
    return VO_OK;
}

} // extern "C"

namespace vo_capi {

int upload_image(vo_ctx *c, int idx, const void *src, int stride, hipMemcpyKind kind, bool idle, const float *pts, int n_pts,
                 hipStream_t on)
{
    if (!c)
        return VO_ERR_ARG;
    if (c->n_images == 0)
        return fail(c, VO_ERR_STATE, "upload before vo_batch_configure");
    if (idx < 0 || idx >= c->n_images || !src || stride < c->w)
        return fail(c, VO_ERR_ARG, "vo_batch_upload_image: bad index / stride");
    if (c->seq.on)
        return fail(c, VO_ERR_STATE, "vo_batch_upload_image inside the sequence loop: use vo_seq_push_pair");
    VO_HIP_TRY(c, hipSetDevice(c->device));
    uint8_t *dst = c->d_pix + (size_t)idx * c->img_bytes + c->loff[0] + (size_t)VO_BY * c->lstride[0] + VO_BX;
    // the pyramid levels, borders and Scharr images of this image now belong to the previous pixels: LK / DETECT
    // refuse to read it until VO_STAGE_PYRAMID has covered it again (the contiguous host copy below also
    // overwrites the level-0 border columns with staging bytes)
    c->img_stale[idx] = 1;
    if (kind == hipMemcpyHostToDevice) {
        // repack to the device pitch in pinned memory, then one contiguous copy from pixel (0, 0) to the
        // last interior pixel.  The bytes between two rows land in border columns, which the pyramid
        // stage's border fill rewrites before anything reads them.
        const size_t pitch = (size_t)c->lstride[0];
        if (c->stage_next == 0 && !idle) // all slots may still be in flight from the previous round of uploads
            VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
        uint8_t *slot = c->h_stage + c->stage_slot * (size_t)c->stage_next;
        c->stage_next = (c->stage_next + 1) % VO_STAGE_SLOTS;
        for (int y = 0; y < c->h; y++)
            memcpy(slot + (size_t)y * pitch, (const uint8_t *)src + (size_t)y * stride, (size_t)c->w);
        const size_t bytes = pitch * (size_t)(c->h - 1) + (size_t)c->w;
        if (idle) { // a synchronous drop-in call: the GPU pulls the slot itself (pyramid.hip, launch_pull_image; the 16-byte
                    // round-up of the last row stays inside the row's pitch)
            if (n_pts >= 0) { // the call's points (frame 0) and their count with this image
                if (n_pts > 0)
                    memcpy(c->h_pts_stage, pts, sizeof(float2) * (size_t)n_pts);
                launch_pull_image(c->d_stage + (slot - c->h_stage), dst, bytes, on ? on : c->stream, c->d_pts_stage, c->d_pts, n_pts, c->d_npts);
            } else {
                launch_pull_image(c->d_stage + (slot - c->h_stage), dst, bytes, on ? on : c->stream);
            }
            VO_HIP_TRY(c, hipGetLastError());
        } else {
            VO_HIP_TRY(c, hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, c->stream));
        }
        return VO_OK;
    }
    VO_HIP_TRY(c, hipMemcpy2DAsync(dst, (size_t)c->lstride[0], src, (size_t)stride, (size_t)c->w, (size_t)c->h,
                                   kind, c->stream));
    return VO_OK;
}

} // namespace vo_capi

extern "C" {

int vo_batch_upload_image(vo_ctx *c, int idx, const uint8_t *host, int stride)
{
    if (c)
        c->tf_base = -1; // the caller owns the image table now (see vo_ctx::tf_base)
    int rc = upload_image(c, idx, host, stride, hipMemcpyHostToDevice);
    if (rc == VO_OK) // pageable host memory: make the call safe to return from
        VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return rc;
}

int vo_batch_upload_image_dev(vo_ctx *c, int idx, const void *dev, int stride)
{
    if (c)
        c->tf_base = -1;
    return upload_image(c, idx, dev, stride, hipMemcpyDeviceToDevice);
}

int vo_batch_set_quads(vo_ctx *c, const int32_t *quads4, int n_frames)
{
    if (!c || !quads4)
        return VO_ERR_ARG;
    if (c->n_images == 0 || n_frames != c->n_frames)
        return fail(c, VO_ERR_STATE, "vo_batch_set_quads: configure first / frame count mismatch");
    for (int i = 0; i < 4 * n_frames; i++)
        if (quads4[i] < 0 || quads4[i] >= c->n_images)
            return fail(c, VO_ERR_ARG, "vo_batch_set_quads: image index out of range");
    if (c->seq.on)
        return fail(c, VO_ERR_STATE, "vo_batch_set_quads inside the sequence loop");
    if (c->quads_set && memcmp(c->h_quads.data(), quads4, sizeof(Quad) * n_frames) == 0)
        return VO_OK; // the table on the device already says so (the single-frame calls set {0, 1, 2, 3} every time)
    VO_HIP_TRY(c, hipSetDevice(c->device));
    VO_HIP_TRY(c, hipMemcpyAsync(c->d_quads, quads4, sizeof(Quad) * n_frames, hipMemcpyHostToDevice, c->stream));
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    memcpy(c->h_quads.data(), quads4, sizeof(Quad) * n_frames);
    c->quads_set = true;
    c->quads_cur = c->d_quads; // (a synchronous drop-in call may have left it at one of its constant quadruples)
    return VO_OK;
}

int vo_batch_set_pyramid_range(vo_ctx *c, int first_image, int n_images)
{
    if (!c)
        return VO_ERR_ARG;
    if (c->n_images == 0)
        return fail(c, VO_ERR_STATE, "vo_batch_set_pyramid_range before vo_batch_configure");
    if (first_image < 0 || n_images < 0 || first_image + n_images > c->n_images)
        return fail(c, VO_ERR_ARG, "vo_batch_set_pyramid_range: range outside the image table");
    c->pyr_first = first_image;
    c->pyr_count = n_images;
    return VO_OK;
}

int vo_batch_set_points(vo_ctx *c, int frame, const float *pts, int n)
{
    if (!c)
        return VO_ERR_ARG;
    if (c->seq.on)
        return fail(c, VO_ERR_STATE, "vo_batch_set_points inside the sequence loop (vo_seq_* owns the feature state)");
    if (frame < 0 || frame >= c->n_frames || n < 0 || n > c->cap || (n > 0 && !pts))
        return fail(c, VO_ERR_ARG, "vo_batch_set_points: bad frame / more points than max_pts");
    int rcs = sync_all(c); // a queued filter of the previous run still reads the points
    if (rcs != VO_OK)
        return rcs;
    if (c->pts_sel >= 0) { // the other frames keep what the last DETECT stage gave them
        const size_t B = (size_t)c->max_frames, cap = (size_t)c->cap;
        VO_HIP_TRY(c, hipMemcpyAsync(c->d_pts, cur_pts(c), sizeof(float2) * B * cap, hipMemcpyDeviceToDevice, c->stream));
        VO_HIP_TRY(c, hipMemcpyAsync(c->d_ages, cur_ages(c), sizeof(int) * B * cap, hipMemcpyDeviceToDevice, c->stream));
        VO_HIP_TRY(c, hipMemcpyAsync(c->d_npts, cur_npts(c), sizeof(int) * B, hipMemcpyDeviceToDevice, c->stream));
        VO_HIP_TRY(c, hipMemcpyAsync(c->h_npts.data(), c->d_npts, sizeof(int) * c->n_frames, hipMemcpyDeviceToHost,
                                     c->stream));
        VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->pts_sel = -1;
    }
    if (n > 0)
        VO_HIP_TRY(c, hipMemcpyAsync(c->d_pts + (size_t)frame * c->cap, pts, sizeof(float2) * n,
                                     hipMemcpyHostToDevice, c->stream));
    c->h_npts[frame] = n;
    c->pts_on_device = false;
    VO_HIP_TRY(c, hipMemcpyAsync(c->d_npts + frame, &c->h_npts[frame], sizeof(int), hipMemcpyHostToDevice,
                                 c->stream));
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->max_pts_set = 0;
    for (int f = 0; f < c->n_frames; f++)
        c->max_pts_set = c->h_npts[f] > c->max_pts_set ? c->h_npts[f] : c->max_pts_set;
    return VO_OK;
}

int vo_batch_set_features(vo_ctx *c, int frame, const float *pts, int n_pts, const int32_t *ages, int n_ages)
{
    if (!c)
        return VO_ERR_ARG;
    if (c->n_images == 0)
        return fail(c, VO_ERR_STATE, "vo_batch_set_features before vo_batch_configure");
    if (c->seq.on)
        return fail(c, VO_ERR_STATE, "vo_batch_set_features inside the sequence loop (vo_seq_* owns the feature state)");
    if (frame < 0 || frame >= c->n_frames || n_pts < 0 || n_ages < n_pts || n_ages > c->fcap ||
        (n_pts > 0 && !pts) || (n_ages > 0 && !ages))
        return fail(c, VO_ERR_ARG, "vo_batch_set_features: bad frame / counts (need n_pts <= n_ages <= capacity)");
    int rcs = sync_all(c);
    if (rcs != VO_OK)
        return rcs;
    // ages beyond n_ages must read as 0 (age of a freshly appended corner, feature.cpp:260)
    VO_HIP_TRY(c, hipMemsetAsync(c->d_fages + (size_t)frame * c->fcap, 0, sizeof(int) * (size_t)c->fcap, c->stream));
    if (n_pts > 0)
        VO_HIP_TRY(c, hipMemcpyAsync(c->d_feat + (size_t)frame * c->fcap, pts, sizeof(float2) * n_pts,
                                     hipMemcpyHostToDevice, c->stream));
    if (n_ages > 0)
        VO_HIP_TRY(c, hipMemcpyAsync(c->d_fages + (size_t)frame * c->fcap, ages, sizeof(int) * n_ages,
                                     hipMemcpyHostToDevice, c->stream));
    c->h_ntracked[frame] = n_pts;
    VO_HIP_TRY(c, hipMemcpyAsync(c->d_ntracked + frame, &c->h_ntracked[frame], sizeof(int), hipMemcpyHostToDevice,
                                 c->stream));
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VO_OK;
}

int vo_batch_set_detect_params(vo_ctx *c, const vo_detect_params *dp)
{
    if (!c)
        return VO_ERR_ARG;
    if (!dp) {
        vo_default_detect_params(&c->dprm);
        return VO_OK;
    }
    if (dp->features_per_bucket < 1 || dp->features_per_bucket > 8 || dp->bucket_size < 0)
        return fail(c, VO_ERR_ARG, "vo_batch_set_detect_params: features_per_bucket must be 1..8, bucket_size >= 0");
    c->dprm = *dp;
    return VO_OK;
}

int vo_batch_get_features(vo_ctx *c, int frame, float *pts, int32_t *ages, int *n)
{
    if (!c || !n)
        return VO_ERR_ARG;
    if (frame < 0 || frame >= c->n_frames)
        return fail(c, VO_ERR_ARG, "vo_batch_get_features: bad frame");
    VO_HIP_TRY(c, hipSetDevice(c->device));
    int k = 0;
    VO_HIP_TRY(c, hipMemcpyAsync(&k, cur_npts(c) + frame, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (pts && k > 0)
        VO_HIP_TRY(c, hipMemcpyAsync(pts, cur_pts(c) + (size_t)frame * c->cap, sizeof(float2) * k, hipMemcpyDeviceToHost,
                                     c->stream));
    if (ages && k > 0)
        VO_HIP_TRY(c, hipMemcpyAsync(ages, cur_ages(c) + (size_t)frame * c->cap, sizeof(int) * k, hipMemcpyDeviceToHost,
                                     c->stream));
    int ovf = 0;
    VO_HIP_TRY(c, hipMemcpyAsync(&ovf, c->d_overflow + frame, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n = k;
    if (ovf)
        return fail(c, VO_ERR_OVERFLOW, ovf & 1 ? "VO_STAGE_DETECT: carried + detected features exceed the feature-list "
                                                  "capacity (4 x max_pts, >= 16384, >= w * h / 16): bucketed set truncated"
                                                : "VO_STAGE_DETECT: the bucketed set exceeds max_pts");
    return VO_OK;
}

int vo_batch_set_projection(vo_ctx *c, const float *P_l, const float *P_r)
{
    if (!c || !P_l || !P_r)
        return VO_ERR_ARG;
    if (c->have_P && memcmp(c->h_P, P_l, 12 * sizeof(float)) == 0 && memcmp(c->h_P + 12, P_r, 12 * sizeof(float)) == 0)
        return VO_OK; // unchanged since the last call (every vo_track_frame passes the same calibration)
    memcpy(c->h_P, P_l, 12 * sizeof(float));
    memcpy(c->h_P + 12, P_r, 12 * sizeof(float));
    VO_HIP_TRY(c, hipSetDevice(c->device));
    VO_HIP_TRY(c, hipMemcpyAsync(c->d_P, c->h_P, 24 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->have_P = true;
    return VO_OK;
}

} // extern "C"

namespace vo_capi {

// working set of the essential-matrix chain, allocated the first time it is asked for
int ensure_em(vo_ctx *c)
{
    if (c->em_ready)
        return VO_OK;
    VO_HIP_TRY(c, hipSetDevice(c->device));
    const size_t B = (size_t)c->max_frames, cap = (size_t)c->cap;
    VO_HIP_TRY(c, hipMalloc((void **)&c->em.q0, sizeof(double2) * B * cap));
    VO_HIP_TRY(c, hipMalloc((void **)&c->em.q1, sizeof(double2) * B * cap));
    VO_HIP_TRY(c, hipMalloc((void **)&c->em.subsets, sizeof(int32_t) * B * EM_MAX_ITERS * 5));
    VO_HIP_TRY(c, hipMalloc((void **)&c->em.rstate, sizeof(RansacState) * B));
    VO_HIP_TRY(c, hipMalloc((void **)&c->em.models, sizeof(double) * B * 128 * 90));
    VO_HIP_TRY(c, hipMalloc((void **)&c->em.nmodels, sizeof(int) * B * 128));
    VO_HIP_TRY(c, hipMalloc((void **)&c->em.counts, sizeof(int) * B * 128 * 10));
    VO_HIP_TRY(c, hipMalloc((void **)&c->em.bestE, sizeof(double) * B * 9));
    VO_HIP_TRY(c, hipMalloc((void **)&c->em.mask, B * cap));
    for (auto &b : c->pb) {
        VO_HIP_TRY(c, hipMalloc((void **)&b.em_results, sizeof(EmResult) * B));
        VO_HIP_TRY(c, hipMemset(b.em_results, 0, sizeof(EmResult) * B));
    }
    c->em_ready = true;
    return VO_OK;
}

} // namespace vo_capi

extern "C" {

int vo_batch_run(vo_ctx *c, int stages)
{
    if (!c)
        return VO_ERR_ARG;
    if (c->seq.on)
        return fail(c, VO_ERR_STATE, "vo_batch_run inside the sequence loop: use vo_seq_step");
    return run_stages_auto(c, stages, false);
}

int vo_batch_run_timed(vo_ctx *c, int stages, float *ms)
{
    if (!c || !ms)
        return VO_ERR_ARG;
    if (c->seq.on)
        return fail(c, VO_ERR_STATE, "vo_batch_run_timed inside the sequence loop: use vo_seq_step");
    int rc = run_stages_auto(c, stages, true);
    if (rc != VO_OK)
        return rc;
    rc = sync_all(c);
    if (rc != VO_OK)
        return rc;
    for (int s = 0; s < VO_NUM_STAGES; s++) // PYRAMID, DETECT, LK on the tracking stream; the rest on the post stream
        VO_HIP_TRY(c, hipEventElapsedTime(&ms[s], c->ev[s == 1 ? VO_NUM_STAGES + 2 : s < 3 ? s : s + 1], c->ev[s < 3 ? s + 1 : s + 2]));
    return VO_OK;
}

int vo_batch_run_slot(vo_ctx *c, int stages, int slot)
{
    if (!c || slot < 0 || slot >= VO_EVENT_SLOTS)
        return VO_ERR_ARG;
    if (c->seq.on)
        return fail(c, VO_ERR_STATE, "vo_batch_run_slot inside the sequence loop: use vo_seq_step");
    return run_stages_auto(c, stages, true, &c->ring[(size_t)slot * (VO_EV_PER_RUN)]);
}

int vo_batch_slot_times(vo_ctx *c, int slot, float *ms)
{
    if (!c || !ms || slot < 0 || slot >= VO_EVENT_SLOTS)
        return VO_ERR_ARG;
    VO_HIP_TRY(c, hipSetDevice(c->device));
    hipEvent_t *evs = &c->ring[(size_t)slot * (VO_EV_PER_RUN)];
    for (int s = 0; s < VO_NUM_STAGES; s++) // PYRAMID, DETECT, LK on the tracking stream; the rest on the post stream
        VO_HIP_TRY(c, hipEventElapsedTime(&ms[s], evs[s == 1 ? VO_NUM_STAGES + 2 : s < 3 ? s : s + 1], evs[s < 3 ? s + 1 : s + 2]));
    return VO_OK;
}

int vo_batch_sync(vo_ctx *c)
{
    if (!c)
        return VO_ERR_ARG;
    return sync_all(c);
}

int vo_batch_get_tracks(vo_ctx *c, int frame, float *r0, float *r1, float *l1, float *l0_ret, uint8_t *status4,
                        int n)
{
    if (!c)
        return VO_ERR_ARG;
    if (frame < 0 || frame >= c->n_frames || n < 0 || n > c->cap)
        return fail(c, VO_ERR_ARG, "vo_batch_get_tracks: bad frame / n");
    VO_HIP_TRY(c, hipSetDevice(c->device));
    const size_t cap = c->cap;
    const float2 *t = c->d_trk2[c->trk_last] + (size_t)frame * 4 * cap;
    D2H(r0, t, sizeof(float2) * n);
    D2H(r1, t + cap, sizeof(float2) * n);
    D2H(l1, t + 2 * cap, sizeof(float2) * n);
    D2H(l0_ret, t + 3 * cap, sizeof(float2) * n);
    if (status4)
        for (int hop = 0; hop < 4; hop++)
            D2H(status4 + (size_t)hop * n, c->d_status2[c->trk_last] + ((size_t)frame * 4 + hop) * cap, (size_t)n);
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VO_OK;
}

int vo_batch_get_filtered(vo_ctx *c, int frame, float *l0, float *r0, float *l1, float *r1, float *xyz,
                          int32_t *keep_idx, int *n_out, int32_t *keep_idx_circ, int *n_circ)
{
    if (!c)
        return VO_ERR_ARG;
    if (frame < 0 || frame >= c->n_frames)
        return fail(c, VO_ERR_ARG, "vo_batch_get_filtered: bad frame");
    int rcs = sync_all(c);
    if (rcs != VO_OK)
        return rcs;
    const vo_ctx::PoseBufs &pb = c->pb[c->last];
    int nAB[2] = {0, 0};
    VO_HIP_TRY(c, hipMemcpyAsync(&nAB[0], c->d_nA + frame, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    VO_HIP_TRY(c, hipMemcpyAsync(&nAB[1], pb.nB + frame, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    const size_t cap = c->cap;
    const int K = nAB[1], M = nAB[0];
    const float2 *b = pb.outB + (size_t)frame * 4 * cap;
    D2H(l0, b, sizeof(float2) * K);
    D2H(r0, b + cap, sizeof(float2) * K);
    D2H(l1, b + 2 * cap, sizeof(float2) * K);
    D2H(r1, b + 3 * cap, sizeof(float2) * K);
    D2H(xyz, pb.xyz + (size_t)frame * cap * 3, sizeof(float) * 3 * K);
    D2H(keep_idx, pb.idxB + (size_t)frame * cap, sizeof(int32_t) * K);
    D2H(keep_idx_circ, c->d_idxA + (size_t)frame * cap, sizeof(int32_t) * M);
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (n_out)
        *n_out = K;
    if (n_circ)
        *n_circ = M;
    return VO_OK;
}

} // extern "C"

namespace vo_capi {

// pnp_rotation: R = Rodrigues(rvec) even under mono_rotation (vo_pnp_ransac).  em_status (optional): status of the
// essential-matrix side of the frame under mono_rotation (1 ok, 0 no model, -1 too few points), 1 otherwise.
int get_pose_impl(vo_ctx *c, int frame, double *rvec, double *tvec, double *R, int32_t *inliers,
                         int *n_inliers, int *status, int32_t *dbg4, bool pnp_rotation, int *em_status, bool io_pose)
{
    if (!c)
        return VO_ERR_ARG;
    if (frame < 0 || frame >= c->n_frames)
        return fail(c, VO_ERR_ARG, "vo_batch_get_pose: bad frame");
    int rcs = sync_all(c);
    if (rcs != VO_OK)
        return rcs;
    const vo_ctx::PoseBufs &pb = c->pb[c->last];
    PnpResult r;
    VO_HIP_TRY(c, hipMemcpyAsync(&r, pb.results + frame, sizeof(r), hipMemcpyDeviceToHost, c->stream));
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (r.status == 0 && r.lm_iters < 0 && io_pose) {
        // four points, P3P without a solution: solvePnP returned false and never wrote rvec / tvec -- the caller's
        // buffers stay as they are; the reference then still runs Rodrigues(rvec, rotation) on what rvec holds.
        // (io_pose = false, vo_batch_get_pose: its rvec / tvec / R are pure OUTPUTS -- a batch frame has no caller pose -- and
        // receive what the kernel left: rvec = 0 like the reference's `cv::Mat::zeros` at visualOdometry.cpp:162, tvec = 0,
        // R = identity, never a Rodrigues of uninitialised memory)
        if (R && rvec && (pnp_rotation || !c->prm.mono_rotation))
            rodrigues_v2m(rvec, R, nullptr);
    } else if (r.status >= 0) {
        if (rvec)
            memcpy(rvec, r.rvec, sizeof(r.rvec));
        if (tvec)
            memcpy(tvec, r.tvec, sizeof(r.tvec));
        if (R && (pnp_rotation || !c->prm.mono_rotation))
            memcpy(R, r.R, sizeof(r.R)); // `if (!mono_rotation) Rodrigues(rvec, rotation)` (visualOdometry.cpp:186-189)
    }
    if (em_status)
        *em_status = 1;
    if (!pnp_rotation && c->prm.mono_rotation && c->em_ready) {
        // rotation = recoverPose's; left untouched when no essential matrix was found (OpenCV throws there)
        EmResult e;
        VO_HIP_TRY(c, hipMemcpy(&e, pb.em_results + frame, sizeof(e), hipMemcpyDeviceToHost));
        if (e.status == 1 && R)
            memcpy(R, e.R, sizeof(e.R));
        if (em_status)
            *em_status = e.status;
    }
    if (inliers && r.n_inliers > 0) {
        D2H(inliers, pb.inliers + (size_t)frame * c->cap, sizeof(int32_t) * r.n_inliers);
        VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    if (n_inliers)
        *n_inliers = r.n_inliers;
    if (status)
        *status = r.status;
    if (dbg4) {
        dbg4[0] = r.niters;
        dbg4[1] = r.best_iter;
        dbg4[2] = r.max_good;
        dbg4[3] = r.lm_iters;
    }
    return VO_OK;
}

} // namespace vo_capi

extern "C" {

int vo_batch_get_pose(vo_ctx *c, int frame, double *rvec, double *tvec, double *R, int32_t *inliers,
                      int *n_inliers, int *status, int32_t *dbg4)
{
    return get_pose_impl(c, frame, rvec, tvec, R, inliers, n_inliers, status, dbg4, false, nullptr, /*io_pose*/ false);
}

int vo_batch_get_essential(vo_ctx *c, int frame, double *E, double *R, double *t, uint8_t *mask, int n,
                           int *n_inliers, int *n_good, int *status, int32_t *dbg2)
{
    if (!c)
        return VO_ERR_ARG;
    if (frame < 0 || frame >= c->n_frames || n < 0 || n > c->cap)
        return fail(c, VO_ERR_ARG, "vo_batch_get_essential: bad frame / n");
    if (!c->em_ready)
        return fail(c, VO_ERR_STATE, "vo_batch_get_essential: no run with vo_params.mono_rotation yet");
    int rcs = sync_all(c);
    if (rcs != VO_OK)
        return rcs;
    EmResult e;
    VO_HIP_TRY(c, hipMemcpy(&e, c->pb[c->last].em_results + frame, sizeof(e), hipMemcpyDeviceToHost));
    if (e.status == 1) {
        if (E)
            memcpy(E, e.E, sizeof(e.E));
        if (R)
            memcpy(R, e.R, sizeof(e.R));
        if (t)
            memcpy(t, e.t, sizeof(e.t));
        if (mask && n > 0)
            VO_HIP_TRY(c, hipMemcpy(mask, c->em.mask + (size_t)frame * c->cap, (size_t)n, hipMemcpyDeviceToHost));
    }
    if (n_inliers)
        *n_inliers = e.n_inliers;
    if (n_good)
        *n_good = e.n_good;
    if (status)
        *status = e.status;
    if (dbg2) {
        dbg2[0] = e.niters;
        dbg2[1] = e.best;
    }
    return VO_OK;
}

int vo_essential_pose(vo_ctx *c, const float *pts0, const float *pts1, int n, double focal, double ppx, double ppy,
                      double prob, double threshold, double *E, double *R, double *t, uint8_t *mask, int *n_good)
{
    if (!c || n < 0 || (n > 0 && (!pts0 || !pts1)) || !(prob > 0 && prob < 1) || !(threshold > 0) || !(focal != 0))
        return VO_ERR_ARG;
    if (n > c->cap)
        return fail(c, VO_ERR_ARG, "more points than max_pts given to vo_create");
    int rcs = sync_all(c);
    if (rcs != VO_OK)
        return rcs;
    rcs = ensure_em(c);
    if (rcs != VO_OK)
        return rcs;
    vo_ctx::PoseBufs &pb = c->pb[c->last];
    const size_t cap = (size_t)c->cap;
    if (n > 0) {
        VO_HIP_TRY(c, hipMemcpyAsync(pb.outB, pts0, sizeof(float2) * n, hipMemcpyHostToDevice, c->stream));
        VO_HIP_TRY(c, hipMemcpyAsync(pb.outB + 2 * cap, pts1, sizeof(float2) * n, hipMemcpyHostToDevice, c->stream));
    }
    VO_HIP_TRY(c, hipMemcpyAsync(pb.nB, &n, sizeof(int), hipMemcpyHostToDevice, c->stream));
    if (c->n_frames < 1)
        c->n_frames = 1;
    EmParams ep;
    ep.focal = focal;
    ep.ppx = ppx;
    ep.ppy = ppy;
    ep.prob = prob;
    ep.threshold = threshold;
    ep.max_iters = EM_MAX_ITERS;
    launch_essential(pb.outB, pb.outB + 2 * cap, 4 * cap, pb.nB, c->cap, 1, ep, c->em, pb.em_results,
                     /*crowded*/ standalone_waves(c) >= 2, c->stream);
    VO_HIP_TRY(c, hipGetLastError());
    int status = 0, good = 0;
    int rc = vo_batch_get_essential(c, 0, E, R, t, mask, n, nullptr, &good, &status, nullptr);
    if (rc != VO_OK)
        return rc;
    if (n_good)
        *n_good = good;
    if (status < 0)
        return fail(c, VO_ERR_TOO_FEW, "fewer than 5 correspondences reached findEssentialMat");
    return status == 1 ? VO_OK : 1;
}

int vo_batch_get_pyramid_level(vo_ctx *c, int idx, int level, uint8_t *out, int *w_l, int *h_l)
{
    if (!c)
        return VO_ERR_ARG;
    if (idx < 0 || idx >= c->n_images || level < 0 || level >= c->levels)
        return fail(c, VO_ERR_ARG, "vo_batch_get_pyramid_level: bad image / level");
    VO_HIP_TRY(c, hipSetDevice(c->device));
    if (w_l)
        *w_l = c->lw[level];
    if (h_l)
        *h_l = c->lh[level];
    if (out) {
        VO_HIP_TRY(c, hipMemcpy2DAsync(out, (size_t)c->lw[level],
                                       c->d_pix + (size_t)idx * c->img_bytes + c->loff[level] +
                                           (size_t)VO_BY * c->lstride[level] + VO_BX,
                                       (size_t)c->lstride[level], (size_t)c->lw[level], (size_t)c->lh[level],
                                       hipMemcpyDeviceToHost, c->stream));
        VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return VO_OK;
}

int vo_model_bytes(const vo_ctx *c, int w, int h, int n_points, double *b)
{
    if (!c || !b || w < 1 || h < 1 || n_points < 0)
        return VO_ERR_ARG;
    // B_pyr = 4 images x (read sum_{l<L} S_l + write sum_{l>=1} S_l)
    int L = c->prm.lk_max_level, cw = w, ch = h;
    double rd = 0, wr = 0;
    int lv = 0;
    for (int l = 0; l <= L; l++) {
        double S = (double)cw * ch;
        if (l < L)
            rd += S;
        if (l >= 1)
            wr += S;
        lv = l;
        int nw = (cw + 1) / 2, nh = (ch + 1) / 2;
        if (nw <= 21 || nh <= 21)
            break;
        cw = nw;
        ch = nh;
    }
    (void)lv;
    b[0] = 4.0 * (rd + wr);
    // B_lk = 4 hops x N x [(L+1) x ((win+3)^2 + (win+1)^2) + 8 + 8 + 1]
    b[1] = 4.0 * n_points * ((L + 1) * (576.0 + 484.0) + 17.0);
    b[2] = 48.0 * n_points + 48.0;
    return VO_OK;
}

} // extern "C"
