// capi_internal.h -- what the translation units of the C ABI share: the context record, the stream set, small helpers and the
// internal functions one unit calls in another.  capi.hip: context, allocation, batch API and getters; capi_run.hip: the stage
// launcher (run_stages); capi_sched.hip: the schedule probe and its export / import; capi_seq.hip: the lock-step sequence
// loop; capi_dropin.hip: the synchronous drop-in calls.  Round 4 split capi.hip (2 800 lines in one unit)
// without changing behaviour.  Nothing declared here is exported: internal functions live in namespace vo_capi.
#pragma once
#include "../../include/vo_hip.h"
#include "vo_kernels.h"
#include "vo_integrate.h"
#include "vo_linalg.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace vo;

#define VO_SEQ_INFLIGHT 8 // steps the host may run ahead of the device
// timing events of one run: [0..3] tracking stream (3 stages), [4..7] post streams (3 stages), [8] start of DETECT on
// the tracking stream (differs from [1] when the pyramid stage runs on the lock-step loop's prepare stream)
#define VO_EV_PER_RUN (VO_NUM_STAGES + 3)
#define VO_SEQ_MAX_RING 3

// the HIP streams of one context (pooled per device, see acquire_streams)
struct StreamSet {
    hipStream_t stream = nullptr, pnp = nullptr, pnp2 = nullptr, filter = nullptr, em = nullptr;
    hipStream_t copy = nullptr, prep = nullptr; // lock-step loop: plain copy stream / highest-priority prepare stream
    // the PARTITIONED twin (capi.hip, ensure_partitioned_streams): stream, pnp, pnp2, filter, em, copy, prep on two disjoint
    // halves of the compute units -- the one-sequence lock-step loop runs on it
    hipStream_t part[7] = {};
    bool part_tried = false;
    int id = 0; // creation rank on its device: the pool hands out the oldest free set first
};

struct vo_ctx {
    int device = 0;
    StreamSet streams;
    int max_w = 0, max_h = 0, cap = 0, max_frames = 0, max_images = 0;
    vo_params prm;
    hipStream_t stream = nullptr; // tracking stream; all streams belong to `streams` (pooled per device)
    hipEvent_t ev[VO_EV_PER_RUN] = {}; // [0..3] tracking stream (3 stages), [4..7] post stream (3 stages)
    std::vector<hipEvent_t> ring; // VO_EVENT_SLOTS x (VO_EV_PER_RUN) for vo_batch_run_slot
    std::string err;

    // batch configuration
    int n_images = 0, n_frames = 0, w = 0, h = 0, levels = 0; // levels = max_level + 1 actually built
    int lw[VO_MAX_LEVELS] = {}, lh[VO_MAX_LEVELS] = {}, lstride[VO_MAX_LEVELS] = {};
    size_t loff[VO_MAX_LEVELS] = {}, img_bytes = 0;
    int pyr_first = 0, pyr_count = 0; // image range VO_STAGE_PYRAMID rebuilds
    int max_pts_set = 0; // largest n over the frames of the batch (or its bound after VO_STAGE_DETECT)
    bool pts_on_device = false, detect_uploaded = false;

    // device memory
    uint8_t *d_pix = nullptr;  // all bordered pyramids, image i at d_pix + i * img_bytes
    uint32_t *d_der = nullptr; // all Scharr pyramids (one dword per pixel), image i at d_der + i * img_bytes
    size_t pix_capacity = 0;   // in pixels (bytes of d_pix, dwords of d_der)
    PyrImage *d_imgs = nullptr;
    Quad *d_quads = nullptr;
    float2 *d_pts = nullptr, *d_outA = nullptr;
    // LK outputs (4 hops of positions + status per frame) are double-buffered: LK of run k + 1 writes one set
    // while the filter of run k still reads the other, so the tracking stream never idles behind the filter
    float2 *d_trk2[2] = {};
    uint8_t *d_status2[2] = {};
    hipEvent_t ev_trk_free[2] = {}; // recorded after the filter has read that set (and d_pts)
    bool trk_busy[2] = {};
    int trk_next = 0, trk_last = 0; // set the next LK writes / set the latest LK wrote
    // The bucketed feature set VO_STAGE_DETECT produces belongs to the same set as the tracks made from it, so
    // DETECT of run k + 1 never waits for the filter of run k either.  pts_sel = -1: the current features are
    // the host-set ones (d_pts / d_npts / d_ages, vo_batch_set_points); else the DETECT output of that set.
    float2 *d_pts_det[2] = {};
    int *d_npts_det[2] = {}, *d_ages_det[2] = {};
    int pts_sel = -1;
    int *d_npts = nullptr, *d_nA = nullptr, *d_idxA = nullptr;
    float *d_P = nullptr; // d_P: P_l (12) then P_r (12)
    // Everything the pose solve reads or writes exists twice: the PnP/RANSAC chain of batch k runs on
    // its own stream while the tracking stages of batch k + 1 already fill the other set.
    struct PoseBufs {
        float2 *outB = nullptr;  // [B][4][cap] l0, r0, l1, r1 after the consistency filter
        int *idxB = nullptr, *nB = nullptr;
        float *xyz = nullptr;
        int32_t *subsets = nullptr, *inliers = nullptr;
        double *models = nullptr;
        int *counts = nullptr;
        RansacState *rstate = nullptr;
        PnpResult *results = nullptr;
        EmResult *em_results = nullptr; // mono_rotation branch (allocated with the rest of `em` on first use)
        double *epnp_ws = nullptr;      // workspace of the four-kernel EPnP (small launches, pnp.hip)
        double *rest_ws = nullptr;      // ransac_rest_kernel's 12 x 12 matrices [min(max_frames, VO_EPNP_WS_MAX_FRAMES)][groups][156][64]
        double *epnp_gws = nullptr;     // developer build: the slim chain's 12 x 12 matrices [max_frames][VO_EPNP_GWS_BLOCKS][156][64]
        hipEvent_t ready = nullptr, tri_done = nullptr, done = nullptr; // LK done / triangulation done / pose solve done
        hipEvent_t em_done = nullptr; // essential-matrix chain done (mono_rotation)
        bool pending = false;                        // `done` has been recorded and not waited for
    } pb[2];
    int cur = 0, last = 0; // set the next run writes / set the last run wrote
    // findEssentialMat + recoverPose working set (vo_params.mono_rotation / vo_essential_pose): one copy, only
    // ever touched on the pose stream, allocated on first use
    EmBufs em;
    bool em_ready = false;
    // detection / bucketing (VO_STAGE_DETECT)
    vo_detect_params dprm;
    int fcap = 0;                  // capacity of the carried + detected feature list of a frame
    unsigned long long *d_nmsmask = nullptr; // [B][max_h][ceil(max_w / 64)] NMS keep ballots
    int *d_rowcnt = nullptr;       // [B][max_h] corners per image row (zero between launches)
    int *d_rowoff = nullptr;       // [B][max_h] exclusive row offsets
    int *d_detect = nullptr, *d_ntracked = nullptr, *d_nnew = nullptr; // [B]
    float2 *d_feat = nullptr;      // [B][fcap] carried features, then the new corners
    int *d_fages = nullptr;        // [B][fcap] ages of d_feat (zero beyond the uploaded ages)
    int *d_ages = nullptr;         // [B][cap] ages of the bucketed set (parallel to d_pts)
    std::vector<int> h_ntracked, h_detect;
    hipStream_t stream_pnp = nullptr, stream_filter = nullptr;
    // second pose stream: in a SMALL batch the pose chain is a few latency-bound waves (1.0-1.3 ms for one frame) and
    // longer than the tracking stages of the next run, so back-to-back runs were throttled by it (lock-step loop with
    // one sequence: 1.35 ms per step, of which 1.3 ms waiting behind the previous step's chain).  Runs alternate between
    // the two buffer sets anyway; giving each set its own stream lets two chains overlap.  Whether that pays is part of
    // the SCHEDULE, which is probed, not looked up (see Schedule below).
    hipStream_t stream_pnp2 = nullptr;
    // How the pose chain is scheduled next to the tracking stages -- three knobs, none of which changes a result:
    //   waves   register budget of the f64 pose kernels as waves per SIMD: 1 = 512 registers (fastest alone, but such a
    //           wave only starts on a completely empty SIMD and keeps the next run's kernels waiting), 2 = 256 registers
    //   streams 1 or 2 pose streams (2: the chains of consecutive runs overlap)
    //   prep    lock-step loop only: the new pairs' pyramids + FAST of their left images on the prepare stream, one step
    //           ahead and off the tracking stream's critical path
    // Round 2 chose them from a table of constants fitted on two point loads at one image size (48 frames, a 49-96
    // sequence band, 65 536 point-frames ...), which sent every other shape wherever the table happened to put it.  Now
    // the first run of a new (mode, image size, frames, point-load) key PROBES the candidates on the caller's own data --
    // a batch run is idempotent, a lock-step step is re-run without its state-carrying kernels -- keeps the fastest and
    // remembers it for the process (tune_*).  vo_set_schedule() pins any knob instead.
    struct Schedule {
        int waves = 2, streams = 1, prep = 1;
        int wide = 4; // four-kernel EPnP for launches of up to this many frames (4 or 16; acts for 5 .. 16 frames per run)
    } sched;
    vo_schedule pin = {0, 0, -1, 0}; // 0 / 0 / -1 / 0 = probe
    long long sched_key[8] = {-1, 0, 0, 0, 0, 0, 0, 0}; // key `sched` was resolved for
    bool sched_probed = false;       // `sched` came out of a probe (here or earlier in the process), not from defaults
    bool tuning = false;             // inside a probe: run_stages must not start another one
    bool sync_call = false;          // the run being scheduled is a synchronous drop-in call (its own probe key: latency)
    Schedule ab_list[8];             // lock-step loop: the candidates being timed over real steps (vo_seq_step): up to four nominees
                                     // + (round 6, from 32 sequences on) every (pose_streams, prepare) pair once more with the OTHER register budget
    long long ab_key[8] = {};
    // what the last probe of this context measured: candidates and their steady-state ms per run (vo_get_probe_log)
    int probe_n = 0;
    vo_schedule probe_cand[VO_PROBE_LOG_MAX] = {};
    float probe_ms[VO_PROBE_LOG_MAX] = {};
    int probe_real[VO_PROBE_LOG_MAX] = {};          // 1: probe_ms[i] was (re)measured over real steps of the lock-step loop
    hipStream_t last_pose_stream = nullptr; // stream the latest pose chain was enqueued on
    hipStream_t stream_em = nullptr; // essential-matrix chain of the mono_rotation branch, next to the PnP chain
    bool partitioned = false; // stream / stream_pnp ... are the partitioned twin of `streams` (select_streams)
    bool quads_set = false; // d_quads holds h_quads (cleared whenever the table is zeroed)
    // synchronous drop-in calls (capi_dropin.hip): image slot of the LEFT image of the stereo pair the last vo_track_frame /
    // vo_circular_match received as its t1 pair (0 or 2; its right image follows it), pyramids built -- the next call may name it
    // as its t0 pair by passing no t0 images (main.cpp:157-158: imageLeft_t0 = imageLeft_t1).  -1: no such pair (no call yet,
    // or the batch / sequence API has touched the image table since)
    int tf_base = -1;
    // (round 6) The t1 pair of a synchronous drop-in call ON THE KEPT PAIR that single_frame_setup has NOT sent yet: hop 0 of
    // the LK chain reads the t0 pair only, so run_stages launches it first (lk_hops_kernel) and lets the t1 pair cross PCIe and
    // get its pyramids on the filter stream beside it; hops 1 .. 3 follow behind ev_t1_ready.  Host pointers of the caller: valid
    // inside the call that set them only -- vo_track_frame / vo_circular_match clear the record on every way out (DeferGuard).
    struct Deferred {
        int n = 0;                         // 2: img[0] / img[1] go to image slots first, first + 1
        const uint8_t *img[2] = {nullptr, nullptr};
        int first = 0, stride = 0;
    } defer;
    hipEvent_t ev_t1_ready = nullptr;
    // identity of that pair (vo_kept_pair_id): bumped whenever a drop-in call publishes a new t1 pair, so that a caller who
    // shares the context with others can tell whether the pair on the device is still the one ITS last call left
    int64_t tf_gen = 0;
    bool last_run_serial = true; // the last run_stages put everything on the tracking stream (else: sync_all before a gather)
    bool serial_pose = false; // -DVO_DEV_VARIANTS + VO_SERIAL_POSE=1: the whole chain on the tracking stream (profiling)
    bool lk_pair = false;     // -DVO_DEV_VARIANTS + VO_LK_PAIR=1: the two-features-per-wavefront LK kernel (lk.hip)
    // pinned staging for host images: rows are repacked to the device pitch on the host and go over
    // PCIe as ONE contiguous copy (a pitched copy from pageable memory moves row by row: 3.3 ms per
    // 1241 x 376 image measured, tools/latency_mode.py)
    uint8_t *h_stage = nullptr, *d_stage = nullptr; // (d_stage: the same memory as the GPU addresses it, launch_pull_image)
    uint8_t *h_gather = nullptr, *d_gather = nullptr; // vo_track_frame's result buffer: host memory, and its device address
    uint8_t *d_pts_stage = nullptr; // (its device address)
    uint8_t *h_feat_stage = nullptr, *d_feat_stage = nullptr; // pinned + its device address: vo_detect_bucket's carried feature set on its way in (fcap float2 + fcap int32)
    uint8_t *h_pts_stage = nullptr; // pinned: the points + count of a synchronous drop-in call on their way to the device (cap float2 + 16 bytes)
    size_t stage_slot = 0; // bytes per slot, VO_STAGE_SLOTS slots
    int stage_next = 0;
    int ransac_cap = 0;
    float h_P[24] = {};
    bool have_P = false;
    std::vector<int> h_npts;
    int *d_overflow = nullptr;     // [B] VO_STAGE_DETECT capacity flags (bit 0: feature list, bit 1: bucketed set)
    Quad *quads_cur = nullptr;     // the quad table the launches read: d_quads, or one phase of seq.d_quads
    std::vector<Quad> h_quads;     // host copy of d_quads (stale-pyramid check)
    std::vector<uint8_t> img_stale; // image re-uploaded since its pyramid was last built
    // ---- lock-step sequence loop (vo_seq_*): S sequences x 1 frame per step, state carried on the device ----
    struct Seq {
        bool on = false;
        int S = 0, ring = 0, max_steps = 0;
        long long step = 0;           // steps enqueued so far
        Quad *d_quads = nullptr;      // [ring][S]: phase r = (t0 in ring slot r, t1 in slot (r + 1) % ring)
        int *d_active = nullptr;      // [VO_SEQ_INFLIGHT][S]
        int *h_active = nullptr;      // pinned, same shape
        double *d_pose = nullptr;     // [S][16]
        double *d_traj = nullptr;     // [S][max_steps][VO_SEQ_ROW]
        SeqFrameInfo *d_info = nullptr; // [S][max_steps]
        int *d_rows = nullptr, *d_rows_carry = nullptr, *d_nages = nullptr; // [S]
        std::vector<uint8_t> pushed, had_prev; // pair pushed for the pending step / for the previous step
        std::vector<uint8_t> ever, gap;        // has had a pair since its reset / resumes after a pause (VO_SEQ_F_GAP)
        std::vector<int> h_rows;               // frames processed per sequence since its reset (host mirror of d_rows)
        bool broken = false;                   // a step failed after it had consumed its pairs: vo_seq_reset(-1) first
        hipStream_t copy = nullptr;
        hipEvent_t ev_upload = nullptr, ev_carry = nullptr, ev_integ = nullptr;
        bool integ_pending = false;
        hipEvent_t ev_slot_free[VO_SEQ_MAX_RING] = {}; // the LK that read ring slot r as its t0 pair has finished
        bool slot_busy[VO_SEQ_MAX_RING] = {};
        bool carry_pending = false;
        hipEvent_t ev_step[VO_SEQ_INFLIGHT] = {};
        bool step_pending[VO_SEQ_INFLIGHT] = {};
        // pinned staging for pageable host images: two generations of [S][2] pitched level-0 images
        uint8_t *h_stage = nullptr;
        // (round 6) the device twin of the staging area: when EVERY sequence's pair of a step is pageable, the step's half of
        // h_stage -- one contiguous block -- crosses the link in ONE copy-engine transfer (57 GB/s, no shader involved:
        // nothing that runs beside it slows down) and the ingest kernel re-pitches it HBM to HBM
        uint8_t *d_stage = nullptr;
        int n_pageable = 0;             // pageable pairs among the pending step's n_ing
        size_t stage_img = 0;
        hipEvent_t ev_stage[2] = {};
        hipEvent_t ev_detect = nullptr; // (round 6) the latest step's detection has run: a PCIe ingest starts behind it
        bool detect_pending = false;
        bool stage_busy[2] = {};
        // "prepare" work of a step runs on the copy stream, off the tracking stream's critical path: ingest of the new pairs,
        // their pyramids, and FAST + non-maximum suppression of their LEFT images -- the corners the NEXT step's
        // appendNewFeatures needs (visualOdometry.cpp:95-101 detects on imageLeft_t0, i.e. on the pair that arrived one
        // step earlier).  Per step the tracking stream is left with: bucketing -> LK -> filter -> carry.
        // (whether the prepare stream is used is vo_ctx::sched.prep; `copy` below is the stream the step's ingest kernel
        // goes to: the prepare stream when it is, a plain copy stream when not)
        float2 *d_corners = nullptr;      // [ring][S][fcap] FAST corners of the left image in each ring slot
        int *d_ncorn = nullptr;           // [ring][S]
        hipEvent_t ev_pyr = nullptr;      // pyramids of the pending step built (prep stream)
        hipEvent_t ev_fast[VO_SEQ_MAX_RING] = {}; // corners of ring slot r ready (prep stream)
        bool fast_pending[VO_SEQ_MAX_RING] = {};
        bool have_corners[VO_SEQ_MAX_RING] = {}; // d_corners of ring slot r belongs to the pair now in that slot
        SeqIngest *h_ing = nullptr, *d_ing = nullptr; // [VO_SEQ_INFLIGHT][S] pairs pushed for a step (pinned / device)
        int n_ing = 0, n_active = 0;    // pairs pushed for / sequences active in the pending step
        bool ing_pcie = false;          // a pair of the pending step lives in host memory (launch_seq_ingest: grid size)
        // A/B of the prepare stream over REAL steps (vo_seq_step): 1 = timing the dry probe's pick, 2 = timing its
        // prepare-flipped twin, ... (ab_cnt candidates), ab_cnt + 1 = decided; ab_left counts down the phase's steps (VO_AB_RAMP
        // untimed ramp steps + ab_n timed)
        int ab_phase = 0, ab_left = 0, ab_n = 0, ab_cnt = 0;
        bool ab_extra = false; // the twins with the other pose_waves have been appended
        hipEvent_t ev_ab[16] = {};
        bool ab_running() const { return ab_phase >= 1 && ab_phase <= ab_cnt; }

        bool begun = false, staged = false;
    } seq;
};

#define VO_STAGE_SLOTS 4
// the quadruples of the synchronous drop-in calls, resident behind the context's own table (d_quads + max_frames) from
// vo_create on: a call selects one (use_const_quad) instead of uploading it -- the stateless frame loop alternates between
// a detection quadruple and a tracking quadruple twice per frame, and an upload is a copy plus a stream synchronisation
#define VO_CONST_QUADS 4
static const vo::Quad VO_CONST_QUAD_TABLE[VO_CONST_QUADS] = {{0, 1, 2, 3}, {2, 3, 0, 1}, {0, 0, 0, 0}, {2, 2, 2, 2}};

#define VO_HIP_TRY(ctx, call)                                                                         \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                           \
            return VO_ERR_HIP;                                                                        \
        }                                                                                             \
    } while (0)

inline int fail(vo_ctx *ctx, int code, const char *msg)
{
    ctx->err = msg;
    return code;
}

inline int align_up(int v, int a) { return (v + a - 1) / a * a; }
// register budget of the pose kernels for the stand-alone calls (vo_pnp_ransac, vo_essential_pose): nothing runs beside
// them, so the full 512 registers unless the caller pinned the other variant
inline int standalone_waves(const vo_ctx *c) { return c->pin.pose_waves ? c->pin.pose_waves : 1; }
inline void use_const_quad(vo_ctx *c, int k) // frame 0 of a one-frame configuration reads VO_CONST_QUAD_TABLE[k]
{
    c->quads_cur = c->d_quads + c->max_frames + k;
    c->h_quads[0] = VO_CONST_QUAD_TABLE[k];
    c->quads_set = false; // d_quads itself no longer says what h_quads says: the next vo_batch_set_quads uploads
}
// the current feature set (see vo_ctx::pts_sel)
inline float2 *cur_pts(vo_ctx *c) { return c->pts_sel < 0 ? c->d_pts : c->d_pts_det[c->pts_sel]; }
inline int *cur_npts(vo_ctx *c) { return c->pts_sel < 0 ? c->d_npts : c->d_npts_det[c->pts_sel]; }
inline int *cur_ages(vo_ctx *c) { return c->pts_sel < 0 ? c->d_ages : c->d_ages_det[c->pts_sel]; }
// row pitch (pixels) of a bordered level: VO_BX left + w + at least VO_BY right, multiple of 16
inline int level_stride(int w) { return align_up(VO_BX + w + VO_BY, 16); }

template <typename T>
hipError_t dmalloc(T **p, size_t n)
{
    return hipMalloc((void **)p, n * sizeof(T));
}

#ifdef VO_DEV_VARIANTS
namespace vo {
int pose_prof_read(long long *out64); // pnp.hip
}
#endif

constexpr int EM_MAX_ITERS = 1000; // maxIters of the findEssentialMat overload the reference calls (OpenCV 4.5)

/* ------------------------------------- schedule probe ------------------------------------ */

struct TuneKey {
    long long k[8];
    bool operator<(const TuneKey &o) const
    {
        for (int i = 0; i < 8; i++)
            if (k[i] != o.k[i])
                return k[i] < o.k[i];
        return false;
    }
};

// the point load a schedule was probed at, in half-octave buckets (1722 .. 2435 points share one): the single-frame
// drop-in calls see a slightly different count every frame and must not probe every time
inline int pts_bucket(long long pts) { return pts <= 0 ? 0 : (int)floor(2.0 * log2((double)pts) + 0.5); }

#define D2H(dst, src, bytes)                                                                           \
    do {                                                                                              \
        if ((dst) && (bytes) > 0)                                                                      \
            VO_HIP_TRY(c, hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, c->stream));   \
    } while (0)

namespace vo_capi {
extern std::mutex g_tune_mu;
extern std::map<TuneKey, vo_ctx::Schedule> g_tuned; // per process: a second context of the same shape starts tuned
int plan_levels(vo_ctx *c, int w, int h);
bool acquire_streams(int device, StreamSet *out);
hipStream_t ensure_copy_stream(StreamSet *s, bool prepare, bool partitioned = false);
bool ensure_partitioned_streams(StreamSet *s, int device);
int select_streams(vo_ctx *c, bool partitioned); // call with every stream idle
void release_streams(int device, const StreamSet &s);
void seq_free(vo_ctx *c);
// idle: the caller has just drained the tracking stream (a synchronous drop-in call); pts / n_pts: the call's points ride along
// on: the stream the pull kernel of an idle upload goes to (default: the tracking stream)
int upload_image(vo_ctx *c, int idx, const void *src, int stride, hipMemcpyKind kind, bool idle = false, const float *pts = nullptr, int n_pts = -1,
                 hipStream_t on = nullptr);
int ensure_em(vo_ctx *c);
int run_stages(vo_ctx *c, int stages, bool timed, hipEvent_t *evs = nullptr, bool dry = false);
int sync_all(vo_ctx *c);
TuneKey tune_key(const vo_ctx *c, int stages);
void apply_pins(const vo_ctx *c, vo_ctx::Schedule *s);
bool all_pinned(const vo_ctx *c);
bool wide_knob_live(const vo_ctx *c);
int set_sched(vo_ctx *c, const vo_ctx::Schedule &s);
int sched_resolve(vo_ctx *c, int stages);
int seq_enqueue_inputs(vo_ctx *c, bool dry);
int seq_lookahead(vo_ctx *c, int r);
int probe_run(vo_ctx *c, int stages, bool timed, hipEvent_t *evs, bool dry);
int probe_candidate(vo_ctx *c, int stages, bool timed, hipEvent_t *evs, bool dry, bool latency, double *ms_per_run);
int tune_schedule(vo_ctx *c, int stages, bool timed, hipEvent_t *evs, bool dry, bool latency = false, bool publish = true);
int run_stages_auto(vo_ctx *c, int stages, bool timed, hipEvent_t *evs = nullptr, bool sync_call = false);
int get_pose_impl(vo_ctx *c, int frame, double *rvec, double *tvec, double *R, int32_t *inliers, int *n_inliers, int *status, int32_t *dbg4, bool pnp_rotation, int *em_status, bool io_pose = true);
int seq_begin_step(vo_ctx *c);
int seq_push(vo_ctx *c, int seq, const void *left, const void *right, int stride, int mode);
int single_frame_setup(vo_ctx *c, const uint8_t *l0, const uint8_t *r0, const uint8_t *l1, const uint8_t *r1, int w, int h, int stride, const float *pts, int n, bool defer_t1 = false);
int single_image_setup(vo_ctx *c, const uint8_t *img, int w, int h, int stride);
int flush_deferred(vo_ctx *c, hipStream_t on);
struct DeferGuard { // the deferred t1 pair never outlives the call whose host pointers it holds
    vo_ctx *c;
    ~DeferGuard() { c->defer.n = 0; }
};
} // namespace vo_capi

using namespace vo_capi;
