// capi.hip -- host side of libvo_hip.so: context, device memory, stream, and the C ABI declared
// in include/vo_hip.h.  No exceptions cross the ABI; every HIP failure becomes VO_ERR_HIP with a
// message retrievable through vo_last_error().
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
    } sched;
    vo_schedule pin = {0, 0, -1};    // 0 / 0 / -1 = probe
    long long sched_key[8] = {-1, 0, 0, 0, 0, 0, 0, 0}; // key `sched` was resolved for
    bool sched_probed = false;       // `sched` came out of a probe (here or earlier in the process), not from defaults
    bool tuning = false;             // inside a probe: run_stages must not start another one
    bool sync_call = false;          // the run being scheduled is a synchronous drop-in call (its own probe key: latency)
    Schedule ab_list[4];             // lock-step loop: the candidates being timed over real steps (vo_seq_step)
    long long ab_key[8] = {};
    // what the last probe of this context measured: candidates and their steady-state ms per run (vo_get_probe_log)
    int probe_n = 0;
    vo_schedule probe_cand[VO_PROBE_LOG_MAX] = {};
    float probe_ms[VO_PROBE_LOG_MAX] = {};
    int probe_real[VO_PROBE_LOG_MAX] = {};          // 1: probe_ms[i] was (re)measured over real steps of the lock-step loop
    hipStream_t last_pose_stream = nullptr; // stream the latest pose chain was enqueued on
    hipStream_t stream_em = nullptr; // essential-matrix chain of the mono_rotation branch, next to the PnP chain
    bool quads_set = false; // d_quads holds h_quads (cleared whenever the table is zeroed)
    bool serial_pose = false; // -DVO_DEV_VARIANTS + VO_SERIAL_POSE=1: the whole chain on the tracking stream (profiling)
    bool lk_pair = false;     // -DVO_DEV_VARIANTS + VO_LK_PAIR=1: the two-features-per-wavefront LK kernel (lk.hip)
    // pinned staging for host images: rows are repacked to the device pitch on the host and go over
    // PCIe as ONE contiguous copy (a pitched copy from pageable memory moves row by row: 3.3 ms per
    // 1241 x 376 image measured, tools/latency_mode.py)
    uint8_t *h_stage = nullptr;
    uint8_t *h_gather = nullptr, *d_gather = nullptr; // vo_track_frame's result buffer: host memory, and its device address
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
        size_t stage_img = 0;
        hipEvent_t ev_stage[2] = {};
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
        // A/B of the prepare stream over REAL steps (vo_seq_step): 1 = timing the dry probe's pick, 2 = timing its
        // prepare-flipped twin, ... (ab_cnt candidates), ab_cnt + 1 = decided; ab_left counts down the phase's steps (3 untimed
        // ramp steps + ab_n timed)
        int ab_phase = 0, ab_left = 0, ab_n = 0, ab_cnt = 0;
        hipEvent_t ev_ab[8] = {};
        bool ab_running() const { return ab_phase >= 1 && ab_phase <= ab_cnt; }

        bool begun = false, staged = false;
    } seq;
};

#define VO_STAGE_SLOTS 4

namespace {

#define VO_HIP_TRY(ctx, call)                                                                         \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                           \
            return VO_ERR_HIP;                                                                        \
        }                                                                                             \
    } while (0)

int fail(vo_ctx *ctx, int code, const char *msg)
{
    ctx->err = msg;
    return code;
}

inline int align_up(int v, int a) { return (v + a - 1) / a * a; }
// register budget of the pose kernels for the stand-alone calls (vo_pnp_ransac, vo_essential_pose): nothing runs beside
// them, so the full 512 registers unless the caller pinned the other variant
inline int standalone_waves(const vo_ctx *c) { return c->pin.pose_waves ? c->pin.pose_waves : 1; }
// the current feature set (see vo_ctx::pts_sel)
inline float2 *cur_pts(vo_ctx *c) { return c->pts_sel < 0 ? c->d_pts : c->d_pts_det[c->pts_sel]; }
inline int *cur_npts(vo_ctx *c) { return c->pts_sel < 0 ? c->d_npts : c->d_npts_det[c->pts_sel]; }
inline int *cur_ages(vo_ctx *c) { return c->pts_sel < 0 ? c->d_ages : c->d_ages_det[c->pts_sel]; }
// row pitch (pixels) of a bordered level: VO_BX left + w + at least VO_BY right, multiple of 16
inline int level_stride(int w) { return align_up(VO_BX + w + VO_BY, 16); }

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

template <typename T>
hipError_t dmalloc(T **p, size_t n)
{
    return hipMalloc((void **)p, n * sizeof(T));
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
hipStream_t ensure_copy_stream(StreamSet *s, bool prepare)
{
    hipStream_t &st = prepare ? s->prep : s->copy;
    if (st)
        return st;
    int least = 0, greatest = 0;
    bool ok = hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess;
    ok = ok && (prepare ? hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest)
                        : hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) == hipSuccess;
    return ok ? st : nullptr;
}

void release_streams(int device, const StreamSet &s)
{
    hipStream_t all[] = {s.stream, s.pnp, s.pnp2, s.filter, s.em, s.copy, s.prep};
    for (hipStream_t st : all)
        if (st)
            (void)hipStreamSynchronize(st);
    if (!s.stream || device >= VO_MAX_DEVICES)
        return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool[device].push_back(s);
}

} // namespace

static int sync_all(vo_ctx *c);

static void seq_free(vo_ctx *c)
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
    hipEvent_t evs[] = {q.ev_upload, q.ev_carry, q.ev_integ, q.ev_pyr, q.ev_stage[0], q.ev_stage[1]};
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

#ifdef VO_DEV_VARIANTS
namespace vo {
int pose_prof_read(long long *out64); // pnp.hip
}
#endif

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
                     b.em_results, b.epnp_ws, b.epnp_gws};
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
    if (c->h_stage)
        (void)hipHostFree(c->h_stage);
    if (c->h_gather)
        (void)hipHostFree(c->h_gather);
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
    ok = ok && hipHostMalloc((void **)&c->h_stage, c->stage_slot * VO_STAGE_SLOTS, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&c->h_gather, frame_gather_bytes(c->cap), hipHostMallocMapped) == hipSuccess;
    ok = ok && hipHostGetDevicePointer((void **)&c->d_gather, c->h_gather, 0) == hipSuccess;
    ok = ok && dmalloc(&c->d_pix, c->pix_capacity) == hipSuccess;
    ok = ok && dmalloc(&c->d_der, c->pix_capacity) == hipSuccess;
    ok = ok && dmalloc(&c->d_imgs, (size_t)c->max_images) == hipSuccess;
    ok = ok && dmalloc(&c->d_quads, B) == hipSuccess;
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
        c->quads_cur = c->d_quads;
        c->n_images = 0; // force the full re-plan below
    }
    if (c->n_images == n_images && c->w == w && c->h == h && c->n_frames == n_frames) {
        c->pyr_first = 0; // a (re)configure always restores "build every pyramid"
        c->pyr_count = n_images;
        return VO_OK;
    }
    c->sched_key[0] = -1; // a new shape: the schedule is resolved again at its first run
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

static int upload_image(vo_ctx *c, int idx, const void *src, int stride, hipMemcpyKind kind)
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
        if (c->stage_next == 0) // all slots may still be in flight from the previous round of uploads
            VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
        uint8_t *slot = c->h_stage + c->stage_slot * (size_t)c->stage_next;
        c->stage_next = (c->stage_next + 1) % VO_STAGE_SLOTS;
        for (int y = 0; y < c->h; y++)
            memcpy(slot + (size_t)y * pitch, (const uint8_t *)src + (size_t)y * stride, (size_t)c->w);
        VO_HIP_TRY(c, hipMemcpyAsync(dst, slot, pitch * (size_t)(c->h - 1) + (size_t)c->w, hipMemcpyHostToDevice,
                                     c->stream));
        return VO_OK;
    }
    VO_HIP_TRY(c, hipMemcpy2DAsync(dst, (size_t)c->lstride[0], src, (size_t)stride, (size_t)c->w, (size_t)c->h,
                                   kind, c->stream));
    return VO_OK;
}

int vo_batch_upload_image(vo_ctx *c, int idx, const uint8_t *host, int stride)
{
    int rc = upload_image(c, idx, host, stride, hipMemcpyHostToDevice);
    if (rc == VO_OK) // pageable host memory: make the call safe to return from
        VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return rc;
}

int vo_batch_upload_image_dev(vo_ctx *c, int idx, const void *dev, int stride)
{
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

constexpr int EM_MAX_ITERS = 1000; // maxIters of the findEssentialMat overload the reference calls (OpenCV 4.5)

// working set of the essential-matrix chain, allocated the first time it is asked for
static int ensure_em(vo_ctx *c)
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

// dry (lock-step loop, schedule probe): everything but the two kernels that advance a sequence's state (seq_carry,
// seq_integrate) -- the step can then be repeated any number of times
static int run_stages(vo_ctx *c, int stages, bool timed, hipEvent_t *evs = nullptr, bool dry = false)
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
    if (timed)
        VO_HIP_TRY(c, hipEventRecord(evs[e], pyrs));
    e++;
    if (stages & VO_STAGE_PYRAMID) {
        const PyrImage *tab = c->d_imgs + c->pyr_first;
        const int ni = c->pyr_count;
        if (ni > 0) {
            // Two launches, no LDS (round 4): level 0 is read once and gives its Scharr image, level 1 and its own border; the
            // small levels follow in one launch, a workgroup per image (pyramid.hip).  (Round 3: eight launches of three
            // kernels that each fetched the level again.)
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
    if (!sq.on && (stages & VO_STAGE_LK)) {
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
        if (bs < 1 || fpb < 1 || fpb > 8 || cells > 1024)
            return fail(c, VO_ERR_ARG, "vo_batch_run: bucket grid beyond 1024 cells / 8 features per bucket");
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
    const bool serial = c->serial_pose || (c->sync_call && !sq.on);
    hipStream_t fs = serial ? c->stream : c->stream_filter;
    const bool two_pose_streams = !serial && !c->prm.mono_rotation && c->sched.streams == 2;
    hipStream_t ps = serial ? c->stream : (two_pose_streams && (c->cur & 1)) ? c->stream_pnp2 : c->stream_pnp;
    if (touches_pose) {
        VO_HIP_TRY(c, hipEventRecord(pb.ready, c->stream));
        VO_HIP_TRY(c, hipStreamWaitEvent(fs, pb.ready, 0));
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
        VO_HIP_TRY(c, hipEventRecord(c->ev_trk_free[c->trk_last], fs));
        c->trk_busy[c->trk_last] = true;
        if (c->pts_sel >= 0 && c->pts_sel != c->trk_last) {
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
        VO_HIP_TRY(c, hipEventRecord(pb.tri_done, fs));
        VO_HIP_TRY(c, hipStreamWaitEvent(ps, pb.tri_done, 0));
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
            VO_HIP_TRY(c, hipStreamWaitEvent(es, pb.tri_done, 0));
            launch_essential(pb.outB, pb.outB + 2 * cap, (size_t)4 * cap, pb.nB, cap, B, ep, c->em, pb.em_results,
                             /*crowded*/ crowded, es);
            VO_HIP_TRY(c, hipEventRecord(pb.em_done, es));
        }
        launch_pnp_ransac(pb.xyz, pb.outB + 2 * cap, (size_t)4 * cap, pb.nB, cap, B, pp, pb.subsets, pb.models, pb.counts,
                          pb.rstate, c->sched.waves, ps, pb.epnp_ws,
                          c->max_frames < VO_EPNP_WS_MAX_FRAMES ? c->max_frames : VO_EPNP_WS_MAX_FRAMES, pb.epnp_gws);
        if (c->prm.mono_rotation)
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
        VO_HIP_TRY(c, hipEventRecord(pb.done, ps));
        pb.pending = true;
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
static int sync_all(vo_ctx *c)
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
    return VO_OK;
}

/* ------------------------------------- schedule probe ------------------------------------ */
namespace {

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
std::mutex g_tune_mu;
std::map<TuneKey, vo_ctx::Schedule> g_tuned; // per process: a second context of the same shape starts tuned

// the point load a schedule was probed at, in half-octave buckets (1722 .. 2435 points share one): the single-frame
// drop-in calls see a slightly different count every frame and must not probe every time
int pts_bucket(long long pts) { return pts <= 0 ? 0 : (int)floor(2.0 * log2((double)pts) + 0.5); }

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

void apply_pins(const vo_ctx *c, vo_ctx::Schedule *s)
{
    if (c->pin.pose_waves)
        s->waves = c->pin.pose_waves;
    if (c->pin.pose_streams)
        s->streams = c->pin.pose_streams;
    if (c->pin.prepare >= 0)
        s->prep = c->pin.prepare;
    if (c->prm.mono_rotation)
        s->streams = 1; // the essential-matrix chain already runs next to the PnP chain on its own stream
    if (!c->seq.on)
        s->prep = 0;
}

bool all_pinned(const vo_ctx *c)
{
    return c->pin.pose_waves && (c->pin.pose_streams || c->prm.mono_rotation) && (!c->seq.on || c->pin.prepare >= 0);
}

} // namespace

// make `s` the schedule the next run uses.  Moving the lock-step loop's ingest between the plain copy stream and the
// prepare stream is only done with every stream idle (the ring slots, the FAST scratch buffers and the staging area are
// ordered per stream).
static int set_sched(vo_ctx *c, const vo_ctx::Schedule &s)
{
    if (c->seq.on && (s.prep != c->sched.prep || !c->seq.copy)) {
        int rc = sync_all(c);
        if (rc != VO_OK)
            return rc;
        c->seq.copy = ensure_copy_stream(&c->streams, s.prep != 0);
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
static int sched_resolve(vo_ctx *c, int stages)
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
    return 0;
}

// What the pending step's kernels read that comes from the host: the pushed pairs -> ring slot step % ring with ONE
// kernel on the copy stream -- after the LK that still reads the slot's previous occupant (ring 2: the previous step's;
// ring 3: the one before, long finished), next to the previous step's kernels -- and the step's activity flags.
// dry (schedule probe): the same transfers again (same bytes to the same places), without the slot / staging bookkeeping.
static int seq_enqueue_inputs(vo_ctx *c, bool dry)
{
    vo_ctx::Seq &q = c->seq;
    const int slot = (int)(q.step % VO_SEQ_INFLIGHT), r = (int)(q.step % q.ring);
    if (q.n_ing > 0) {
        if (!dry && q.slot_busy[r]) {
            VO_HIP_TRY(c, hipStreamWaitEvent(q.copy, q.ev_slot_free[r], 0));
            q.slot_busy[r] = false;
        }
        SeqIngest *d_tab = q.d_ing + (size_t)slot * q.S;
        VO_HIP_TRY(c, hipMemcpyAsync(d_tab, q.h_ing + (size_t)slot * q.S, sizeof(SeqIngest) * q.n_ing,
                                     hipMemcpyHostToDevice, q.copy));
        launch_seq_ingest(d_tab, q.n_ing, c->w, c->h, c->lstride[0],
                          c->d_pix + c->loff[0] + (size_t)VO_BY * c->lstride[0] + VO_BX, c->img_bytes, q.copy);
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
static int seq_lookahead(vo_ctx *c, int r)
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
static int probe_run(vo_ctx *c, int stages, bool timed, hipEvent_t *evs, bool dry)
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
static int probe_candidate(vo_ctx *c, int stages, bool timed, hipEvent_t *evs, bool dry, bool latency, double *ms_per_run)
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
static int tune_schedule(vo_ctx *c, int stages, bool timed, hipEvent_t *evs, bool dry, bool latency = false)
{
    const TuneKey key = tune_key(c, stages);
    std::vector<vo_ctx::Schedule> cands;
    for (int waves = 1; waves <= 2; waves++)
        for (int streams = 1; streams <= (c->sync_call && !c->seq.on ? 1 : 2); streams++) // (a synchronous call runs on one stream)
            for (int prep = 1; prep >= 0; prep--) {
                vo_ctx::Schedule s, t;
                s.waves = waves;
                s.streams = streams;
                s.prep = prep;
                t = s;
                apply_pins(c, &t);
                if (t.waves != s.waves || t.streams != s.streams || t.prep != s.prep)
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
    c->tuning = true;
    int rc = VO_OK, best = 0;
    double best_ms = 0;
    for (size_t i = 0; i < cands.size() && rc == VO_OK; i++) {
        rc = set_sched(c, cands[i]);
        double ms = 0;
        if (rc == VO_OK)
            rc = cands.size() > 1 ? probe_candidate(c, stages, timed, evs, dry, latency, &ms) : VO_OK;
        if (rc == VO_OK && (i == 0 || ms < best_ms)) {
            best = (int)i;
            best_ms = ms;
        }
        if (i < VO_PROBE_LOG_MAX) {
            c->probe_cand[i] = vo_schedule{cands[i].waves, cands[i].streams, cands[i].prep};
            c->probe_ms[i] = (float)ms;
            c->probe_real[i] = 0;
        }
    }
    c->probe_n = (int)(cands.size() < VO_PROBE_LOG_MAX ? cands.size() : VO_PROBE_LOG_MAX);
    c->tuning = false;
    if (rc != VO_OK)
        return rc;
    rc = set_sched(c, cands[best]);
    if (rc != VO_OK)
        return rc;
    {
        std::lock_guard<std::mutex> lk(g_tune_mu);
        g_tuned[key] = cands[best];
    }
    memcpy(c->sched_key, key.k, sizeof(key.k));
    c->sched_probed = true;
    return VO_OK;
}

// run_stages for the batch entry points: settles the schedule first (cached, pinned or probed) when the run has a pose chain
// sync_call: a drop-in call that returns results -- the caller waits for every run, so candidates are compared by latency
static int run_stages_auto(vo_ctx *c, int stages, bool timed, hipEvent_t *evs = nullptr, bool sync_call = false)
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

int vo_set_schedule(vo_ctx *c, const vo_schedule *s)
{
    if (!c)
        return VO_ERR_ARG;
    vo_schedule p = {0, 0, -1};
    if (s)
        p = *s;
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

#define D2H(dst, src, bytes)                                                                           \
    do {                                                                                              \
        if ((dst) && (bytes) > 0)                                                                      \
            VO_HIP_TRY(c, hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, c->stream));   \
    } while (0)

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

// stage-A arrays (deleteUnmatchFeaturesCircle output) of one frame
static int get_stage_a(vo_ctx *c, int frame, float *l0, float *r0, float *r1, float *l1, float *l0r,
                       int32_t *keep_idx, int *n_out)
{
    int M = 0;
    int rcs = sync_all(c); // the filter runs on the post stream
    if (rcs != VO_OK)
        return rcs;
    VO_HIP_TRY(c, hipMemcpyAsync(&M, c->d_nA + frame, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    const size_t cap = c->cap;
    const float2 *a = c->d_outA + (size_t)frame * 5 * cap;
    D2H(l0, a, sizeof(float2) * M);
    D2H(r0, a + cap, sizeof(float2) * M);
    D2H(r1, a + 2 * cap, sizeof(float2) * M);
    D2H(l1, a + 3 * cap, sizeof(float2) * M);
    D2H(l0r, a + 4 * cap, sizeof(float2) * M);
    D2H(keep_idx, c->d_idxA + (size_t)frame * cap, sizeof(int32_t) * M);
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n_out = M;
    return VO_OK;
}

// pnp_rotation: R = Rodrigues(rvec) even under mono_rotation (vo_pnp_ransac).  em_status (optional): status of the
// essential-matrix side of the frame under mono_rotation (1 ok, 0 no model, -1 too few points), 1 otherwise.
static int get_pose_impl(vo_ctx *c, int frame, double *rvec, double *tvec, double *R, int32_t *inliers,
                         int *n_inliers, int *status, int32_t *dbg4, bool pnp_rotation, int *em_status, bool io_pose = true)
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
        q.copy = ensure_copy_stream(&c->streams, sc.prep != 0);
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

// First touch of the pending step (a push or the step call itself): its slot of the pinned per-step tables must
// have been consumed (step - VO_SEQ_INFLIGHT has finished), which also bounds the host's run-ahead.
static int seq_begin_step(vo_ctx *c)
{
    vo_ctx::Seq &q = c->seq;
    if (q.begun)
        return VO_OK;
    const int slot = (int)(q.step % VO_SEQ_INFLIGHT);
    if (q.step_pending[slot]) {
        VO_HIP_TRY(c, hipEventSynchronize(q.ev_step[slot]));
        q.step_pending[slot] = false;
    }
    q.n_ing = 0;
    q.begun = true;
    return VO_OK;
}

// A push only records where the pair is; vo_seq_step moves all pairs of the step with ONE kernel on the copy stream
// (seq_ingest_kernel).  mode 0: pageable host memory, copied into the pinned staging area now so the caller's buffer
// is free on return; 1: page-locked host memory, read by the GPU over PCIe when the step runs; 2: device memory.
static int seq_push(vo_ctx *c, int seq, const void *left, const void *right, int stride, int mode)
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
            }
            q.stage_img = img;
            VO_HIP_TRY(c, hipHostMalloc((void **)&q.h_stage, img * 2 * 2 * (size_t)q.S, hipHostMallocDefault));
        }
        if (q.stage_busy[g]) { // the ingest kernel of step - 2 still reads this half of the staging area
            VO_HIP_TRY(c, hipEventSynchronize(q.ev_stage[g]));
            q.stage_busy[g] = false;
        }
        uint8_t *sl = q.h_stage + (((size_t)g * q.S + seq) * 2) * img, *sr = sl + img;
        const uint8_t *srcs[2] = {(const uint8_t *)left, (const uint8_t *)right};
        uint8_t *dsts[2] = {sl, sr};
        for (int side = 0; side < 2; side++) {
            if (stride == c->w)
                memcpy(dsts[side], srcs[side], img);
            else
                for (int y = 0; y < c->h; y++)
                    memcpy(dsts[side] + (size_t)y * c->w, srcs[side] + (size_t)y * stride, (size_t)c->w);
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
    q.h_ing[(size_t)(q.step % VO_SEQ_INFLIGHT) * q.S + q.n_ing++] = e;
    q.pushed[seq] = 1;
    return VO_OK;
}

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
    for (int i = 0; i < n; i++) {
        int rc = seq_push(c, seq_ids[i], left[i], right[i], stride, kind);
        if (rc != VO_OK)
            return rc;
    }
    return VO_OK;
}

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
        if (bs < 1 || fpb < 1 || fpb > 8 || (long long)(c->h / bs + 1) * (c->w / bs + 1) > 1024)
            return fail(c, VO_ERR_ARG, "vo_seq_step: bucket grid beyond 1024 cells / 8 features per bucket");
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
            rc = tune_schedule(c, stages, true, step_evs, /*dry*/ true);
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
                            c->probe_cand[i].prepare == x.prep)
                            return (double)c->probe_ms[i];
                    return -1.0;
                };
                // one candidate per (pose_streams, prepare) -- the two knobs the dry runs misjudge -- each with the register
                // budget the dry runs prefer for it; the dry pick first (it stays if the loop ends before the comparison does)
                int n = 0;
                c->ab_list[n++] = c->sched;
                for (int st = 1; st <= 2; st++)
                    for (int pr = 1; pr >= 0; pr--) {
                        if (st == c->sched.streams && pr == c->sched.prep)
                            continue;
                        int bi = -1;
                        for (int i = 0; i < c->probe_n; i++)
                            if (c->probe_cand[i].pose_streams == st && c->probe_cand[i].prepare == pr &&
                                (bi < 0 || c->probe_ms[i] < c->probe_ms[bi]))
                                bi = i;
                        if (bi >= 0 && n < 4) {
                            c->ab_list[n].waves = c->probe_cand[bi].pose_waves;
                            c->ab_list[n].streams = st;
                            c->ab_list[n].prep = pr;
                            n++;
                        }
                    }
                if (n > 1) {
                    double ms = dry_ms(c->sched);
                    ms = ms > 0.02 ? ms : 0.02;
                    q.ab_n = (int)ceil(25.0 / ms);
                    q.ab_n = q.ab_n < 12 ? 12 : q.ab_n > 48 ? 48 : q.ab_n;
                    q.ab_cnt = n;
                    q.ab_phase = 1;
                    q.ab_left = 3 + q.ab_n;
                    memcpy(c->ab_key, c->sched_key, sizeof(c->ab_key));
                    c->sched_probed = false; // "in progress" (vo_get_schedule reports 2)
                }
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
                q.ab_left = 3 + q.ab_n;
            } else {
                VO_HIP_TRY(c, hipEventSynchronize(q.ev_ab[2 * ph + 1]));
                int best = 0;
                float t[4] = {0, 0, 0, 0};
                for (int i = 0; i < q.ab_cnt; i++) {
                    VO_HIP_TRY(c, hipEventElapsedTime(&t[i], q.ev_ab[2 * i], q.ev_ab[2 * i + 1]));
                    if (t[i] < t[best])
                        best = i;
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
                for (int k = 0; k < q.ab_cnt; k++) // the log shows what was measured over real steps
                    for (int i = 0; i < c->probe_n; i++)
                        if (c->probe_cand[i].pose_waves == c->ab_list[k].waves && c->probe_cand[i].pose_streams == c->ab_list[k].streams &&
                            c->probe_cand[i].prepare == c->ab_list[k].prep) {
                            c->probe_ms[i] = t[k] / q.ab_n;
                            c->probe_real[i] = 1;
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

/* ---------------------------------- drop-in calls ---------------------------------------- */

static int single_frame_setup(vo_ctx *c, const uint8_t *l0, const uint8_t *r0, const uint8_t *l1,
                              const uint8_t *r1, int w, int h, int stride, const float *pts, int n)
{
    if (!l0 || !r0 || !l1 || !r1 || n < 0 || (n > 0 && !pts))
        return fail(c, VO_ERR_ARG, "null image / points");
    if (n > c->cap)
        return fail(c, VO_ERR_ARG, "more points than max_pts given to vo_create");
    int rc = vo_batch_configure(c, 4, w, h, 1);
    if (rc != VO_OK)
        return rc;
    const uint8_t *imgs[4] = {l0, r0, l1, r1};
    for (int i = 0; i < 4; i++) {
        rc = upload_image(c, i, imgs[i], stride, hipMemcpyHostToDevice);
        if (rc != VO_OK)
            return rc;
    }
    const int32_t quad[4] = {0, 1, 2, 3};
    rc = vo_batch_set_quads(c, quad, 1);
    if (rc != VO_OK)
        return rc;
    return vo_batch_set_points(c, 0, pts, n);
}

int vo_circular_match(vo_ctx *c, const uint8_t *l0, const uint8_t *r0, const uint8_t *l1, const uint8_t *r1,
                      int w, int h, int stride, const float *pts, int n, float *out_l0, float *out_r0,
                      float *out_r1, float *out_l1, float *out_l0_ret, uint8_t *status4, int32_t *keep_idx,
                      int *n_out, int apply_consistency)
{
    if (!c || !n_out)
        return VO_ERR_ARG;
    int rc = single_frame_setup(c, l0, r0, l1, r1, w, h, stride, pts, n);
    if (rc != VO_OK)
        return rc;
    rc = run_stages(c, VO_STAGE_PYRAMID | VO_STAGE_LK | VO_STAGE_FILTER, false);
    if (rc != VO_OK)
        return rc;
    if (status4) {
        rc = vo_batch_get_tracks(c, 0, nullptr, nullptr, nullptr, nullptr, status4, n);
        if (rc != VO_OK)
            return rc;
    }
    if (!apply_consistency)
        return get_stage_a(c, 0, out_l0, out_r0, out_r1, out_l1, out_l0_ret, keep_idx, n_out);
    // stage B drops l0_ret (removeInvalidPoints is not applied to it); fill it from stage A by index
    int K = 0;
    rc = vo_batch_get_filtered(c, 0, out_l0, out_r0, out_l1, out_r1, nullptr, keep_idx, &K, nullptr, nullptr);
    if (rc != VO_OK)
        return rc;
    if (out_l0_ret && K > 0) {
        std::vector<int32_t> idx((size_t)K);
        std::vector<float> ret((size_t)2 * (n > 0 ? n : 1));
        VO_HIP_TRY(c, hipMemcpy(idx.data(), c->pb[c->last].idxB, sizeof(int32_t) * K, hipMemcpyDeviceToHost));
        VO_HIP_TRY(c, hipMemcpy(ret.data(), c->d_trk2[c->trk_last] + (size_t)3 * c->cap, sizeof(float2) * n,
                                hipMemcpyDeviceToHost));
        for (int i = 0; i < K; i++) {
            out_l0_ret[2 * i] = ret[2 * idx[i]];
            out_l0_ret[2 * i + 1] = ret[2 * idx[i] + 1];
        }
    }
    *n_out = K;
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
    // frame 0, stage-B rows 0 (left) and 1 (right)
    vo_ctx::PoseBufs &pb = c->pb[c->last];
    VO_HIP_TRY(c, hipMemcpyAsync(pb.outB, pl, sizeof(float2) * n, hipMemcpyHostToDevice, c->stream));
    VO_HIP_TRY(c, hipMemcpyAsync(pb.outB + c->cap, pr, sizeof(float2) * n, hipMemcpyHostToDevice, c->stream));
    VO_HIP_TRY(c, hipMemcpyAsync(pb.nB, &n, sizeof(int), hipMemcpyHostToDevice, c->stream));
    launch_triangulate(c->d_P, c->d_P + 12, pb.outB, pb.outB + c->cap, (size_t)4 * c->cap, pb.nB, c->cap, n, 1,
                       pb.xyz, c->stream);
    VO_HIP_TRY(c, hipGetLastError());
    VO_HIP_TRY(c, hipMemcpyAsync(xyz_out, pb.xyz, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, c->stream));
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VO_OK;
}

static int fetch_pose(vo_ctx *c, double *rvec_io, double *tvec_io, double *R_out, int32_t *inliers,
                      int *n_inliers, bool pnp_rotation)
{
    int status = 0, ninl = 0, em_status = 1;
    int rc = get_pose_impl(c, 0, rvec_io, tvec_io, R_out, inliers, &ninl, &status, nullptr, pnp_rotation, &em_status);
    if (rc != VO_OK)
        return rc;
    if (n_inliers)
        *n_inliers = ninl;
    if (status < 0)
        return fail(c, VO_ERR_TOO_FEW, "fewer than 4 correspondences reached solvePnPRansac (CV_Assert(npoints >= 4))");
    if (em_status != 1) // mono_rotation and findEssentialMat found nothing: R_out was left untouched
        return VO_NO_ESSENTIAL;
    return status == 1 ? VO_OK : VO_NO_MODEL;
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
    if (n > 0) {
        VO_HIP_TRY(c, hipMemcpyAsync(pb.xyz, xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, c->stream));
        VO_HIP_TRY(c, hipMemcpyAsync(pb.outB + 2 * (size_t)c->cap, uv, sizeof(float2) * n, hipMemcpyHostToDevice,
                                     c->stream));
    }
    VO_HIP_TRY(c, hipMemcpyAsync(pb.nB, &n, sizeof(int), hipMemcpyHostToDevice, c->stream));
    if (c->n_frames < 1)
        c->n_frames = 1;
    launch_pnp(pb.xyz, pb.outB + 2 * (size_t)c->cap, (size_t)4 * c->cap, pb.nB, c->cap, 1, pp, pb.subsets,
               pb.models, pb.counts, pb.rstate, pb.inliers, pb.results, standalone_waves(c), c->stream, pb.epnp_ws, 1, pb.epnp_gws);
    VO_HIP_TRY(c, hipGetLastError());
    return fetch_pose(c, rvec_io, tvec_io, R_out, inliers, n_inliers, /*pnp_rotation*/ true);
}

// one image as a 1-frame batch whose quad points at image 0 four times
static int single_image_setup(vo_ctx *c, const uint8_t *img, int w, int h, int stride)
{
    if (!img)
        return fail(c, VO_ERR_ARG, "null image");
    int rc = vo_batch_configure(c, 4, w, h, 1);
    if (rc != VO_OK)
        return rc;
    rc = upload_image(c, 0, img, stride, hipMemcpyHostToDevice);
    if (rc != VO_OK)
        return rc;
    const int32_t quad[4] = {0, 0, 0, 0};
    return vo_batch_set_quads(c, quad, 1);
}

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
    const int one = 1, zero = 0;
    VO_HIP_TRY(c, hipMemcpyAsync(c->d_detect, &one, sizeof(int), hipMemcpyHostToDevice, c->stream));
    VO_HIP_TRY(c, hipMemcpyAsync(c->d_ntracked, &zero, sizeof(int), hipMemcpyHostToDevice, c->stream));
    c->h_ntracked[0] = 0;
    c->detect_uploaded = false;
    threshold = threshold < 0 ? 0 : threshold > 255 ? 255 : threshold;
    launch_detect_bucket(c->d_imgs, c->d_quads, c->d_detect, 1, w, h, threshold, nonmax, c->d_nmsmask, c->d_rowcnt, c->d_rowoff,
                         c->d_ntracked, c->d_nnew, c->fcap, c->d_feat, c->d_fages, /*bucket_size*/ 0, 1, nullptr,
                         nullptr, nullptr, 0, nullptr, nullptr, c->stream);
    VO_HIP_TRY(c, hipGetLastError());
    int n = 0;
    VO_HIP_TRY(c, hipMemcpyAsync(&n, c->d_nnew, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    int k = n < cap ? n : cap;
    k = k < c->fcap ? k : c->fcap;
    if (k > 0)
        VO_HIP_TRY(c, hipMemcpy(pts_out, c->d_feat, sizeof(float2) * k, hipMemcpyDeviceToHost));
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
    int rc = single_image_setup(c, img, w, h, stride);
    if (rc != VO_OK)
        return rc;
    const vo_detect_params saved = c->dprm;
    rc = vo_batch_set_detect_params(c, dp);
    if (rc == VO_OK)
        rc = vo_batch_set_features(c, 0, pts_io, *n_pts, ages_io, *n_ages);
    if (rc == VO_OK)
        rc = run_stages(c, VO_STAGE_DETECT, false);
    int k = 0;
    if (rc == VO_OK) {
        VO_HIP_TRY(c, hipMemcpyAsync(&k, cur_npts(c), sizeof(int), hipMemcpyDeviceToHost, c->stream));
        VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (k > cap) {
            c->dprm = saved;
            return fail(c, VO_ERR_ARG, "vo_detect_bucket: bucketed set larger than the caller's capacity");
        }
        rc = vo_batch_get_features(c, 0, pts_io, ages_io, &k);
    }
    c->dprm = saved;
    if (rc != VO_OK)
        return rc;
    *n_pts = k;
    *n_ages = k;
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
    int rc = single_frame_setup(c, l0, r0, l1, r1, w, h, stride, pts, n);
    if (rc != VO_OK)
        return rc;
    rc = vo_batch_set_projection(c, P_l, P_r);
    if (rc != VO_OK)
        return rc;
    rc = run_stages_auto(c, VO_STAGE_ALL, false, nullptr, /*sync_call*/ true);
    if (rc != VO_OK)
        return rc;
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
    VO_HIP_TRY(c, hipStreamWaitEvent(c->stream, pb.done, 0)); // `done` covers the filter, both pose chains
    launch_frame_gather(g, c->d_gather, c->stream);
    VO_HIP_TRY(c, hipStreamSynchronize(c->stream));
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
    if (r.status < 0)
        return fail(c, VO_ERR_TOO_FEW, "fewer than 4 correspondences reached solvePnPRansac (CV_Assert(npoints >= 4))");
    if (em_status != 1) // mono_rotation and findEssentialMat found nothing: R_out was left untouched
        return VO_NO_ESSENTIAL;
    return r.status == 1 ? VO_OK : VO_NO_MODEL;
}

#ifdef VO_DEV_VARIANTS
// developer build only: the 100 MHz stamps the pose kernels left for frame 0 / hypothesis 0 (pnp.hip, tools/pose_phases.py)
int vo_dev_pose_prof(vo_ctx *c, long long *out64)
{
    if (!c || !out64 || sync_all(c) != VO_OK)
        return VO_ERR_ARG;
    return vo::pose_prof_read(out64) == 0 ? VO_OK : VO_ERR_HIP;
}
#endif
} // extern "C"
