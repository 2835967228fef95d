// pnp.hip -- PnP + RANSAC pose solve on the device.
//
// Replaces cv::solvePnPRansac + cv::Rodrigues as called by trackingFrame2Frame()
// (reference src/visualOdometry.cpp:161-189: useExtrinsicGuess = true, 500 iterations,
// reprojection error 0.5 px, confidence 0.999f, SOLVEPNP_ITERATIVE, zero distortion).
//
// OpenCV's RANSAC loop is sequential with an adaptive iteration count.  Its random 5-subsets do
// not depend on earlier hypotheses (cv::RNG(-1) stream, duplicates redrawn), so all hypotheses are
// evaluated in parallel and the sequential control flow is replayed afterwards on the vote counts:
//   1. ransac_subsets_kernel   one thread per frame replays the RNG stream -> [iters][5] indices
//   then, in chunks of RANSAC_CHUNK hypotheses (a frame whose adaptive iteration count has already
//   dropped below a chunk's first hypothesis skips it -- with >= 60 % inliers OpenCV stops before 128):
//   2. epnp_kernel             one thread per hypothesis: 5-point EPnP (vo_epnp.h) -> rvec|tvec
//   3. vote_kernel             one wavefront per hypothesis: project all K points (f64 -> f32),
//                              squared error <= thr^2, wave sum -> inlier count
//   4. ransac_replay_kernel    one thread per frame continues "keep first strictly better,
//                              niters = RANSACUpdateNumIters(...)" over the new counts
//   (frames with exactly 4 points take no part in 1-4: OpenCV switches to P3P and returns solvePnP's answer
//    directly -- p3p_frame, thread 0 of the frame's select_refine workgroup, vo_p3p.h)
//   small launches: everything behind the first chunk is ONE launch (ransac_rest_kernel)
//   5. select_refine_kernel    one workgroup per frame: winning hypothesis and the last one OpenCV
//                              would have evaluated (its pose is the start of the final refinement
//                              because rvec/tvec are shared buffers), inlier mask, the CvLevMarq
//                              state machine with block-wide reductions of J^T J / J^T e, rvec -> R.
#include "vo_kernels.h"
#ifdef VO_DEV_VARIANTS
// developer build: 100 MHz time stamps of one EPnP hypothesis / one refinement (tools/pose_phases.py)
__device__ long long g_pose_prof[64];
#ifdef __HIP_DEVICE_COMPILE__
#define VO_EPNP_STAMP(i)                                                      \
    do {                                                                      \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {         \
            g_pose_prof[i] = (long long)wall_clock64();                       \
        }                                                                     \
    } while (0)
#define VO_POSE_NOW() ((long long)wall_clock64())
#endif
#endif
#ifndef VO_POSE_NOW
#define VO_POSE_NOW() 0ll
#endif
#include "vo_epnp.h"
#include "vo_svd_wide.h"
#include "vo_p3p.h"
#include "vo_seqtail.h"

#include <mutex>

#include <float.h>
#include <stdlib.h>

// Register budget of the two f64-heavy kernels, as minimum waves per SIMD (launch_bounds' second
// argument), a template parameter with two instantiations:
//   WAVES = 1  all 512 registers, hardly any spills: 1.1 ms stand-alone pose chain.  Used for small
//              batches (the single-frame drop-in calls), where the GPU is otherwise idle.
//   WAVES = 4  128 registers, heavy spilling, 2.2 ms stand-alone -- but such a wave fits on a SIMD next
//              to LK waves, whereas a 512-register wave can only start on a completely EMPTY SIMD, which
//              the next batch's LK launch (one hundred thousand workgroups) never leaves: next to LK the
//              WAVES = 1 chain took 5-10 ms and the following run ended up waiting for it
//              (gpurun_out/sweep1, r11 - r13).  Used when the caller says the GPU is crowded: the LK
//              launch of the same batch has enough features to keep every SIMD full for many rounds
//              (capi.hip: frames x points >= 65536); at the reference-default 340 points per frame LK
//              does not, and the 512-register chain is the faster one there (37 k vs 25 k frames/s).
//   WAVES = 2  256 registers: the middle ground for batches whose tracking stages are short but busy (the
//              reference-default 340-point load, the lock-step sequence loop): a 512-register wave needs a whole SIMD
//              to itself and keeps the NEXT step's pyramid / detection kernels waiting (bucket_kernel 0.54 ms instead
//              of 0.02 behind select_refine_kernel<1>, pyr_down 0.77 instead of 0.13 ms; profiles/r02), a 256-register
//              wave leaves half of it to them.

namespace vo {

// dynamic LDS of the workgroup (tests/host_check/kernel_emu.cpp runs these kernels on the CPU emulator, which hands the
// block's buffer over through emu::dyn_shared())
#ifdef VO_HOST_EMUL
#define VO_DYN_LDS(type, name) type *name = (type *)emu::dyn_shared()
#else
#define VO_DYN_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
#endif

constexpr int RANSAC_CHUNK = VO_EPNP_WS_HYPS;
constexpr int EPNP_UT_DOUBLES = VO_EPNP_UT_DOUBLES; // M^T M + its 12 column norms, per hypothesis
#define VO_SLIM_WAVES 6 // register budget of the slim pose kernels as waves per SIMD (6: 80 registers)

// ---- random 5-subsets: RANSACPointSetRegistrator::getSubset with cv::RNG(-1) ----
// cv::RNG is a multiply-with-carry generator whose raw 32-bit outputs do not depend on anything the caller passes: the
// stream of a solvePnPRansac call is the same in every call; only `next() % count` and the redraws of duplicates within
// a subset depend on the frame.  So the raw stream is tabulated once per device (rng_table_kernel), and a WAVEFRONT
// derives 64 subsets at a time from it: lane j first assumes that no earlier subset of the batch needed a redraw (its
// draws start at position 5 j), every lane reports how many draws it consumed, an exclusive prefix sum gives the true
// starts, lanes whose start moved recompute -- until nothing moves (lane 0 is right at once, every round settles at
// least one more lane; with 340 points 3 % of the subsets redraw, so two or three rounds).  Round 2 replayed the stream
// with one thread per frame: 0.14 ms at the head of every pose solve; this takes a few microseconds.
// Exact by construction; a frame that would run past the table falls back to the serial generator.
constexpr int RNG_TABLE = 1 << 15; // raw draws kept per device (128 KB): 500 subsets of 5 need ~2 600, 1 000 of them with 6 points ~9 000

__global__ void rng_table_kernel(uint32_t *__restrict__ raw, int n)
{
    if (blockIdx.x != 0 || threadIdx.x != 0)
        return;
    uint64_t state = 0xffffffffffffffffull; // cv::RNG rng((uint64)-1)
    for (int i = 0; i < n; i++) {
        state = (uint64_t)(uint32_t)state * 4164903690U + (uint32_t)(state >> 32);
        raw[i] = (uint32_t)state;
    }
}

// the serial generator (round 2's kernel body): subsets [first, last) continuing from stream position `pos`
__device__ void subsets_serial(uint32_t pos, int first, int last, int count, int32_t *__restrict__ out, uint32_t *pos_out)
{
    uint64_t state = 0xffffffffffffffffull;
    for (uint32_t i = 0; i < pos; i++)
        state = (uint64_t)(uint32_t)state * 4164903690U + (uint32_t)(state >> 32);
    const uint64_t recip = 0xFFFFFFFFFFFFFFFFull / (uint32_t)count + 1;
    for (int it = first; it < last; it++) {
        int idx[5];
        for (int i = 0; i < 5; i++) {
            int idx_i;
            for (;;) {
                state = (uint64_t)(uint32_t)state * 4164903690U + (uint32_t)(state >> 32);
                pos++;
                idx_i = (int)__umul64hi(recip * (uint32_t)state, (uint64_t)(uint32_t)count);
                bool dup = false;
                for (int k = 0; k < i; k++)
                    dup |= idx[k] == idx_i;
                if (!dup)
                    break;
            }
            idx[i] = idx_i;
        }
        for (int i = 0; i < 5; i++)
            out[it * 5 + i] = idx[i];
    }
    *pos_out = pos;
}

// subsets [first, last) of one frame by ONE WAVEFRONT, continuing the stream at position pos0; returns the position behind them
__device__ __forceinline__ uint32_t subsets_wave(uint32_t pos0, int first, int last, int count, const uint32_t *__restrict__ raw,
                                                 int n_raw, int32_t *__restrict__ out, int lane)
{
    // rng.uniform(0, count) = next() % count with a divisor that is fixed for the whole stream: Lemire's exact
    // remainder by a precomputed 64-bit reciprocal (two multiplies) instead of a 32-bit division per draw
    const uint64_t recip = 0xFFFFFFFFFFFFFFFFull / (uint32_t)count + 1;
    bool overflow = false;
    int base = first;
    for (; base < last; base += 64) {
        const int m = min(64, last - base);
        const bool act = lane < m;
        uint32_t off = pos0 + 5u * (uint32_t)lane, used = 0;
        int idx[5] = {0, 0, 0, 0, 0};
        bool dirty = act, ovf = false;
        for (;;) {
            if (dirty) {
                uint32_t p = off;
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    int v = 0;
                    for (;;) {
                        if (p >= (uint32_t)n_raw) {
                            ovf = true;
                            break;
                        }
                        v = (int)__umul64hi(recip * raw[p++], (uint64_t)(uint32_t)count);
                        bool dup = false;
#pragma unroll
                        for (int k = 0; k < 5; k++)
                            dup |= (k < i) && idx[k] == v;
                        if (!dup)
                            break;
                    }
                    idx[i] = v;
                }
                used = p - off;
            }
            if (__any(ovf))
                break;
            uint32_t incl = act ? used : 0u; // inclusive prefix sum over the wavefront
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d, 64);
                if (lane >= d)
                    incl += t;
            }
            const uint32_t start = pos0 + incl - (act ? used : 0u);
            dirty = act && start != off;
            off = start;
            if (!__any(dirty)) {
                pos0 += __shfl(incl, 63, 64);
                break;
            }
        }
        if (__any(ovf)) {
            overflow = true;
            break;
        }
        if (act) {
#pragma unroll
            for (int i = 0; i < 5; i++)
                out[(base + lane) * 5 + i] = idx[i];
        }
    }
    if (overflow) { // past the table (never seen: it holds 3.5 x the worst expected demand): lane 0's serial generator
        if (lane == 0)
            subsets_serial(pos0, base, last, count, out, &pos0);
        pos0 = __shfl(pos0, 0, 64);
    }
    return pos0;
}

// one wavefront per frame.  The chunk [h0, h0 + hn) continues the frame's stream where the previous chunk left it
// (RansacState::rng holds the stream POSITION), and only as far as the frame's adaptive iteration count still reaches.
__global__ __launch_bounds__(64) void ransac_subsets_kernel(const int *__restrict__ n_pts, int n_frames, int iters, int h0,
                                                            int hn, const uint32_t *__restrict__ raw, int n_raw,
                                                            int32_t *__restrict__ subsets /* [B][iters][5] */,
                                                            RansacState *__restrict__ rstate)
{
    const int frame = blockIdx.x, lane = threadIdx.x;
    if (frame >= n_frames)
        return;
    RansacState st;
    if (h0 == 0) {
        st.it = 0;
        st.niters = iters > 1 ? iters : 1;
        st.max_good = 0;
        st.best = -1;
        st.rng = 0;
        st.arrive = 0;
        st.pad_ = 0;
    } else {
        st = rstate[frame];
    }
    const int count = n_pts[frame];
    int32_t *out = subsets + (size_t)frame * iters * 5;
    if (count == 5 && h0 == 0 && lane < 5) // model_points == npoints: solvePnP on all points in order
        out[lane] = lane;
    if (count > 5)
        st.rng = subsets_wave((uint32_t)st.rng, h0, min(min(h0 + hn, iters), st.niters), count, raw, n_raw, out, lane);
    if (lane == 0)
        rstate[frame] = st;
}

// GWS = true ("slim", round 4): the same lane-interleaved matrix in a GLOBAL workspace (one 156 x 64 block of doubles per
// workgroup, L2 / Infinity-Cache resident while the wave runs) instead of LDS, and a small register budget -- a wave that
// needs neither LDS nor half a SIMD's registers starts in the slot any retiring LK wave leaves (see launch_pnp_ransac).
// hypothesis h of a frame by ONE LANE: its five points, 5-point EPnP, the model record.  ut: the lane's column of the
// lane-interleaved 12 x 12 workspace (element idx at ut[idx * 64])
__device__ __forceinline__ void epnp_hypothesis(const float *__restrict__ xyz, const float2 *__restrict__ uv, size_t uv_stride,
                                                int cap, const int32_t *__restrict__ subsets, const PnpParams &prm, int frame,
                                                int h, double *ut, double *__restrict__ models)
{
    const int32_t *idx = subsets + ((size_t)frame * prm.iters + h) * 5;
    float x5[15], u5[10];
    for (int i = 0; i < 5; i++) {
        const int k = idx[i];
        const float *p = xyz + ((size_t)frame * cap + k) * 3;
        x5[3 * i] = p[0];
        x5[3 * i + 1] = p[1];
        x5[3 * i + 2] = p[2];
        const float2 q = uv[frame * uv_stride + k];
        u5[2 * i] = q.x;
        u5[2 * i + 1] = q.y;
    }
    double rv[3], tv[3];
    epnp5_solve_t<64>(x5, u5, prm.K, rv, tv, ut);
    double *m = models + ((size_t)frame * prm.iters + h) * 6;
    m[0] = rv[0];
    m[1] = rv[1];
    m[2] = rv[2];
    m[3] = tv[0];
    m[4] = tv[1];
    m[5] = tv[2];
}

template <int WAVES, bool GWS>
__global__ __launch_bounds__(64, WAVES) void epnp_kernel(const float *__restrict__ xyz,   // [B][cap][3]
                                                  const float2 *__restrict__ uv,    // frame f at uv + f*uv_stride
                                                  size_t uv_stride, const int *__restrict__ n_pts, int cap,
                                                  const int32_t *__restrict__ subsets, PnpParams prm,
                                                  const RansacState *__restrict__ rstate, int h0, int hn,
                                                  double *__restrict__ models /* [B][iters][6] */,
                                                  double *__restrict__ gws /* [frames][VO_EPNP_GWS_BLOCKS][156][64] or null */)
{
    // M^T M (12 x 12) + its column norms of every lane, lane-interleaved: element idx of lane l at
    // s_ut[idx * 64 + l] -> consecutive lanes hit consecutive 8-byte words (conflict-free ds_*_b64)
    // (dynamic LDS, (144 + 12) * 64 doubles: with a static array the compiler derives one wave per SIMD
    // from the LDS footprint and spends all 512 registers, ignoring the launch bound above)
    VO_DYN_LDS(double, s_lds);
    double *s_ut = GWS ? gws + ((size_t)blockIdx.y * VO_EPNP_GWS_BLOCKS + blockIdx.x) * (EPNP_UT_DOUBLES * 64) : s_lds;
    const int frame = blockIdx.y, h = h0 + blockIdx.x * 64 + threadIdx.x;
    const int count = n_pts[frame];
    if (count < 5 || h >= h0 + hn)
        return;
    // hypotheses beyond the iteration count the replay has already settled on are never looked at
    const int nh = count == 5 ? 1 : min(prm.iters, rstate[frame].niters);
    if (h >= nh)
        return;
    epnp_hypothesis(xyz, uv, uv_stride, cap, subsets, prm, frame, h, s_ut + threadIdx.x, models);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same solve for SMALL launches (a single frame, a lock-step step of a few sequences), where the pose chain is the latency
// of the call and the GPU is otherwise idle: FOUR kernels instead of one, each shaped for its part (round 3; developer-build
// time stamps of the kernel above, tools/pose_phases.py: 16 us set-up, 337 us 12 x 12 SVD, 108 us the three approximations,
// 6 us selection = 466 us per hypothesis, whatever the number of hypotheses):
//   epnp_prepare_kernel  a lane per hypothesis as above: control points, barycentric coordinates, M^T M -> workspace
//   svd12_wave_kernel    TWO WAVEFRONTS per hypothesis (vo_svd_wide.h): a lane per matrix column, every sum in the serial
//                        order through DPP row broadcasts, up to six independent pairs per time slot on the eight DPP rows,
//                        the next sweep started while this one finishes -- bit-identical
//   epnp_approx_kernel   a lane per (hypothesis, approximation): the three beta approximations + Gauss-Newton + R, t are
//                        independent of each other; blockIdx.z picks the approximation, so a workgroup runs one code path
//   epnp_select_kernel   a lane per hypothesis: best of three, R -> rvec, the model record
// Why kernels and not phases of one kernel: the solver is ~35 k instructions of mostly straight-line code; a first version
// with a DPP row per hypothesis inside ONE kernel (4 wavefronts per workgroup) ran its one-lane phases at HALF speed --
// wavefronts of a CU that drift apart evict each other's loops from the instruction cache (465 us per hypothesis, no gain;
// with one wavefront per workgroup 327 us, but only up to ~128 wavefronts per launch: profiles/r03_pose_chain_experiments.md).  Here the
// wide part is a kernel of a few hundred instructions, and the long code runs in a few wavefronts as before.
// Costs several times the VALU time of the kernel above per hypothesis (12 of 64 lanes do useful work in the sweeps):
// launch_pnp_ransac uses it while that is free.
constexpr int EPNP_WS = VO_EPNP_WS_DOUBLES; // doubles per hypothesis: Epnp5 (88) | At 144 | 12 spare | 3 x (rep, R[9], t[3]) | pad
constexpr int EPNP_WS_AT = 88, EPNP_WS_RES = 244; // (232 .. 243 spare)
static_assert(sizeof(Epnp5) == 88 * sizeof(double), "workspace layout");

// (as in epnp_kernel: hypotheses beyond the iteration count the replay has settled on are never looked at)
__device__ __forceinline__ bool epnp_hyp_active(int count, int h, int h0, int hn, const PnpParams &prm, const RansacState *rs)
{
    return count >= 5 && h < h0 + hn && h < (count == 5 ? 1 : min(prm.iters, rs->niters));
}

__global__ __launch_bounds__(64, 1) void epnp_prepare_kernel(const float *__restrict__ xyz, const float2 *__restrict__ uv,
                                                             size_t uv_stride, const int *__restrict__ n_pts, int cap,
                                                             const int32_t *__restrict__ subsets, PnpParams prm,
                                                             const RansacState *__restrict__ rstate, int h0, int hn,
                                                             double *__restrict__ ws /* [B][hn][EPNP_WS] */)
{
    VO_DYN_LDS(double, s_ut); // 144 x 64, lane-interleaved as in epnp_kernel
    const int frame = blockIdx.y, hl = blockIdx.x * 64 + threadIdx.x, h = h0 + hl;
    if (!epnp_hyp_active(n_pts[frame], h, h0, hn, prm, rstate + frame))
        return;
    const int32_t *idx = subsets + ((size_t)frame * prm.iters + h) * 5;
    float x5[15], u5[10];
    for (int i = 0; i < 5; i++) {
        const int k = idx[i];
        const float *p = xyz + ((size_t)frame * cap + k) * 3;
        x5[3 * i] = p[0];
        x5[3 * i + 1] = p[1];
        x5[3 * i + 2] = p[2];
        const float2 q = uv[frame * uv_stride + k];
        u5[2 * i] = q.x;
        u5[2 * i + 1] = q.y;
    }
    double *w = ws + ((size_t)frame * hn + hl) * EPNP_WS;
    Epnp5 e;
    epnp5_prepare<64>(x5, u5, prm.K, e, s_ut + threadIdx.x);
    *(Epnp5 *)w = e;
#pragma unroll 4
    for (int i = 0; i < 144; i++)
        w[EPNP_WS_AT + i] = s_ut[i * 64 + threadIdx.x];
}

__global__ __launch_bounds__(128) void svd12_wave_kernel(const int *__restrict__ n_pts, PnpParams prm,
                                                         const RansacState *__restrict__ rstate, int h0, int hn,
                                                         double *__restrict__ ws)
{
    __shared__ __attribute__((aligned(16))) double s_at[144];
    __shared__ double s_w[12];
    __shared__ int s_flag;
    const int frame = blockIdx.y, hl = blockIdx.x, tid = threadIdx.x;
    if (!epnp_hyp_active(n_pts[frame], h0 + hl, h0, hn, prm, rstate + frame)) // (uniform over the workgroup)
        return;
    double *w = ws + ((size_t)frame * hn + hl) * EPNP_WS + EPNP_WS_AT;
    for (int i = tid; i < 144; i += 128)
        s_at[i] = w[i];
    __syncthreads();
    jacobi12_pipe_sweeps(s_at, s_w, &s_flag, tid); // two wavefronts: up to six independent pairs per time slot
    __syncthreads();
    if (tid == 0)
        jacobi12_finish(s_at, s_w);
    __syncthreads();
    if (tid < 48) // rows 8 .. 11: the null-space basis is all the rest of the solver reads
        w[96 + tid] = s_at[96 + tid];
}

__global__ __launch_bounds__(64, 1) void epnp_approx_kernel(const int *__restrict__ n_pts, PnpParams prm,
                                                            const RansacState *__restrict__ rstate, int h0, int hn,
                                                            double *__restrict__ ws)
{
    const int frame = blockIdx.y, hl = blockIdx.x * 64 + threadIdx.x;
    if (!epnp_hyp_active(n_pts[frame], h0 + hl, h0, hn, prm, rstate + frame))
        return;
    double *w = ws + ((size_t)frame * hn + hl) * EPNP_WS;
    Epnp5 e = *(const Epnp5 *)w;
    const double *ut = w + EPNP_WS_AT;
    double L[60], rho[6], R[9], t[3], rep;
    epnp5_L_rho<1>(e, ut, L, rho);
    if (blockIdx.z == 0)
        rep = epnp5_approx<1, 0>(e, ut, L, rho, R, t);
    else if (blockIdx.z == 1)
        rep = epnp5_approx<1, 1>(e, ut, L, rho, R, t);
    else
        rep = epnp5_approx<1, 2>(e, ut, L, rho, R, t);
    double *o = w + EPNP_WS_RES + 13 * blockIdx.z;
    o[0] = rep;
#pragma unroll
    for (int k = 0; k < 9; k++)
        o[1 + k] = R[k];
#pragma unroll
    for (int k = 0; k < 3; k++)
        o[10 + k] = t[k];
}

__global__ __launch_bounds__(64) void epnp_select_kernel(const int *__restrict__ n_pts, PnpParams prm,
                                                         const RansacState *__restrict__ rstate, int h0, int hn,
                                                         const double *__restrict__ ws, double *__restrict__ models)
{
    const int frame = blockIdx.y, hl = blockIdx.x * 64 + threadIdx.x, h = h0 + hl;
    if (!epnp_hyp_active(n_pts[frame], h, h0, hn, prm, rstate + frame))
        return;
    const double *o = ws + ((size_t)frame * hn + hl) * EPNP_WS + EPNP_WS_RES;
    double rep[3], R[3][9], t[3][3], rv[3], tv[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        rep[a] = o[13 * a];
#pragma unroll
        for (int k = 0; k < 9; k++)
            R[a][k] = o[13 * a + 1 + k];
#pragma unroll
        for (int k = 0; k < 3; k++)
            t[a][k] = o[13 * a + 10 + k];
    }
    epnp5_select(rep, R[0], R[1], R[2], t[0], t[1], t[2], rv, tv);
    double *m = models + ((size_t)frame * prm.iters + h) * 6;
    m[0] = rv[0];
    m[1] = rv[1];
    m[2] = rv[2];
    m[3] = tv[0];
    m[4] = tv[1];
    m[5] = tv[2];
}

// squared reprojection error exactly as PnPRansacCallback::computeError: projection in f64,
// stored as f32, difference and squared norm in f32
__device__ __forceinline__ bool is_inlier(const double *R, const double *t, double fx, double fy, double cx,
                                          double cy, const float *p, float2 q, float thr2)
{
    double uvd[2];
    project_point(R, t, nullptr, fx, fy, cx, cy, (double)p[0], (double)p[1], (double)p[2], uvd, nullptr,
                  nullptr);
    const float dx = q.x - (float)uvd[0], dy = q.y - (float)uvd[1];
    const float e = dx * dx + dy * dy;
    return e <= thr2;
}

// inlier count of hypothesis h over the frame's `count` points by ONE WAVEFRONT (every lane returns the sum)
__device__ __forceinline__ int vote_hypothesis(const float *__restrict__ xyz, const float2 *__restrict__ uv, size_t uv_stride,
                                               int cap, const PnpParams &prm, const double *__restrict__ models, int frame, int h,
                                               int count, int lane)
{
    const double *m = models + ((size_t)frame * prm.iters + h) * 6;
    double R[9], t[3] = {m[3], m[4], m[5]};
    rodrigues_v2m(m, R, nullptr);
    const double fx = prm.K[0], fy = prm.K[4], cx = prm.K[2], cy = prm.K[5];
    const double thr = (double)prm.reproj;
    const float thr2 = (float)(thr * thr);
    int good = 0;
    for (int i = lane; i < count; i += 64)
        good += is_inlier(R, t, fx, fy, cx, cy, xyz + ((size_t)frame * cap + i) * 3, uv[frame * uv_stride + i],
                          thr2);
#pragma unroll
    for (int mm = 32; mm >= 1; mm >>= 1)
        good += __shfl_xor(good, mm, 64);
    return good;
}

__global__ __launch_bounds__(64) void vote_kernel(const float *__restrict__ xyz, const float2 *__restrict__ uv,
                                                  size_t uv_stride, const int *__restrict__ n_pts, int cap,
                                                  PnpParams prm, const double *__restrict__ models,
                                                  const RansacState *__restrict__ rstate, int h0,
                                                  int *__restrict__ counts /* [B][iters] */)
{
    const int frame = blockIdx.y, h = h0 + blockIdx.x, lane = threadIdx.x;
    const int count = n_pts[frame];
    if (count <= 5 || h >= prm.iters || h >= rstate[frame].niters)
        return;
    const int good = vote_hypothesis(xyz, uv, uv_stride, cap, prm, models, frame, h, count, lane);
    if (lane == 0)
        counts[(size_t)frame * prm.iters + h] = good;
}

// calib3d/ptsetreg.cpp RANSACUpdateNumIters
__device__ int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters)
{
    p = p > 0. ? p : 0.;
    p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.;
    ep = ep < 1. ? ep : 1.;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, (double)modelPoints);
    if (denom < DBL_MIN)
        return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)rint(num / denom);
}

// RANSACPointSetRegistrator::run on the vote counts of hypotheses st.it .. end - 1:
//     for (; it < niters && it < end; it++) if (good[it] > max(max_good, 4)) { max_good = good[it]; best = it; niters = update(...); }
// by ONE WAVEFRONT, 64 counts at a time: the lanes load them side by side, a ballot finds the first one that beats the
// running best, every lane takes it over (the new bound is a function of wave-uniform values), the search goes on behind it.
// One thread walking the counts took 8-14 us per chunk of 128 -- one dependent load and branch per hypothesis (round 5).
// Every lane returns the same state.
__device__ __forceinline__ void replay_frame(RansacState &st, int count, const int *__restrict__ cf, int end, const PnpParams &prm,
                                             int lane)
{
    int it = st.it;
    while (it < st.niters && it < end) {
        int lim = st.niters < end ? st.niters : end; // (> it)
        const int h = it + lane;
        const int good = h < lim ? cf[h] : -1;
        int from = 0, stop = -1; // lanes below `from` have been passed; stop: the loop ends behind this hypothesis
        for (;;) {
            const int thr = st.max_good > 4 ? st.max_good : 4;
            const unsigned long long m = VO_BALLOT(lane >= from && h < lim && good > thr);
            if (m == 0ull)
                break;
            const int j = __builtin_ctzll(m);
            const int g = VO_READLANE(good, j);
            st.max_good = g;
            st.best = it + j;
            st.niters = ransac_update_num_iters(prm.confidence, (double)(count - g) / count, 5, st.niters);
            lim = st.niters < end ? st.niters : end;
            if (lim <= it + j + 1) { // the new bound is already behind us: `it++`, then the loop condition fails
                stop = it + j + 1;
                break;
            }
            from = j + 1;
        }
        if (stop >= 0) {
            it = stop;
            break;
        }
        it = it + 64 < lim ? it + 64 : lim;
    }
    st.it = it;
}

// solvePnPRansac with exactly four correspondences: `npoints == 4 -> model_points = 4, SOLVEPNP_P3P`, and model_points
// being npoints the call IS solvePnP(P3P): first of solveP3P's sorted solutions, no refinement, all four points inliers;
// no solution -> false, rvec / tvec untouched (lm_iters = -1 marks "pose buffers untouched" for the host side and the
// lock-step loop), inliers released.  One thread.
__device__ void p3p_frame(const float *__restrict__ xyz, const float2 *__restrict__ uv, size_t uv_stride, int cap, int frame,
                          const PnpParams &prm, int32_t *__restrict__ inliers, PnpResult *__restrict__ results)
{
    float x4[12], u4[8];
    for (int i = 0; i < 4; i++) {
        const float *p = xyz + ((size_t)frame * cap + i) * 3;
        x4[3 * i] = p[0];
        x4[3 * i + 1] = p[1];
        x4[3 * i + 2] = p[2];
        const float2 q = uv[frame * uv_stride + i];
        u4[2 * i] = q.x;
        u4[2 * i + 1] = q.y;
    }
    PnpResult &res = results[frame];
    double rv[3] = {0, 0, 0}, tv[3] = {0, 0, 0};
    const int ns = p3p4_solve(x4, u4, prm.K, rv, tv);
    for (int k = 0; k < 3; k++) {
        res.rvec[k] = rv[k];
        res.tvec[k] = tv[k];
    }
    rodrigues_v2m(rv, res.R, nullptr);
    res.niters = 1;
    res.max_good = ns > 0 ? 4 : 0;
    res.best_iter = ns > 0 ? 0 : -1;
    res.lm_iters = ns > 0 ? 0 : -1;
    res.n_inliers = ns > 0 ? 4 : 0;
    res.status = ns > 0 ? 1 : 0;
    if (ns > 0)
        for (int i = 0; i < 4; i++)
            inliers[(size_t)frame * cap + i] = i;
}

// RANSACPointSetRegistrator::run continued over a chunk's vote counts, one wavefront per frame.  (A light kernel on purpose:
// in batch mode it runs next to the following step's LK launch and must fit into whatever an LK wave leaves free -- with the
// four-point solve inlined here it needed 128 registers + 1 KB of scratch per lane and its 256 wavefronts took 3 ms on average
// to find room, profiles/r05_kernel_stats_batch_p3p_in_replay.csv.  The four-point frames are solved by the refinement kernel.)
__global__ __launch_bounds__(64) void ransac_replay_kernel(const int *__restrict__ n_pts, int n_frames, PnpParams prm, int h_end,
                                                            const int *__restrict__ counts, RansacState *__restrict__ rstate)
{
    const int frame = blockIdx.x, lane = threadIdx.x;
    if (frame >= n_frames)
        return;
    const int count = n_pts[frame];
    if (count <= 5)
        return;
    RansacState st = rstate[frame];
    replay_frame(st, count, counts + (size_t)frame * prm.iters, min(h_end, prm.iters), prm, lane);
    if (lane == 0)
        rstate[frame] = st;
}

// Everything of a solve BEHIND the first chunk, in ONE launch (round 5; small launches): subsets, EPnP, votes and the replay of
// hypotheses h0 .. h0 + hn - 1 for the frames whose adaptive iteration count reaches that far.  With >= 55 % inliers OpenCV
// stops inside the first 128 iterations, so for an ordinary frame every workgroup reads the frame's state and returns -- one
// idle launch where round 4 had four (ransac_subsets + epnp_kernel + vote + replay: 20 us of every synchronous call,
// profiles/r04_track_frame_timeline.txt).  A frame that does go on: workgroup x (one wavefront) of the frame
//   1. draws the subsets of the WHOLE chunk (every workgroup the same ones: the stream is serial, a wavefront does 64 subsets
//      in a few microseconds, and nobody waits for anybody),
//   2. solves its 64 hypotheses, a lane each (epnp_hypothesis, as epnp_kernel),
//   3. counts their inliers, the wavefront one hypothesis at a time (vote_hypothesis, as vote_kernel),
//   4. arrives (RansacState::arrive); the workgroup that arrives last replays the control flow over all counts and writes the
//      frame's state -- every other workgroup has read the state before it arrived, so nobody sees the new one too early.
// Same device functions as the four kernels, same results.
// The 12 x 12 matrices of its 64 hypotheses live in a GLOBAL workspace (`rest_ws`: the lane-interleaved block epnp_kernel keeps
// in 78 KB of LDS) and the kernel is built for two waves per SIMD: next to the following frame's LK launch (the lock-step loop)
// a wavefront that wants a whole SIMD's registers and half a CU's LDS waited 22-46 us for them -- to find out that it has nothing
// to do (gpurun_out/r5_21).  The rare frame that does go on pays for it with a slower solve.
template <int WAVES>
__global__ __launch_bounds__(64, WAVES) void ransac_rest_kernel(const float *__restrict__ xyz, const float2 *__restrict__ uv,
                                                                size_t uv_stride, const int *__restrict__ n_pts, int cap,
                                                                int32_t *__restrict__ subsets, PnpParams prm,
                                                                RansacState *__restrict__ rstate, int h0, int hn,
                                                                double *__restrict__ models, int *__restrict__ counts,
                                                                const uint32_t *__restrict__ raw, int n_raw,
                                                                int n_groups /* workgroups per frame = gridDim.x */,
                                                                double *__restrict__ rest_ws /* [frames][n_groups][156][64] */)
{
    const int frame = blockIdx.y, lane = threadIdx.x;
    double *s_ut = rest_ws + ((size_t)frame * n_groups + blockIdx.x) * (EPNP_UT_DOUBLES * 64);
    const int count = n_pts[frame];
    if (count <= 5)
        return;
    RansacState st = rstate[frame];
    if (st.it >= st.niters || st.it >= prm.iters)
        return; // the first chunk settled this frame
    const int last = min(min(h0 + hn, prm.iters), st.niters);
    st.rng = subsets_wave((uint32_t)st.rng, h0, last, count, raw, n_raw, subsets + (size_t)frame * prm.iters * 5, lane);
    wide_sync(true); // the lanes read subsets other lanes of this wavefront wrote
    const int hb = h0 + blockIdx.x * 64;
    if (hb + lane < last)
        epnp_hypothesis(xyz, uv, uv_stride, cap, subsets, prm, frame, hb + lane, s_ut + lane, models);
    wide_sync(true); // ... and models other lanes wrote
    for (int h = hb; h < min(hb + 64, last); h++) {
        const int good = vote_hypothesis(xyz, uv, uv_stride, cap, prm, models, frame, h, count, lane);
        if (lane == 0)
            counts[(size_t)frame * prm.iters + h] = good;
    }
    __threadfence(); // this workgroup's counts before its arrival
    int arrived = 0;
    if (lane == 0)
        arrived = atomicAdd(&rstate[frame].arrive, 1);
    if (VO_READFIRSTLANE(arrived) != n_groups - 1)
        return;
    __threadfence(); // every other workgroup's counts
    replay_frame(st, count, counts + (size_t)frame * prm.iters, min(h0 + hn, prm.iters), prm, lane);
    if (lane == 0) {
        st.arrive = 0;
        rstate[frame] = st;
    }
}

constexpr int LM_NRED = 28; // 21 (upper JtJ) + 6 (JtErr) + 1 (|err|^2)
constexpr int LM_PITCH = 29; // row pitch of the per-thread partial sums in LDS

// the work of one frame's workgroup (256 threads, all of them call it; every `return` below is block-uniform)
__device__ __forceinline__ void select_refine_frame(const float *__restrict__ xyz, const float2 *__restrict__ uv,
                                                    size_t uv_stride, const int *__restrict__ n_pts, int cap,
                                                    const PnpParams &prm, const double *__restrict__ models,
                                                    const RansacState *__restrict__ rstate,
                                                    int32_t *__restrict__ inliers /* [B][cap] */,
                                                    PnpResult *__restrict__ results)
{
    __shared__ int s_best, s_last, s_niters, s_maxgood, s_ninl;
    __shared__ int s_wave[4];
    __shared__ double s_param[6];
    __shared__ double s_red[8][LM_NRED];
    __shared__ double s_sum[LM_NRED];
    __shared__ int s_flags[3]; // [0] proceed & want_err, [1] want_J, [2] a 6 x 6 solve is due
    __shared__ __attribute__((aligned(16))) double s_acc[256 * LM_PITCH]; // per-thread partial sums of a pass
    __shared__ double s_At[36], s_Vt[36], s_W6[6], s_b[6];             // the Levenberg-Marquardt step's linear system

    const int frame = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int count = n_pts[frame];
    PnpResult &res = results[frame];
#ifdef VO_DEV_VARIANTS
    const bool prof = frame == 0 && tid == 0;
    long long t_solve = 0, n_solve = 0, t_pass[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (prof)
        g_pose_prof[16] = VO_POSE_NOW();
#endif
    if (count < 5) {
        if (tid == 0) {
            if (count == 4) { // OpenCV's SOLVEPNP_P3P switch: the whole solve by one thread (round 5: a kernel of its own in
                              // front of this one -- 5 us of every synchronous call for a case an ordinary frame never is)
                p3p_frame(xyz, uv, uv_stride, cap, frame, prm, inliers, results);
            } else {
                res.status = -1; // CV_Assert(npoints >= 4)
                res.n_inliers = 0;
                res.niters = res.best_iter = res.max_good = res.lm_iters = 0;
            }
        }
        return;
    }
    const double *mf = models + (size_t)frame * prm.iters * 6;
    const float *X = xyz + (size_t)frame * cap * 3;
    const float2 *U = uv + frame * uv_stride;
    int32_t *inl = inliers + (size_t)frame * cap;

    if (count == 5) { // direct solvePnP(EPNP) on the 5 points, all inliers, no refinement
        if (tid < 5)
            inl[tid] = tid;
        if (tid == 0) {
            for (int k = 0; k < 3; k++) {
                res.rvec[k] = mf[k];
                res.tvec[k] = mf[3 + k];
            }
            rodrigues_v2m(res.rvec, res.R, nullptr);
            res.n_inliers = 5;
            res.status = 1;
            res.niters = 1;
            res.best_iter = 0;
            res.max_good = 5;
            res.lm_iters = 0;
        }
        return;
    }

    // ---- outcome of RANSACPointSetRegistrator::run (ransac_replay_kernel) ----
    if (tid == 0) {
        const RansacState st = rstate[frame];
        s_best = st.best;
        s_last = st.it - 1;
        s_niters = st.it;
        s_maxgood = st.max_good;
        s_ninl = 0;
    }
    __syncthreads();
    const int best = s_best, last = s_last;
    const double fx = prm.K[0], fy = prm.K[4], cx = prm.K[2], cy = prm.K[5];

    if (best < 0) { // no model: rvec/tvec hold the last evaluated hypothesis, inliers released
        if (tid == 0) {
            for (int k = 0; k < 3; k++) {
                res.rvec[k] = mf[last * 6 + k];
                res.tvec[k] = mf[last * 6 + 3 + k];
            }
            rodrigues_v2m(res.rvec, res.R, nullptr);
            res.n_inliers = 0;
            res.status = 0;
            res.niters = s_niters;
            res.best_iter = -1;
            res.max_good = 0;
            res.lm_iters = 0;
        }
        return;
    }

    // ---- inlier mask of the winning hypothesis, stable compaction into inl[] ----
    {
        double R[9], t[3] = {mf[best * 6 + 3], mf[best * 6 + 4], mf[best * 6 + 5]};
        rodrigues_v2m(mf + best * 6, R, nullptr);
        const double thr = (double)prm.reproj;
        const float thr2 = (float)(thr * thr);
        for (int start = 0; start < count; start += 256) {
            const int i = start + tid;
            const bool in = i < count && is_inlier(R, t, fx, fy, cx, cy, X + (size_t)i * 3, U[i], thr2);
            const unsigned long long m = __ballot(in);
            if (lane == 0)
                s_wave[wv] = __popcll(m);
            __syncthreads();
            int off = s_ninl;
            for (int w = 0; w < wv; w++)
                off += s_wave[w];
            if (in)
                inl[off + __popcll(m & ((1ull << lane) - 1ull))] = i;
            __syncthreads();
            if (tid == 0)
                s_ninl += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
            __syncthreads();
        }
    }
    const int n1 = s_ninl;
#ifdef VO_DEV_VARIANTS
    if (prof)
        g_pose_prof[17] = VO_POSE_NOW();
#endif

    // ---- CvLevMarq (cvFindExtrinsicCameraParams2, useExtrinsicGuess) from the LAST hypothesis ----
    // Thread 0 runs the solver's state machine; what it asks for is done by the workgroup:
    //   * residuals (and Jacobians) of the inliers at s_param, summed over the workgroup (below);
    //   * the step's 6 x 6 solve through the SVD (cv::solve(DECOMP_SVD)): its Jacobi sweeps by wavefront 0 (vo_svd_wide.h:
    //     15 pairs in 9 steps on the four DPP rows, V accumulated, every sum in the serial order -- bit-identical to
    //     solve_svd<6, 6>; one lane took 28 us per solve, four solves per frame: half of this kernel).
    enum { LM_DONE = 0, LM_STARTED = 1, LM_CALC_J = 2, LM_CHECK_ERR = 3 };
    // thread-0 private solver state
    double prevParam[6], JtJ[36], JtErr[6];
    double prevErrNorm = DBL_MAX, errNorm = DBL_MAX;
    int lambdaLg10 = -3, state = LM_STARTED, iters = 0;
    const int max_iter = 20;
    const double epsilon = (double)FLT_EPSILON;
    if (tid == 0)
        for (int k = 0; k < 6; k++)
            s_param[k] = mf[last * 6 + k];
    __syncthreads();
    auto take_J = [&]() { // thread 0: the workgroup's sums -> J^T J, J^T e; the parameters they were formed at -> prevParam
        int q = 0;
        for (int i = 0; i < 6; i++)
            for (int j = i; j < 6; j++) {
                JtJ[i * 6 + j] = s_sum[q];
                JtJ[j * 6 + i] = s_sum[q];
                q++;
            }
        for (int k = 0; k < 6; k++) {
            JtErr[k] = s_sum[21 + k];
            prevParam[k] = s_param[k];
        }
    };

    for (;;) {
#ifdef VO_DEV_VARIANTS
        const long long t_in = VO_POSE_NOW();
#endif
        if (tid == 0) {
            int want_J = 0, want_err = 0, proceed = 1, need_solve = 0;
            if (state == LM_DONE) {
                proceed = 0;
            } else if (state == LM_STARTED) {
                want_J = want_err = 1;
                state = LM_CALC_J;
            } else if (state == LM_CALC_J) {
                take_J();
                need_solve = 1;
                if (iters == 0)
                    prevErrNorm = sqrt(s_sum[27]);
                want_err = 1;
                state = LM_CHECK_ERR;
            } else { // LM_CHECK_ERR
                errNorm = sqrt(s_sum[27]);
                bool handled = false;
                if (errNorm > prevErrNorm) {
                    if (++lambdaLg10 <= 16) {
                        need_solve = 1;
                        want_err = 1;
                        state = LM_CHECK_ERR;
                        handled = true;
                    }
                }
                if (!handled) {
                    lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
                    double dn = 0, pn = 0;
                    for (int k = 0; k < 6; k++) {
                        const double d = s_param[k] - prevParam[k];
                        dn += d * d;
                        pn += prevParam[k] * prevParam[k];
                    }
                    if (++iters >= max_iter || sqrt(dn) / (sqrt(pn) + DBL_EPSILON) < epsilon) {
                        state = LM_DONE; // update() returns true with _err == 0 -> caller breaks
                    } else {
                        // CvLevMarq now asks for J^T J, J^T e and the error at the accepted parameters (state CALC_J) -- the
                        // very parameters the pass that has just delivered errNorm ran at.  That pass formed the Jacobian sums
                        // on the way (every pass does, round 5: same points, same order, hence the same bits as a pass of
                        // their own), so the step follows at once: one pass per iteration instead of two.
                        prevErrNorm = errNorm;
                        take_J();
                        need_solve = 1;
                        want_err = 1;
                        state = LM_CHECK_ERR;
                    }
                }
            }
            if (need_solve) { // CvLevMarq::step: (J^T J with its diagonal scaled by 1 + lambda) x = J^T e; At = A^T as solve_svd
                const double lambda = vo_lm_lambda(lambdaLg10); // exp(lambdaLg10 * log(10.)), glibc's bits (vo_math.h)
                for (int i = 0; i < 6; i++)
                    for (int k = 0; k < 6; k++)
                        s_At[i * 6 + k] = k == i ? JtJ[k * 6 + i] * (1. + lambda) : JtJ[k * 6 + i];
                for (int k = 0; k < 6; k++)
                    s_b[k] = JtErr[k];
            }
            s_flags[0] = proceed && want_err;
            s_flags[1] = want_J || want_err; // (the Jacobian sums with every pass, see above)
            s_flags[2] = need_solve;
        }
        __syncthreads();
        if (s_flags[2]) { // (uniform over the workgroup)
            if (wv == 0)
                jacobi6v_wave_sweeps(s_At, s_W6, s_Vt, lane);
            __syncthreads();
            if (tid == 0) {
                double x[6];
                jacobi_finish<6, true>(s_At, s_W6, s_Vt);
                svd_backsubst<6>(s_At, s_W6, s_Vt, s_b, x);
                for (int k = 0; k < 6; k++)
                    s_param[k] = prevParam[k] - x[k];
            }
            __syncthreads();
#ifdef VO_DEV_VARIANTS
            t_solve += VO_POSE_NOW() - t_in;
            n_solve++;
#endif
        }
        if (!s_flags[0])
            break;
        const bool want_J = s_flags[1] != 0;

        // ---- residuals (and Jacobian) of every inlier at s_param ----
        double acc[LM_NRED];
#pragma unroll
        for (int k = 0; k < LM_NRED; k++)
            acc[k] = 0;
#ifdef VO_DEV_VARIANTS
        const long long tp0 = VO_POSE_NOW();
        long long tp1 = 0, tp2 = 0;
#endif
        {
            double R[9], dRdr[27];
            const double rv[3] = {s_param[0], s_param[1], s_param[2]};
            const double t[3] = {s_param[3], s_param[4], s_param[5]};
            rodrigues_v2m(rv, R, want_J ? dRdr : nullptr);
#ifdef VO_DEV_VARIANTS
            tp1 = VO_POSE_NOW();
#endif
            // (the point of the NEXT round is fetched -- index, then coordinates: two dependent loads -- while this one's
            // projection is computed: the loop was bound by that latency, 1.3 us per point)
            int k = tid;
            float px = 0, py = 0, pz = 0;
            float2 q = make_float2(0, 0);
            if (k < n1) {
                const int i = inl[k];
                const float *p = X + (size_t)i * 3;
                px = p[0];
                py = p[1];
                pz = p[2];
                q = U[i];
            }
            for (; k < n1; k += 256) {
                const float cx_ = px, cy_ = py, cz_ = pz;
                const float2 cq = q;
                if (k + 256 < n1) {
                    const int i = inl[k + 256];
                    const float *p = X + (size_t)i * 3;
                    px = p[0];
                    py = p[1];
                    pz = p[2];
                    q = U[i];
                }
                double uvd[2], Ju[6], Jv[6];
                project_point(R, t, dRdr, fx, fy, cx, cy, (double)cx_, (double)cy_, (double)cz_, uvd, want_J ? Ju : nullptr,
                              want_J ? Jv : nullptr);
                const double eu = uvd[0] - (double)cq.x, ev = uvd[1] - (double)cq.y;
                acc[27] += eu * eu + ev * ev;
                if (want_J) {
                    int qq = 0;
#pragma unroll
                    for (int a = 0; a < 6; a++) {
#pragma unroll
                        for (int b = a; b < 6; b++)
                            acc[qq++] += Ju[a] * Ju[b] + Jv[a] * Jv[b];
                        acc[21 + a] += Ju[a] * eu + Jv[a] * ev;
                    }
                }
            }
        }
#ifdef VO_DEV_VARIANTS
        tp2 = VO_POSE_NOW();
#endif
        // ---- sum over the workgroup, in a fixed order: thread-major partial sums in LDS (row pitch 29: neighbouring threads
        // land in different banks), 8 x 28 threads add 32 of them each, 28 threads the 8 results.  (Butterfly sums through
        // ds_bpermute -- 12 per value and level set, 28 values -- took 6 us per pass, eight passes per frame.)
        if (want_J) {
#pragma unroll
            for (int k = 0; k < LM_NRED; k++)
                s_acc[tid * LM_PITCH + k] = acc[k];
        } else {
            s_acc[tid * LM_PITCH + 27] = acc[27];
        }
        __syncthreads();
        {
            const int part = tid / LM_NRED, k = tid - part * LM_NRED;
            if (part < 8 && (want_J || k == 27)) {
                double sum = 0;
                for (int i = 0; i < 32; i++)
                    sum += s_acc[(part * 32 + i) * LM_PITCH + k];
                s_red[part][k] = sum;
            }
        }
        __syncthreads();
        if (tid < LM_NRED && (want_J || tid == 27)) {
            double sum = 0;
            for (int part = 0; part < 8; part++)
                sum += s_red[part][tid];
            s_sum[tid] = sum;
        }
        __syncthreads();
#ifdef VO_DEV_VARIANTS
        if (prof) { // [22 + 4 j ..]: rodrigues, point loop, reduction, count -- j = 1 for passes with the Jacobian
            const int o = want_J ? 4 : 0;
            t_pass[o] += tp1 - tp0;
            t_pass[o + 1] += tp2 - tp1;
            t_pass[o + 2] += VO_POSE_NOW() - tp2;
            t_pass[o + 3] += 1;
        }
#endif
    }

    if (tid == 0) {
        for (int k = 0; k < 3; k++) {
            res.rvec[k] = s_param[k];
            res.tvec[k] = s_param[3 + k];
        }
        rodrigues_v2m(res.rvec, res.R, nullptr); // cv::Rodrigues(rvec, rotation), visualOdometry.cpp:188
        res.n_inliers = n1;
        res.status = 1;
        res.niters = s_niters;
        res.best_iter = best;
        res.max_good = s_maxgood;
        res.lm_iters = iters;
#ifdef VO_DEV_VARIANTS
        if (prof) {
            g_pose_prof[18] = t_solve;
            g_pose_prof[19] = n_solve;
            g_pose_prof[20] = VO_POSE_NOW();
            g_pose_prof[21] = n1;
            for (int k = 0; k < 8; k++)
                g_pose_prof[22 + k] = t_pass[k];
        }
#endif
    }
}

// one workgroup per frame.  Lock-step loop (tail.active != nullptr): thread 0 goes on with the tail of the reference's
// frame loop for this sequence -- euler gates + integrateOdometryStereo + one trajectory row (vo_seqtail.h) -- instead of
// a separate kernel behind the chain (round 2's seq_integrate_kernel: 7 us of work that waited up to 1.7 ms for a SIMD slot
// next to the following step's LK waves).
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void select_refine_kernel(const float *__restrict__ xyz,
                                                            const float2 *__restrict__ uv, size_t uv_stride,
                                                            const int *__restrict__ n_pts, int cap,
                                                            PnpParams prm, const double *__restrict__ models,
                                                            const RansacState *__restrict__ rstate,
                                                            int32_t *__restrict__ inliers /* [B][cap] */,
                                                            PnpResult *__restrict__ results, SeqTail tail)
{
    select_refine_frame(xyz, uv, uv_stride, n_pts, cap, prm, models, rstate, inliers, results);
    if (tail.active && threadIdx.x == 0 && tail.active[blockIdx.x])
        seq_integrate_frame(tail, blockIdx.x, results[blockIdx.x], tail.active[blockIdx.x]);
}

#ifndef VO_HOST_EMUL // ---- host side: launches ----
// the raw cv::RNG(-1) stream of the device the calling thread has selected (created on first use)
static const uint32_t *rng_table(hipStream_t stream)
{
    static std::mutex mu;
    static uint32_t *tab[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!tab[dev]) {
        uint32_t *p = nullptr;
        if (hipMalloc((void **)&p, sizeof(uint32_t) * RNG_TABLE) != hipSuccess)
            return nullptr;
        hipLaunchKernelGGL(rng_table_kernel, dim3(1), dim3(1), 0, stream, p, RNG_TABLE);
        (void)hipStreamSynchronize(stream); // once per device and process: later launches on other streams read it
        tab[dev] = p;
    }
    return tab[dev];
}

int pnp_init_device(hipStream_t stream) { return rng_table(stream) ? 0 : -1; }

void launch_ransac_subsets(const int *n_pts, int n_frames, int iters, int h0, int hn, int32_t *subsets,
                           RansacState *rstate, hipStream_t stream)
{
    // (no table -- hipMalloc failed, or a device ordinal beyond the cache: n_raw = 0 sends every frame to the kernel's exact
    // serial generator instead of dereferencing a null table)
    const uint32_t *tab = rng_table(stream);
    hipLaunchKernelGGL(ransac_subsets_kernel, dim3(n_frames), dim3(64), 0, stream, n_pts, n_frames, iters, h0, hn, tab,
                       tab ? RNG_TABLE : 0, subsets, rstate);
}

// RANSAC hypotheses + votes + control-flow replay, everything up to the choice of the winner.  Two chunks: the first
// RANSAC_CHUNK hypotheses (with >= 60 % inliers OpenCV stops before 128 iterations, so this is normally all), then ALL the
// remaining ones at once for the frames whose adaptive iteration count reaches further -- round 2 went on in steps of
// 128, i.e. 16 dependent launches per solve of which 12 found nothing to do.
void launch_pnp_ransac(const float *xyz, const float2 *uv, size_t uv_stride, const int *n_pts, int cap, int n_frames,
                       const PnpParams &prm, int32_t *subsets, double *models, int *counts, RansacState *state,
                       int waves /* 1 or 2 per SIMD: 512 / 256 registers; 4: the slim form (needs gws) */, hipStream_t stream,
                       double *epnp_ws /* [ws_frames][VO_EPNP_WS_HYPS][VO_EPNP_WS_DOUBLES] or null */, int ws_frames,
                       double *gws /* [n_frames][VO_EPNP_GWS_BLOCKS][156][64] or null */,
                       int wide_frames /* four-kernel form for launches of up to this many frames (the schedule's knob) */,
                       double *rest_ws /* [ws_frames][pnp_rest_groups(iters)][156][64]: ransac_rest_kernel's matrices, or null */)
{
    if (n_frames <= 0)
        return;
    // The 12 x 12 matrices of 64 hypotheses take 78 KB of LDS, so two such workgroups fill a CU's 160 KB completely and no
    // other kernel that uses LDS at all (pyr_down, FAST, LK with 1.9 KB per wave) can start a workgroup there until one of
    // them retires: when the chain starts on an idle GPU, the next LDS-using kernel of the tracking stream sits behind it
    // for ~0.6 ms (round-2 trace of the lock-step loop, 256 sequences).  Capping the solver at ONE workgroup per CU by asking
    // for more than half of the LDS (VO_EPNP_LDS_KB=82, developer build) was measured and is worse -- the chain gets longer
    // than two steps -- so the tracking stream instead starts each step with its LDS-free kernels (capi.hip, PYRAMID stage).
    static const size_t lds = [] {
        size_t need = (144 + 12) * 64 * sizeof(double), want = 0;
#ifdef VO_DEV_VARIANTS
        if (const char *e = getenv("VO_EPNP_LDS_KB"))
            want = (size_t)atoi(e) * 1024;
#endif
        return want > need ? want : need;
    }();
    // Small launches: the four-kernel form (epnp_prepare_kernel ...) for the first chunk; the rarely needed second chunk stays
    // with the one kernel (four launches that mostly find nothing to do would cost more than they save).
    int split_max = wide_frames > 0 ? wide_frames : VO_EPNP_SPLIT_DEFAULT_FRAMES;
#ifdef VO_DEV_VARIANTS
    static const int split_env = [] { const char *e = getenv("VO_EPNP_SPLIT_MAX"); return e ? atoi(e) : -1; }();
    if (split_env >= 0)
        split_max = split_env;
#endif
    const bool split = epnp_ws && n_frames <= split_max && n_frames <= ws_frames;
    // First chunk: 128 hypotheses cover what OpenCV's adaptive count asks for down to ~55 % inliers in ONE round of dependent
    // launches -- right where the chain's latency counts.  From 128 frames on the chain is hidden behind the next run's
    // tracking kernels and what counts is how much of the chip the EPnP workgroups hold while they are resident (78 KB of
    // LDS and half a SIMD's registers each): 64 first, the rest only for the frames that ask for more (measured, 256-frame
    // batch, ms per step with 128 | 64 | 32: 12.84 | 12.55 | 13.83 at ~2000 points, 3.57 | 3.18 | 3.64 at 340; lock-step loop
    // 256 sequences 4.03 | 4.04 | 4.23, 64 sequences 1.25 | 1.39 | 1.41 -- profiles/r03_pose_chain_experiments.md).
    int first_chunk = n_frames >= 128 ? 64 : RANSAC_CHUNK;
#ifdef VO_DEV_VARIANTS
    static const int chunk_env = [] { const char *e = getenv("VO_RANSAC_CHUNK"); return e ? atoi(e) : 0; }();
    if (chunk_env > 0 && chunk_env <= RANSAC_CHUNK && !split)
        first_chunk = chunk_env;
#endif
    for (int h0 = 0; h0 < prm.iters;) {
        const int hn = h0 == 0 ? min(first_chunk, prm.iters) : prm.iters - h0;
        const dim3 eg((hn + 63) / 64, n_frames);
        if (split && h0 > 0 && rest_ws) { // small launches: the rest of the solve in one launch, which an ordinary frame leaves at once
            const uint32_t *tab = rng_table(stream);
            hipLaunchKernelGGL(ransac_rest_kernel<2>, eg, dim3(64), 0, stream, xyz, uv, uv_stride, n_pts, cap, subsets, prm, state,
                               h0, hn, models, counts, tab, tab ? RNG_TABLE : 0, (int)eg.x, rest_ws);
            break;
        }
        launch_ransac_subsets(n_pts, n_frames, prm.iters, h0, hn, subsets, state, stream);
        if (split && h0 == 0) {
            hipLaunchKernelGGL(epnp_prepare_kernel, eg, dim3(64), 144 * 64 * sizeof(double), stream, xyz, uv, uv_stride, n_pts,
                               cap, subsets, prm, state, h0, hn, epnp_ws);
            hipLaunchKernelGGL(svd12_wave_kernel, dim3(hn, n_frames), dim3(128), 0, stream, n_pts, prm, state, h0, hn, epnp_ws);
            hipLaunchKernelGGL(epnp_approx_kernel, dim3(eg.x, n_frames, 3), dim3(64), 0, stream, n_pts, prm, state, h0, hn,
                               epnp_ws);
            hipLaunchKernelGGL(epnp_select_kernel, eg, dim3(64), 0, stream, n_pts, prm, state, h0, hn, epnp_ws, models);
        } else
#ifdef VO_DEV_VARIANTS
        if (waves >= 4 && gws && (int)eg.x <= VO_EPNP_GWS_BLOCKS) {
            // SLIM (round-4 experiment, developer build only): 12 x 12 matrices in a global workspace, VO_SLIM_WAVES waves per
            // SIMD worth of registers, no LDS -- such a wave starts wherever ONE LK wave has retired instead of waiting for half
            // an empty SIMD and 78 KB of LDS.  Bit-identical, and SLOWER in every configuration measured
            // (profiles/r04_experiments.md: headline step 12.45 -> 12.68 ms, 340-point step 3.15 -> 3.32 ... 3.72, lock-step loop
            // 3.71 -> 4.10): the scratch-resident solver takes 2 x as long and costs LK more than the fat one does.
            static const int sw = [] { const char *e = getenv("VO_SLIM_WAVES"); return e ? atoi(e) : VO_SLIM_WAVES; }();
            if (sw == 4)
                hipLaunchKernelGGL((epnp_kernel<4, true>), eg, dim3(64), 0, stream, xyz, uv, uv_stride, n_pts, cap, subsets, prm,
                                   state, h0, hn, models, gws);
            else if (sw == 5)
                hipLaunchKernelGGL((epnp_kernel<5, true>), eg, dim3(64), 0, stream, xyz, uv, uv_stride, n_pts, cap, subsets, prm,
                                   state, h0, hn, models, gws);
            else if (sw == 7)
                hipLaunchKernelGGL((epnp_kernel<7, true>), eg, dim3(64), 0, stream, xyz, uv, uv_stride, n_pts, cap, subsets, prm,
                                   state, h0, hn, models, gws);
            else
                hipLaunchKernelGGL((epnp_kernel<VO_SLIM_WAVES, true>), eg, dim3(64), 0, stream, xyz, uv, uv_stride, n_pts, cap,
                                   subsets, prm, state, h0, hn, models, gws);
        } else
#endif
        if (waves >= 2)
            hipLaunchKernelGGL((epnp_kernel<2, false>), eg, dim3(64), lds, stream, xyz, uv, uv_stride, n_pts, cap, subsets, prm, state,
                               h0, hn, models, (double *)nullptr);
        else
            hipLaunchKernelGGL((epnp_kernel<1, false>), eg, dim3(64), lds, stream, xyz, uv, uv_stride, n_pts, cap, subsets, prm, state,
                               h0, hn, models, (double *)nullptr);
        hipLaunchKernelGGL(vote_kernel, dim3(hn, n_frames), dim3(64), 0, stream, xyz, uv, uv_stride, n_pts, cap, prm, models,
                           state, h0, counts);
        hipLaunchKernelGGL(ransac_replay_kernel, dim3(n_frames), dim3(64), 0, stream, n_pts, n_frames, prm, h0 + hn, counts, state);
        h0 += hn;
    }
}

// the four-point frames (P3P), winner / inlier mask / Levenberg-Marquardt refinement / Rodrigues -- and, in the lock-step loop,
// the pose integration of every sequence (tail)
void launch_pnp_refine(const float *xyz, const float2 *uv, size_t uv_stride, const int *n_pts, int cap, int n_frames,
                       const PnpParams &prm, const double *models, const RansacState *state, int32_t *inliers,
                       PnpResult *results, int waves, const SeqTail &tail, hipStream_t stream)
{
    if (n_frames <= 0)
        return;
#ifdef VO_DEV_VARIANTS
    if (waves >= 4) // slim: 128 registers
        hipLaunchKernelGGL(select_refine_kernel<4>, dim3(n_frames), dim3(256), 0, stream, xyz, uv, uv_stride, n_pts,
                           cap, prm, models, state, inliers, results, tail);
    else
#endif
    if (waves >= 2)
        hipLaunchKernelGGL(select_refine_kernel<2>, dim3(n_frames), dim3(256), 0, stream, xyz, uv, uv_stride, n_pts,
                           cap, prm, models, state, inliers, results, tail);
    else
        hipLaunchKernelGGL(select_refine_kernel<1>, dim3(n_frames), dim3(256), 0, stream, xyz, uv, uv_stride, n_pts,
                           cap, prm, models, state, inliers, results, tail);
}

void launch_pnp(const float *xyz, const float2 *uv, size_t uv_stride, const int *n_pts, int cap, int n_frames,
                const PnpParams &prm, int32_t *subsets, double *models, int *counts, RansacState *state,
                int32_t *inliers, PnpResult *results, int waves, hipStream_t stream, double *epnp_ws, int ws_frames, double *gws,
                double *rest_ws)
{
    launch_pnp_ransac(xyz, uv, uv_stride, n_pts, cap, n_frames, prm, subsets, models, counts, state, waves, stream, epnp_ws,
                      ws_frames, gws, VO_EPNP_SPLIT_DEFAULT_FRAMES, rest_ws);
    launch_pnp_refine(xyz, uv, uv_stride, n_pts, cap, n_frames, prm, models, state, inliers, results, waves, SeqTail(), stream);
}

#endif // VO_HOST_EMUL

#ifdef VO_DEV_VARIANTS
int pose_prof_read(long long *out64)
{
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_pose_prof), sizeof(long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

} // namespace vo
