// vo_svd_wide.h -- the Jacobi sweeps of EPnP's 12 x 12 SVD (and of the refinement's 6 x 6 solve) run by one or two
// WAVEFRONTS per matrix (round 3).
//
// Why: one hypothesis per lane (vo_epnp.h + jacobi_svd<12, 12, false, 64>) costs 466 us per solve, 337 us of them in
// this SVD (developer-build time stamps, tools/pose_phases.py): ~307 rotations + ~114 skipped pairs, each a chain of
// dependent f64 instructions issued by one lane.  A single frame (vo_track_frame, a lock-step step of a few sequences)
// has 128 hypotheses and an otherwise idle GPU: there the SVD is the latency of the call.  Two things are spread here,
// neither of which changes a bit of the result:
//   * inside a pair (i, j): lane k < 12 of a DPP row owns COLUMN k (element k of every row of At), so the rotation of
//     the two rows and the products of a dot product are one instruction each.  What cannot be spread is the ORDER of a
//     floating-point sum: s = ((0 + x_0) + x_1) + ... + x_11 is formed by twelve row-broadcast + add steps
//     (v_mov_b64_dpp row_newbcast:k feeds lane k's term to the whole row); the rotation parameters are then computed
//     redundantly by every lane of the row from those sums;
//   * across pairs: the serial order (0,1), (0,2) ... (10,11) makes pair (i, j) wait for (i, j - 1) and (i - 1, j) only
//     -- the last pairs before it that touch row i / row j -- so all pairs with the same i + j are independent and see
//     exactly the rows they would see in the serial order -- and the next sweep's early pairs do not have to wait for this
//     sweep's late ones either (JacobiPipe below): 12 time slots per sweep instead of 66 pairs, each slot up to six pairs
//     on the DPP rows of two wavefronts.
// Bit-identical to jacobi_svd<12, 12, false> by construction and by tests/test_kernel_emulation.py (CPU emulator: this
// file against the serial routine on random, rank-deficient and degenerate matrices) and the GPU parity tests of the
// pose solve.
//
// Layout: At[144] (row i at At + 12 i) and W[12] (squared row norms) in LDS or any memory the wavefront shares; rows of
// the matrix move between DPP rows through that memory, in program order of the one wavefront (VO_WAVE_SYNC keeps the
// compiler from moving memory operations across a step boundary; the hardware executes a wavefront's LDS operations in
// order).  After the sweeps ONE lane runs jacobi12_finish (final norms, descending selection sort, normalisation).
#pragma once

#include "vo_linalg.h"

namespace vo {

#if defined(VO_HOST_EMUL)
// CPU emulator (tests/host_check/hip_emu.h): lanes exchange through the emulator's per-block buffer
static inline double row_bcast_f64(double v, int src_in_row)
{
    const int lane = emu::lane_id(), src = (lane & ~15) + src_in_row;
    uint64_t u;
    memcpy(&u, &v, 8);
    const uint32_t lo = emu::exchange((uint32_t)u, src, 0), hi = emu::exchange((uint32_t)(u >> 32), src, 0);
    u = ((uint64_t)hi << 32) | lo;
    memcpy(&v, &u, 8);
    return v;
}
// s = ((0 + x_0) + x_1) + ... + x_{N-1} over the first N lanes of the caller's DPP row
template <int N>
static inline double row_ordered_sum(double x)
{
    double s = 0.0;
    for (int k = 0; k < N; k++)
        s = s + row_bcast_f64(x, k);
    return s;
}
template <int N>
static inline void row_ordered_sum_x2(double x, double y, double &sx, double &sy)
{
    sx = row_ordered_sum<N>(x);
    sy = row_ordered_sum<N>(y);
}
#elif defined(__HIPCC__)
#define VO_BC_ADD(K, S, X)                                                                   \
    "v_mov_b64_dpp %[t], %[" X "] row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"        \
    "v_add_f64 %[" S "], %[" S "], %[t]\n\t"
#define VO_BC_ADD6(S, X) VO_BC_ADD(0, S, X) VO_BC_ADD(1, S, X) VO_BC_ADD(2, S, X) VO_BC_ADD(3, S, X) VO_BC_ADD(4, S, X) VO_BC_ADD(5, S, X)
#define VO_BC_ADD6B(S, X) VO_BC_ADD(6, S, X) VO_BC_ADD(7, S, X) VO_BC_ADD(8, S, X) VO_BC_ADD(9, S, X) VO_BC_ADD(10, S, X) VO_BC_ADD(11, S, X)
#define VO_BC_ADD2(K) VO_BC_ADD(K, "s", "x") VO_BC_ADD(K, "r", "y")
// (s_nop 4: a DPP instruction must not read a VGPR in the two slots after the VALU write that produced it, nor follow a
// VALU write of EXEC by fewer than five -- the compiler keeps those distances for its own instructions but does not look
// inside an asm block; five idle cycles per chain of 24 to 48 instructions)
template <int N>
__device__ __forceinline__ double row_ordered_sum(double x)
{
    static_assert(N == 6 || N == 12, "chain lengths built below");
    double s = 0.0;
#if defined(__HIP_DEVICE_COMPILE__) // (the host pass of hipcc only needs the declaration)
    double t;
    if (N == 6)
        asm volatile("s_nop 4\n\t" VO_BC_ADD6("s", "x") : [s] "+v"(s), [t] "=&v"(t) : [x] "v"(x));
    else
        asm volatile("s_nop 4\n\t" VO_BC_ADD6("s", "x") VO_BC_ADD6B("s", "x") : [s] "+v"(s), [t] "=&v"(t) : [x] "v"(x));
#else
    s = x;
#endif
    return s;
}
// two independent sums in one block (their dependent adds interleave)
template <int N>
__device__ __forceinline__ void row_ordered_sum_x2(double x, double y, double &sx, double &sy)
{
    static_assert(N == 6 || N == 12, "chain lengths built below");
    double s = 0.0, r = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
    double t;
    if (N == 6)
        asm volatile("s_nop 4\n\t" VO_BC_ADD2(0) VO_BC_ADD2(1) VO_BC_ADD2(2) VO_BC_ADD2(3) VO_BC_ADD2(4) VO_BC_ADD2(5)
                     : [s] "+v"(s), [r] "+v"(r), [t] "=&v"(t)
                     : [x] "v"(x), [y] "v"(y));
    else
        asm volatile("s_nop 4\n\t" VO_BC_ADD2(0) VO_BC_ADD2(1) VO_BC_ADD2(2) VO_BC_ADD2(3) VO_BC_ADD2(4) VO_BC_ADD2(5) VO_BC_ADD2(6)
                         VO_BC_ADD2(7) VO_BC_ADD2(8) VO_BC_ADD2(9) VO_BC_ADD2(10) VO_BC_ADD2(11)
                     : [s] "+v"(s), [r] "+v"(r), [t] "=&v"(t)
                     : [x] "v"(x), [y] "v"(y));
#else
    s = x;
    r = y;
#endif
    sx = s;
    sy = r;
}
#undef VO_BC_ADD2
#undef VO_BC_ADD6B
#undef VO_BC_ADD6
#undef VO_BC_ADD
#endif

#if defined(VO_HOST_EMUL) || defined(__HIPCC__)
#if defined(VO_HOST_EMUL)
#define VO_WIDE_FN static inline
static inline void wide_sync(bool one_wave)
{
    if (one_wave)
        emu::wave_barrier();
    else
        emu::barrier();
}
static inline bool wave_any(bool v) { return emu_ballot(v) != 0; }
#else
#define VO_WIDE_FN __device__ __forceinline__
// one wavefront: its LDS operations execute in order, the fences keep the compiler from moving memory operations across
// the step boundary; several wavefronts: a workgroup barrier
__device__ __forceinline__ void wide_sync(bool one_wave)
{
    if (one_wave) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}
__device__ __forceinline__ bool wave_any(bool v) { return __ballot(v) != 0ull; }
#endif

// The schedule.  Serial order: sweep after sweep, in a sweep the pairs (0,1), (0,2) ... (N-2,N-1).  Pair (i, j) reads and
// writes rows i and j only, so it has to wait for the LAST earlier pair that touched row i and for the last one that touched
// row j -- nothing else:
//   * inside a sweep those are (i, j - 1) and (i - 1, j): all pairs on an anti-diagonal d = i + j are independent and may
//     run side by side, d = 1 .. D = 2N - 3 one after the other;
//   * across sweeps row r is last touched by (r, N-1) on diagonal r + N - 1 (row N-1 by (N-2, N-1) on D), so pair (i, j) of
//     the NEXT sweep may run P = N diagonals after the same pair of this one: diagonal d of sweep s + 1 together with
//     diagonal d + P of sweep s.  The two never share a row (rows <= d against rows >= d + 1), and every pair sees exactly
//     the rows the serial order would show it.
// A sweep that rotates nothing ends the algorithm; the next sweep has by then started speculatively -- and, reading the very
// rows the idle sweep left untouched, has skipped every pair as well: nothing to undo.
// One time slot per P diagonals-of-the-newest-sweep: up to ROWS pairs, one per DPP row (N = 12: at most 6 -> two
// wavefronts, 12 slots per sweep instead of the serial 66 pairs; N = 6: at most 3 -> one wavefront, 6 slots instead of 15).
template <int N, int ROWS>
struct JacobiPipe {
    static constexpr int P = N, D = 2 * N - 3;
    // slot phi = 1 .. P, DPP row r: i | j << 4 | (1 = pair of the OLDER sweep) << 8 | (1 = a pair to do) << 9.  One 16-bit entry
    // per (slot, row), fetched ONE SLOT AHEAD (round 5): the table lives in global memory, and the three byte loads of the
    // first version sat at the head of every slot with a wait behind each (two round trips to the cache per slot, 77 slots per
    // 12 x 12 matrix, ~41 per 6 x 6 solve).
    uint16_t e[P][ROWS];
};
template <int N, int ROWS>
constexpr JacobiPipe<N, ROWS> jacobi_pipe()
{
    JacobiPipe<N, ROWS> t = {};
    for (int phi = 1; phi <= N; phi++) {
        int n = 0;
        for (int old = 0; old < 2; old++) {
            const int d = phi + old * N;
            if (d > 2 * N - 3)
                continue;
            for (int i = 0; i < N; i++) {
                const int j = d - i;
                if (j <= i || j > N - 1)
                    continue;
                t.e[phi - 1][n] = (uint16_t)(i | j << 4 | old << 8 | 1 << 9); // (n < ROWS: an index past the array stops the constant evaluation)
                n++;
            }
        }
        for (; n < ROWS; n++)
            t.e[phi - 1][n] = 0;
    }
    return t;
}
#if defined(VO_HOST_EMUL)
static const JacobiPipe<12, 8> JACOBI_PIPE_12 = jacobi_pipe<12, 8>();
static const JacobiPipe<6, 4> JACOBI_PIPE_6 = jacobi_pipe<6, 4>();
#else
__device__ const JacobiPipe<12, 8> JACOBI_PIPE_12 = jacobi_pipe<12, 8>();
__device__ const JacobiPipe<6, 4> JACOBI_PIPE_6 = jacobi_pipe<6, 4>();
#endif

// Squared row norms (+ Vt = I) and the Jacobi sweeps of jacobi_svd<N, N, WANT_V>.  Called by all 16 * ROWS threads that work
// on ONE matrix (ROWS = 4: a wavefront, possibly one of several in its workgroup -- no workgroup barrier is used; ROWS = 8:
// a 128-thread workgroup); `tid` = 0 .. 16 ROWS - 1; lanes N .. 15 of a DPP row follow their row and never store.  Every
// branch below is uniform over a DPP row: its conditions are functions of the row's broadcast sums or of the table.
// At, Vt: N x N row-major, W: N, flag: one int the threads share (ROWS = 8 only).
template <int N, bool WANT_V, int ROWS>
VO_WIDE_FN void jacobi_pipe_sweeps(const JacobiPipe<N, ROWS> &tab, double *At, double *W, double *Vt, int *flag, int tid)
{
    constexpr int P = JacobiPipe<N, ROWS>::P, D = JacobiPipe<N, ROWS>::D;
    constexpr bool ONE_WAVE = ROWS <= 4;
    const int row = tid >> 4, l16 = tid & 15, k = l16 < N ? l16 : N - 1;
    const bool owner = l16 < N;
    const double eps = DBL_EPSILON * 10;
    const int max_iter = N > 30 ? N : 30;
    for (int i = row; i < N; i += ROWS) {
        const double t = At[i * N + k];
        const double sd = row_ordered_sum<N>(t * t);
        if (l16 == 0)
            W[i] = sd;
        if (WANT_V && owner)
            Vt[i * N + k] = i == k ? 1.0 : 0.0;
    }
    wide_sync(ONE_WAVE);
    bool chg_old = false, chg_new = false; // this thread's row rotated something in the older / the newest sweep
    uint32_t ent_next = tab.e[0][row];
    for (int sweep = 0;; sweep++) {         // `sweep` = index of the newest sweep in flight
        const bool new_on = sweep < max_iter;
        for (int phi = 1; phi <= P; phi++) {
            const uint32_t ent = ent_next;
            ent_next = tab.e[phi == P ? 0 : phi][row]; // the next slot's entry: in flight while this slot computes
            const int i = ent & 15, j = ent >> 4 & 15;
            const bool old = (ent >> 8 & 1) != 0, todo = (ent >> 9 & 1) != 0;
            if (todo && (old ? sweep >= 1 : new_on)) {
                const double ai = At[i * N + k], aj = At[j * N + k];
                double a = W[i], b = W[j];
                double p = row_ordered_sum<N>(ai * aj);
                // (forming the rotation's hypot before this branch, next to the skip test's sqrt -- two independent chains of
                // ~20 dependent f64 instructions -- was measured and gains nothing: 127.7 vs 127.9 us, gpurun_out/r3_31)
                if (!(fabs(p) <= eps * sqrt(a * b))) {
                    p *= 2;
                    const double beta = a - b, gamma = vo_hypot(p, beta);
                    double c, s;
                    if (beta < 0) {
                        const double delta = (gamma - beta) * 0.5;
                        s = sqrt(delta / gamma);
                        c = p / (gamma * s * 2);
                    } else {
                        c = sqrt((gamma + beta) / (gamma * 2));
                        s = p / (gamma * c * 2);
                    }
                    const double t0 = c * ai + s * aj;
                    const double t1 = -s * ai + c * aj;
                    if (owner) {
                        At[i * N + k] = t0;
                        At[j * N + k] = t1;
                    }
                    row_ordered_sum_x2<N>(t0 * t0, t1 * t1, a, b);
                    if (l16 == 0) {
                        W[i] = a;
                        W[j] = b;
                    }
                    if (WANT_V && owner) {
                        const double vi = Vt[i * N + k], vj = Vt[j * N + k];
                        Vt[i * N + k] = c * vi + s * vj;
                        Vt[j * N + k] = -s * vi + c * vj;
                    }
                    if (old)
                        chg_old = true;
                    else
                        chg_new = true;
                }
            }
            wide_sync(ONE_WAVE);
            if (phi == D - P && sweep >= 1) { // the older sweep has just run its last diagonal: did it rotate anything?
                bool any;
                if (ONE_WAVE) {
                    any = wave_any(chg_old);
                } else {
                    if (tid == 0)
                        *flag = 0;
                    wide_sync(false);
                    if (chg_old)
                        *flag = 1;
                    wide_sync(false);
                    any = *flag != 0;
                    wide_sync(false);
                }
                if (!any || !new_on)
                    return;
            }
        }
        chg_old = chg_new;
        chg_new = false;
    }
}
VO_WIDE_FN void jacobi12_pipe_sweeps(double *At, double *W, int *flag, int tid /* 0 .. 127 */)
{
    jacobi_pipe_sweeps<12, false, 8>(JACOBI_PIPE_12, At, W, nullptr, flag, tid);
}
VO_WIDE_FN void jacobi6v_wave_sweeps(double *At, double *W, double *Vt, int lane /* 0 .. 63 */)
{
    jacobi_pipe_sweeps<6, true, 4>(JACOBI_PIPE_6, At, W, Vt, nullptr, lane);
}
#undef VO_WIDE_FN
#endif

// What jacobi_svd<N, N, WANT_V> does after its sweeps, for ONE lane on matrices in memory: singular values = row norms,
// descending selection sort (rows of At and Vt follow; OpenCV swaps row i with the FIRST index of the running maximum),
// normalised rows of At; an exactly-zero singular value gets the deterministic pseudo-random vector OpenCV fills in
// (cv::RNG(0x12345678)).  W: N doubles, on return the singular values.
template <int N, bool WANT_V>
VO_HD void jacobi_finish(double *At, double *W, double *Vt)
{
    const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
    for (int i = 0; i < N; i++) {
        double sd = 0;
        for (int k = 0; k < N; k++) {
            const double t = At[i * N + k];
            sd += t * t;
        }
        W[i] = sqrt(sd);
    }
    for (int i = 0; i < N - 1; i++) {
        int j = i;
        for (int k = i + 1; k < N; k++)
            if (W[j] < W[k])
                j = k;
        if (i != j) {
            double t = W[i];
            W[i] = W[j];
            W[j] = t;
            for (int k = 0; k < N; k++) {
                t = At[i * N + k];
                At[i * N + k] = At[j * N + k];
                At[j * N + k] = t;
            }
            if (WANT_V)
                for (int k = 0; k < N; k++) {
                    t = Vt[i * N + k];
                    Vt[i * N + k] = Vt[j * N + k];
                    Vt[j * N + k] = t;
                }
        }
    }
    uint64_t rng = 0x12345678;
    for (int i = 0; i < N; i++) {
        double sd = W[i];
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const double val0 = 1. / N;
            for (int k = 0; k < N; k++) {
                rng = (uint64_t)(uint32_t)rng * 4164903690U + (uint32_t)(rng >> 32);
                At[i * N + k] = ((uint32_t)rng & 256) != 0 ? val0 : -val0;
            }
            for (int iter = 0; iter < 2; iter++)
                for (int j = 0; j < i; j++) {
                    sd = 0;
                    for (int k = 0; k < N; k++)
                        sd += At[i * N + k] * At[j * N + k];
                    double asum = 0;
                    for (int k = 0; k < N; k++) {
                        const double t = At[i * N + k] - sd * At[j * N + k];
                        At[i * N + k] = t;
                        asum += fabs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (int k = 0; k < N; k++)
                        At[i * N + k] *= asum;
                }
            sd = 0;
            for (int k = 0; k < N; k++) {
                const double t = At[i * N + k];
                sd += t * t;
            }
            sd = sqrt(sd);
        }
        const double s = sd > minval ? 1 / sd : 0.;
        for (int k = 0; k < N; k++)
            At[i * N + k] *= s;
    }
}
VO_HD void jacobi12_finish(double *At, double *W) { jacobi_finish<12, false>(At, W, nullptr); }

// x = pinv(A) b as solve_svd<N, N> forms it, from the factors jacobi_finish left: At rows = U^T, W, Vt
template <int N>
VO_HD void svd_backsubst(const double *At, const double *W, const double *Vt, const double *b, double *x)
{
    double threshold = 0;
    for (int i = 0; i < N; i++)
        x[i] = 0;
    for (int i = 0; i < N; i++)
        threshold += W[i];
    threshold *= DBL_EPSILON * 2;
    for (int i = 0; i < N; i++) {
        double wi = W[i];
        if (fabs(wi) <= threshold)
            continue;
        wi = 1 / wi;
        double s = 0;
        for (int j = 0; j < N; j++)
            s += At[i * N + j] * b[j];
        s *= wi;
        for (int j = 0; j < N; j++)
            x[j] = x[j] + s * Vt[i * N + j];
    }
}

} // namespace vo
