// vo_svd_wide.h -- the Jacobi sweeps of EPnP's 12 x 12 SVD run by a whole WAVEFRONT per matrix (round 3).
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
//     exactly the rows they would see in the serial order.  The four DPP rows of the wavefront take up to four of them
//     at a time: 26 steps per sweep instead of 66 (the table below).
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
// (s_nop 1: a DPP instruction must not read a VGPR in the two slots after the VALU write that produced it -- the compiler
// keeps that distance for its own instructions but does not look inside an asm block)
template <int N>
__device__ __forceinline__ double row_ordered_sum(double x)
{
    static_assert(N == 6 || N == 12, "chain lengths built below");
    double s = 0.0;
#if defined(__HIP_DEVICE_COMPILE__) // (the host pass of hipcc only needs the declaration)
    double t;
    if (N == 6)
        asm volatile("s_nop 1\n\t" VO_BC_ADD6("s", "x") : [s] "+v"(s), [t] "=&v"(t) : [x] "v"(x));
    else
        asm volatile("s_nop 1\n\t" VO_BC_ADD6("s", "x") VO_BC_ADD6B("s", "x") : [s] "+v"(s), [t] "=&v"(t) : [x] "v"(x));
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
        asm volatile("s_nop 1\n\t" VO_BC_ADD2(0) VO_BC_ADD2(1) VO_BC_ADD2(2) VO_BC_ADD2(3) VO_BC_ADD2(4) VO_BC_ADD2(5)
                     : [s] "+v"(s), [r] "+v"(r), [t] "=&v"(t)
                     : [x] "v"(x), [y] "v"(y));
    else
        asm volatile("s_nop 1\n\t" VO_BC_ADD2(0) VO_BC_ADD2(1) VO_BC_ADD2(2) VO_BC_ADD2(3) VO_BC_ADD2(4) VO_BC_ADD2(5) VO_BC_ADD2(6)
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
#define VO_WAVE_SYNC() emu::barrier()
static inline bool wave_any(bool v) { return emu_ballot(v) != 0; }
#else
#define VO_WIDE_FN __device__ __forceinline__
#define VO_WAVE_SYNC() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"), __builtin_amdgcn_wave_barrier(), __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront")
__device__ __forceinline__ bool wave_any(bool v) { return __ballot(v) != 0ull; }
#endif

// Pairs (i, j), i < j < N, by anti-diagonal i + j, four per step: {i, j} of DPP row 0 .. 3, 255 = nothing to do in this
// step.  N = 12: 26 steps for 66 pairs, N = 6: 9 steps for 15.
template <int N>
struct JacobiSteps {
    uint8_t ij[N * (N - 1) / 2][4][2];
    int n;
};
template <int N>
constexpr JacobiSteps<N> jacobi_steps()
{
    JacobiSteps<N> t = {};
    int step = 0;
    for (int d = 1; d <= 2 * N - 3; d++) {
        int n = 0;
        for (int i = 0; i < N; i++) {
            const int j = d - i;
            if (j <= i || j > N - 1)
                continue;
            if (n == 4) {
                step++;
                n = 0;
            }
            t.ij[step][n][0] = (uint8_t)i;
            t.ij[step][n][1] = (uint8_t)j;
            n++;
        }
        for (; n < 4; n++)
            t.ij[step][n][0] = t.ij[step][n][1] = 255;
        step++;
    }
    t.n = step;
    return t;
}
#if defined(VO_HOST_EMUL)
static const JacobiSteps<12> JACOBI_STEPS_12 = jacobi_steps<12>();
static const JacobiSteps<6> JACOBI_STEPS_6 = jacobi_steps<6>();
#else
__device__ const JacobiSteps<12> JACOBI_STEPS_12 = jacobi_steps<12>();
__device__ const JacobiSteps<6> JACOBI_STEPS_6 = jacobi_steps<6>();
#endif
// Squared row norms (+ Vt = I) and the Jacobi sweeps of jacobi_svd<N, N, WANT_V>.  Called by all 64 lanes of ONE wavefront per
// matrix; lanes N .. 15 of a DPP row shadow lane N - 1.  Every branch below is uniform over a DPP row: its conditions are
// functions of the row's broadcast sums (or of the step table) only.  At, Vt: N x N row-major, W: N.
template <int N, bool WANT_V>
VO_WIDE_FN void jacobi_wave_sweeps(const JacobiSteps<N> &tab, double *At, double *W, double *Vt, int lane)
{
    const int row = (lane >> 4) & 3, l16 = lane & 15, k = l16 < N ? l16 : N - 1;
    const bool owner = l16 < N; // lanes N .. 15 follow the row's control flow and feed nothing: they never store
    const double eps = DBL_EPSILON * 10;
    for (int i = row; i < N; i += 4) {
        const double t = At[i * N + k];
        const double sd = row_ordered_sum<N>(t * t);
        if (l16 == 0)
            W[i] = sd;
        if (WANT_V && owner)
            Vt[i * N + k] = i == k ? 1.0 : 0.0;
    }
    VO_WAVE_SYNC();
    const int max_iter = N > 30 ? N : 30;
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (int step = 0; step < tab.n; step++) {
            const int i = tab.ij[step][row][0], j = tab.ij[step][row][1];
            if (i != 255) {
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
                    changed = true;
                }
            }
            VO_WAVE_SYNC();
        }
        if (!wave_any(changed))
            break;
    }
}
VO_WIDE_FN void jacobi12_wave_sweeps(double *At, double *W, int lane)
{
    jacobi_wave_sweeps<12, false>(JACOBI_STEPS_12, At, W, nullptr, lane);
}
VO_WIDE_FN void jacobi6v_wave_sweeps(double *At, double *W, double *Vt, int lane)
{
    jacobi_wave_sweeps<6, true>(JACOBI_STEPS_6, At, W, Vt, lane);
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
