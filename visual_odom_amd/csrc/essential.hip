// essential.hip -- essential matrix by five-point RANSAC + pose recovery on the device.
//
// Replaces the `mono_rotation` branch of trackingFrame2Frame() (reference src/visualOdometry.cpp:146-157):
//     E = cv::findEssentialMat(pointsLeft_t0, pointsLeft_t1, focal, pp, cv::RANSAC, 0.999, 1.0, mask);
//     cv::recoverPose(E, pointsLeft_t0, pointsLeft_t1, rotation, translation_mono, focal, pp, mask);
//
// Same parallelisation as the PnP solve (pnp.hip): OpenCV's RANSAC loop is sequential with an adaptive
// iteration bound, but its random 5-subsets do not depend on earlier hypotheses, so hypotheses are solved
// and scored in parallel, RANSAC_CHUNK at a time, and the sequential control flow ("keep the first model
// with strictly more inliers, then shrink the bound") is replayed on the counts afterwards.  One sample
// yields up to 10 essential matrices; they are replayed in OpenCV's order (sample, then model index).
//   em_normalise_kernel   pixel -> normalised f64 coordinates, as the MatExpr (p - c) / f evaluates
//   ransac_subsets_kernel (pnp.hip) the shared cv::RNG(-1) subset stream, maxIters = 1000
//   em_solve_kernel       one thread per sample: vo_fivept.h five_point_solve -> models, model count
//   em_vote_kernel        one wavefront per (sample, model): Sampson error of all points, inlier count
//   em_replay_kernel      one thread per frame: RANSACPointSetRegistrator::run on the counts; keeps E
//   em_finish_kernel      one workgroup per frame: inlier mask of the winner, decomposeEssentialMat,
//                         the four cheirality counts (DLT triangulation of every point, f64), selection
#include "vo_kernels.h"
#include "vo_fivept.h"

#include <float.h>

namespace vo {

constexpr int EM_CHUNK = 128; // == RANSAC_CHUNK of pnp.hip (the subset kernel draws chunk by chunk)
constexpr int EM_MAX_MODELS = 10;

__global__ void em_normalise_kernel(const float2 *__restrict__ p0, const float2 *__restrict__ p1, size_t stride,
                                     const int *__restrict__ n_pts, int cap, EmParams prm,
                                     double2 *__restrict__ q0, double2 *__restrict__ q1)
{
    const int frame = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pts[frame])
        return;
    // points.col(0) = (points.col(0) - cx) / fx  ==  x * (1 / fx) + (-cx * (1 / fx))
    const double ax = 1. / prm.focal, bx = -prm.ppx * ax, by = -prm.ppy * ax;
    const float2 a = p0[frame * stride + i], b = p1[frame * stride + i];
    q0[(size_t)frame * cap + i] = make_double2((double)a.x * ax + bx, (double)a.y * ax + by);
    q1[(size_t)frame * cap + i] = make_double2((double)b.x * ax + bx, (double)b.y * ax + by);
}

// WAVES: minimum waves per SIMD = register budget, as for the PnP kernels (pnp.hip): 1 = all 512 registers for
// the stand-alone calls, 4 = 128 registers so that the wave fits next to the LK waves of a crowded batch
template <int WAVES>
__global__ __launch_bounds__(64, WAVES) void em_solve_kernel(const double2 *__restrict__ q0, const double2 *__restrict__ q1,
                                                      const int *__restrict__ n_pts, int cap, int iters, int chunk,
                                                      const int32_t *__restrict__ subsets,
                                                      const RansacState *__restrict__ rstate,
                                                      double *__restrict__ models /* [B][EM_CHUNK][10][9] */,
                                                      int *__restrict__ nmodels /* [B][EM_CHUNK] */)
{
    const int frame = blockIdx.y, s = blockIdx.x * blockDim.x + threadIdx.x, h = chunk * EM_CHUNK + s;
    const int count = n_pts[frame];
    if (s >= EM_CHUNK || count < 5)
        return;
    const bool all_points = count == 5; // count == modelPoints: one runKernel on the points in order
    if (all_points ? h != 0 : (h >= iters || h >= rstate[frame].niters))
        return;
    const int32_t *idx = subsets + ((size_t)frame * iters + h) * 5;
    double s0[10], s1[10], Es[EM_MAX_MODELS * 9];
    for (int i = 0; i < 5; i++) {
        const int k = all_points ? i : idx[i];
        const double2 a = q0[(size_t)frame * cap + k], b = q1[(size_t)frame * cap + k];
        s0[2 * i] = a.x;
        s0[2 * i + 1] = a.y;
        s1[2 * i] = b.x;
        s1[2 * i + 1] = b.y;
    }
    const int nm = five_point_solve(s0, s1, Es);
    double *out = models + ((size_t)frame * EM_CHUNK + s) * (EM_MAX_MODELS * 9);
    for (int i = 0; i < nm * 9; i++)
        out[i] = Es[i];
    nmodels[frame * EM_CHUNK + s] = nm;
}

__global__ __launch_bounds__(64) void em_vote_kernel(const double2 *__restrict__ q0, const double2 *__restrict__ q1,
                                                     const int *__restrict__ n_pts, int cap, int iters, int chunk,
                                                     float thr2, const RansacState *__restrict__ rstate,
                                                     const double *__restrict__ models,
                                                     const int *__restrict__ nmodels,
                                                     int *__restrict__ counts /* [B][EM_CHUNK][10] */)
{
    const int frame = blockIdx.y, s = blockIdx.x / EM_MAX_MODELS, m = blockIdx.x - s * EM_MAX_MODELS;
    const int h = chunk * EM_CHUNK + s, lane = threadIdx.x;
    const int count = n_pts[frame];
    if (count <= 5 || h >= iters || h >= rstate[frame].niters || m >= nmodels[frame * EM_CHUNK + s])
        return;
    double E[9];
    const double *src = models + (((size_t)frame * EM_CHUNK + s) * EM_MAX_MODELS + m) * 9;
    for (int k = 0; k < 9; k++)
        E[k] = src[k];
    int good = 0;
    for (int i = lane; i < count; i += 64) {
        const double2 a = q0[(size_t)frame * cap + i], b = q1[(size_t)frame * cap + i];
        good += em_sampson_error(E, a.x, a.y, b.x, b.y) <= thr2;
    }
#pragma unroll
    for (int mm = 32; mm >= 1; mm >>= 1)
        good += __shfl_xor(good, mm, 64);
    if (lane == 0)
        counts[((size_t)frame * EM_CHUNK + s) * EM_MAX_MODELS + m] = good;
}

// calib3d/ptsetreg.cpp RANSACUpdateNumIters
__device__ static int em_update_num_iters(double p, double ep, int modelPoints, int maxIters)
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

__global__ void em_replay_kernel(const int *__restrict__ n_pts, int n_frames, int iters, double prob, int chunk,
                                  const double *__restrict__ models, const int *__restrict__ nmodels,
                                  const int *__restrict__ counts, RansacState *__restrict__ rstate,
                                  double *__restrict__ bestE /* [B][9] */)
{
    const int frame = blockIdx.x * blockDim.x + threadIdx.x;
    if (frame >= n_frames)
        return;
    const int count = n_pts[frame];
    if (count < 5)
        return;
    RansacState st = rstate[frame];
    if (count == 5) { // bestModel = the models of the single runKernel (first one is what a 3 x 3 E holds)
        if (chunk == 0) {
            const int nm = nmodels[frame * EM_CHUNK];
            st.it = 1;
            st.max_good = nm > 0 ? 5 : 0;
            st.best = nm > 0 ? 0 : -1;
            if (nm > 0)
                for (int k = 0; k < 9; k++)
                    bestE[frame * 9 + k] = models[(size_t)frame * EM_CHUNK * EM_MAX_MODELS * 9 + k];
            rstate[frame] = st;
        }
        return;
    }
    const int end = min((chunk + 1) * EM_CHUNK, iters);
    int it = st.it;
    for (; it < st.niters && it < end; it++) {
        const int s = it - chunk * EM_CHUNK;
        const int nm = nmodels[frame * EM_CHUNK + s];
        for (int m = 0; m < nm; m++) {
            const int good = counts[((size_t)frame * EM_CHUNK + s) * EM_MAX_MODELS + m];
            if (good > (st.max_good > 4 ? st.max_good : 4)) {
                st.max_good = good;
                st.best = it * EM_MAX_MODELS + m;
                const double *src = models + (((size_t)frame * EM_CHUNK + s) * EM_MAX_MODELS + m) * 9;
                for (int k = 0; k < 9; k++)
                    bestE[frame * 9 + k] = src[k];
                st.niters = em_update_num_iters(prob, (double)(count - good) / count, 5, st.niters);
            }
        }
    }
    st.it = it;
    rstate[frame] = st;
}

__global__ __launch_bounds__(256) void em_finish_kernel(const double2 *__restrict__ q0, const double2 *__restrict__ q1,
                                                        const int *__restrict__ n_pts, int cap, float thr2,
                                                        const RansacState *__restrict__ rstate,
                                                        const double *__restrict__ bestE,
                                                        uint8_t *__restrict__ mask /* [B][cap] */,
                                                        EmResult *__restrict__ results)
{
    __shared__ double s_P[4][12];
    __shared__ int s_good[4];
    const int frame = blockIdx.x, tid = threadIdx.x;
    const int count = n_pts[frame];
    EmResult &res = results[frame];
    const RansacState st = rstate[frame];
    if (count < 5 || st.max_good <= 0) {
        if (tid == 0) {
            res.status = count < 5 ? -1 : 0;
            res.n_inliers = 0;
            res.n_good = 0;
            res.niters = count < 5 ? 0 : st.it;
            res.best = -1;
        }
        return;
    }
    double E[9];
    for (int k = 0; k < 9; k++)
        E[k] = bestE[frame * 9 + k];
    if (tid == 0) {
        double R1[9], R2[9], t[3];
        em_decompose(E, R1, R2, t);
        for (int c = 0; c < 4; c++) {
            const double *Rc = (c & 1) ? R2 : R1;
            const double sg = c >= 2 ? -1.0 : 1.0;
            for (int r = 0; r < 3; r++) {
                for (int k = 0; k < 3; k++)
                    s_P[c][4 * r + k] = Rc[3 * r + k] * 1.0;
                s_P[c][4 * r + 3] = sg * t[r] * 1.0;
            }
        }
    }
    if (tid < 4)
        s_good[tid] = 0;
    __syncthreads();
    // RANSAC mask of the winner (all ones when count == modelPoints), then the four cheirality masks ANDed
    // with it; the per-point flags are packed 4 bits per point into the mask buffer for the selection pass
    int g[4] = {0, 0, 0, 0}, n_inl = 0;
    uint8_t *mk = mask + (size_t)frame * cap;
    for (int i = tid; i < count; i += blockDim.x) {
        const double2 a = q0[(size_t)frame * cap + i], b = q1[(size_t)frame * cap + i];
        const bool inl = count == 5 ? true : em_sampson_error(E, a.x, a.y, b.x, b.y) <= thr2;
        n_inl += inl;
        unsigned bits = 0;
        for (int c = 0; c < 4; c++) {
            const bool ok = em_cheirality(s_P[c], a.x, a.y, b.x, b.y, 50.0) && inl;
            bits |= (unsigned)ok << c;
            g[c] += ok;
        }
        mk[i] = (uint8_t)(bits | (inl ? 0x10u : 0u));
    }
    for (int c = 0; c < 4; c++)
        atomicAdd(&s_good[c], g[c]);
    __shared__ int s_ninl;
    if (tid == 0)
        s_ninl = 0;
    __syncthreads();
    atomicAdd(&s_ninl, n_inl);
    __syncthreads();
    const int g0 = s_good[0], g1 = s_good[1], g2 = s_good[2], g3 = s_good[3];
    int sel;
    if (g0 >= g1 && g0 >= g2 && g0 >= g3)
        sel = 0;
    else if (g1 >= g0 && g1 >= g2 && g1 >= g3)
        sel = 1;
    else if (g2 >= g0 && g2 >= g1 && g2 >= g3)
        sel = 2;
    else
        sel = 3;
    for (int i = tid; i < count; i += blockDim.x)
        mk[i] = (mk[i] >> sel) & 1; // bitwise_and(0/1 RANSAC mask, 0/255 compare result)
    if (tid == 0) {
        for (int k = 0; k < 9; k++)
            res.E[k] = E[k];
        for (int r = 0; r < 3; r++) {
            for (int k = 0; k < 3; k++)
                res.R[3 * r + k] = s_P[sel][4 * r + k];
            res.t[r] = s_P[sel][4 * r + 3];
        }
        res.status = 1;
        res.n_inliers = s_ninl;
        res.n_good = sel == 0 ? g0 : sel == 1 ? g1 : sel == 2 ? g2 : g3;
        res.niters = st.it;
        res.best = st.best;
    }
}

#ifndef VO_HOST_EMUL // (the CPU emulator of tests/host_check launches the kernels above itself)
void launch_essential(const float2 *p0, const float2 *p1, size_t stride, const int *n_pts, int cap, int n_frames,
                      const EmParams &prm, const EmBufs &eb, EmResult *results, bool crowded, hipStream_t stream)
{
    if (n_frames <= 0)
        return;
    const int iters = prm.max_iters;
    const double thr = prm.threshold / ((prm.focal + prm.focal) / 2);
    const float thr2 = (float)(thr * thr);
    hipLaunchKernelGGL(em_normalise_kernel, dim3((cap + 255) / 256, n_frames), dim3(256), 0, stream, p0, p1, stride,
                       n_pts, cap, prm, eb.q0, eb.q1);
    const int n_chunks = (iters + EM_CHUNK - 1) / EM_CHUNK;
    for (int chunk = 0; chunk < n_chunks; chunk++) {
        launch_ransac_subsets(n_pts, n_frames, iters, chunk * EM_CHUNK, EM_CHUNK, eb.subsets, eb.rstate, stream);
        if (crowded)
            hipLaunchKernelGGL(em_solve_kernel<4>, dim3(EM_CHUNK / 64, n_frames), dim3(64), 0, stream, eb.q0, eb.q1,
                               n_pts, cap, iters, chunk, eb.subsets, eb.rstate, eb.models, eb.nmodels);
        else
            hipLaunchKernelGGL(em_solve_kernel<1>, dim3(EM_CHUNK / 64, n_frames), dim3(64), 0, stream, eb.q0, eb.q1,
                               n_pts, cap, iters, chunk, eb.subsets, eb.rstate, eb.models, eb.nmodels);
        hipLaunchKernelGGL(em_vote_kernel, dim3(EM_CHUNK * EM_MAX_MODELS, n_frames), dim3(64), 0, stream, eb.q0, eb.q1,
                           n_pts, cap, iters, chunk, thr2, eb.rstate, eb.models, eb.nmodels, eb.counts);
        hipLaunchKernelGGL(em_replay_kernel, dim3((n_frames + 63) / 64), dim3(64), 0, stream, n_pts, n_frames, iters,
                           prm.prob, chunk, eb.models, eb.nmodels, eb.counts, eb.rstate, eb.bestE);
    }
    hipLaunchKernelGGL(em_finish_kernel, dim3(n_frames), dim3(256), 0, stream, eb.q0, eb.q1, n_pts, cap, thr2,
                       eb.rstate, eb.bestE, eb.mask, results);
}
#endif // VO_HOST_EMUL

} // namespace vo
