// post.hip -- epilogue of circularMatching(): status / sign filter, circular-consistency check,
// order-preserving compaction, and stereo triangulation.
//
// Replaces (reference file:line)
//   deleteUnmatchFeaturesCircle   src/feature.cpp:76-116   (stage A: 5 arrays + ages by 4 statuses,
//                                                           also drops x<0 || y<0 of pt0..pt3)
//   checkValidMatch + removeInvalidPoints x4   src/visualOdometry.cpp:44-77,119-125 (stage B)
//   cv::triangulatePoints + convertPointsFromHomogeneous   src/main.cpp:169-171
// vector::erase semantics = stable compaction, done with a wave-ballot prefix sum per frame.
#include "vo_kernels.h"
#include "vo_tri.h"

namespace vo {

// one workgroup per frame, of any whole number of wavefronts up to 16 (round 5: small launches use 1024 threads -- a frame's
// ~2000 features in two rounds of loads instead of eight, 15 -> 8 us in the synchronous call's timeline; the rounds are bound
// by the latency of their loads, not by their arithmetic; launch_compact)
__global__ __launch_bounds__(1024) void compact_kernel(const float2 *__restrict__ pts_in,   // [B][cap]
                                                      const float2 *__restrict__ trk,      // [B][4][cap]
                                                      const uint8_t *__restrict__ status,  // [B][4][cap]
                                                      const int *__restrict__ n_pts, int cap, int threshold,
                                                      float2 *__restrict__ outA,  // [B][5][cap] l0,r0,r1,l1,l0ret
                                                      int *__restrict__ idxA,     // [B][cap]
                                                      int *__restrict__ nA,       // [B]
                                                      float2 *__restrict__ outB,  // [B][4][cap] l0,r0,l1,r1
                                                      int *__restrict__ idxB,     // [B][cap]
                                                      int *__restrict__ nB)       // [B]
{
    __shared__ int s_wave[2][16];
    __shared__ int s_base[2];
    const int frame = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nthr = blockDim.x, nwv = nthr >> 6;
    const int n = n_pts[frame];
    const float2 *p0 = pts_in + (size_t)frame * cap;
    const float2 *t0 = trk + (size_t)frame * 4 * cap;
    const uint8_t *s0 = status + (size_t)frame * 4 * cap;
    if (tid < 2)
        s_base[tid] = 0;
    __syncthreads();

    for (int start = 0; start < n; start += nthr) {
        const int i = start + tid;
        bool ka = false, kb = false;
        float2 l0 = {0, 0}, r0 = {0, 0}, r1 = {0, 0}, l1 = {0, 0}, lr = {0, 0};
        if (i < n) {
            l0 = p0[i];
            r0 = t0[i];
            r1 = t0[cap + i];
            l1 = t0[2 * cap + i];
            lr = t0[3 * cap + i];
            bool bad = s0[3 * cap + i] == 0 || l1.x < 0 || l1.y < 0 || s0[2 * cap + i] == 0 || r1.x < 0 ||
                       r1.y < 0 || s0[cap + i] == 0 || r0.x < 0 || r0.y < 0 || s0[i] == 0 || l0.x < 0 ||
                       l0.y < 0;
            ka = !bad;
            // int offset = max(|dx|, |dy|) truncated; keep iff offset <= threshold
            float ax = fabsf(l0.x - lr.x), ay = fabsf(l0.y - lr.y);
            int offset = vo_f2i(ax < ay ? ay : ax); // (survivors of stage A: finite and within a window of the image)
            kb = ka && !(offset > threshold);
        }
        const unsigned long long ma = __ballot(ka), mb = __ballot(kb);
        const unsigned long long below = (1ull << lane) - 1ull;
        const int ra = __popcll(ma & below), rb = __popcll(mb & below);
        if (lane == 0) {
            s_wave[0][wv] = __popcll(ma);
            s_wave[1][wv] = __popcll(mb);
        }
        __syncthreads();
        int offa = s_base[0], offb = s_base[1];
        for (int w = 0; w < wv; w++) {
            offa += s_wave[0][w];
            offb += s_wave[1][w];
        }
        if (ka) {
            const int o = offa + ra;
            float2 *oa = outA + (size_t)frame * 5 * cap;
            oa[o] = l0;
            oa[cap + o] = r0;
            oa[2 * cap + o] = r1;
            oa[3 * cap + o] = l1;
            oa[4 * cap + o] = lr;
            idxA[(size_t)frame * cap + o] = i;
        }
        if (kb) {
            const int o = offb + rb;
            float2 *ob = outB + (size_t)frame * 4 * cap;
            ob[o] = l0;
            ob[cap + o] = r0;
            ob[2 * cap + o] = l1;
            ob[3 * cap + o] = r1;
            idxB[(size_t)frame * cap + o] = i;
        }
        __syncthreads();
        if (tid < 2) {
            int sum = s_base[tid];
            for (int w = 0; w < nwv; w++)
                sum += s_wave[tid][w];
            s_base[tid] = sum;
        }
        __syncthreads();
    }
    if (tid == 0) {
        nA[frame] = s_base[0];
        nB[frame] = s_base[1];
    }
}

// thread per point; pl/pr are rows 0 and 1 of the stage-B arrays of each frame
__global__ __launch_bounds__(256) void triangulate_kernel(const float *__restrict__ Pl,
                                                          const float *__restrict__ Pr,
                                                          const float2 *__restrict__ pl,
                                                          const float2 *__restrict__ pr,
                                                          size_t frame_stride /* float2 units */,
                                                          const int *__restrict__ n_pts, int cap,
                                                          float *__restrict__ xyz /* [B][cap][3] */)
{
    const int frame = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pts[frame])
        return;
    float P0[12], P1[12];
#pragma unroll
    for (int k = 0; k < 12; k++) {
        P0[k] = Pl[k];
        P1[k] = Pr[k];
    }
    const float2 a = pl[frame * frame_stride + i], b = pr[frame * frame_stride + i];
    float out[3];
    triangulate_one(P0, P1, a.x, a.y, b.x, b.y, out);
    float *o = xyz + ((size_t)frame * cap + i) * 3;
    o[0] = out[0];
    o[1] = out[1];
    o[2] = out[2];
}

#ifndef VO_HOST_EMUL // (the CPU emulator of tests/host_check runs the two kernels above)
// see FrameGather (vo_kernels.h); `out` is page-locked host memory mapped into the device's address space
__global__ __launch_bounds__(256) void frame_gather_kernel(FrameGather g, uint8_t *__restrict__ out)
{
    const int K = g.pose_only ? 0 : g.nB[0], M = g.pose_only ? 0 : g.nA[0], cap = g.cap; // (pose_only: vo_pnp_ransac)
    const PnpResult r = g.result[0];
    const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    if (tid == 0) {
        int *h = reinterpret_cast<int *>(out);
        h[0] = M;
        h[1] = K;
        h[2] = g.em ? 1 : 0;
        *reinterpret_cast<PnpResult *>(out + 16) = r;
        if (g.em)
            *reinterpret_cast<EmResult *>(out + 256) = g.em[0];
    }
    float2 *o2 = reinterpret_cast<float2 *>(out + VO_GATHER_HEADER);
    float *ox = reinterpret_cast<float *>(o2 + 4 * (size_t)cap);
    int32_t *ok = reinterpret_cast<int32_t *>(ox + 3 * (size_t)cap), *oc = ok + cap, *oi = oc + cap;
    for (int i = tid; i < K; i += nth) {
#pragma unroll
        for (int row = 0; row < 4; row++)
            o2[(size_t)row * cap + i] = g.outB[(size_t)row * cap + i];
        ox[3 * i] = g.xyz[3 * i];
        ox[3 * i + 1] = g.xyz[3 * i + 1];
        ox[3 * i + 2] = g.xyz[3 * i + 2];
        ok[i] = g.idxB[i];
    }
    for (int i = tid; i < M; i += nth)
        oc[i] = g.idxA[i];
    const int ninl = r.n_inliers < cap ? r.n_inliers : cap;
    for (int i = tid; i < ninl; i += nth)
        oi[i] = g.inliers[i];
}

// vo_detect_bucket's hand-over in both directions without a copy call (round 5; capi_dropin.hip).  IN: the carried feature
// set of frame 0 read out of page-locked host memory (pts [n_pts] float2, then ages [n_ages] int32 at `ages_off` bytes) into the
// DETECT stage's lists -- ages beyond n_ages read 0, the age of a freshly appended corner (feature.cpp:260) -- with the
// frame's tracked count and its "detect again" flag.  OUT: the bucketed set and the overflow flags into page-locked host
// memory: k, overflow at bytes 0 / 4, pts at byte 16, ages behind cap points.
__global__ __launch_bounds__(256) void features_in_kernel(const uint8_t *__restrict__ src, size_t ages_off, int n_pts, int n_ages,
                                                          int detect, float2 *__restrict__ feat, int *__restrict__ fages,
                                                          int fcap, int *__restrict__ n_tracked, int *__restrict__ detect_flag)
{
    const float2 *sp = reinterpret_cast<const float2 *>(src);
    const int *sa = reinterpret_cast<const int *>(src + ages_off);
    const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    for (int i = tid; i < fcap; i += nth) {
        if (i < n_pts)
            feat[i] = sp[i];
        fages[i] = i < n_ages ? sa[i] : 0;
    }
    if (tid == 0) {
        n_tracked[0] = n_pts;
        detect_flag[0] = detect;
    }
}

__global__ __launch_bounds__(256) void features_out_kernel(const float2 *__restrict__ pts, const int *__restrict__ ages,
                                                           const int *__restrict__ n, const int *__restrict__ overflow, int cap,
                                                           uint8_t *__restrict__ out)
{
    const int k = n[0] < cap ? n[0] : cap;
    const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    if (tid == 0) {
        reinterpret_cast<int *>(out)[0] = n[0];
        reinterpret_cast<int *>(out)[1] = overflow[0];
    }
    float2 *op = reinterpret_cast<float2 *>(out + 16);
    int *oa = reinterpret_cast<int *>(op + cap);
    for (int i = tid; i < k; i += nth) {
        op[i] = pts[i];
        oa[i] = ages[i];
    }
}

// the same for plain arrays (vo_triangulate, vo_pnp_ransac): n0 then n1 32-bit words out of page-locked host memory into two
// device arrays, with a count; n words of a device array into page-locked host memory
__global__ __launch_bounds__(256) void words_in_kernel(const uint32_t *__restrict__ src, int n0, uint32_t *__restrict__ dst0, int n1,
                                                       uint32_t *__restrict__ dst1, int *__restrict__ count_dst, int count)
{
    const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    for (int i = tid; i < n0; i += nth)
        dst0[i] = src[i];
    for (int i = tid; i < n1; i += nth)
        dst1[i] = src[n0 + i];
    if (tid == 0)
        count_dst[0] = count;
}

__global__ __launch_bounds__(256) void words_out_kernel(const uint32_t *__restrict__ src, int n, uint32_t *__restrict__ dst)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        dst[i] = src[i];
}

// vo_circular_match's results (see CircGather, vo_kernels.h)
__global__ __launch_bounds__(256) void circ_gather_kernel(CircGather g, uint8_t *__restrict__ out)
{
    const int cap = g.cap, count = g.consistency ? g.nB[0] : g.nA[0];
    const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    if (tid == 0)
        reinterpret_cast<int *>(out)[0] = count;
    float2 *o = reinterpret_cast<float2 *>(out + 16);
    int32_t *oi = reinterpret_cast<int32_t *>(o + 5 * (size_t)cap);
    uint8_t *os = reinterpret_cast<uint8_t *>(oi + cap);
    for (int i = tid; i < count; i += nth) {
        if (!g.consistency) { // stage A rows: l0, r0, r1, l1, l0_ret
#pragma unroll
            for (int row = 0; row < 5; row++)
                o[(size_t)row * cap + i] = g.outA[(size_t)row * cap + i];
            oi[i] = g.idxA[i];
        } else { // stage B rows are l0, r0, l1, r1; it drops l0_ret (removeInvalidPoints is not applied to it): by index
            const int k = g.idxB[i];
            o[i] = g.outB[i];
            o[(size_t)cap + i] = g.outB[(size_t)cap + i];
            o[2 * (size_t)cap + i] = g.outB[3 * (size_t)cap + i];
            o[3 * (size_t)cap + i] = g.outB[2 * (size_t)cap + i];
            o[4 * (size_t)cap + i] = g.trk[3 * (size_t)cap + k];
            oi[i] = k;
        }
    }
    for (int i = tid; i < g.n; i += nth)
#pragma unroll
        for (int hop = 0; hop < 4; hop++)
            os[(size_t)hop * cap + i] = g.status[(size_t)hop * cap + i];
}

void launch_words_in(const void *src, int n0, void *dst0, int n1, void *dst1, int *count_dst, int count, hipStream_t stream)
{
    hipLaunchKernelGGL(words_in_kernel, dim3(8), dim3(256), 0, stream, static_cast<const uint32_t *>(src), n0,
                       static_cast<uint32_t *>(dst0), n1, static_cast<uint32_t *>(dst1), count_dst, count);
}

void launch_words_out(const void *src, int n, void *dst, hipStream_t stream)
{
    hipLaunchKernelGGL(words_out_kernel, dim3(8), dim3(256), 0, stream, static_cast<const uint32_t *>(src), n,
                       static_cast<uint32_t *>(dst));
}

void launch_circ_gather(const CircGather &g, uint8_t *out, hipStream_t stream)
{
    hipLaunchKernelGGL(circ_gather_kernel, dim3(8), dim3(256), 0, stream, g, out);
}

void launch_features_in(const uint8_t *src, size_t ages_off, int n_pts, int n_ages, int detect, float2 *feat, int *fages, int fcap,
                        int *n_tracked, int *detect_flag, hipStream_t stream)
{
    hipLaunchKernelGGL(features_in_kernel, dim3(16), dim3(256), 0, stream, src, ages_off, n_pts, n_ages, detect, feat, fages, fcap,
                       n_tracked, detect_flag);
}

void launch_features_out(const float2 *pts, const int *ages, const int *n, const int *overflow, int cap, uint8_t *out,
                         hipStream_t stream)
{
    hipLaunchKernelGGL(features_out_kernel, dim3(8), dim3(256), 0, stream, pts, ages, n, overflow, cap, out);
}

void launch_frame_gather(const FrameGather &g, uint8_t *out, hipStream_t stream)
{
    hipLaunchKernelGGL(frame_gather_kernel, dim3(8), dim3(256), 0, stream, g, out);
}

void launch_compact(const float2 *pts_in, const float2 *trk, const uint8_t *status, const int *n_pts, int cap,
                    int threshold, float2 *outA, int *idxA, int *nA, float2 *outB, int *idxB, int *nB,
                    int n_frames, hipStream_t stream)
{
    if (n_frames <= 0)
        return;
    // 16 wavefronts per frame where the launch is the latency of a call (a handful of frames on an idle GPU); 4 in big launches,
    // which run next to the following step's LK: a 1024-thread workgroup needs four free wave slots on every SIMD of a CU at
    // once and waited 4 ms on average for them there (profiles/r05_kernel_stats_batch.csv of gpurun_out/r5_final2)
    const int threads = n_frames <= 4 ? 1024 : 256;
    hipLaunchKernelGGL(compact_kernel, dim3(n_frames), dim3(threads), 0, stream, pts_in, trk, status, n_pts, cap,
                       threshold, outA, idxA, nA, outB, idxB, nB);
}

void launch_triangulate(const float *Pl, const float *Pr, const float2 *pl, const float2 *pr,
                        size_t frame_stride, const int *n_pts, int cap, int max_pts, int n_frames, float *xyz,
                        hipStream_t stream)
{
    if (n_frames <= 0 || max_pts <= 0)
        return;
    hipLaunchKernelGGL(triangulate_kernel, dim3((max_pts + 255) / 256, n_frames), dim3(256), 0, stream, Pl, Pr,
                       pl, pr, frame_stride, n_pts, cap, xyz);
}
#endif // VO_HOST_EMUL

} // namespace vo
