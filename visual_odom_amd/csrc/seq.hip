// seq.hip -- device-side state hand-off of the reference's frame loop, for S sequences in lock step.
//
// The reference processes ONE sequence, one frame after another (main.cpp:123-224): frame k + 1 starts from the
// feature set frame k left behind (currentVOFeatures.points = pointsLeft_t1, visualOdometry.cpp:127; ages compacted by
// deleteUnmatchFeaturesCircle only, feature.cpp:83-86,111 -- quirk B3 of SURVEY.md), from the images of frame k
// (main.cpp:157-158) and from frame_pose (utils.cpp:84).  Within a sequence that chain is serial, so the exact way
// to fill the chip is S independent sequences x 1 frame per step (vo_seq_* in include/vo_hip.h).  These kernels
// keep the chain on the device; per step the host only hands over the new stereo pairs:
//   seq_prepare_kernel    `if (currentVOFeatures.size() < 2000) appendNewFeatures(...)`  visualOdometry.cpp:95-101
//   seq_carry_kernel      currentVOFeatures after the frame: points = pointsLeft_t1 (the K consistency-filter
//                         survivors), ages = (bucketed ages + 1) compacted with the M >= K circular-matching survivors
//                         -- the ages array keeps the longer length, so the next frame's first M - K new corners
//                         inherit stale ages exactly like the reference
//   (the pose tail -- rotationMatrixToEulerAngles + gates + integrateOdometryStereo, main.cpp:196-208, utils.cpp:57-131,
//    one trajectory row per processed frame -- runs inside select_refine_kernel: vo_seqtail.h)
#include "vo_kernels.h"
#include <stdlib.h>

namespace vo {

// n_corners (optional): the FAST corners of every sequence's t0 image were detected one step ahead (vo_seq_step);
// a sequence that does not re-detect in this frame simply does not append them (n_new = 0)
__global__ void seq_prepare_kernel(const int *__restrict__ active, const int *__restrict__ n_tracked,
                                   int redetect_below, int *__restrict__ detect,
                                   const int *__restrict__ n_corners, int *__restrict__ n_new, int n_seq)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n_seq) {
        const int d = (active[f] && n_tracked[f] < redetect_below) ? 1 : 0;
        detect[f] = d;
        if (n_corners)
            n_new[f] = d ? n_corners[f] : 0;
    }
}

// one 256-thread workgroup per sequence
__global__ __launch_bounds__(256) void seq_carry_kernel(const int *__restrict__ active,
                                                        const float2 *__restrict__ outB /* [S][4][cap] */,
                                                        const int *__restrict__ nB, const int *__restrict__ idxA,
                                                        const int *__restrict__ nA,
                                                        const int *__restrict__ ages /* [S][cap] bucketed */,
                                                        const int *__restrict__ n_bucketed, int cap, int fcap,
                                                        float2 *__restrict__ feat /* [S][fcap] */,
                                                        int *__restrict__ fages /* [S][fcap] */,
                                                        int *__restrict__ n_tracked,
                                                        const int *__restrict__ overflow,
                                                        int *__restrict__ n_rows_carry,
                                                        int *__restrict__ n_ages /* [S] length of `ages` */,
                                                        SeqFrameInfo *__restrict__ info /* [S][max_steps] */,
                                                        int max_steps)
{
    const int f = blockIdx.x, tid = threadIdx.x;
    if (!active[f])
        return;
    const int nb = nB[f], na = nA[f];
    const float2 *__restrict__ l1 = outB + ((size_t)f * 4 + 2) * cap; // stage-B row 2 = pointsLeft_t1
    float2 *__restrict__ F = feat + (size_t)f * fcap;
    int *__restrict__ A = fages + (size_t)f * fcap;
    const int *__restrict__ ia = idxA + (size_t)f * cap;
    const int *__restrict__ ag = ages + (size_t)f * cap;
    for (int i = tid; i < nb; i += 256)
        F[i] = l1[i];
    // ages[i] += 1 for all, then erase-compacted with the circular-matching survivors (feature.cpp:83-86,111);
    // entries beyond it read as 0 = the age appendNewFeatures gives a fresh corner (feature.cpp:260)
    for (int i = tid; i < fcap; i += 256)
        A[i] = i < na ? ag[ia[i]] + 1 : 0;
    if (tid == 0) {
        n_tracked[f] = nb;
        n_ages[f] = na;
        // rows are counted twice -- here and in seq_integrate_kernel, which runs on the pose stream and may lag a
        // step behind: both count the frames this sequence has processed
        const int row = n_rows_carry[f];
        if (row < max_steps) {
            SeqFrameInfo &o = info[(size_t)f * max_steps + row];
            o.n_bucketed = n_bucketed[f];
            o.n_circ = na;
            o.n_tracked = nb;
            o.overflow = overflow[f];
        }
        n_rows_carry[f] = row + 1;
    }
}

// One launch moves every new stereo pair of a step into its ring slot.  Sources are row-major 8-bit images with a byte
// stride, in device memory or in page-locked host memory the GPU reads over PCIe directly (one kernel instead of 2 S
// pitched copies: at S = 256 the 512 hipMemcpy2DAsync calls alone cost the host 8 ms per step -- more than the step's
// kernels -- and the copy engine moves a 1241-byte-wide pitched image at 0.1 GB/s).  8 bytes per lane: unaligned source
// loads (rows of a 1241-pixel image start anywhere), the row's last 8 bytes by an overlapping access instead of a byte loop.
//
// Round 6 -- a PERSISTENT grid sized for the link, not for the GPU.  The transfer of step k + 1 runs under step k's kernels.
// Rounds 2-5 launched one workgroup per four rows (48 k workgroups per step at 256 sequences): the copy kept the link's rate,
// but everything that ran BESIDE it crawled -- with page-locked host pairs 256 sequences at the 2 000-point load took 13.9 ms
// per step against 10.0 with resident ones (pyramids 0.5 -> 2.0 ms, detection 0.5 -> 1.9, LK 8.8 -> 9.5), at any stream
// priority (gpurun_out/r6_ingab).  tools/ubench/ingest_under_load.hip isolates it (profiles/r06_ingest_under_load.txt): what
// hurts the neighbour is the number of PCIe reads in flight.  Every wave parks 512 bytes of requests in the L2's queues for
// the microseconds a host read takes; thousands of waves (the flood: as many as find a slot) hold them for ~100 us each and a
// kernel whose waves wait for their own L2 fills -- LK -- slows from 9.8 to 13.8 ms; so do 1 024 persistent waves (13.6 ms)
// and 4 096 (15.0), on the same CUs or on others (CU masks: 13.6).  256 waves = 128 KB in flight are what 50 GB/s x 2.5 us
// need: the link still runs at 49.8 GB/s and the neighbour at 10.0 ms, untouched.  (The copy ENGINE would be better still --
// 57 GB/s, no shader at all -- but only for ONE contiguous copy: 512 linear copies reach 26 GB/s, pitched ones 0.1.)
// In the loop itself LK is touchier than the probe's stand-in (gpurun_out/r6_ingw, 256 sequences, page-locked pairs, 2 000 / 340
// points per frame, k frames/s): G = 64 17.8 / 22.5, 128 23.2 / 38.3, 192 23.1 / 40.7, 256 21.7 / 37.7, 384 20.1 / 34.0,
// 512 18.8 / 33.2 -- against 18.3 / 37.8 for the flood.
// So: G single-wave workgroups walk over the rows, row r = blockIdx.x, + G, ...; G = 192 when any pair of the step comes
// over PCIe, 8192 for a step of device-resident pairs (an HBM-to-HBM copy wants more loads in flight, they are short, and it
// is over in a fraction of a millisecond).
struct __attribute__((packed, aligned(1))) IngU2 {
    uint32_t lo, hi;
};

__global__ __launch_bounds__(64) void seq_ingest_kernel(const SeqIngest *__restrict__ tab, int n_rows /* 2 * pairs * h */,
                                                        int n_waves /* = the grid */, int w, int h, int pitch,
                                                        uint8_t *__restrict__ pix0 /* pixel (0,0) of image 0 */, size_t img_bytes)
{
    const int last = w - 8; // (w >= 32: vo_batch_configure) the lane that would cross the row end re-reads the row's last 8 bytes
    for (int r = blockIdx.x; r < n_rows; r += n_waves) { // (wave-uniform: image, side, row and both row addresses are scalars)
        const int img = r / h, row = r - img * h, side = img & 1;
        const SeqIngest e = tab[img >> 1];
        const VO_GLOBAL uint8_t *__restrict__ s = (const VO_GLOBAL uint8_t *)(side ? e.right : e.left) + (size_t)row * e.stride;
        VO_GLOBAL uint8_t *__restrict__ d = (VO_GLOBAL uint8_t *)pix0 + (size_t)(e.image0 + side) * img_bytes + (size_t)row * pitch;
        for (int x0 = 0; x0 < w; x0 += 512) {
            int x = x0 + (int)threadIdx.x * 8;
            if (x < w) {
                x = x < last ? x : last;
                const IngU2 v = *reinterpret_cast<const VO_GLOBAL IngU2 *>(s + (uint32_t)x);
                *reinterpret_cast<VO_GLOBAL IngU2 *>(d + (uint32_t)x) = v;
            }
        }
    }
}

#ifndef VO_HOST_EMUL // (the CPU emulator of tests/host_check launches the kernels above itself)
void launch_seq_ingest(const SeqIngest *tab, int n_pairs, int w, int h, int pitch, uint8_t *pix0, size_t img_bytes,
                       bool over_pcie, hipStream_t stream)
{
    if (n_pairs <= 0)
        return;
    int want = over_pcie ? 192 : 8192;
#ifdef VO_DEV_VARIANTS
    if (const char *e = getenv(over_pcie ? "VO_INGEST_WAVES" : "VO_INGEST_WAVES_DEV")) // developer build: A/B of the grid size
        want = atoi(e) > 0 ? atoi(e) : want;
#endif
    const int n_rows = 2 * n_pairs * h;
    const int n_waves = n_rows < want ? n_rows : want;
    hipLaunchKernelGGL(seq_ingest_kernel, dim3(n_waves), dim3(64), 0, stream, tab, n_rows, n_waves, w, h, pitch, pix0, img_bytes);
}

void launch_seq_prepare(const int *active, const int *n_tracked, int redetect_below, int *detect,
                        const int *n_corners, int *n_new, int n_seq, hipStream_t stream)
{
    hipLaunchKernelGGL(seq_prepare_kernel, dim3((n_seq + 63) / 64), dim3(64), 0, stream, active, n_tracked,
                       redetect_below, detect, n_corners, n_new, n_seq);
}

void launch_seq_carry(const int *active, const float2 *outB, const int *nB, const int *idxA, const int *nA,
                      const int *ages, const int *n_bucketed, int cap, int fcap, float2 *feat, int *fages,
                      int *n_tracked, const int *overflow, int *n_rows_carry, int *n_ages, SeqFrameInfo *info,
                      int max_steps, int n_seq, hipStream_t stream)
{
    hipLaunchKernelGGL(seq_carry_kernel, dim3(n_seq), dim3(256), 0, stream, active, outB, nB, idxA, nA, ages,
                       n_bucketed, cap, fcap, feat, fages, n_tracked, overflow, n_rows_carry, n_ages, info, max_steps);
}

#endif // VO_HOST_EMUL

} // namespace vo
