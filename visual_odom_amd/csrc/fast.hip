// fast.hip -- feature detection + bucketing on the device.
//
// Replaces, for every frame of a batch, what matchingFeatures() does before circularMatching()
// (reference src/visualOdometry.cpp:95-110):
//   appendNewFeatures(imageLeft_t0, features)      feature.cpp:255-262
//     -> featureDetectionFast: cv::FAST(image, keypoints, 20, true)    feature.cpp:39-47
//        (FastFeatureDetector TYPE_9_16: >= 9 contiguous circle pixels all brighter than p + t or all
//         darker than p - t, 3-pixel image margin; score = cornerScore<16>, the largest threshold
//         for which the pixel is still a corner; non-maximum suppression keeps a corner whose score
//         is strictly greater than its 8 neighbours'; keypoints in row-major scan order)
//   bucketingFeatures(image, features, rows/10, features_per_bucket)   feature.cpp:206-253,
//     bucket.cpp:14-51 -- including its quirks: buckets are indexed
//     hidx * (cols/bucket) + widx with widx in [0, cols/bucket] so the last column aliases the next
//     row's first bucket, the read-back visits (rows/bucket + 1) x (cols/bucket + 1) cells and so
//     emits aliased buckets twice, a full bucket always overwrites its slot 0 with the incoming
//     feature, features of age >= 10 are dropped, and ages[i] pairs with points[i] even when the
//     ages array is longer than the points array (new corners then inherit stale ages).
//
// Kernels:
//   fast_tile_kernel     workgroup per 64 x 16 (fast_tile_tall_kernel: 64 x 32) tile of the level-0 image already resident for LK: pixels staged in
//                        LDS; per position 16 circle pixels -> 2 x 16-bit masks -> 9-contiguous test by shift-and,
//                        cornerScore<16> for corners; NMS predicate on the LDS score tile, one 64-bit ballot per
//                        64-pixel row segment, row counts by atomicAdd
//   fast_rowscan_kernel  workgroup per frame: exclusive scan of the row counts
//   fast_nms_write_kernel wavefront per 64 / segs image rows, lane per 64-pixel segment: (x, y) at rows_before + rank from the stored
//                        ballots (row-major)
//   bucket_kernel        workgroup per frame.  The sequential bucket fill is restated as order
//                        statistics: a bucket ends up holding (slot 0) its LAST eligible feature if
//                        more than fpb are eligible, else its first; (slots 1..) its 2nd..fpb-th
//                        eligible feature -- found with LDS atomicMin / atomicMax rounds -- then an
//                        exclusive scan over the visited cells gives the emission offsets.
#include "vo_kernels.h"
#include "vo_lkmath.h"

#include <limits.h>
#include <stdlib.h>

namespace vo {

struct __attribute__((packed, aligned(4))) FastU32x4 {
    uint32_t a, b, c, d;
};

// (acc << 1) | (x < 0): one v_alignbit_b32 shifts a comparison's sign bit into a ring mask (a compare + select + or
// per bit cost 2.5 x as much issue time, profiles/r02_valu_issue_cost.txt)
__device__ __forceinline__ uint32_t shift_in_sign(uint32_t acc, int x)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VO_HOST_EMUL)
    return __builtin_amdgcn_alignbit(acc, (uint32_t)x, 31);
#else
    return (acc << 1) | ((uint32_t)x >> 31);
#endif
}

// the 16 differences centre - circle pixel: Bresenham circle of radius 3, the 16 offsets of FAST_t<16> starting at (0, 3),
// clockwise
__device__ __forceinline__ void fast_ring(const uint8_t *__restrict__ p, int stride, int *__restrict__ d)
{
    const int v = p[0];
    d[0] = v - p[3 * stride];
    d[1] = v - p[3 * stride + 1];
    d[2] = v - p[2 * stride + 2];
    d[3] = v - p[stride + 3];
    d[4] = v - p[3];
    d[5] = v - p[-stride + 3];
    d[6] = v - p[-2 * stride + 2];
    d[7] = v - p[-3 * stride + 1];
    d[8] = v - p[-3 * stride];
    d[9] = v - p[-3 * stride - 1];
    d[10] = v - p[-2 * stride - 2];
    d[11] = v - p[-stride - 3];
    d[12] = v - p[-3];
    d[13] = v - p[stride - 3];
    d[14] = v - p[2 * stride - 2];
    d[15] = v - p[3 * stride - 1];
}

// Necessary condition of the TYPE_9_16 test on the four compass pixels of the circle (ring positions 0, 4, 8, 12): any 9
// contiguous ring positions contain two NEIGHBOURING compass positions, so a corner has two neighbouring compass pixels
// that are both brighter than p + t or both darker than p - t.  5 LDS bytes and ~25 instructions instead of 17 and ~100;
// 8.7 % of the positions of the benchmark's frames pass (3.6 % are corners).
__device__ __forceinline__ bool fast_compass_candidate(const uint8_t *__restrict__ p, int stride, int threshold)
{
    const int v = p[0], hi = v + threshold, lo = v - threshold;
    const int c0 = p[3 * stride], c4 = p[3], c8 = p[-3 * stride], c12 = p[-3];
    uint32_t mb = 0, md = 0; // bit 3 .. 0 = compass position 0, 4, 8, 12
    mb = shift_in_sign(mb, hi - c0);
    md = shift_in_sign(md, c0 - lo);
    mb = shift_in_sign(mb, hi - c4);
    md = shift_in_sign(md, c4 - lo);
    mb = shift_in_sign(mb, hi - c8);
    md = shift_in_sign(md, c8 - lo);
    mb = shift_in_sign(mb, hi - c12);
    md = shift_in_sign(md, c12 - lo);
    // two neighbours on the 4-ring, both polarities at once: bits 0-3 bright, 8-11 dark, each doubled by 4 for the wrap
    uint32_t m = mb | md << 8;
    m |= m << 4;
    return ((m & (m >> 1)) & 0x0f0fu) != 0;
}

// fast_compass_candidate for TWO horizontally adjacent positions, pixels as u16 pairs (v_perm_b32 lifts): a lane of the
// result is nonzero iff that position passes.  bright_k = sat(c_k - (v + t)) != 0 <=> c_k > v + t; dark_k = sat(sat(v - t) -
// c_k) != 0 <=> c_k < v - t; "both" of two neighbours = min != 0.  25 packed instructions per pair against 2 x 22 scalar ones.
__device__ __forceinline__ uint32_t fast_compass_pair(uint32_t v, uint32_t c0, uint32_t c4, uint32_t c8, uint32_t c12, uint32_t t2)
{
    const uint32_t hi = pk_add_u16(v, t2), lo = pk_subsat_u16(v, t2);
    const uint32_t b0 = pk_subsat_u16(c0, hi), b4 = pk_subsat_u16(c4, hi), b8 = pk_subsat_u16(c8, hi), b12 = pk_subsat_u16(c12, hi);
    const uint32_t d0 = pk_subsat_u16(lo, c0), d4 = pk_subsat_u16(lo, c4), d8 = pk_subsat_u16(lo, c8), d12 = pk_subsat_u16(lo, c12);
    return pk_min_u16(b0, b4) | pk_min_u16(b4, b8) | pk_min_u16(b8, b12) | pk_min_u16(b12, b0) | pk_min_u16(d0, d4) | pk_min_u16(d4, d8) |
           pk_min_u16(d8, d12) | pk_min_u16(d12, d0);
}

// FastFeatureDetector TYPE_9_16 corner test: >= 9 contiguous circle pixels all brighter than p + t or all darker than p - t
__device__ __forceinline__ bool fast_is_corner(const uint8_t *__restrict__ p, int stride, int threshold)
{
    int d[16];
    fast_ring(p, stride, d);
    // d > t : circle pixel darker than the centre; -d > t : brighter.  The bits enter from the low end, i.e. the ring is
    // stored mirrored -- "9 contiguous" does not care about orientation.
    uint32_t mb = 0, md = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mb = shift_in_sign(mb, threshold - d[k]);
        md = shift_in_sign(md, threshold + d[k]);
    }
    // >= 9 contiguous set bits on the 16-bit ring
    auto ring9 = [](uint32_t m) {
        m |= m << 16;
        uint32_t x = m & (m >> 1);
        x &= x >> 2;
        x &= x >> 4;
        x &= m >> 8;
        return x != 0;
    };
    return ring9(mb) || ring9(md);
}

// cornerScore<16> (features2d/src/fast_score.cpp) of a position that passed fast_is_corner
__device__ __forceinline__ int fast_corner_score(const uint8_t *__restrict__ p, int stride, int threshold)
{
    int d[25];
    fast_ring(p, stride, d);
#pragma unroll
    for (int k = 16; k < 25; k++)
        d[k] = d[k - 16];
    int a0 = threshold;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        int a = min(d[k + 1], d[k + 2]);
        a = min(a, d[k + 3]);
        if (a <= a0)
            continue;
        a = min(a, min(min(d[k + 4], d[k + 5]), min(d[k + 6], min(d[k + 7], d[k + 8]))));
        a0 = max(a0, min(a, d[k]));
        a0 = max(a0, min(a, d[k + 9]));
    }
    int b0 = -a0;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        int b = max(max(d[k + 1], d[k + 2]), max(d[k + 3], max(d[k + 4], d[k + 5])));
        if (b >= b0)
            continue;
        b = max(b, max(d[k + 6], max(d[k + 7], d[k + 8])));
        b0 = min(b0, max(b, d[k]));
        b0 = min(b0, max(b, d[k + 9]));
    }
    return (-b0 - 1) & 0xff;
}

// Fused score + non-maximum suppression, one 256-thread workgroup per 64 x 16 pixel tile (round 2; the first version
// was a thread-per-pixel score kernel -- 17 global byte loads per pixel -- writing a u16 score map that a second kernel
// read back nine times: 0.77 + 0.49 ms per 256 KITTI frames, the map alone 239 MB):
//   A  the tile's pixels + 4-pixel apron (80 x 24 bytes, origin (x0 - 4, y0 - 4): 4-byte aligned in the bordered
//      level-0 image, always inside its allocation) go to LDS, 16 bytes per thread (one round trip to memory: round 4);
//   B0 compass test (a necessary condition on 4 of the 16 circle pixels) of the 66 x 18 positions of the tile and its
//      1-pixel halo (positions inside FAST's 3-pixel image margin or outside the image are no corners), FOUR positions
//      per lane from aligned LDS dwords, two per packed 16-bit instruction (round 4: fast_compass_pair); the positions
//      that pass are appended to a list in LDS (one LDS atomic per wavefront and round, ballot ranks);
//   B1 the full corner test of the listed candidates (8.7 % of the positions on the benchmark's frames), corners to a
//      second list;
//   B2 cornerScore<16> of the listed positions only, on densely packed lanes.  The score is ~4 x the work of the corner
//      test and only a few per cent of the positions are corners, but in the one-pass form nearly every wavefront held
//      at least one corner and so executed it for all 64 lanes (1.02 ms per 256 KITTI frames, round-2 trace);
//   C  keep predicate (score strictly above its 8 neighbours') of the LISTED corners inside the tile (round 4; over all
//      tile positions before); a kept corner sets its bit in its row segment's 64-bit mask in LDS; the masks are stored
//      exactly where fast_nms_write_kernel expects them, row counts by atomicAdd (the row-scan pass turns them into offsets
//      and zeroes them again).
// Results are identical to the two-kernel form by construction (same predicate, same neighbour scores).
// Round 4 (PMC per dispatch of 256 KITTI frames, gpurun_out/r4_34 ... r4_37): 206 M -> 152 M vector, 121 M -> 52 M scalar,
// 32 M -> 14 M LDS instructions; 398 -> 290 us.
// Tile size: a template parameter, chosen by launch_fast_corners (64 x 16 for fewer than 8 frames, 64 x 32 above; 128 x 32
// was measured and is slower).  The list phases B1 / B2 keep one or two wavefronts of the workgroup busy for ~100 / ~500
// instructions whatever the tile size (a 64 x 16 tile holds ~100 candidates and ~40 corners on the benchmark's frames).
constexpr int BK_PF = 8; // list entries a thread of bucket_kernel holds in registers at a time
[[maybe_unused]] constexpr int FAST_MAX_SEGS = 64; // 64-pixel segments per row: images up to 4096 pixels wide

template <int SEGS, int H>
__device__ __forceinline__ void fast_tile_body(const PyrImage *__restrict__ imgs, const Quad *__restrict__ quads,
                                               const int *__restrict__ detect, int threshold, int nonmax,
                                               unsigned long long *__restrict__ mask, int segs, int *__restrict__ rowcnt)
{
    constexpr int W = 64 * SEGS;               // output tile W x H
    constexpr int PW = W + 16, PH = H + 8;     // pixel tile in LDS (bytes x rows)
    constexpr int SW = W + 2, SH = H + 2;      // score tile incl. the 1-pixel halo
    __shared__ __attribute__((aligned(16))) uint8_t s_px[PH * PW];
    __shared__ __attribute__((aligned(4))) uint16_t s_sc[SH * SW];
    __shared__ uint16_t s_cand[SH * SW]; // positions (index into s_sc) that passed the compass test
    __shared__ uint16_t s_list[SH * SW]; // ... and the corner test
    __shared__ unsigned long long s_rowmask[H * SEGS]; // kept corners of each 64-pixel row segment of the tile
    __shared__ int s_ncand, s_ncorner;
    const int frame = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
    if (detect && !detect[frame])
        return;
    const PyrImage &im = imgs[quads[frame].l0];
    const int w = im.w[0], h = im.h[0], stride = im.stride[0];
    const int x0 = blockIdx.x * W, y0 = blockIdx.y * H;
    const VO_GLOBAL uint8_t *__restrict__ base = (const VO_GLOBAL uint8_t *)im.lvl[0] + (x0 - 4);
    const int last_row = h + VO_BY - 1; // rows past the bordered allocation are never used: read the last one instead
    // (16 bytes per thread and trip: the tile's rows are 80 or 144 bytes; the global address is only 4-byte aligned.  With a dword
    // per thread and trip every thread made 4 dependent round trips to memory -- the compiler waits for a load before the LDS
    // write that follows it.)
    // No load leaves its row of the bordered allocation (vo_dev.h, "reads stay inside their row"): the tile of the last tile
    // column reaches up to W + 11 columns past the image, more than the right border holds, so a chunk that would cross the
    // row end reads the row's last 16 bytes instead.  Those chunks start past column w + VO_BY - 16 >= w + 8, and no position
    // tested below reads a pixel right of column w - 1 (positions stop at w - 4, the circle's radius is 3): their bytes are
    // never used.  (Round 4 read on into the next row -- on the allocation's last row into whatever followed the image.)
    static_assert(PW % 16 == 0, "16-byte tile loads");
    static_assert(VO_BX % 4 == 0 && VO_BY >= 16 + 4, "the clamped chunk stays 4-byte aligned and right of the last pixel read");
    const int last_chunk = stride - VO_BX - 16 - (x0 - 4); // byte offset (from base) of the last 16 bytes of a row, a multiple of 4
    for (int i = tid; i < PH * (PW / 16); i += 256) {
        const int row = i / (PW / 16), c = i - row * (PW / 16);
        const int gy = y0 - 4 + row < last_row ? y0 - 4 + row : last_row;
        const int gc = 16 * c < last_chunk ? 16 * c : last_chunk;
        *reinterpret_cast<FastU32x4 *>(&s_px[row * PW + 16 * c]) =
            *reinterpret_cast<const VO_GLOBAL FastU32x4 *>(base + ((ptrdiff_t)gy * stride + gc));
    }
    if (tid == 0)
        s_ncand = s_ncorner = 0;
    for (int q = tid; q < H * SEGS; q += 256)
        s_rowmask[q] = 0;
    __syncthreads();
    // B0: compass test of every position, FOUR positions per lane: the lane owns one aligned dword of centre pixels of an LDS
    // row (columns 4 m .. 4 m + 3 of the pixel tile = score-tile columns 4 m - 3 .. 4 m), reads the dwords left and right of it
    // and the ones 3 rows above / below, and tests two positions per packed instruction.  The positions that pass are appended to
    // the candidate list lane by lane (the order of the list does not matter: the results reach the
    // output through the score tile and the row masks).  (Round 3: one position per lane and round, 5 LDS byte reads + ~45
    // instructions each, ballot-ranked appends -- half of the kernel's instructions.)
    static_assert((SH * SW) % 2 == 0, "the score tile is cleared by dwords");
    for (int i = tid; i < SH * SW / 2; i += 256)
        reinterpret_cast<uint32_t *>(s_sc)[i] = 0;
    {
        constexpr int GW = (SW + 3 + 3) / 4; // dword groups of a score-tile row: LDS columns 3 .. SW + 2
        const uint32_t t2 = (uint32_t)threshold | (uint32_t)threshold << 16;
        const uint32_t *__restrict__ px4 = reinterpret_cast<const uint32_t *>(s_px);
        for (int q0 = 0; q0 < SH * GW; q0 += 256) { // wave-uniform trip count: the ballots need all lanes
            const int q = q0 + tid;
            const int sy = q / GW, m = q - sy * GW;
            const int gy = y0 - 1 + sy;
            // position j sits at score-tile column sx = 4 m - 3 + j, image column gx = x0 - 1 + sx: inside the score tile and
            // inside FAST's 3-pixel margin for jlo <= j < jhi
            const int sx0 = 4 * m - 3, gx0 = x0 - 1 + sx0;
            const int jlo = max(max(-sx0, 3 - gx0), 0), jhi = min(min(SW - sx0, w - 3 - gx0), 4);
            uint32_t pass = 0;
            if (q < SH * GW && gy >= 3 && gy < h - 3 && jhi > jlo) {
                const uint32_t *__restrict__ row = px4 + (sy + 3) * (PW / 4) + m;
                const uint32_t dl = row[m > 0 ? -1 : 0], dm = row[0], dr = row[1];
                const uint32_t up = row[-3 * (PW / 4)], dn = row[3 * (PW / 4)];
                // bytes b0 .. b11 of {dl, dm, dr}: position j = 0 .. 3 has its centre at b[4 + j], left at b[1 + j], right at b[7 + j]
                const uint32_t ra = fast_compass_pair(perm_b32(0, dm, 0x0c010c00u), perm_b32(0, dn, 0x0c010c00u), perm_b32(dr, dm, 0x0c040c03u),
                                                      perm_b32(0, up, 0x0c010c00u), perm_b32(0, dl, 0x0c020c01u), t2);
                const uint32_t rb = fast_compass_pair(perm_b32(0, dm, 0x0c030c02u), perm_b32(0, dn, 0x0c030c02u), perm_b32(0, dr, 0x0c020c01u),
                                                      perm_b32(0, up, 0x0c030c02u), perm_b32(dm, dl, 0x0c040c03u), t2);
                // lanes nonzero -> 1, then bits 0, 16 (ra) and 2, 18 (rb) -> bits 0 .. 3
                pass = pk_min_u16(ra, 0x00010001u) | pk_min_u16(rb, 0x00010001u) << 2;
                pass = (pass | pass >> 15) & ((1u << jhi) - 1u) & ~((1u << jlo) - 1u);
            }
            // append (lane-major): one LDS atomic per wavefront and round, ranks from four ballots.  (A per-lane atomicAdd on
            // the counter is rewritten by the compiler into a scalar loop over the active lanes -- 9 scalar instructions per
            // lane with a candidate.)
            unsigned long long bm[4];
            uint32_t rank = 0, total = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                bm[j] = VO_BALLOT((pass >> j & 1u) != 0);
                rank = VO_MBCNT(bm[j], rank, lane);
                total += (uint32_t)VO_POPCLL(bm[j]);
            }
            if (total == 0)
                continue;
            int base_k = 0;
            if (lane == 0)
                base_k = atomicAdd(&s_ncand, (int)total);
            int k = uni(base_k) + (int)rank;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (pass >> j & 1)
                    s_cand[k++] = (uint16_t)(sy * SW + sx0 + j);
        }
    }
    __syncthreads();
    // B1: the full corner test on the candidates only, corners to a second list
    const int ncand = s_ncand;
    for (int k0 = 0; k0 < ncand; k0 += 256) {
        const int k = k0 + tid;
        bool corner = false;
        int i = 0;
        if (k < ncand) {
            i = s_cand[k];
            const int sy = i / SW, sx = i - sy * SW;
            corner = fast_is_corner(&s_px[(sy + 3) * PW + sx + 3], PW, threshold);
        }
        const unsigned long long m = VO_BALLOT(corner);
        int base_k = 0;
        if (lane == 0 && m)
            base_k = atomicAdd(&s_ncorner, (int)VO_POPCLL(m));
        base_k = uni(base_k);
        if (corner)
            s_list[base_k + (int)VO_POPCLL(m & ((1ull << lane) - 1ull))] = (uint16_t)i;
    }
    __syncthreads();
    // B2: cornerScore of the corners
    const int ncorner = s_ncorner;
    for (int k = tid; k < ncorner; k += 256) {
        const int i = s_list[k];
        const int sy = i / SW, sx = i - sy * SW;
        s_sc[i] = (uint16_t)(0x100 | fast_corner_score(&s_px[(sy + 3) * PW + sx + 3], PW, threshold));
    }
    __syncthreads();
    // C: non-maximum suppression ON THE CORNER LIST -- a corner inside the tile that is kept sets its bit in the 64-bit mask of
    // its row segment (LDS); then one thread per row segment stores the mask where fast_nms_write_kernel expects it and adds
    // its population to the row count.  (Round 3 evaluated the keep predicate -- 9 LDS reads, 8 compares -- at all W x H tile
    // positions, a wavefront per row segment: a quarter of the kernel's instructions for the ~2 % of positions that are
    // corners.)
    for (int k = tid; k < ncorner; k += 256) {
        const int i = s_list[k];
        const int sy = i / SW, sx = i - sy * SW;
        if (sx < 1 || sx > W || sy < 1 || sy > H) // a halo corner: scored for its neighbours' sake only
            continue;
        const uint16_t *__restrict__ r = &s_sc[i];
        const int sc = r[0] & 0xff;
        const bool keep = !nonmax || (sc > (r[1] & 0xff) && sc > (r[-1] & 0xff) && sc > (r[-SW - 1] & 0xff) &&
                                      sc > (r[-SW] & 0xff) && sc > (r[-SW + 1] & 0xff) && sc > (r[SW - 1] & 0xff) &&
                                      sc > (r[SW] & 0xff) && sc > (r[SW + 1] & 0xff));
        if (keep)
            atomicOr(&s_rowmask[(sy - 1) * SEGS + ((sx - 1) >> 6)], 1ull << ((sx - 1) & 63));
    }
    __syncthreads();
    for (int q = tid; q < H * SEGS; q += 256) {
        const int ly = q / SEGS, sg = q - ly * SEGS;
        const int gy = y0 + ly, seg = (int)blockIdx.x * SEGS + sg;
        if (gy >= h || seg >= segs)
            continue;
        const unsigned long long m = s_rowmask[q];
        mask[((size_t)frame * h + gy) * segs + seg] = m;
        if (m)
            atomicAdd(&rowcnt[(size_t)frame * h + gy], (int)VO_POPCLL(m));
    }
}

__global__ __launch_bounds__(256) void fast_tile_kernel(const PyrImage *__restrict__ imgs,
                                                        const Quad *__restrict__ quads,
                                                        const int *__restrict__ detect /* or null = every frame */,
                                                        int threshold, int nonmax,
                                                        unsigned long long *__restrict__ mask /* [B][h][segs] */,
                                                        int segs, int *__restrict__ rowcnt /* [B][h], zero on entry */)
{
    fast_tile_body<1, 16>(imgs, quads, detect, threshold, nonmax, mask, segs, rowcnt);
}

__global__ __launch_bounds__(256) void fast_tile_tall_kernel(const PyrImage *__restrict__ imgs,
                                                             const Quad *__restrict__ quads,
                                                             const int *__restrict__ detect, int threshold, int nonmax,
                                                             unsigned long long *__restrict__ mask, int segs,
                                                             int *__restrict__ rowcnt)
{
    fast_tile_body<1, 32>(imgs, quads, detect, threshold, nonmax, mask, segs, rowcnt);
}

#if defined(VO_DEV_VARIANTS) || defined(VO_HOST_EMUL) // the 128 x 32 tile: measured slower than both product tiles (launch_fast_corners)
__global__ __launch_bounds__(256) void fast_tile_big_kernel(const PyrImage *__restrict__ imgs,
                                                            const Quad *__restrict__ quads,
                                                            const int *__restrict__ detect, int threshold, int nonmax,
                                                            unsigned long long *__restrict__ mask, int segs,
                                                            int *__restrict__ rowcnt)
{
    fast_tile_body<2, 32>(imgs, quads, detect, threshold, nonmax, mask, segs, rowcnt);
}
#endif

// Corner list from the stored ballots: a wavefront per 64 / segs image rows, a lane per 64-pixel segment.  The
// lanes fetch the row's ballots with one coalesced load, the lane's exclusive prefix of the per-segment counts comes from
// ballots of the counts' bits, then every lane walks the set bits of its own ballot: (x, y) at rows_before + rank -- row-major order =
// cv::FAST's keypoint order.  (The first version spent most of its 0.18 ms per 256 frames in one thread's chain of 20
// dependent global loads per row.)
// rows of an image one wavefront of fast_nms_write_kernel covers: as many as fit its 64 lanes at `segs` lanes per row
inline int fast_nms_rows_per_wave(int segs) { return segs >= 64 ? 1 : 64 / segs; }

__global__ __launch_bounds__(256) void fast_nms_write_kernel(const unsigned long long *__restrict__ mask, int segs,
                                                              int h, const int *__restrict__ detect,
                                                              const int *__restrict__ rowoff /* [B][h] exclusive */,
                                                              const int *__restrict__ n_tracked, int cap,
                                                              float2 *__restrict__ feat /* [B][cap] */)
{
    const int frame = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (detect && !detect[frame])
        return;
    // a wavefront = rpw consecutive image rows, `segs` lanes each (a 1241-pixel row has 20 segments: three rows per wavefront;
    // one row per wavefront left 44 of its 64 lanes idle)
    const int rpw = segs >= 64 ? 1 : 64 / segs;
    const int rw = lane / segs, seg = lane - rw * segs; // (rw >= rpw: a spare lane)
    const int y = ((int)blockIdx.x * 4 + wv) * rpw + rw;
    unsigned long long m = 0;
    if (rw < rpw && y < h)
        m = mask[((size_t)frame * h + y) * segs + seg];
    // corners of the row's segments left of this lane's: the exclusive prefix of the per-lane counts (0 .. 64) WITHIN the lanes
    // of the row, bit slice by bit slice -- a ballot of bit b of every lane's count, masked to the row's lanes, v_mbcnt of it,
    // shifted.  (Round 3 put the counts in LDS and every lane summed the entries before its own in a loop of dependent LDS
    // reads behind a workgroup barrier.)
    const uint32_t c = (uint32_t)VO_POPCLL(m);
    if (VO_BALLOT(c != 0) == 0ull)
        return;
    const unsigned long long row_lanes = (segs >= 64 ? ~0ull : (1ull << segs) - 1ull) << (rw * segs & 63);
    uint32_t before = 0;
#pragma unroll
    for (int bit = 0; bit < 7; bit++)
        before += VO_MBCNT(VO_BALLOT((c >> bit & 1u) != 0) & row_lanes, 0u, lane) << bit;
    if (m == 0ull)
        return;
    int o = (n_tracked ? n_tracked[frame] : 0) + rowoff[(size_t)frame * h + y] + (int)before;
    float2 *__restrict__ out = feat + (size_t)frame * cap;
    do {
        const int b = __builtin_ctzll(m);
        if (o < cap)
            out[o] = make_float2((float)(seg * 64 + b), (float)y);
        o++;
        m &= m - 1ull;
    } while (m);
}

// one 256-thread workgroup per frame: row counts -> exclusive offsets (separate array), n_new = total (0 when not
// detecting); the counts are zeroed again for the next fast_tile_kernel launch
__global__ __launch_bounds__(256) void fast_rowscan_kernel(int *__restrict__ rowcnt, int *__restrict__ rowoff, int h,
                                                           const int *__restrict__ detect,
                                                           int *__restrict__ n_new)
{
    __shared__ int s_part[256];
    const int frame = blockIdx.x, tid = threadIdx.x;
    if (detect && !detect[frame]) {
        if (tid == 0)
            n_new[frame] = 0;
        return;
    }
    int *__restrict__ rc = rowcnt + (size_t)frame * h;
    int *__restrict__ ro = rowoff + (size_t)frame * h;
    const int per = (h + 255) / 256, r0 = tid * per, r1 = min(h, r0 + per);
    int sum = 0;
    for (int r = r0; r < r1; r++)
        sum += rc[r];
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) { // 256 partials: a serial scan is a few hundred cycles
        int acc = 0;
        for (int k = 0; k < 256; k++) {
            const int t = s_part[k];
            s_part[k] = acc;
            acc += t;
        }
        n_new[frame] = acc;
    }
    __syncthreads();
    int acc = s_part[tid];
    for (int r = r0; r < r1; r++) {
        const int t = rc[r];
        ro[r] = acc;
        rc[r] = 0;
        acc += t;
    }
}

constexpr int BK_MAX_CELLS = 1024; // (rows/bucket + 1) * (cols/bucket + 1) of the ordinary grid: any features_per_bucket <= 8
constexpr int BK_MAX_FPB = 8;
constexpr int BK_CELL_CACHE = 8192; // list entries whose cell is kept in LDS between the passes (16 KB)
// Round 6 (the fuzz of vo_detect_bucket walked into the 1 024-bucket limit with the reference's own rule rows / 10 on a wide,
// low image): a second instantiation for FINE grids -- up to 4 096 buckets whose "first q" table is laid out [q][n_buckets]
// in the same 32 KB, so n_buckets x features_per_bucket <= 8 192 (a 4 096 x 376 panorama at rows / 10: 1 221 buckets, up to 6
// per bucket; KITTI at a 20-pixel bucket: 1 197).  The ordinary instantiation is the round-5 kernel, layout and LDS unchanged.
constexpr int BK_FINE_CELLS = 4096;

// one workgroup per frame, of 256 threads or (round 5, launches of a few frames: the kernel is on the critical path of a
// one-sequence step and of vo_detect_bucket, 35 us per KITTI frame at 6 features per bucket) of 1024: the walks over the
// list -- 1 + (features_per_bucket - 1) of them -- go four times as wide; the emission, 4 cells per thread, stays on the
// first 256 threads.  Every phase is order-free (counts, minima, maxima), so the width does not change the result.
template <int CELLS>
__global__ __launch_bounds__(1024) void bucket_kernel(const float2 *__restrict__ feat /* [B][cap] */,
                                                     const int *__restrict__ ages /* [B][cap] */,
                                                     const int *__restrict__ n_tracked,
                                                     const int *__restrict__ n_new, int cap, int rows, int cols,
                                                     int bucket_size, int fpb, float2 *__restrict__ out_pts,
                                                     int *__restrict__ out_ages, int *__restrict__ out_n,
                                                     int out_cap, const int *__restrict__ active /* or null */,
                                                     int *__restrict__ overflow /* or null */,
                                                     const float2 *__restrict__ corners /* or null: [B][cap] */)
{
    __shared__ int s_cnt[CELLS], s_last[CELLS];
    __shared__ int s_first_flat[BK_MAX_FPB * BK_MAX_CELLS]; // [q][BK_MAX_CELLS], or [q][nb] on a fine grid
    __shared__ int s_scan[256];
    // the cell of the first BK_CELL_CACHE list entries, computed once: with features_per_bucket = f the list is walked f times,
    // and every walk re-loaded point + age and re-did the two float divisions -- 70 us for ONE frame at f = 6, between the
    // pyramids and LK on the critical path of a lock-step step (gpurun_out/r4_09 timeline)
    __shared__ int16_t s_cell[BK_CELL_CACHE];
    const int frame = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int bh = rows / bucket_size, bw = cols / bucket_size;
    const int nb = (bh + 1) * (bw + 1); // the reference allocates this many buckets ("<=" loops)
    const int fstride = CELLS == BK_MAX_CELLS ? BK_MAX_CELLS : nb;
#define s_first(q, b) s_first_flat[(q) * fstride + (b)]
    // The list appendNewFeatures leaves (feature.cpp:255-262) is "carried features, then the new corners".  Combined
    // form (corners == null): both live in `feat`.  Split form: the corners were detected ahead of time into their own
    // array (lock-step loop: FAST of a frame's left image runs one step early, off the critical path), element i of the
    // list is feat[i] for i < n_tracked and corners[i - n_tracked] after that.
    const float2 *__restrict__ P = feat + (size_t)frame * cap;
    const float2 *__restrict__ Cn = corners ? corners + (size_t)frame * cap : nullptr;
    const int *__restrict__ A = ages + (size_t)frame * cap;
    const int nt = n_tracked[frame];
    auto point = [&](int i) { return (Cn && i >= nt) ? Cn[i - nt] : P[i]; };
    auto age = [&](int i) { return i < cap ? A[i] : 0; };
    if (active && !active[frame]) { // lock-step sequence loop: this sequence has no frame in this step
        if (tid == 0) {
            out_n[frame] = 0;
            if (overflow)
                overflow[frame] = 0;
        }
        return;
    }
    // Capacity: carried + detected features beyond `cap` were not stored (fast_nms_write_kernel), and bucketing keeps
    // the LAST eligible feature of a cell, so a truncated list changes the result -- reported, never silent
    int n_in;
    bool list_overflow;
    if (Cn) {
        const int nn = n_new[frame];
        list_overflow = nn > cap;
        n_in = nt + (nn < cap ? nn : cap);
    } else {
        n_in = nt + n_new[frame];
        list_overflow = n_in > cap;
        n_in = n_in < cap ? n_in : cap;
    }

    for (int b = tid; b < nb; b += nthr) {
        s_cnt[b] = 0;
        s_last[b] = -1;
        for (int q = 0; q < fpb; q++)
            s_first(q, b) = INT_MAX;
    }
    __syncthreads();
    // bucket of feature i, or -1 when Bucket::add_feature ignores it (age >= 10) / the reference
    // would index outside its bucket vector (undefined behaviour there; never hit by in-image points)
    // The quotients are float divisions truncated to int (feature.cpp:233-234).  A NaN / infinite / huge coordinate converts
    // differently on x86 (INT_MIN) and on gfx950 (0 / saturated), and the reference then indexes its vector out of range either
    // way: such a feature is ignored, decided on the floats so that no conversion of an unrepresentable value takes part.
    auto cell_of = [&](float2 p, int a) {
        const float qy = p.y / (float)bucket_size, qx = p.x / (float)bucket_size;
        if (!(fabsf(qy) < 32768.f && fabsf(qx) < 32768.f))
            return -1;
        const int idx = vo_f2i(qy) * bw + vo_f2i(qx);
        return (idx < 0 || idx >= nb || a >= 10) ? -1 : idx;
    };
    auto cell = [&](int i) { return cell_of(point(i), age(i)); };
    // BK_PF list entries per thread at a time: their points and ages are requested before the first one is used (the walk was a
    // chain of dependent global loads -- one memory round trip per 256 features, ~14 per frame -- in a kernel that has the chip
    // to itself: one workgroup per frame).
    // LDS atomics per RUN, not per entry (round 5): the list is "carried features, then FAST's corners in row-major order", so
    // the 64 consecutive entries a wavefront holds fall into a handful of cells in long runs -- and 64 lanes adding to the same
    // LDS word are executed one after the other (a frame's ~5 000 entries x 3 atomics: most of the kernel's 32 us in the
    // one-sequence timeline).  The first lane of a run of equal cells adds the run's length, offers its own index as the
    // cell's first and the run's last index as its last; counts, first and last are order-free, so the result is the same.
    const int lane = tid & 63;
    auto run_of = [&](int b, int *len) { // is this lane the head of a run of equal cells among the wavefront's lanes?  (all 64 lanes call)
        const int prevb = __shfl_up(b, 1, 64);
        const bool head = lane == 0 || b != prevb;
        const unsigned long long H = VO_BALLOT(head), above = H & ~((2ull << lane) - 1ull);
        *len = (above ? __builtin_ctzll(above) : 64) - lane;
        return head && b >= 0;
    };
    for (int base = 0; base < n_in; base += nthr * BK_PF) { // (wave-uniform trip count: the ballots need all lanes)
        const int i0 = base + tid;
        float2 pt[BK_PF];
        int ag[BK_PF];
#pragma unroll
        for (int k = 0; k < BK_PF; k++) {
            const int i = i0 + nthr * k;
            if (i < n_in) {
                pt[k] = point(i);
                ag[k] = age(i);
            }
        }
#pragma unroll
        for (int k = 0; k < BK_PF; k++) {
            const int i = i0 + nthr * k; // (the wavefront's lanes hold 64 CONSECUTIVE list entries)
            int b = -2;                 // beyond the list: never equal to a cell or to "ignored" (-1)
            if (i < n_in) {
                b = cell_of(pt[k], ag[k]);
                if (i < BK_CELL_CACHE)
                    s_cell[i] = (int16_t)b; // (nb <= 4 096 cells: fits)
            }
            int len;
            if (run_of(b, &len)) {
                atomicAdd(&s_cnt[b], len);
                atomicMax(&s_last[b], i + len - 1);
                atomicMin(&s_first(0, b), i);
            }
        }
    }
    __syncthreads();
    for (int q = 1; q < fpb; q++) { // q-th eligible feature of every bucket, in list order
        for (int base = 0; base < n_in; base += nthr) {
            const int i = base + tid;
            const int b = i >= n_in ? -2 : i < BK_CELL_CACHE ? (int)s_cell[i] : cell(i);
            int len;
            if (run_of(b, &len)) { // the run's first entry behind the cell's (q - 1)-th feature, if it has one
                const int prev = s_first(q - 1, b), cand = i > prev ? i : prev == INT_MAX ? INT_MAX : prev + 1;
                if (cand <= i + len - 1)
                    atomicMin(&s_first(q, b), cand);
            }
        }
        __syncthreads();
    }
    // emission: cells (hh, ww), hh <= bh, ww <= bw, in that order, each emits bucket hh * bw + ww
    // (on the first 256 threads whatever the width of the workgroup)
    const int per = (nb + 255) / 256, v0 = min(nb, tid * per), v1 = tid < 256 ? min(nb, v0 + per) : v0;
    int sum = 0;
    for (int v = v0; v < v1; v++) {
        const int idx = (v / (bw + 1)) * bw + (v % (bw + 1));
        sum += min(s_cnt[idx], fpb);
    }
    if (tid < 256)
        s_scan[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int k = 0; k < 256; k++) {
            const int t = s_scan[k];
            s_scan[k] = acc;
            acc += t;
        }
        out_n[frame] = acc < out_cap ? acc : out_cap;
        if (overflow)
            overflow[frame] = (list_overflow ? 1 : 0) | (acc > out_cap ? 2 : 0);
    }
    __syncthreads();
    int off = tid < 256 ? s_scan[tid] : 0;
    float2 *__restrict__ OP = out_pts + (size_t)frame * out_cap;
    int *__restrict__ OA = out_ages + (size_t)frame * out_cap;
    for (int v = v0; v < v1; v++) {
        const int idx = (v / (bw + 1)) * bw + (v % (bw + 1));
        const int c = s_cnt[idx], m = min(c, fpb);
        for (int q = 0; q < m; q++) {
            const int src = q == 0 ? (c > fpb ? s_last[idx] : s_first(0, idx)) : s_first(q, idx);
            if (off + q < out_cap) {
                OP[off + q] = point(src);
                OA[off + q] = age(src);
            }
        }
        off += m;
    }
#undef s_first
}

// which grids the device bucketing takes (the hosts' argument checks and launch_bucket share it)
bool bucket_grid_ok(int w, int h, int bucket_size, int fpb)
{
    if (bucket_size < 1 || fpb < 1 || fpb > BK_MAX_FPB)
        return false;
    const long long nb = (long long)(h / bucket_size + 1) * (w / bucket_size + 1);
    return nb <= BK_MAX_CELLS || (nb <= BK_FINE_CELLS && nb * fpb <= (long long)BK_MAX_FPB * BK_MAX_CELLS);
}

#ifndef VO_HOST_EMUL
// cv::FAST of every frame's left t0 image: corners in row-major order at out[frame][base ...], count in n_new[frame]
// (base = n_tracked[frame] when given -- the combined list -- else 0)
void launch_fast_corners(const PyrImage *d_imgs, const Quad *d_quads, const int *d_detect, int n_frames, int w, int h,
                         int threshold, int nonmax, unsigned long long *d_nmsmask, int *d_rowcnt, int *d_rowoff,
                         const int *d_ntracked, int *d_nnew, int cap, float2 *d_out, hipStream_t stream)
{
    if (n_frames <= 0)
        return;
    const int segs = (w + 63) / 64;
    // tile form 0: 64 x 16, 1: 64 x 32, 2: 128 x 32 (VO_FAST_TILE forces one).  Measured per 256 KITTI frames in the lock-step
    // loop: detection stage 0.67 / 0.58 / 0.94 ms -- the 64 x 32 tile amortises the two list phases (one or two busy
    // wavefronts per workgroup) over twice the full-width work, the 128 x 32 tile's 32 KB of LDS costs more occupancy than
    // that saves.  Fewer than 8 frames keep the small tile (more workgroups than CUs for one 1241 x 376 image).
    int tile = n_frames >= 8 ? 1 : 0;
#ifdef VO_DEV_VARIANTS
    static const int forced = [] { const char *e = getenv("VO_FAST_TILE"); return e ? atoi(e) : -1; }();
    tile = forced >= 0 ? forced : tile;
    if (tile == 2)
        hipLaunchKernelGGL(fast_tile_big_kernel, dim3((segs + 1) / 2, (h + 31) / 32, n_frames), dim3(256), 0, stream, d_imgs,
                           d_quads, d_detect, threshold, nonmax, d_nmsmask, segs, d_rowcnt);
    else
#endif
    if (tile == 1)
        hipLaunchKernelGGL(fast_tile_tall_kernel, dim3(segs, (h + 31) / 32, n_frames), dim3(256), 0, stream, d_imgs,
                           d_quads, d_detect, threshold, nonmax, d_nmsmask, segs, d_rowcnt);
    else
        hipLaunchKernelGGL(fast_tile_kernel, dim3(segs, (h + 15) / 16, n_frames), dim3(256), 0, stream, d_imgs,
                           d_quads, d_detect, threshold, nonmax, d_nmsmask, segs, d_rowcnt);
    hipLaunchKernelGGL(fast_rowscan_kernel, dim3(n_frames), dim3(256), 0, stream, d_rowcnt, d_rowoff, h, d_detect,
                       d_nnew);
    const int rows_per_wg = 4 * fast_nms_rows_per_wave(segs);
    hipLaunchKernelGGL(fast_nms_write_kernel, dim3((h + rows_per_wg - 1) / rows_per_wg, n_frames), dim3(256), 0, stream, d_nmsmask, segs, h, d_detect,
                       d_rowoff, d_ntracked, cap, d_out);
}

void launch_bucket(const float2 *d_feat, const float2 *d_corners, const int *d_ages, const int *d_ntracked,
                   const int *d_nnew, int cap, int w, int h, int bucket_size, int fpb, float2 *d_out_pts, int *d_out_ages,
                   int *d_out_n, int out_cap, const int *d_active, int *d_overflow, int n_frames, hipStream_t stream)
{
    if (n_frames <= 0)
        return;
    // 16 wavefronts per frame where the kernel is the latency of a step (a handful of frames on an otherwise idle GPU), 4 in
    // big launches that run next to other work (the rule of launch_compact, post.hip)
    const int threads = n_frames <= 4 ? 1024 : 256;
    if ((long long)(h / bucket_size + 1) * (w / bucket_size + 1) <= BK_MAX_CELLS)
        hipLaunchKernelGGL(bucket_kernel<BK_MAX_CELLS>, dim3(n_frames), dim3(threads), 0, stream, d_feat, d_ages, d_ntracked, d_nnew, cap, h, w,
                           bucket_size, fpb, d_out_pts, d_out_ages, d_out_n, out_cap, d_active, d_overflow, d_corners);
    else // (bucket_grid_ok was the callers' check)
        hipLaunchKernelGGL(bucket_kernel<BK_FINE_CELLS>, dim3(n_frames), dim3(threads), 0, stream, d_feat, d_ages, d_ntracked, d_nnew, cap, h, w,
                           bucket_size, fpb, d_out_pts, d_out_ages, d_out_n, out_cap, d_active, d_overflow, d_corners);
}

void launch_detect_bucket(const PyrImage *d_imgs, const Quad *d_quads, const int *d_detect, int n_frames, int w,
                          int h, int threshold, int nonmax, unsigned long long *d_nmsmask,
                          int *d_rowcnt, int *d_rowoff,
                          const int *d_ntracked, int *d_nnew, int cap, float2 *d_feat, const int *d_ages,
                          int bucket_size, int fpb, float2 *d_out_pts, int *d_out_ages, int *d_out_n, int out_cap,
                          const int *d_active, int *d_overflow, hipStream_t stream)
{
    launch_fast_corners(d_imgs, d_quads, d_detect, n_frames, w, h, threshold, nonmax, d_nmsmask, d_rowcnt, d_rowoff,
                        d_ntracked, d_nnew, cap, d_feat, stream);
    if (bucket_size > 0)
        launch_bucket(d_feat, nullptr, d_ages, d_ntracked, d_nnew, cap, w, h, bucket_size, fpb, d_out_pts, d_out_ages,
                      d_out_n, out_cap, d_active, d_overflow, n_frames, stream);
}
#endif

} // namespace vo
