// lk.hip -- fused 4-hop pyramidal Lucas-Kanade "circular matching" kernel.
//
// Replaces the four chained cv::calcOpticalFlowPyrLK calls of circularMatching()
// (reference src/feature.cpp:136-139: l0->r0, r0->r1, r1->l1, l1->l0; winSize 21x21, maxLevel 3,
// TermCriteria(COUNT+EPS, 30, 0.01), flags 0, minEigThreshold 1e-3, err vector requested).
// Semantics follow OpenCV's CPU LKTrackerInvoker: 14-bit fixed-point bilinear weights, int16
// template (I*32, Scharr Ix/Iy), 2x2 structure tensor, min-eigenvalue test, <=30 Gauss-Newton
// iterations with the eps^2 and oscillation stops, +-winSize admissibility window, final in-bounds
// status check.  The five sums A11/A12/A22/b1/b2 are accumulated exactly in integers (so the result
// does not depend on reduction order) and rounded to f32 once; the 2x2 solve is plain f32 with
// contraction off.
//
// Mapping (CDNA4): ONE WAVEFRONT PER FEATURE, one single-wave workgroup per feature.  A feature is
// a serial chain (4 hops x 4 levels x <=30 iterations) but independent of every other feature, so
// the wave keeps all per-feature state in registers and never synchronises with another wave.
//   lane l -> window row r = l/3, column segment s = l%3 (7 px): 63 lanes cover the 21x21 window.
//   Template: each lane loads its 2 x 8 pixels and 2 x 8 Scharr samples straight from the bordered
//   pyramid (no border path, no staging) and keeps the 7 samples of I, Ix, Iy packed two per VGPR
//   (12 VGPRs) for the whole level.
//   Search image J: a 40 x 48 byte tile in LDS (1920 B per wave), filled with 16-byte loads and
//   re-fetched only when the window leaves it; one iteration reads two unaligned 8-byte rows.
//   Pixel arithmetic: v_perm_b32 / v_dot2_u32_u16 / v_dot2_i32_i16 / v_pk_sub_i16 (vo_lkmath.h).
//   Reductions: v_permlane32/16_swap + v_add_u32_dpp butterflies, two sums per tree (vo_dev.h
//   wave_sum2_exact_f32), no LDS round trips.
// Grid: blocks are numbered so that (dispatcher: block b -> XCD b % 8) all features of one frame
// run on ONE XCD, i.e. the frame's four pyramids are pulled into one L2 only, eight frames at a time.
// Batches of fewer than 8 frames split each frame's feature list into contiguous parts over 8/fpg XCDs.
// lk_hops_kernel (round 6): the same body for hops [begin, end) of the chain -- the synchronous drop-in calls on the kept pair
// launch hop 0 while the new pair is still crossing PCIe (capi_run.hip); lk_circular_kernel is what every other launch runs.
#include "vo_kernels.h"
#include "vo_lkmath.h"

#include <float.h>

namespace vo {

constexpr int LK_WIN = 21;
constexpr int LK_JT_W = 48, LK_JT_H = 40; // search tile (bytes x rows), LDS row stride = LK_JT_W
constexpr int LK_W_BITS = 14;

struct __attribute__((packed, aligned(1))) LkU2 {
    uint32_t lo, hi;
};
struct __attribute__((packed, aligned(4))) LkU4 {
    uint32_t a, b, c, d;
};

// 14-bit fixed-point bilinear weights of OpenCV's LKTrackerInvoker from the fractional parts of the
// window corner: iw00 = cvRound((1-a)*(1-b)*2^14), iw01 = cvRound(a*(1-b)*2^14), iw10 = cvRound((1-a)*b*2^14),
// iw11 = 2^14 - iw00 - iw01 - iw10, returned as the packed int16 pairs wt = (iw00, iw01), wb = (iw10, iw11).
//   * the scale is folded into the first factor ((1-a)*2^14 is exact, so the rounded product is identical);
//   * cvRound (round half to even) = adding 1.5 * 2^23: the f32 sum has ulp 1, so the add rounds the
//     product to the nearest-even integer and leaves it in the low mantissa bits.  The raw bit patterns
//     are packed / summed directly (0x4B400000 has no low 16 bits), no v_rndne / v_cvt per weight.
__device__ __forceinline__ void lk_weights(float a, float b, uint32_t &wt, uint32_t &wb)
{
    const float s = (float)(1 << LK_W_BITS), magic = 12582912.f; // 1.5 * 2^23 = 0x4B400000
    const float a1 = (1.f - a) * s, a0 = a * s, b1 = 1.f - b;
    const uint32_t r00 = (uint32_t)__float_as_int(a1 * b1 + magic);
    const uint32_t r01 = (uint32_t)__float_as_int(a0 * b1 + magic);
    const uint32_t r10 = (uint32_t)__float_as_int(a1 * b + magic);
    // iw11 = 2^14 - (r00 + r01 + r10 - 3 * 0x4B400000)   (mod 2^32)
    const uint32_t iw11 = ((1u << LK_W_BITS) + 3u * 0x4B400000u) - (r00 + r01 + r10);
    wt = perm_b32(r01, r00, VO_SEL_LO16);
    wb = perm_b32(iw11, r10, VO_SEL_LO16); // signed lanes: iw11 may be -1
}

// 7 waves per SIMD = at most 72 VGPRs.  Round 1 had the bound at 6 waves and the allocator happened to land on 70
// registers (so 7 waves were resident anyway); with the round-2 loop it took 78 under that bound -- one wave fewer per
// SIMD, LK 11.9 -> 12.2 ms (gpurun_out/r2_04) -- so the bound now says what is meant: 71 registers, no spills.
// (8 waves = 64 VGPRs spills inside the iteration loop and is slower, gpurun_out/lksweep1.)
#ifndef VO_LK_ATTRS
#define VO_LK_ATTRS __launch_bounds__(64, 7)
#endif
// The body of the kernel.  SPLIT = false: the whole chain (lk_circular_kernel, every batch / lock-step launch).  SPLIT = true
// (lk_hops_kernel, round 6, the synchronous drop-in calls only): hops hop_begin .. hop_end - 1 of the chain -- a launch that does
// not start at hop 0 continues from what the launch before it wrote for hop_begin - 1 (position, status), with the same "this
// feature is going to be dropped anyway" rule, so two launches [0, 1) + [1, 4) write bit for bit what one launch writes.
template <bool SPLIT>
__device__ __forceinline__ void lk_circular_body(const PyrImage *__restrict__ imgs, const Quad *__restrict__ quads,
                                                 const float2 *__restrict__ pts_in, const int *__restrict__ n_pts, int cap,
                                                 int n_frames, int fpg /* 1, 2, 4 or 8 */, int ppp /* features per part */,
                                                 float2 *__restrict__ trk,     // [B][4][cap]
                                                 uint8_t *__restrict__ status, // [B][4][cap]
                                                 const LkParams prm, int hop_begin, int hop_end)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_jt[LK_JT_H * LK_JT_W];

    // XCD-aware block numbering: block id b runs on XCD b % 8 (observed dispatcher behaviour, used
    // for L2 affinity only): XCD x works on frames x, x + 8, x + 16, ... one frame after another
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int grp = slot / ppp, fi = slot - grp * ppp;
    const int frame = grp * fpg + (xcd & (fpg - 1));
    const int f = (xcd / fpg) * ppp + fi;
    if (frame >= n_frames || f >= n_pts[frame])
        return;
    const int lane = threadIdx.x;
    // lane -> (row, 7-pixel segment); lane 63 duplicates lane 62's addresses, contributions zeroed
    const bool live = lane < 63;
    const int lr = live ? lane : 62;
    const int r = lr / 3, c0 = 7 * (lr - 3 * r);
    const int lane_off = r * LK_JT_W + c0; // this lane's row segment inside the search tile

    const Quad q = quads[frame];
    const float halfWin = (LK_WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float eps_lo = (float)(prm.epsilon * 0.999999), eps_hi = (float)(prm.epsilon * 1.000001);
    const float min_eig_n = prm.min_eig * (1.001f * (float)(2 * LK_WIN * LK_WIN)); // see the min-eigenvalue test

    const float2 p = pts_in[(size_t)frame * cap + f];
    float prevPtX = unif(p.x), prevPtY = unif(p.y);
    if (SPLIT && hop_begin > 0) {
        // the chain so far: hop_begin - 1 as the previous launch left it.  A feature that launch saw dead (see the end of the
        // hop loop) has its remaining hops reported already.
        const size_t o = ((size_t)frame * 4 + (hop_begin - 1)) * cap + f;
        const float2 q0 = trk[o];
        const bool dead0 = status[o] == 0 || q0.x < 0.f || q0.y < 0.f || (hop_begin == 1 && (p.x < 0.f || p.y < 0.f));
        if (dead0 && !prm.full_chain)
            return;
        prevPtX = unif(q0.x);
        prevPtY = unif(q0.y);
    }

    for (int hop = SPLIT ? hop_begin : 0; hop < (SPLIT ? hop_end : 4); hop++) {
        // hop chain: l0 -> r0 -> r1 -> l1 -> l0
        const int pi = hop == 0 ? q.l0 : hop == 1 ? q.r0 : hop == 2 ? q.r1 : q.l1;
        const int ni = hop == 0 ? q.r0 : hop == 1 ? q.r1 : hop == 2 ? q.l1 : q.l0;
        const PyrImage &I = imgs[pi];
        const PyrImage &J = imgs[ni];
        float outX = 0.f, outY = 0.f;
        int st = 1;

        for (int level = prm.max_level; level >= 0; level--) {
            const float scale = __int_as_float((127 - level) << 23); // 2^-level, exact (no divide)
            float prevX = prevPtX * scale, prevY = prevPtY * scale;
            float nextX, nextY;
            if (level == prm.max_level) {
                nextX = prevX;
                nextY = prevY;
            } else {
                nextX = outX * 2.f;
                nextY = outY * 2.f;
            }
            outX = nextX;
            outY = nextY;

            const int iw = I.w[level], ih = I.h[level], istride = I.stride[level];
            const int jw = J.w[level], jh = J.h[level], jstride = J.stride[level];
            const VO_GLOBAL uint8_t *__restrict__ Iimg = (const VO_GLOBAL uint8_t *)I.lvl[level];
            const VO_GLOBAL uint32_t *__restrict__ Ider = (const VO_GLOBAL uint32_t *)I.der[level];
            const VO_GLOBAL uint8_t *__restrict__ Jimg = (const VO_GLOBAL uint8_t *)J.lvl[level];

            prevX -= halfWin;
            prevY -= halfWin;
            const float fpx = floorf(prevX), fpy = floorf(prevY);
            const int ipx = uni(vo_f2i(fpx)), ipy = uni(vo_f2i(fpy));
            // x86's cvFloor(NaN) is INT_MIN (cvttss2si), i.e. "left of the window": OpenCV rejects a NaN coordinate here.
            // v_cvt_i32_f32(NaN) is 0, which would ADMIT it -- so NaN is rejected on the float (one
            // v_cmp_u_f32 into a scalar pair).  +-inf and finite values beyond int32 saturate to the rejected side on both.
            if (VO_BALLOT(__builtin_isunordered(fpx, fpy)) != 0ull || ipx < -LK_WIN || ipx >= iw || ipy < -LK_WIN || ipy >= ih) {
                if (level == 0)
                    st = 0;
                continue;
            }
            uint32_t wt, wb; // (float)ipx == floorf(prevX): the fractional part needs no int round trip
            lk_weights(prevX - fpx, prevY - fpy, wt, wb);

            // ---- 21 x 21 template straight from the bordered pyramid (registers) + structure tensor --
            // lane: pixels (ipx + c0 .. + 7, ipy + r) and the row below; the bordered layout makes
            // every admissible window an in-bounds read (reflected pixels, zero derivatives)
            // b1 = sum (J - I) * Ix is accumulated as sum J * Ix - sum I * Ix: the lane's own (exact, integer)
            // sum I * Ix is formed once per level and seeds the iteration's accumulator, so the inner loop
            // neither keeps I nor subtracts it
            uint32_t Ixp[4], Iyp[4];
            int a11 = 0, a12 = 0, a22 = 0, c1 = 0, c2 = 0;
            {
                // Addresses as a wave-uniform base (first element of the bordered allocation, scalar registers) plus a
                // per-lane 32-bit offset that is never negative: what global_load's scalar-base form takes, instead
                // of 64-bit per-lane pointer arithmetic.
                const ptrdiff_t org = (ptrdiff_t)VO_BY * istride + VO_BX;
                const VO_GLOBAL uint8_t *Ib = Iimg - org;
                const VO_GLOBAL uint8_t *Db = (const VO_GLOBAL uint8_t *)(Ider - org); // byte offsets: 4 * od fits 32 bits
                const uint32_t o = (uint32_t)((ipy + r + VO_BY) * istride + ipx + c0 + VO_BX);
                // lane 63 owns no pixel: it reads its Scharr samples from the (all-zero) top-left border
                // corner, so Ix = Iy = 0 there and every sum it feeds is 0 without any select
                const uint32_t od = live ? 4u * o : 0u, drow = 4u * (uint32_t)istride;
                const LkU2 t = *(const VO_GLOBAL LkU2 *)(Ib + o);
                const LkU2 u = *(const VO_GLOBAL LkU2 *)(Ib + (o + (uint32_t)istride));
                const LkU4 dt0 = *(const VO_GLOBAL LkU4 *)(Db + od);
                const LkU4 dt1 = *(const VO_GLOBAL LkU4 *)(Db + (od + 16u));
                const LkU4 db0 = *(const VO_GLOBAL LkU4 *)(Db + (od + drow));
                const LkU4 db1 = *(const VO_GLOBAL LkU4 *)(Db + (od + drow + 16u));
                const uint32_t dt[8] = {dt0.a, dt0.b, dt0.c, dt0.d, dt1.a, dt1.b, dt1.c, dt1.d};
                const uint32_t db[8] = {db0.a, db0.b, db0.c, db0.d, db1.a, db1.b, db1.c, db1.d};
                uint32_t Ip[4];
                bilinear7_u8(t.lo, t.hi, u.lo, u.hi, wt, wb, Ip);
                bilinear7_deriv(dt, db, wt, wb, Ixp, Iyp);
                a11 = sdot2_first(Ixp[0], Ixp[0], 0);
                a12 = sdot2_first(Ixp[0], Iyp[0], 0);
                a22 = sdot2_first(Iyp[0], Iyp[0], 0);
                c1 = sdot2_first(Ip[0], Ixp[0], 0);
                c2 = sdot2_first(Ip[0], Iyp[0], 0);
#pragma unroll
                for (int m = 1; m < 4; m++) {
                    a11 = sdot2(Ixp[m], Ixp[m], a11);
                    a12 = sdot2(Ixp[m], Iyp[m], a12);
                    a22 = sdot2(Iyp[m], Iyp[m], a22);
                    c1 = sdot2(Ip[m], Ixp[m], c1);
                    c2 = sdot2(Ip[m], Iyp[m], c2);
                }
            }
            float A11, A12, A22;
            wave_sum3_exact_f32(a11, a12, a22, A11, A12, A22);
            A11 *= FLT_SCALE;
            A12 *= FLT_SCALE;
            A22 *= FLT_SCALE;

            float D = A11 * A22 - A12 * A12;
            // OpenCV: minEig = (A22 + A11 - sqrt((A11-A22)^2 + 4 A12^2)) / (2 * 21 * 21), rejected when < minEigThreshold.
            // The correctly rounded sqrt and divide cost ~30 wave-uniform VALU instructions, and the test is almost
            // never close: with t = A22 + A11, s2 = the radicand and u = t - (1.001 * 882 * thr + t * 2^-18),
            // "u > 0 and s2 < 0.9999 u^2" implies sqrt(s2) < u (1 - 4e-5), hence the rounded t - sqrt(s2) exceeds
            // 1.0009 * 882 * thr and the rounded quotient exceeds thr (every rounding involved is below 2^-23
            // relative, the margins are 1e-3 and 2^-18 t): the exact expression cannot reject.  Otherwise evaluate it.
            const float t = A22 + A11;
            const float s2 = (A11 - A22) * (A11 - A22) + 4.f * A12 * A12;
            const float u = t - (min_eig_n + t * 3.814697265625e-6f);
            bool eig_ok = u > 0.f && s2 < u * u * 0.9999f;
            if (__builtin_expect(!eig_ok, 0)) {
#ifndef VO_HOST_EMUL
                asm volatile("" ::: "memory"); // keep the sqrt / divide out of the common path
#endif
                const float minEig = (t - sqrtf(s2)) / (float)(2 * LK_WIN * LK_WIN);
                eig_ok = !(minEig < prm.min_eig);
            }
            if (!eig_ok || D < FLT_EPSILON) {
                if (level == 0)
                    st = 0;
                continue;
            }
            D = 1.f / D;
            // b1, b2 carry the same 2^-20 as the A's; a power of two moves through the rounded products unchanged
            // (A12 * (S * 2^-20) == (A12 * 2^-20) * S bit for bit), so it is folded into the A's once per level
            const float A11s = A11 * FLT_SCALE, A12s = A12 * FLT_SCALE, A22s = A22 * FLT_SCALE;
            const int nc1 = -c1, nc2 = -c2;

            nextX -= halfWin;
            nextY -= halfWin;
            float prevDX = 0.f, prevDY = 0.f;
            int jx0 = 0, jy0 = 0;
            bool have_tile = false;
            // tile origins that keep the 40 x 48 tile inside the level's rectangle (vo_dev.h, "reads stay inside their level"):
            // the last tile ends exactly at the row end (column jstride - VO_BX - 1) and at the last border row
            static_assert(LK_JT_W % 16 == 0 && VO_BX % 4 == 0, "16-byte tile chunks at 4-byte aligned origins");
            static_assert(VO_BY >= LK_WIN && LK_JT_W >= LK_WIN + 1 + 12 + 3 && LK_JT_H >= LK_WIN + 1 + 9,
                          "a tile clamped to the row end / the last border row still covers the window of every corner the "
                          "reference admits (up to w - 1, h - 1: the borders are at least a window wide)");
            const int jx_max = jstride - VO_BX - LK_JT_W, jy_max = jh + VO_BY - LK_JT_H;

            // The window corner stays in the same pixel cell for two iterations out of three, so the Gauss-Newton
            // loop is written as two loops: the outer one is entered once per cell and does everything that only
            // depends on the cell -- admissibility test, tile test / refill, LDS address, the two row reads and the 14
            // v_perm_b32 that lift the pixel pairs into registers; the inner one iterates while the corner stays put.
            int j = 0;
            float fnx = floorf(nextX), fny = floorf(nextY);
            bool run = prm.max_count > 0;
            while (run) {
                const int inx = uni(vo_f2i(fnx)), iny = uni(vo_f2i(fny));
                if (VO_BALLOT(__builtin_isunordered(fnx, fny)) != 0ull || inx < -LK_WIN || inx >= jw || iny < -LK_WIN || iny >= jh) {
                    if (level == 0)
                        st = 0;
                    break;
                }
                // search tile must cover cols inx..inx+21, rows iny..iny+21
                if (!have_tile || inx < jx0 || inx + LK_WIN + 1 > jx0 + LK_JT_W || iny < jy0 ||
                    iny + LK_WIN + 1 > jy0 + LK_JT_H) {
                    jx0 = (inx - 12) & ~3;
                    jy0 = iny - 9;
                    jx0 = jx0 < -VO_BX ? -VO_BX : jx0 > jx_max ? jx_max : jx0;
                    jy0 = jy0 < -VO_BY ? -VO_BY : jy0 > jy_max ? jy_max : jy0;
                    __syncthreads(); // single-wave workgroup: orders the LDS reads before the refill
                    const VO_GLOBAL uint8_t *tb = Jimg + (ptrdiff_t)jy0 * jstride + jx0;
                    for (int c = lane; c < LK_JT_H * (LK_JT_W / 16); c += 64) {
                        const int row = c / (LK_JT_W / 16), col = c - row * (LK_JT_W / 16);
                        const LkU4 v = *(const VO_GLOBAL LkU4 *)(tb + (uint32_t)(row * jstride + 16 * col));
                        *reinterpret_cast<uint4 *>(&s_jt[row * LK_JT_W + 16 * col]) = make_uint4(v.a, v.b, v.c, v.d);
                    }
                    __syncthreads();
                    have_tile = true;
                }
                uint32_t Jt[7], Jb[7]; // the cell's pixel pairs (two window rows per lane)
                {
                    const int off = (iny - jy0) * LK_JT_W + (inx - jx0) + lane_off; // uniform part on the scalar unit
                    // two unaligned 8-byte LDS reads (gfx950 handles misaligned ds_read_b64; measured
                    // equal to three aligned dwords + v_alignbyte_b32 per row, profiles/r01 notes)
                    const LkU2 t = *reinterpret_cast<const LkU2 *>(&s_jt[off]);
                    const LkU2 u = *reinterpret_cast<const LkU2 *>(&s_jt[off + LK_JT_W]);
                    lift7(t.lo, t.hi, Jt);
                    lift7(u.lo, u.hi, Jb);
                }
                // fractional position of the window corner inside its pixel cell: the bilinear weights come from it, and
                // the corner is still inside the cell exactly as long as both parts are in [0, 1).  fl(nextX - fnx) is what
                // OpenCV itself computes (nextPt.x - inextPt.x); a true difference >= 1 or < 0 can never round into [0, 1),
                // so "inside" is never wrong -- a spurious "left" (difference just below 1 rounding up to 1.0) only
                // re-enters the same cell through the outer loop.  As raw bits, [0, 1) is "below 0x3f800000, unsigned"
                // (negative values have the sign bit set), so one unsigned max + one compare replace two floors, two
                // compares and two selects per iteration.
                for (;;) {
                    lk_weights(nextX - fnx, nextY - fny, wt, wb);
                    int b1, b2;
                    {
                        uint32_t Jp[4];
                        blend7(Jt, Jb, wt, wb, Jp);
                        b1 = sdot2_first(Jp[0], Ixp[0], nc1); // seeds: minus the lane's sum I * Ix, sum I * Iy
                        b2 = sdot2_first(Jp[0], Iyp[0], nc2);
#pragma unroll
                        for (int m = 1; m < 4; m++) {
                            b1 = sdot2(Jp[m], Ixp[m], b1);
                            b2 = sdot2(Jp[m], Iyp[m], b2);
                        }
                    }
                    float fb1, fb2;
                    wave_sum2_exact_f32(b1, b2, fb1, fb2);
                    const float dx = (A12s * fb2 - A22s * fb1) * D;
                    const float dy = (A12s * fb1 - A11s * fb2) * D;
                    nextX += dx;
                    nextY += dy;
                    outX = nextX + halfWin;
                    outY = nextY + halfWin;
                    // OpenCV: delta.ddot(delta) <= epsilon in f64.  The f32 value n2 is within 2^-23 of it, so it
                    // decides on its own unless it falls inside a 1e-6 band around epsilon (then the f64 form).  The hot
                    // path carries ONE compare ("clearly not converged"); everything else happens once per level.
                    const float n2 = fmaf(dy, dy, dx * dx);
                    if (__builtin_expect(!(n2 > eps_hi), 0)) {
#ifndef VO_HOST_EMUL
                        asm volatile("" ::: "memory"); // keep the rare evaluation out of the hot path
#endif
                        if (n2 < eps_lo || (double)dx * dx + (double)dy * dy <= prm.epsilon) {
                            run = false;
                            break;
                        }
                    }
                    // OpenCV: std::abs(delta.x + prevDelta.x) < 0.01 (f32 sum compared as double).  0.01f is
                    // the largest f32 below the double 0.01, so for an f32 s:  |s| < 0.01  <=>  |s| <= 0.01f
                    if (j > 0 && fabsf(dx + prevDX) <= 0.01f && fabsf(dy + prevDY) <= 0.01f) {
#ifndef VO_HOST_EMUL
                        asm volatile("" ::: "memory"); // do not speculate the half-step into every iteration
#endif
                        outX -= dx * 0.5f;
                        outY -= dy * 0.5f;
                        run = false;
                        break;
                    }
                    prevDX = dx;
                    prevDY = dy;
                    if (++j >= prm.max_count) {
                        run = false;
                        break;
                    }
                    const uint32_t ua = (uint32_t)__float_as_int(nextX - fnx), ub = (uint32_t)__float_as_int(nextY - fny);
                    if (VO_BALLOT((ua > ub ? ua : ub) >= 0x3f800000u) != 0ull) { // the corner left the cell
                        fnx = floorf(nextX);
                        fny = floorf(nextY);
                        break;
                    }
                }
            }

            // final in-bounds check OpenCV performs at level 0 when an err vector is requested
            // (st == 1 here means the last cell entry was admitted and every delta since was finite: outX / outY are finite,
            // possibly huge -- both conversions then land on a rejected side, x86's INT_MIN and gfx950's saturated INT_MAX)
            if (st && level == 0) {
                const int fx = vo_f2i(floorf(outX - halfWin)), fy = vo_f2i(floorf(outY - halfWin));
                if (fx < -LK_WIN || fx >= jw || fy < -LK_WIN || fy >= jh)
                    st = 0;
            }
        }

        if (lane == 0) {
            trk[((size_t)frame * 4 + hop) * cap + f] = make_float2(outX, outY);
            status[((size_t)frame * 4 + hop) * cap + f] = (uint8_t)st;
        }
        // deleteUnmatchFeaturesCircle (feature.cpp:96-104) drops the feature if any hop failed or any
        // of pt0..pt3 has a negative coordinate: nothing computed after that point can reach an output
        // of circularMatching(), so the wave retires (remaining hops reported as status 0)
        const bool dead = st == 0 || outX < 0.f || outY < 0.f || (hop == 0 && (p.x < 0.f || p.y < 0.f));
        if (dead && !prm.full_chain && hop < 3) {
            if (lane == 0)
                for (int k = hop + 1; k < 4; k++) {
                    trk[((size_t)frame * 4 + k) * cap + f] = make_float2(-1.f, -1.f);
                    status[((size_t)frame * 4 + k) * cap + f] = 0;
                }
            return;
        }
        prevPtX = unif(outX);
        prevPtY = unif(outY);
    }
}

__global__ VO_LK_ATTRS void lk_circular_kernel(const PyrImage *__restrict__ imgs,
                                                          const Quad *__restrict__ quads,
                                                          const float2 *__restrict__ pts_in,
                                                          const int *__restrict__ n_pts, int cap,
                                                          int n_frames, int fpg /* 1, 2, 4 or 8 */,
                                                          int ppp /* features per part */,
                                                          float2 *__restrict__ trk,      // [B][4][cap]
                                                          uint8_t *__restrict__ status,  // [B][4][cap]
                                                          LkParams prm)
{
    lk_circular_body<false>(imgs, quads, pts_in, n_pts, cap, n_frames, fpg, ppp, trk, status, prm, 0, 4);
}

// Hops hop_begin .. hop_end - 1 only: the synchronous drop-in calls (capi_run.hip) launch hop 0 -- which reads the t0 pair
// only -- while the t1 pair is still crossing PCIe and its pyramids are being built on another stream, then hops 1 .. 3.
__global__ VO_LK_ATTRS void lk_hops_kernel(const PyrImage *__restrict__ imgs, const Quad *__restrict__ quads,
                                           const float2 *__restrict__ pts_in, const int *__restrict__ n_pts, int cap,
                                           int n_frames, int fpg, int ppp, float2 *__restrict__ trk,
                                           uint8_t *__restrict__ status, LkParams prm, int hop_begin, int hop_end)
{
    lk_circular_body<true>(imgs, quads, pts_in, n_pts, cap, n_frames, fpg, ppp, trk, status, prm, hop_begin, hop_end);
}

// Developer variant: compiled into libvo_hip_dev.so (python -m visual_odom_amd.build --dev) and into the CPU emulator of
// tests/host_check only -- measured 4-6 % slower than lk_circular_kernel, so the product library does not carry it.
#if defined(VO_DEV_VARIANTS) || defined(VO_HOST_EMUL)
// ---------------------------------------------------------------------------------------------------------------------
// TWO FEATURES PER WAVEFRONT ("build and measure the 2-features-per-wave variant").
// 65 % of the iteration's VALU slots above are wave-uniform arithmetic replicated over 64 lanes (weights, reduction tail,
// 2 x 2 solve, convergence).  Here a wavefront tracks two features of one frame, one per 32-lane half: lane hl of a half owns
// the row segments 2 hl and 2 hl + 1 (14 pixels), so the per-pixel work per feature is unchanged while the uniform part and
// the reduction tree are shared by two features.  The halves run in lock step per (hop, level): a level's Gauss-Newton loop
// runs until both halves have stopped (a stopped half only idles: every state update is predicated), cell entries (tile
// test, lift of the pixel pairs) happen when either half needs one.  All per-feature quantities are per-lane values that
// are uniform within a half; only the images, the level and the iteration counter stay wave-uniform.  The arithmetic of a
// running half is operation for operation that of lk_circular_kernel (same helpers, same expressions), so the results are
// bit-identical (tests: the CPU emulator run of this source and the GPU run against the one-feature kernel).
//
// per-half exact sums of two per-lane int32 partials (|v| < 2^29: 14 products per lane), as f32 rounded once from the exact
// integer.  v_permlane16_swap folds a | b into the row pairs (even rows: a, odd rows: b), one quad step keeps the 4-value sums
// below 2^31, then the signed-high / unsigned-low halves finish the 16-lane rows (3 DPP steps each); one fma recombines
// and a second swap hands every lane of a half both totals.
__device__ __forceinline__ void half_sum2_exact_f32(int a, int b, float &fa, float &fb)
{
    VO_PERMLANE16_SWAP(a, b);
    int t = a + b;
    t = dpp_add<VO_DPP_QUAD_XOR1, 0xf>(t);
    int hi = t >> 16, lo = t & 0xffff;
    hi = dpp_add<VO_DPP_QUAD_XOR2, 0xf>(hi);
    lo = dpp_add<VO_DPP_QUAD_XOR2, 0xf>(lo);
    hi = dpp_add<VO_DPP_ROW_HALF_MIRROR, 0xf>(hi);
    lo = dpp_add<VO_DPP_ROW_HALF_MIRROR, 0xf>(lo);
    hi = dpp_add<VO_DPP_ROW_MIRROR, 0xf>(hi);
    lo = dpp_add<VO_DPP_ROW_MIRROR, 0xf>(lo);
    const float res = fmaf((float)hi, 65536.f, (float)lo);
    int x = __float_as_int(res), y = x;
    VO_PERMLANE16_SWAP(x, y); // x: the even row's value (a) in both rows of a pair, y: the odd row's (b)
    fa = __int_as_float(x);
    fb = __int_as_float(y);
}

#ifndef VO_LK_PAIR_ATTRS
#define VO_LK_PAIR_ATTRS __launch_bounds__(64, 4)
#endif
__global__ VO_LK_PAIR_ATTRS void lk_circular_pair_kernel(const PyrImage *__restrict__ imgs,
                                                         const Quad *__restrict__ quads,
                                                         const float2 *__restrict__ pts_in,
                                                         const int *__restrict__ n_pts, int cap, int n_frames,
                                                         int fpg /* 1, 2, 4 or 8 */, int ppp /* PAIRS per part */,
                                                         float2 *__restrict__ trk,     // [B][4][cap]
                                                         uint8_t *__restrict__ status, // [B][4][cap]
                                                         LkParams prm)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_jt[2 * LK_JT_H * LK_JT_W];

    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int grp = slot / ppp, fi = slot - grp * ppp;
    const int frame = grp * fpg + (xcd & (fpg - 1));
    if (frame >= n_frames)
        return;
    const int lane = threadIdx.x, half = lane >> 5, hl = lane & 31;
    const int f = 2 * ((xcd / fpg) * ppp + fi) + half;
    bool act = f < n_pts[frame]; // this half still tracks its feature
    if (VO_BALLOT(act) == 0ull)
        return;
    // lane -> two (row, 7-pixel segment) slots; slot 63 (second slot of lane 31) owns no pixel: it duplicates slot 62's
    // pixel addresses and reads its Scharr samples from the all-zero border corner
    const int sA = 2 * hl, sB = 2 * hl + 1;
    const bool liveB = sB < 63;
    const int sBc = liveB ? sB : 62;
    const int rA = sA / 3, cA = 7 * (sA - 3 * rA), rB = sBc / 3, cB = 7 * (sBc - 3 * rB);
    const int offA = rA * LK_JT_W + cA, offB = rB * LK_JT_W + cB;
    uint8_t *const tile = s_jt + half * (LK_JT_H * LK_JT_W);

    const Quad q = quads[frame];
    const float halfWin = (LK_WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float eps_lo = (float)(prm.epsilon * 0.999999), eps_hi = (float)(prm.epsilon * 1.000001);
    const float min_eig_n = prm.min_eig * (1.001f * (float)(2 * LK_WIN * LK_WIN));

    float2 p = make_float2(0.f, 0.f);
    if (act)
        p = pts_in[(size_t)frame * cap + f];
    float prevPtX = p.x, prevPtY = p.y;

    for (int hop = 0; hop < 4; hop++) {
        const int pi = hop == 0 ? q.l0 : hop == 1 ? q.r0 : hop == 2 ? q.r1 : q.l1;
        const int ni = hop == 0 ? q.r0 : hop == 1 ? q.r1 : hop == 2 ? q.l1 : q.l0;
        const PyrImage &I = imgs[pi];
        const PyrImage &J = imgs[ni];
        float outX = 0.f, outY = 0.f;
        int st = 1;

        for (int level = prm.max_level; level >= 0; level--) {
            const float scale = __int_as_float((127 - level) << 23);
            float prevX = prevPtX * scale, prevY = prevPtY * scale;
            float nextX, nextY;
            if (level == prm.max_level) {
                nextX = prevX;
                nextY = prevY;
            } else {
                nextX = outX * 2.f;
                nextY = outY * 2.f;
            }
            outX = nextX;
            outY = nextY;

            const int iw = I.w[level], ih = I.h[level], istride = I.stride[level];
            const int jw = J.w[level], jh = J.h[level], jstride = J.stride[level];
            const VO_GLOBAL uint8_t *__restrict__ Iimg = (const VO_GLOBAL uint8_t *)I.lvl[level];
            const VO_GLOBAL uint32_t *__restrict__ Ider = (const VO_GLOBAL uint32_t *)I.der[level];
            const VO_GLOBAL uint8_t *__restrict__ Jimg = (const VO_GLOBAL uint8_t *)J.lvl[level];

            prevX -= halfWin;
            prevY -= halfWin;
            const float fpx = floorf(prevX), fpy = floorf(prevY);
            const int ipx = vo_f2i(fpx), ipy = vo_f2i(fpy);
            const bool in = act && !(__builtin_isunordered(fpx, fpy) || ipx < -LK_WIN || ipx >= iw || ipy < -LK_WIN || ipy >= ih);
            if (act && !in && level == 0)
                st = 0;
            if (VO_BALLOT(in) == 0ull)
                continue;
            uint32_t wt, wb;
            lk_weights(prevX - fpx, prevY - fpy, wt, wb);

            // ---- template of both slots + structure tensor (a half that sits this level out reads window (0, 0)) ----
            uint32_t IxA[4], IyA[4], IxB[4], IyB[4];
            int a11, a12, a22, c1, c2;
            {
                const int ipxs = in ? ipx : 0, ipys = in ? ipy : 0;
                const ptrdiff_t org = (ptrdiff_t)VO_BY * istride + VO_BX;
                const VO_GLOBAL uint8_t *Ib = Iimg - org;
                const VO_GLOBAL uint8_t *Db = (const VO_GLOBAL uint8_t *)(Ider - org);
                const uint32_t drow = 4u * (uint32_t)istride;
#define VO_LK_SLOT(r, c0, live, Ixp, Iyp, FIRST)                                                                       \
    {                                                                                                                  \
        const uint32_t o = (uint32_t)((ipys + (r) + VO_BY) * istride + ipxs + (c0) + VO_BX);                           \
        const uint32_t od = (live) ? 4u * o : 0u;                                                                      \
        const LkU2 t = *(const VO_GLOBAL LkU2 *)(Ib + o);                                                              \
        const LkU2 u = *(const VO_GLOBAL LkU2 *)(Ib + (o + (uint32_t)istride));                                        \
        const LkU4 dt0 = *(const VO_GLOBAL LkU4 *)(Db + od);                                                           \
        const LkU4 dt1 = *(const VO_GLOBAL LkU4 *)(Db + (od + 16u));                                                   \
        const LkU4 db0 = *(const VO_GLOBAL LkU4 *)(Db + (od + drow));                                                  \
        const LkU4 db1 = *(const VO_GLOBAL LkU4 *)(Db + (od + drow + 16u));                                            \
        const uint32_t dt[8] = {dt0.a, dt0.b, dt0.c, dt0.d, dt1.a, dt1.b, dt1.c, dt1.d};                               \
        const uint32_t db[8] = {db0.a, db0.b, db0.c, db0.d, db1.a, db1.b, db1.c, db1.d};                               \
        uint32_t Ip[4];                                                                                                \
        bilinear7_u8(t.lo, t.hi, u.lo, u.hi, wt, wb, Ip);                                                              \
        bilinear7_deriv(dt, db, wt, wb, Ixp, Iyp);                                                                     \
        _Pragma("unroll") for (int m = 0; m < 4; m++)                                                                  \
        {                                                                                                              \
            if ((FIRST) && m == 0) {                                                                                   \
                a11 = sdot2_first(Ixp[0], Ixp[0], 0);                                                                  \
                a12 = sdot2_first(Ixp[0], Iyp[0], 0);                                                                  \
                a22 = sdot2_first(Iyp[0], Iyp[0], 0);                                                                  \
                c1 = sdot2_first(Ip[0], Ixp[0], 0);                                                                    \
                c2 = sdot2_first(Ip[0], Iyp[0], 0);                                                                    \
            } else {                                                                                                   \
                a11 = sdot2(Ixp[m], Ixp[m], a11);                                                                      \
                a12 = sdot2(Ixp[m], Iyp[m], a12);                                                                      \
                a22 = sdot2(Iyp[m], Iyp[m], a22);                                                                      \
                c1 = sdot2(Ip[m], Ixp[m], c1);                                                                         \
                c2 = sdot2(Ip[m], Iyp[m], c2);                                                                         \
            }                                                                                                          \
        }                                                                                                              \
    }
                VO_LK_SLOT(rA, cA, true, IxA, IyA, true)
                VO_LK_SLOT(rB, cB, liveB, IxB, IyB, false)
#undef VO_LK_SLOT
            }
            float A11, A12, A22, unused;
            half_sum2_exact_f32(a11, a12, A11, A12);
            half_sum2_exact_f32(a22, 0, A22, unused);
            A11 *= FLT_SCALE;
            A12 *= FLT_SCALE;
            A22 *= FLT_SCALE;

            float D = A11 * A22 - A12 * A12;
            // min-eigenvalue test: the conservative quick form of lk_circular_kernel, the exact expression where it does not decide
            const float t = A22 + A11;
            const float s2 = (A11 - A22) * (A11 - A22) + 4.f * A12 * A12;
            const float u = t - (min_eig_n + t * 3.814697265625e-6f);
            bool eig_ok = u > 0.f && s2 < u * u * 0.9999f;
            if (VO_BALLOT(in && !eig_ok) != 0ull) {
                const float minEig = (t - sqrtf(s2)) / (float)(2 * LK_WIN * LK_WIN);
                eig_ok = eig_ok || !(minEig < prm.min_eig); // the quick form implies the exact one
            }
            const bool ok = in && eig_ok && !(D < FLT_EPSILON);
            if (in && !ok && level == 0)
                st = 0;
            if (VO_BALLOT(ok) == 0ull)
                continue;
            D = 1.f / D;
            const float A11s = A11 * FLT_SCALE, A12s = A12 * FLT_SCALE, A22s = A22 * FLT_SCALE;
            const int nc1 = -c1, nc2 = -c2;

            nextX -= halfWin;
            nextY -= halfWin;
            float prevDX = 0.f, prevDY = 0.f;
            int jx0 = 0, jy0 = 0, offT = 0;
            bool have_tile = false;
            const int jx_max = jstride - VO_BX - LK_JT_W, jy_max = jh + VO_BY - LK_JT_H;

            int j = 0; // iterations done at this level: the same for every half that still runs
            float fnx = floorf(nextX), fny = floorf(nextY);
            bool run = ok && prm.max_count > 0;
            bool entry = run; // this half has to (re-)enter a pixel cell before its next iteration
            uint32_t JtA[7], JbA[7], JtB[7], JbB[7];
            while (VO_BALLOT(run) != 0ull) {
                if (VO_BALLOT(entry) != 0ull) {
                    const int inx = vo_f2i(fnx), iny = vo_f2i(fny);
                    const bool oob = entry && (__builtin_isunordered(fnx, fny) || inx < -LK_WIN || inx >= jw || iny < -LK_WIN || iny >= jh);
                    if (oob) {
                        if (level == 0)
                            st = 0;
                        run = false;
                    }
                    const bool e2 = entry && !oob;
                    const bool need = e2 && (!have_tile || inx < jx0 || inx + LK_WIN + 1 > jx0 + LK_JT_W || iny < jy0 ||
                                             iny + LK_WIN + 1 > jy0 + LK_JT_H);
                    if (VO_BALLOT(need) != 0ull) {
                        if (need) {
                            jx0 = (inx - 12) & ~3;
                            jy0 = iny - 9;
                            jx0 = jx0 < -VO_BX ? -VO_BX : jx0 > jx_max ? jx_max : jx0;
                            jy0 = jy0 < -VO_BY ? -VO_BY : jy0 > jy_max ? jy_max : jy0;
                        }
                        __syncthreads(); // single-wave workgroup: orders the LDS reads before the refill
                        if (need) {
                            const VO_GLOBAL uint8_t *tb = Jimg + ((ptrdiff_t)jy0 * jstride + jx0);
                            for (int c = hl; c < LK_JT_H * (LK_JT_W / 16); c += 32) {
                                const int row = c / (LK_JT_W / 16), col = c - row * (LK_JT_W / 16);
                                const LkU4 v = *(const VO_GLOBAL LkU4 *)(tb + (uint32_t)(row * jstride + 16 * col));
                                *reinterpret_cast<uint4 *>(&tile[row * LK_JT_W + 16 * col]) = make_uint4(v.a, v.b, v.c, v.d);
                            }
                            have_tile = true;
                        }
                        __syncthreads();
                    }
                    if (e2)
                        offT = (iny - jy0) * LK_JT_W + (inx - jx0);
                    {
                        // every lane lifts its current cell again (a half that did not move re-reads the same bytes; a
                        // half without a tile reads offset 0 and never uses the result)
                        const LkU2 tA = *reinterpret_cast<const LkU2 *>(&tile[offT + offA]);
                        const LkU2 uA = *reinterpret_cast<const LkU2 *>(&tile[offT + offA + LK_JT_W]);
                        const LkU2 tB = *reinterpret_cast<const LkU2 *>(&tile[offT + offB]);
                        const LkU2 uB = *reinterpret_cast<const LkU2 *>(&tile[offT + offB + LK_JT_W]);
                        lift7(tA.lo, tA.hi, JtA);
                        lift7(uA.lo, uA.hi, JbA);
                        lift7(tB.lo, tB.hi, JtB);
                        lift7(uB.lo, uB.hi, JbB);
                    }
                    entry = false;
                    if (VO_BALLOT(run) == 0ull)
                        break;
                }
                lk_weights(nextX - fnx, nextY - fny, wt, wb);
                int b1, b2;
                {
                    uint32_t Jp[4];
                    blend7(JtA, JbA, wt, wb, Jp);
                    b1 = sdot2_first(Jp[0], IxA[0], nc1);
                    b2 = sdot2_first(Jp[0], IyA[0], nc2);
#pragma unroll
                    for (int m = 1; m < 4; m++) {
                        b1 = sdot2(Jp[m], IxA[m], b1);
                        b2 = sdot2(Jp[m], IyA[m], b2);
                    }
                    blend7(JtB, JbB, wt, wb, Jp);
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        b1 = sdot2(Jp[m], IxB[m], b1);
                        b2 = sdot2(Jp[m], IyB[m], b2);
                    }
                }
                float fb1, fb2;
                half_sum2_exact_f32(b1, b2, fb1, fb2);
                const float dx = (A12s * fb2 - A22s * fb1) * D;
                const float dy = (A12s * fb1 - A11s * fb2) * D;
                if (run) {
                    nextX += dx;
                    nextY += dy;
                    outX = nextX + halfWin;
                    outY = nextY + halfWin;
                    const float n2 = fmaf(dy, dy, dx * dx);
                    bool stop = false;
                    if (!(n2 > eps_hi))
                        stop = n2 < eps_lo || (double)dx * dx + (double)dy * dy <= prm.epsilon;
                    if (!stop && j > 0 && fabsf(dx + prevDX) <= 0.01f && fabsf(dy + prevDY) <= 0.01f) {
                        outX -= dx * 0.5f;
                        outY -= dy * 0.5f;
                        stop = true;
                    }
                    prevDX = dx;
                    prevDY = dy;
                    if (stop)
                        run = false;
                }
                if (++j >= prm.max_count)
                    run = false;
                if (run) {
                    const uint32_t ua = (uint32_t)__float_as_int(nextX - fnx), ub = (uint32_t)__float_as_int(nextY - fny);
                    if ((ua > ub ? ua : ub) >= 0x3f800000u) { // the corner left the cell
                        fnx = floorf(nextX);
                        fny = floorf(nextY);
                        entry = true;
                    }
                }
            }

            // final in-bounds check OpenCV performs at level 0 when an err vector is requested
            if (ok && st && level == 0) {
                const int fx = vo_f2i(floorf(outX - halfWin)), fy = vo_f2i(floorf(outY - halfWin));
                if (fx < -LK_WIN || fx >= jw || fy < -LK_WIN || fy >= jh)
                    st = 0;
            }
        }

        if (act && hl == 0) {
            trk[((size_t)frame * 4 + hop) * cap + f] = make_float2(outX, outY);
            status[((size_t)frame * 4 + hop) * cap + f] = (uint8_t)st;
        }
        // a feature deleteUnmatchFeaturesCircle is going to drop stops here (see lk_circular_kernel)
        const bool dead = st == 0 || outX < 0.f || outY < 0.f || (hop == 0 && (p.x < 0.f || p.y < 0.f));
        if (act && dead && !prm.full_chain && hop < 3) {
            if (hl == 0)
                for (int k = hop + 1; k < 4; k++) {
                    trk[((size_t)frame * 4 + k) * cap + f] = make_float2(-1.f, -1.f);
                    status[((size_t)frame * 4 + k) * cap + f] = 0;
                }
            act = false;
        }
        if (VO_BALLOT(act) == 0ull)
            return;
        prevPtX = outX;
        prevPtY = outY;
    }
}

#endif // VO_DEV_VARIANTS || VO_HOST_EMUL
#ifndef VO_HOST_EMUL
void launch_lk_circular(const PyrImage *d_imgs, const Quad *d_quads, const float2 *d_pts, const int *d_npts,
                        int cap, int max_pts, int n_frames, float2 *d_trk, uint8_t *d_status,
                        const LkParams &prm, hipStream_t stream)
{
    if (max_pts <= 0 || n_frames <= 0)
        return;
    // frames per group of 8 XCDs: largest power of two <= min(8, n_frames)
    const int fpg = n_frames >= 8 ? 8 : n_frames >= 4 ? 4 : n_frames >= 2 ? 2 : 1;
    const int parts = 8 / fpg, ppp = (max_pts + parts - 1) / parts;
    const int groups = (n_frames + fpg - 1) / fpg;
    dim3 grid((unsigned)(8 * groups * ppp));
    hipLaunchKernelGGL(lk_circular_kernel, grid, dim3(64), 0, stream, d_imgs, d_quads, d_pts, d_npts, cap, n_frames,
                       fpg, ppp, d_trk, d_status, prm);
}

void launch_lk_hops(const PyrImage *d_imgs, const Quad *d_quads, const float2 *d_pts, const int *d_npts, int cap, int max_pts,
                    int n_frames, float2 *d_trk, uint8_t *d_status, const LkParams &prm, int hop_begin, int hop_end,
                    hipStream_t stream)
{
    if (max_pts <= 0 || n_frames <= 0 || hop_begin < 0 || hop_end > 4 || hop_begin >= hop_end)
        return;
    const int fpg = n_frames >= 8 ? 8 : n_frames >= 4 ? 4 : n_frames >= 2 ? 2 : 1;
    const int parts = 8 / fpg, ppp = (max_pts + parts - 1) / parts;
    const int groups = (n_frames + fpg - 1) / fpg;
    dim3 grid((unsigned)(8 * groups * ppp));
    hipLaunchKernelGGL(lk_hops_kernel, grid, dim3(64), 0, stream, d_imgs, d_quads, d_pts, d_npts, cap, n_frames, fpg, ppp,
                       d_trk, d_status, prm, hop_begin, hop_end);
}

#ifdef VO_DEV_VARIANTS
void launch_lk_circular_pair(const PyrImage *d_imgs, const Quad *d_quads, const float2 *d_pts, const int *d_npts,
                             int cap, int max_pts, int n_frames, float2 *d_trk, uint8_t *d_status,
                             const LkParams &prm, hipStream_t stream)
{
    if (max_pts <= 0 || n_frames <= 0)
        return;
    const int fpg = n_frames >= 8 ? 8 : n_frames >= 4 ? 4 : n_frames >= 2 ? 2 : 1;
    const int parts = 8 / fpg, pairs = (max_pts + 1) / 2, ppp = (pairs + parts - 1) / parts;
    const int groups = (n_frames + fpg - 1) / fpg;
    dim3 grid((unsigned)(8 * groups * ppp));
    hipLaunchKernelGGL(lk_circular_pair_kernel, grid, dim3(64), 0, stream, d_imgs, d_quads, d_pts, d_npts, cap, n_frames,
                       fpg, ppp, d_trk, d_status, prm);
}
#endif // VO_DEV_VARIANTS

#endif // VO_HOST_EMUL

} // namespace vo
