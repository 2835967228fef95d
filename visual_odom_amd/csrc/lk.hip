// lk.hip -- fused 4-hop pyramidal Lucas-Kanade "circular matching" kernel.
//
// Replaces the four chained cv::calcOpticalFlowPyrLK calls of circularMatching()
// (reference src/feature.cpp:136-139: l0->r0, r0->r1, r1->l1, l1->l0; winSize 21x21, maxLevel 3,
// TermCriteria(COUNT+EPS, 30, 0.01), flags 0, minEigThreshold 1e-3, err vector requested).
// Semantics follow OpenCV's CPU LKTrackerInvoker: 14-bit fixed-point bilinear weights, int16
// template (I*32, Scharr Ix/Iy), 2x2 structure tensor, min-eigenvalue test, <=30 Gauss-Newton
// iterations with the eps^2 and oscillation stops, +-winSize admissibility window, final in-bounds
// status check.  The five sums A11/A12/A22/b1/b2 are accumulated exactly (int32 per lane, int64
// across the wave) so the result does not depend on reduction order; the 2x2 solve is plain f32
// with contraction off.
//
// Mapping (CDNA4): ONE WAVEFRONT PER FEATURE, one single-wave workgroup per feature.  A feature is
// a serial chain (4 hops x 4 levels x <=30 iterations) but independent of every other feature, so
// the wave keeps all per-feature state in registers and never synchronises with another wave.
//   lane l -> window row r = l/3, column segment s = l%3 (7 px): 63 lanes cover the 21x21 window,
//   each lane keeps its 7 template samples (I, Ix, Iy) in VGPRs for the whole level.
//   LDS per wave: 24x32 u8 template source tile, 22x22 packed (Ix,Iy) Scharr tile, 40x48 u8 search
//   tile of J that is re-fetched only when the window leaves it.  Tiles are filled with aligned
//   dword loads (rows of the pyramid are 16-B aligned); tiles touching the image border take a
//   byte path that applies REFLECT_101.
#include "vo_kernels.h"

#include <float.h>

namespace vo {

constexpr int LK_WIN = 21;
constexpr int LK_IT_W = 32, LK_IT_H = 24, LK_IT_STRIDE = 36; // template source tile (bytes)
constexpr int LK_D_W = 22;                                    // derivative tile 22 x 22 dwords
constexpr int LK_JT_W = 48, LK_JT_H = 40, LK_JT_STRIDE = 52;  // search tile (bytes)
constexpr int LK_W_BITS = 14;

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// fill a (rows x wbytes) u8 tile at LDS `dst` (row stride dstride) from image coords (x0, y0);
// x0 must be a multiple of 4 when the aligned path is taken
template <int WBYTES, int ROWS, int DSTRIDE>
__device__ __forceinline__ void load_tile(uint8_t *dst, const uint8_t *__restrict__ img, int stride, int w,
                                          int h, int x0, int y0, int lane)
{
    const bool interior = x0 >= 0 && x0 + WBYTES <= w && y0 >= 0 && y0 + ROWS <= h;
    if (interior) {
        constexpr int DW = WBYTES / 4;
        const uint8_t *base = img + (size_t)y0 * stride + x0;
        for (int i = lane; i < ROWS * DW; i += 64) {
            int r = i / DW, c = i - r * DW;
            uint32_t v = *reinterpret_cast<const uint32_t *>(base + (size_t)r * stride + 4 * c);
            *reinterpret_cast<uint32_t *>(dst + r * DSTRIDE + 4 * c) = v;
        }
    } else {
        for (int i = lane; i < ROWS * WBYTES; i += 64) {
            int r = i / WBYTES, c = i - r * WBYTES;
            int y = reflect101(y0 + r, h), x = reflect101(x0 + c, w);
            dst[r * DSTRIDE + c] = img[(size_t)y * stride + x];
        }
    }
}

__global__ __launch_bounds__(64) void lk_circular_kernel(const PyrImage *__restrict__ imgs,
                                                          const Quad *__restrict__ quads,
                                                          const float2 *__restrict__ pts_in,
                                                          const int *__restrict__ n_pts, int cap,
                                                          float2 *__restrict__ trk,      // [B][4][cap]
                                                          uint8_t *__restrict__ status,  // [B][4][cap]
                                                          LkParams prm)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_it[LK_IT_H * LK_IT_STRIDE];
    __shared__ __attribute__((aligned(16))) int s_d[LK_D_W * LK_D_W];
    __shared__ __attribute__((aligned(16))) uint8_t s_jt[LK_JT_H * LK_JT_STRIDE];

    const int frame = blockIdx.y, f = blockIdx.x;
    if (f >= n_pts[frame])
        return;
    const int lane = threadIdx.x;
    // lane -> (row, 7-pixel segment); lane 63 has no pixels (addresses clamped, contributions zeroed)
    const bool live = lane < 63;
    const int r = live ? lane / 3 : 20, seg = live ? lane - 3 * (lane / 3) : 0;
    const int c0 = 7 * seg;

    const Quad q = quads[frame];
    const float halfWin = (LK_WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);

    float2 p = pts_in[(size_t)frame * cap + f];
    float prevPtX = unif(p.x), prevPtY = unif(p.y);

    for (int hop = 0; hop < 4; hop++) {
        // hop chain: l0 -> r0 -> r1 -> l1 -> l0
        const int pi = hop == 0 ? q.l0 : hop == 1 ? q.r0 : hop == 2 ? q.r1 : q.l1;
        const int ni = hop == 0 ? q.r0 : hop == 1 ? q.r1 : hop == 2 ? q.l1 : q.l0;
        const PyrImage &I = imgs[pi];
        const PyrImage &J = imgs[ni];
        float outX = 0.f, outY = 0.f;
        int st = 1;

        for (int level = prm.max_level; level >= 0; level--) {
            const float scale = 1.f / (float)(1 << level);
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
            const uint8_t *__restrict__ Iimg = I.lvl[level];
            const uint8_t *__restrict__ Jimg = J.lvl[level];

            prevX -= halfWin;
            prevY -= halfWin;
            const int ipx = uni((int)floorf(prevX)), ipy = uni((int)floorf(prevY));
            if (ipx < -LK_WIN || ipx >= iw || ipy < -LK_WIN || ipy >= ih) {
                if (level == 0)
                    st = 0;
                continue;
            }
            float a = prevX - ipx, b = prevY - ipy;
            int iw00 = uni(__float2int_rn((1.f - a) * (1.f - b) * (1 << LK_W_BITS)));
            int iw01 = uni(__float2int_rn(a * (1.f - b) * (1 << LK_W_BITS)));
            int iw10 = uni(__float2int_rn((1.f - a) * b * (1 << LK_W_BITS)));
            int iw11 = (1 << LK_W_BITS) - iw00 - iw01 - iw10;

            // ---- template source tile: image rows ipy-1 .. ipy+22, cols x0 .. x0+31 ----
            const int tx0 = (ipx - 1) & ~3, ty0 = ipy - 1;
            __syncthreads(); // previous level's readers of s_it / s_d / s_jt are done
            load_tile<LK_IT_W, LK_IT_H, LK_IT_STRIDE>(s_it, Iimg, istride, iw, ih, tx0, ty0, lane);
            __syncthreads();

            // ---- Scharr derivative at the 22 x 22 integer positions (ipx+x, ipy+y) ----
            // zero outside the image (derivative buffer has a CONSTANT border), reflected inside
            for (int i = lane; i < LK_D_W * LK_D_W; i += 64) {
                int y = i / LK_D_W, x = i - y * LK_D_W;
                int gx = ipx + x, gy = ipy + y;
                int packed = 0;
                if (gx >= 0 && gx < iw && gy >= 0 && gy < ih) {
                    const uint8_t *c = &s_it[(y + 1) * LK_IT_STRIDE + (gx - tx0)];
                    int p00 = c[-LK_IT_STRIDE - 1], p01 = c[-LK_IT_STRIDE], p02 = c[-LK_IT_STRIDE + 1];
                    int p10 = c[-1], p12 = c[1];
                    int p20 = c[LK_IT_STRIDE - 1], p21 = c[LK_IT_STRIDE], p22 = c[LK_IT_STRIDE + 1];
                    // t0(x) = (row-1 + row+1)*3 + row*10 ; t1(x) = row+1 - row-1
                    int ix = ((p02 + p22) * 3 + p12 * 10) - ((p00 + p20) * 3 + p10 * 10);
                    int iy = ((p22 - p02) + (p20 - p00)) * 3 + (p21 - p01) * 10;
                    packed = (ix & 0xffff) | (iy << 16);
                }
                s_d[i] = packed;
            }
            __syncthreads();

            // ---- 21 x 21 template (registers) + structure tensor ----
            int Ival[7], Ixv[7], Iyv[7];
            int a11 = 0, a12 = 0, a22 = 0;
            {
                const uint8_t *row0 = &s_it[(r + 1) * LK_IT_STRIDE + (ipx - tx0) + c0];
                const uint8_t *row1 = row0 + LK_IT_STRIDE;
                const int *d0 = &s_d[r * LK_D_W + c0], *d1 = d0 + LK_D_W;
                int pa = row0[0], pb = row1[0];
                int da = d0[0], db = d1[0];
#pragma unroll
                for (int j = 0; j < 7; j++) {
                    int pa1 = row0[j + 1], pb1 = row1[j + 1];
                    int da1 = d0[j + 1], db1 = d1[j + 1];
                    int ival = descale(pa * iw00 + pa1 * iw01 + pb * iw10 + pb1 * iw11, LK_W_BITS - 5);
                    int ixval = descale((short)da * iw00 + (short)da1 * iw01 + (short)db * iw10 +
                                            (short)db1 * iw11,
                                        LK_W_BITS);
                    int iyval = descale((da >> 16) * iw00 + (da1 >> 16) * iw01 + (db >> 16) * iw10 +
                                            (db1 >> 16) * iw11,
                                        LK_W_BITS);
                    if (!live)
                        ival = ixval = iyval = 0;
                    Ival[j] = ival;
                    Ixv[j] = ixval;
                    Iyv[j] = iyval;
                    a11 += ixval * ixval;
                    a12 += ixval * iyval;
                    a22 += iyval * iyval;
                    pa = pa1;
                    pb = pb1;
                    da = da1;
                    db = db1;
                }
            }
            const float A11 = (float)wave_sum_i64(a11) * FLT_SCALE;
            const float A12 = (float)wave_sum_i64(a12) * FLT_SCALE;
            const float A22 = (float)wave_sum_i64(a22) * FLT_SCALE;

            float D = A11 * A22 - A12 * A12;
            const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                                 (float)(2 * LK_WIN * LK_WIN);
            if (minEig < prm.min_eig || D < FLT_EPSILON) {
                if (level == 0)
                    st = 0;
                continue;
            }
            D = 1.f / D;

            nextX -= halfWin;
            nextY -= halfWin;
            float prevDX = 0.f, prevDY = 0.f;
            int jx0 = 0, jy0 = 0;
            bool have_tile = false;

            for (int j = 0; j < prm.max_count; j++) {
                const int inx = uni((int)floorf(nextX)), iny = uni((int)floorf(nextY));
                if (inx < -LK_WIN || inx >= jw || iny < -LK_WIN || iny >= jh) {
                    if (level == 0)
                        st = 0;
                    break;
                }
                // search tile must cover cols inx..inx+21, rows iny..iny+21
                if (!have_tile || inx < jx0 || inx + LK_WIN + 1 > jx0 + LK_JT_W || iny < jy0 ||
                    iny + LK_WIN + 1 > jy0 + LK_JT_H) {
                    jx0 = (inx - 12) & ~3;
                    jy0 = iny - 9;
                    __syncthreads();
                    load_tile<LK_JT_W, LK_JT_H, LK_JT_STRIDE>(s_jt, Jimg, jstride, jw, jh, jx0, jy0, lane);
                    __syncthreads();
                    have_tile = true;
                }
                a = nextX - inx;
                b = nextY - iny;
                iw00 = uni(__float2int_rn((1.f - a) * (1.f - b) * (1 << LK_W_BITS)));
                iw01 = uni(__float2int_rn(a * (1.f - b) * (1 << LK_W_BITS)));
                iw10 = uni(__float2int_rn((1.f - a) * b * (1 << LK_W_BITS)));
                iw11 = (1 << LK_W_BITS) - iw00 - iw01 - iw10;

                int b1 = 0, b2 = 0;
                {
                    const uint8_t *row0 = &s_jt[(iny - jy0 + r) * LK_JT_STRIDE + (inx - jx0) + c0];
                    const uint8_t *row1 = row0 + LK_JT_STRIDE;
                    int pa = row0[0], pb = row1[0];
#pragma unroll
                    for (int k = 0; k < 7; k++) {
                        int pa1 = row0[k + 1], pb1 = row1[k + 1];
                        int diff = descale(pa * iw00 + pa1 * iw01 + pb * iw10 + pb1 * iw11, LK_W_BITS - 5) -
                                   Ival[k];
                        b1 += diff * Ixv[k];
                        b2 += diff * Iyv[k];
                        pa = pa1;
                        pb = pb1;
                    }
                }
                const float fb1 = (float)wave_sum_i64(b1) * FLT_SCALE;
                const float fb2 = (float)wave_sum_i64(b2) * FLT_SCALE;
                const float dx = (A12 * fb2 - A22 * fb1) * D;
                const float dy = (A12 * fb1 - A11 * fb2) * D;
                nextX += dx;
                nextY += dy;
                outX = nextX + halfWin;
                outY = nextY + halfWin;
                if ((double)dx * dx + (double)dy * dy <= prm.epsilon)
                    break;
                if (j > 0 && fabs((double)(dx + prevDX)) < 0.01 && fabs((double)(dy + prevDY)) < 0.01) {
                    outX -= dx * 0.5f;
                    outY -= dy * 0.5f;
                    break;
                }
                prevDX = dx;
                prevDY = dy;
            }

            // final in-bounds check OpenCV performs at level 0 when an err vector is requested
            if (st && level == 0) {
                const int fx = (int)floorf(outX - halfWin), fy = (int)floorf(outY - halfWin);
                if (fx < -LK_WIN || fx >= jw || fy < -LK_WIN || fy >= jh)
                    st = 0;
            }
        }

        if (lane == 0) {
            trk[((size_t)frame * 4 + hop) * cap + f] = make_float2(outX, outY);
            status[((size_t)frame * 4 + hop) * cap + f] = (uint8_t)st;
        }
        prevPtX = unif(outX);
        prevPtY = unif(outY);
    }
}

void launch_lk_circular(const PyrImage *d_imgs, const Quad *d_quads, const float2 *d_pts, const int *d_npts,
                        int cap, int max_pts, int n_frames, float2 *d_trk, uint8_t *d_status,
                        const LkParams &prm, hipStream_t stream)
{
    if (max_pts <= 0 || n_frames <= 0)
        return;
    dim3 grid(max_pts, n_frames);
    hipLaunchKernelGGL(lk_circular_kernel, grid, dim3(64), 0, stream, d_imgs, d_quads, d_pts, d_npts, cap,
                       d_trk, d_status, prm);
}

} // namespace vo
