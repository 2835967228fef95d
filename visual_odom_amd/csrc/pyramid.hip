// pyramid.hip -- bordered Gaussian image pyramid + Scharr images, batched over the image table.
//
// Replaces what every cv::calcOpticalFlowPyrLK call of the reference rebuilds internally
// (feature.cpp:136-139): buildOpticalFlowPyramid (cv::pyrDown chain + winSize REFLECT_101 border
// around every level) for both images and calcSharrDeriv of the previous image's levels (zero
// border).  The reference does 8 pyramid builds and 16 Scharr passes per frame; here every image is
// processed once per batch run.  Integer arithmetic, bit-exact by construction:
//   pyrDown: horizontal [1 4 6 4 1] in int, vertical [1 4 6 4 1], (v + 128) >> 8, REFLECT_101 on the
//   level itself; Scharr: 3x3 unnormalised, REFLECT_101 (= reading the level's own border).
//
// Kernels (all HBM-streaming; see DESIGN.md for the byte counts):
//   border_fill_kernel  writes the REFLECT_101 border of one level from its interior (workgroup per row)
//   pyr_down_kernel     one 256-thread workgroup -> 64 x 16 output tile; the 136 x 35 source tile is
//                       staged in LDS with aligned dword loads (the source border makes every tile an
//                       in-bounds read), u16 horizontal partials in LDS, 4 pixels per 32-bit store
//   scharr_kernel       4 pixels per thread: three unaligned 8-byte row loads, one 16-byte store of
//                       (4*Ix | 4*Iy << 16) x 4
#include "vo_kernels.h"
#include "vo_lkmath.h"

namespace vo {

struct __attribute__((packed, aligned(1))) U8x8 {
    uint32_t lo, hi;
};

// ---------------------------------------------------------------------------------------------------
// one 64-thread workgroup per bordered row: rows above / below the image copy a whole reflected row,
// image rows only write their VO_BX left and (stride - VO_BX - w) right border pixels
__global__ __launch_bounds__(64) void border_fill_kernel(const PyrImage *__restrict__ imgs, int level)
{
    const PyrImage &im = imgs[blockIdx.y];
    const int w = im.w[level], h = im.h[level], stride = im.stride[level];
    VO_GLOBAL uint8_t *__restrict__ p = (VO_GLOBAL uint8_t *)im.lvl[level];
    const int y = (int)blockIdx.x - VO_BY; // -VO_BY .. h + VO_BY - 1
    const VO_GLOBAL uint8_t *__restrict__ src = p + (ptrdiff_t)reflect101(y, h) * stride;
    VO_GLOBAL uint8_t *__restrict__ dst = p + (ptrdiff_t)y * stride;
    const int right = stride - VO_BX - w; // >= VO_BY
    if (y >= 0 && y < h) {
        for (int i = threadIdx.x; i < VO_BX + right; i += 64) {
            const int x = i < VO_BX ? i - VO_BX : w + (i - VO_BX);
            dst[x] = src[reflect101(x, w)];
        }
    } else {
        for (int x = (int)threadIdx.x - VO_BX; x < stride - VO_BX; x += 64)
            dst[x] = src[reflect101(x, w)];
    }
}

// ---------------------------------------------------------------------------------------------------
constexpr int PD_TW = 64, PD_TH = 16;             // output tile
constexpr int PD_SW = 136, PD_SH = 2 * PD_TH + 3; // source tile (bytes x rows), x origin = 2*ox-4
constexpr int PD_SSTRIDE = 140;                   // LDS row stride of the source tile (bytes)

__global__ __launch_bounds__(256) void pyr_down_kernel(const PyrImage *__restrict__ imgs, int level)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_src[PD_SH * PD_SSTRIDE];
    __shared__ uint16_t s_h[PD_SH * PD_TW];

    const PyrImage &im = imgs[blockIdx.z];
    const int sh = im.h[level], sstride = im.stride[level];
    const int dw = im.w[level + 1], dh = im.h[level + 1], dstride = im.stride[level + 1];
    const VO_GLOBAL uint8_t *__restrict__ src = (const VO_GLOBAL uint8_t *)im.lvl[level];
    VO_GLOBAL uint8_t *__restrict__ dst = (VO_GLOBAL uint8_t *)im.lvl[level + 1];
    const int ox = blockIdx.x * PD_TW, oy = blockIdx.y * PD_TH;
    if (ox >= dw || oy >= dh)
        return;
    const int tid = threadIdx.x;
    const int sx0 = 2 * ox - 4, sy0 = 2 * oy - 2;   // >= -4 / -2: inside the source border
    const int xmax = sstride - VO_BX, ymax = sh + VO_BY; // first column / row outside the allocation

    // 35 rows x 34 dwords, coalesced along rows; columns / rows past the allocation are never used
    // by a valid output pixel (those need source x <= 2 dw <= w + 1, y <= h + 1) and read as 0
    for (int i = tid; i < PD_SH * (PD_SW / 4); i += 256) {
        const int r = i / (PD_SW / 4), c = i - r * (PD_SW / 4);
        const int x = sx0 + 4 * c, y = sy0 + r;
        uint32_t v = 0;
        if (x + 4 <= xmax && y < ymax)
            v = *(const VO_GLOBAL uint32_t *)(src + (ptrdiff_t)y * sstride + x);
        *reinterpret_cast<uint32_t *>(&s_src[r * PD_SSTRIDE + 4 * c]) = v;
    }
    __syncthreads();

    // horizontal 5-tap; output column x reads source columns 2x-2 .. 2x+2 = tile columns 2x+2 .. 2x+6
    for (int i = tid; i < PD_SH * PD_TW; i += 256) {
        const int r = i / PD_TW, x = i - r * PD_TW;
        const uint8_t *p = &s_src[r * PD_SSTRIDE + 2 * x + 2];
        s_h[i] = (uint16_t)(p[2] * 6 + (p[1] + p[3]) * 4 + p[0] + p[4]);
    }
    __syncthreads();

    // vertical 5-tap; thread -> (row y, 4 adjacent columns); columns >= dw land in the right border
    // (stride - VO_BX - dw >= VO_BY there) and are overwritten by border_fill_kernel afterwards
    const int y = tid >> 4, x4 = (tid & 15) * 4;
    if (oy + y < dh && ox + x4 < dw) {
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint16_t *q = &s_h[(2 * y) * PD_TW + x4 + k];
            int v = q[2 * PD_TW] * 6 + (q[PD_TW] + q[3 * PD_TW]) * 4 + q[0] + q[4 * PD_TW];
            packed |= (uint32_t)((v + 128) >> 8) << (8 * k);
        }
        *(VO_GLOBAL uint32_t *)(dst + (ptrdiff_t)(oy + y) * dstride + ox + x4) = packed;
    }
}

// ---------------------------------------------------------------------------------------------------
// all levels of all images in one launch: blockIdx.y = image, blockIdx.x = tile of 256 x 4 pixels numbered
// level by level (every image of the table has the same geometry, so the per-level tile counts are launch
// constants: no workgroup is launched for tiles a smaller level does not have)
struct ScharrTiles {
    int first[VO_MAX_LEVELS + 1]; // first[l] = tiles of the levels before l
    int tiles_x[VO_MAX_LEVELS];
};

__global__ __launch_bounds__(256) void scharr_kernel(const PyrImage *__restrict__ imgs, int n_levels, ScharrTiles st)
{
    int level = 0;
    while (level + 1 < n_levels && (int)blockIdx.x >= st.first[level + 1])
        level++;
    const int tile = (int)blockIdx.x - st.first[level];
    const int ty = tile / st.tiles_x[level], tx = tile - ty * st.tiles_x[level];
    const PyrImage &im = imgs[blockIdx.y];
    const int w = im.w[level], h = im.h[level], stride = im.stride[level];
    const int x4 = (int)(tx * 64 + (threadIdx.x & 63)) * 4;
    const int y = (int)(ty * 4 + (threadIdx.x >> 6));
    if (x4 >= w || y >= h)
        return;
    const VO_GLOBAL uint8_t *__restrict__ p = (const VO_GLOBAL uint8_t *)im.lvl[level] + (ptrdiff_t)y * stride + x4 - 1; // pixel (x4-1, y)
    const U8x8 a = *(const VO_GLOBAL U8x8 *)(p - stride);
    const U8x8 b = *(const VO_GLOBAL U8x8 *)(p);
    const U8x8 c = *(const VO_GLOBAL U8x8 *)(p + stride);
    const uint64_t ra = ((uint64_t)a.hi << 32) | a.lo, rb = ((uint64_t)b.hi << 32) | b.lo,
                   rc = ((uint64_t)c.hi << 32) | c.lo;
    uint32_t out[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int s = 8 * k;
        out[k] = scharr4_packed((int)((ra >> s) & 0xff), (int)((ra >> (s + 8)) & 0xff), (int)((ra >> (s + 16)) & 0xff),
                                (int)((rb >> s) & 0xff), (int)((rb >> (s + 16)) & 0xff), (int)((rc >> s) & 0xff),
                                (int)((rc >> (s + 8)) & 0xff), (int)((rc >> (s + 16)) & 0xff));
    }
    // pixels >= w of the last quad fall into the (zero) right border: keep them zero
#pragma unroll
    for (int k = 1; k < 4; k++)
        if (x4 + k >= w)
            out[k] = 0;
    *(VO_GLOBAL uint4 *)((VO_GLOBAL uint32_t *)im.der[level] + (ptrdiff_t)y * stride + x4) = make_uint4(out[0], out[1], out[2], out[3]);
}

#ifndef VO_HOST_EMUL
void launch_border_fill(const PyrImage *d_imgs, int n_images, int level, int stride, int h, hipStream_t stream)
{
    (void)stride;
    dim3 grid(h + 2 * VO_BY, n_images);
    hipLaunchKernelGGL(border_fill_kernel, grid, dim3(64), 0, stream, d_imgs, level);
}

void launch_pyr_down(const PyrImage *d_imgs, int n_images, int level, int dw, int dh, hipStream_t stream)
{
    dim3 grid((dw + PD_TW - 1) / PD_TW, (dh + PD_TH - 1) / PD_TH, n_images);
    hipLaunchKernelGGL(pyr_down_kernel, grid, dim3(256), 0, stream, d_imgs, level);
}

void launch_scharr(const PyrImage *d_imgs, int n_images, int n_levels, const int *lw, const int *lh,
                   hipStream_t stream)
{
    ScharrTiles st = {};
    for (int l = 0; l < n_levels; l++) {
        st.tiles_x[l] = (lw[l] + 255) / 256;
        st.first[l + 1] = st.first[l] + st.tiles_x[l] * ((lh[l] + 3) / 4);
    }
    dim3 grid(st.first[n_levels], n_images);
    hipLaunchKernelGGL(scharr_kernel, grid, dim3(256), 0, stream, d_imgs, n_levels, st);
}

#endif // VO_HOST_EMUL

} // namespace vo
