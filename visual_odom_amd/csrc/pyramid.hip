// pyramid.hip -- Gaussian image pyramid (one level per launch, batched over the image table).
//
// Replaces the buildOpticalFlowPyramid -> cv::pyrDown chain that every cv::calcOpticalFlowPyrLK
// call of the reference runs internally (feature.cpp:136-139; 8 pyramid builds per frame there,
// 4 here -- each image once).  Integer arithmetic, bit-exact by construction:
//   horizontal [1 4 6 4 1] in int, vertical [1 4 6 4 1], (v + 128) >> 8, REFLECT_101 borders.
//
// Mapping: one 256-thread workgroup produces a 64 x 16 output tile.  The 136 x 35 source tile is
// staged in LDS with aligned dword loads (byte loads + index reflection only for tiles touching
// the image border), the horizontal pass writes u16 partial rows back to LDS, the vertical pass
// emits 4 adjacent pixels per thread as one 32-bit store.  HBM-bound: reads S_l, writes S_l/4.
#include "vo_kernels.h"

namespace vo {

constexpr int PD_TW = 64, PD_TH = 16;           // output tile
constexpr int PD_SW = 136, PD_SH = 2 * PD_TH + 3; // source tile (bytes x rows), x origin = 2*ox-4
constexpr int PD_SSTRIDE = 140;                 // LDS row stride of the source tile (bytes)

__global__ __launch_bounds__(256) void pyr_down_kernel(const PyrImage *__restrict__ imgs, int level)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_src[PD_SH * PD_SSTRIDE];
    __shared__ uint16_t s_h[PD_SH * PD_TW];

    const PyrImage &im = imgs[blockIdx.z];
    const int sw = im.w[level], sh = im.h[level], sstride = im.stride[level];
    const int dw = im.w[level + 1], dh = im.h[level + 1], dstride = im.stride[level + 1];
    const uint8_t *__restrict__ src = im.lvl[level];
    uint8_t *__restrict__ dst = im.lvl[level + 1];
    const int ox = blockIdx.x * PD_TW, oy = blockIdx.y * PD_TH;
    if (ox >= dw || oy >= dh)
        return;
    const int tid = threadIdx.x;
    const int sx0 = 2 * ox - 4, sy0 = 2 * oy - 2;

    const bool interior = sx0 >= 0 && sx0 + PD_SW <= sw && sy0 >= 0 && sy0 + PD_SH <= sh;
    if (interior) {
        // 35 rows x 34 dwords, coalesced along rows
        for (int i = tid; i < PD_SH * (PD_SW / 4); i += 256) {
            int r = i / (PD_SW / 4), c = i - r * (PD_SW / 4);
            uint32_t v = *reinterpret_cast<const uint32_t *>(src + (size_t)(sy0 + r) * sstride + sx0 + 4 * c);
            *reinterpret_cast<uint32_t *>(&s_src[r * PD_SSTRIDE + 4 * c]) = v;
        }
    } else {
        for (int i = tid; i < PD_SH * PD_SW; i += 256) {
            int r = i / PD_SW, c = i - r * PD_SW;
            int y = reflect101(sy0 + r, sh), x = reflect101(sx0 + c, sw);
            s_src[r * PD_SSTRIDE + c] = src[(size_t)y * sstride + x];
        }
    }
    __syncthreads();

    // horizontal 5-tap; output column x reads source columns 2x-2 .. 2x+2 = tile columns 2x+2 .. 2x+6
    for (int i = tid; i < PD_SH * PD_TW; i += 256) {
        int r = i / PD_TW, x = i - r * PD_TW;
        const uint8_t *p = &s_src[r * PD_SSTRIDE + 2 * x + 2];
        s_h[i] = (uint16_t)(p[2] * 6 + (p[1] + p[3]) * 4 + p[0] + p[4]);
    }
    __syncthreads();

    // vertical 5-tap; thread -> (row y, 4 adjacent columns)
    const int y = tid >> 4, x4 = (tid & 15) * 4;
    if (oy + y < dh && ox + x4 < dstride) {
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint16_t *q = &s_h[(2 * y) * PD_TW + x4 + k];
            int v = q[2 * PD_TW] * 6 + (q[PD_TW] + q[3 * PD_TW]) * 4 + q[0] + q[4 * PD_TW];
            packed |= (uint32_t)((v + 128) >> 8) << (8 * k);
        }
        *reinterpret_cast<uint32_t *>(dst + (size_t)(oy + y) * dstride + ox + x4) = packed;
    }
}

void launch_pyr_down(const PyrImage *d_imgs, int n_images, int level, int dw, int dh, hipStream_t stream)
{
    dim3 grid((dw + PD_TW - 1) / PD_TW, (dh + PD_TH - 1) / PD_TH, n_images);
    hipLaunchKernelGGL(pyr_down_kernel, grid, dim3(256), 0, stream, d_imgs, level);
}

} // namespace vo
