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
// Kernels (all HBM-streaming; see DESIGN.md for the byte counts).  Round 2 rewrote all three -- the first versions were
// instruction-bound far below the memory rate (byte-granular loads / stores, one LDS byte read per filter tap, 64-bit
// shifts to pull pixels apart): 512 KITTI images took 0.11-0.17 + 0.20 + 0.60 ms, against 0.02 / 0.07 / 0.35 ms of HBM time:
//   pyr_down_kernel     one 256-thread workgroup -> 64 x 16 output tile; the 144 x 35 source tile is staged in LDS with
//                       16-byte loads (rows through REFLECT_101, the two reflected columns an edge tile needs patched in
//                       LDS: a level's border is never read here, so the three launches only depend on each other);
//                       horizontal [1 4 6 4 1]:
//                       a thread reads 16 bytes and forms 4 partials with v_alignbyte_b32 + v_dot4_u32_u8 (u16 in LDS);
//                       vertical: packed 16-bit multiply-adds (the sum + 128 stays below 2^16), 4 pixels per 32-bit store
//   border_fill_kernel  a range of levels in one launch: one thread per 16-byte chunk that holds a REFLECT_101 border
//                       pixel (all chunks of the rows above / below the image, the left / right border chunks of image
//                       rows): chunks inside the image span are aligned copies of the reflected row, the rest gathers
//                       16 reflected bytes
//   scharr_kernel       8 pixels per thread: three unaligned 12-byte row loads, the pixels lifted into u16 pairs
//                       (v_perm_b32), the separable form t0 = 3 (above + below) + 10 row, t1 = below - above in packed
//                       16-bit arithmetic with the x4 pre-scale folded into the constants, two 16-byte stores of
//                       (4*Ix | 4*Iy << 16) x 4
#include "vo_kernels.h"
#include "vo_lkmath.h"

#include <stdlib.h>

namespace vo {

struct __attribute__((packed, aligned(1))) U8x12 {
    uint32_t a, b, c;
};
struct __attribute__((packed, aligned(4))) U32x4 {
    uint32_t a, b, c, d;
};
struct __attribute__((packed, aligned(1))) U8x16 { // 16 bytes at any address
    uint32_t a, b, c, d;
};

#if defined(VO_DEV_VARIANTS) || defined(VO_HOST_EMUL)
// ROUND-3 CHAIN (developer build and CPU emulator only: the A/B partner of the fused passes below, VO_PYR_FUSED=0):
// border_fill_kernel -> scharr -> pyr_down x (L - 1) -> border_fill_kernel -> scharr, eight launches per pyramid build.
// ---------------------------------------------------------------------------------------------------
// Border words of one level.  Work items: first the 2 * VO_BY rows above / below the image (stride / 4 words each), then,
// per image row, the VO_BX / 4 words left of the image and the words from the one holding pixel w - 1 (or starting at w)
// to the end of the row.  A word that straddles the image edge rewrites its interior bytes with the values they already have.
constexpr int BF_MAX_ROW_CHUNKS = VO_BX / 16 + 4; // right border < 40 pixels + up to 15 interior ones (level_stride, capi.hip)

// One launch covers a range of levels of all images (nothing on the path reads a level's border before the whole pyramid
// exists -- pyr_down_kernel reflects on its own): blockIdx.y = image, blockIdx.x = 256-chunk block numbered level by level.
// Work items are 16-byte chunks (rows start 16-byte aligned, the stride is a multiple of 16): first all chunks of the
// 2 * VO_BY rows above / below the image, then, per image row, the VO_BX / 16 chunks left of the image and the chunks from
// the one holding pixel w - 1 (or starting at w) to the end of the row.  A chunk that lies inside the image span is an
// aligned 16-byte copy of the reflected row, any other gathers its 16 reflected bytes (a chunk that straddles the image
// edge rewrites its interior bytes with the values they already have).  (One 32-bit word per thread, the first round-2
// version, was bound by the latency of its one load: 0.11 ms for level 0 of 512 KITTI images.)
struct BorderBlocks {
    int first[VO_MAX_LEVELS + 1]; // first[l] = blocks of the levels before l
};

inline BorderBlocks border_blocks(int first_level, int n_levels, const int *lstride, const int *lh)
{
    BorderBlocks bb = {}; // levels below first_level get no blocks
    for (int l = first_level; l < n_levels; l++)
        bb.first[l + 1] = bb.first[l] + (2 * VO_BY * (lstride[l] / 16) + lh[l] * BF_MAX_ROW_CHUNKS + 255) / 256;
    return bb;
}

__global__ __launch_bounds__(256) void border_fill_kernel(const PyrImage *__restrict__ imgs, int n_levels, BorderBlocks bb)
{
    int level = 0; // bb.first[l + 1] == bb.first[l] for levels that are not part of this launch
    while (level + 1 < n_levels && (int)blockIdx.x >= bb.first[level + 1])
        level++;
    const PyrImage &im = imgs[blockIdx.y];
    const int w = im.w[level], h = im.h[level], stride = im.stride[level];
    VO_GLOBAL uint8_t *__restrict__ p = (VO_GLOBAL uint8_t *)im.lvl[level];
    const int cpr = stride >> 4;                                 // chunks per bordered row
    const int xr0 = w & ~15;                                     // first chunk with a right-border pixel
    const int nb = VO_BX / 16 + ((stride - VO_BX - xr0) >> 4);   // border chunks of an image row
    const int n_out = 2 * VO_BY * cpr;
    int item = (int)(((int)blockIdx.x - bb.first[level]) * 256 + threadIdx.x);
    int y, x0;
    if (item < n_out) {
        const int r = item / cpr;
        y = r < VO_BY ? r - VO_BY : h + (r - VO_BY);
        x0 = 16 * (item - r * cpr) - VO_BX;
    } else {
        item -= n_out;
        const int r = item / nb, k = item - r * nb;
        if (r >= h)
            return;
        y = r;
        x0 = k < VO_BX / 16 ? 16 * k - VO_BX : xr0 + 16 * (k - VO_BX / 16);
    }
    const VO_GLOBAL uint8_t *__restrict__ src = p + (ptrdiff_t)reflect101(y, h) * stride;
    uint32_t v[4];
    if (x0 >= 0 && x0 + 15 < w) {
        const U32x4 t = *(const VO_GLOBAL U32x4 *)(src + x0);
        v[0] = t.a;
        v[1] = t.b;
        v[2] = t.c;
        v[3] = t.d;
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int x = x0 + 4 * q;
            v[q] = (uint32_t)src[reflect101(x, w)] | (uint32_t)src[reflect101(x + 1, w)] << 8 |
                   (uint32_t)src[reflect101(x + 2, w)] << 16 | (uint32_t)src[reflect101(x + 3, w)] << 24;
        }
    }
    *(VO_GLOBAL uint4 *)(p + (ptrdiff_t)y * stride + x0) = make_uint4(v[0], v[1], v[2], v[3]);
}

// ---------------------------------------------------------------------------------------------------
// pyr_down WITHOUT LDS (round 3).  The tile kernel below keeps 9.5 KB of LDS per workgroup -- and the pose chain's
// epnp_kernel fills the CUs' LDS completely while it runs (two 78 KB workgroups per CU), so next to a pose
// chain its workgroups waited for LDS: 12 us stand-alone, 195 us on average and up to 0.7 ms in the benchmark's kernel
// trace (profiles/r03.md), three launches per step.  Here a thread owns 4 adjacent output columns and walks DOWN the image:
// per source row one 16-byte load, the horizontal [1 4 6 4 1] of its 4 outputs (the same v_alignbyte + v_dot4 forms, packed
// two per register), a 5-row window of those in registers, and every second row the vertical filter + one 4-byte store.
// No LDS, no barrier; neighbouring threads' loads overlap in the L1.  REFLECT_101 rows by index, the (at most two) columns
// beyond the image edge by a per-byte gather in the two threads of a row that need them.  Bit-identical by construction
// (same integer arithmetic, same order) and by the emulator / GPU pyramid tests.
// Measured against the tile kernel (developer build, VO_PYR_LDS=1; gpurun_out/r3_14, pyramid stage ms | frames/s), 4 output
// rows per thread: 256-frame batch at 340 points 1.26 -> 1.02 | 70.4 k -> 73.3 k, lock-step loop with 256 sequences
// 1.56 -> 1.22 | 62.4 k -> 64.6 k, headline batch 1.33 -> 1.08 | 19.70 k -> 19.78 k; alone (no pose chain beside it) the tile
// kernel is the faster one: `--stages lk` 0.70 -> 0.76, 1080p 1.60 -> 1.62.  8 / 16 / 32 rows per thread: 1.03 / 1.08 / 1.21 ms
// at 340 points -- more threads beat fewer redundant rows.
#ifndef VO_PN_ROWS
#define VO_PN_ROWS 4
#endif
constexpr int PN_ROWS = VO_PN_ROWS;        // output rows per thread
constexpr int PN_TW = 64, PN_TH = 16 * PN_ROWS; // output tile of a 256-thread workgroup: 16 x 16 threads

template <bool EDGE>
__device__ __forceinline__ uint2 pyr_hrow(const VO_GLOBAL uint8_t *__restrict__ row, int c0, int sw)
{
    uint32_t w0, w1, w2, w3;
    if (!EDGE) {
        const U32x4 v = *(const VO_GLOBAL U32x4 *)(row + c0);
        w0 = v.a;
        w1 = v.b;
        w2 = v.c;
        w3 = v.d;
    } else { // source columns c0 + 2 .. c0 + 12 through REFLECT_101 (the level's border is not read as data)
        uint32_t b[16];
#pragma unroll
        for (int k = 0; k < 16; k++)
            b[k] = (k >= 2 && k <= 12) ? (uint32_t)row[reflect101(c0 + k, sw)] : 0u;
        w0 = b[0] | b[1] << 8 | b[2] << 16 | b[3] << 24;
        w1 = b[4] | b[5] << 8 | b[6] << 16 | b[7] << 24;
        w2 = b[8] | b[9] << 8 | b[10] << 16 | b[11] << 24;
        w3 = b[12] | b[13] << 8 | b[14] << 16 | b[15] << 24;
    }
    // outputs x4 .. x4 + 3 read bytes 2 + 2k .. 6 + 2k of the 16 bytes at source column 2 x4 - 4
    const uint32_t taps = 0x04060401u; // weights of bytes 0..3 of the aligned group; the fifth tap is the next byte
    const uint32_t h0 = udot4(alignbyte(w1, w0, 2), taps, udot4(w1, 0x00010000u, 0));
    const uint32_t h1 = udot4(w1, taps, udot4(w2, 0x00000001u, 0));
    const uint32_t h2 = udot4(alignbyte(w2, w1, 2), taps, udot4(w2, 0x00010000u, 0));
    const uint32_t h3 = udot4(w2, taps, udot4(w3, 0x00000001u, 0));
    return make_uint2(h0 | h1 << 16, h2 | h3 << 16);
}

// one thread's column: source rows 2 y0 - 2 .. 2 (y0 + PN_ROWS - 1) + 2; output row y0 + j is complete after row 2 j + 4
template <bool EDGE>
__device__ __forceinline__ void pyr_column(const VO_GLOBAL uint8_t *__restrict__ src, VO_GLOBAL uint8_t *__restrict__ dst,
                                           int sw, int sh, int sstride, int dh, int dstride, int x4, int y0)
{
    const int c0 = 2 * x4 - 4; // source column of byte 0 of the 16-byte window
    uint2 q0, q1, q2, q3, q4;
    q0 = q1 = q2 = q3 = q4 = make_uint2(0, 0);
    constexpr int UNROLL = EDGE ? 1 : 2 * PN_ROWS + 3; // the rare edge columns keep the loop (and its per-byte gathers) rolled
#pragma unroll UNROLL
    for (int r = 0; r < 2 * PN_ROWS + 3; r++) {
        if (r >= 5 && y0 + (r - 3) / 2 >= dh) // no further output row of this thread exists
            break;
        const int sy = reflect101(2 * y0 - 2 + r, sh);
        q0 = q1;
        q1 = q2;
        q2 = q3;
        q3 = q4;
        q4 = pyr_hrow<EDGE>(src + (ptrdiff_t)sy * sstride, c0, sw);
        const int j = (r - 4) / 2; // output row this source row completes (r even, r >= 4)
        if (r >= 4 && (r & 1) == 0 && y0 + j < dh) {
            // vertical 5-tap on two packed u16 pairs: 6 q2 + 4 (q1 + q3) + q0 + q4 + 128 <= 65408 fits 16 bits, the result
            // is its high byte.  Columns >= dw land in the right border (stride - VO_BX - dw >= VO_BY there) and are
            // overwritten by border_fill_kernel afterwards
            const uint32_t va = pk_mad_u16(q2.x, 6, pk_mad_u16(pk_add_u16(q1.x, q3.x), 4, pk_add_u16(pk_add_u16(q0.x, q4.x), 0x00800080u)));
            const uint32_t vb = pk_mad_u16(q2.y, 6, pk_mad_u16(pk_add_u16(q1.y, q3.y), 4, pk_add_u16(pk_add_u16(q0.y, q4.y), 0x00800080u)));
            *(VO_GLOBAL uint32_t *)(dst + (ptrdiff_t)(y0 + j) * dstride + x4) = perm_b32(vb, va, 0x07050301u);
        }
    }
}

__global__ __launch_bounds__(256) void pyr_down_kernel(const PyrImage *__restrict__ imgs, int level)
{
    const PyrImage &im = imgs[blockIdx.z];
    const int sw = im.w[level], sh = im.h[level], sstride = im.stride[level];
    const int dw = im.w[level + 1], dh = im.h[level + 1], dstride = im.stride[level + 1];
    const VO_GLOBAL uint8_t *__restrict__ src = (const VO_GLOBAL uint8_t *)im.lvl[level];
    VO_GLOBAL uint8_t *__restrict__ dst = (VO_GLOBAL uint8_t *)im.lvl[level + 1];
    const int tid = threadIdx.x;
    const int x4 = blockIdx.x * PN_TW + (tid & 15) * 4;              // first of this thread's 4 output columns
    const int y0 = blockIdx.y * PN_TH + (tid >> 4) * PN_ROWS;        // first of its output rows
    if (x4 >= dw || y0 >= dh)
        return;
    // a needed source column (2 x4 - 2 .. 2 x4 + 8) lies outside the image: the first thread of a row, and the last one or two
    if (2 * x4 - 2 < 0 || 2 * x4 + 8 >= sw)
        pyr_column<true>(src, dst, sw, sh, sstride, dh, dstride, x4, y0);
    else
        pyr_column<false>(src, dst, sw, sh, sstride, dh, dstride, x4, y0);
}

// Round 2's LDS tile kernel: still the one for SMALL launches (a single frame, a few sequences), where no pose chain of any
// size runs beside it and latency is what counts -- 4 images: 3 x 4 us against 20 + 16 + 12 us for the column walk above
// (profiles/r03_track_frame_timeline.txt vs gpurun_out/r3_15).
// ---------------------------------------------------------------------------------------------------
constexpr int PD_TW = 64, PD_TH = 16;             // output tile
constexpr int PD_SW = 144, PD_SH = 2 * PD_TH + 3; // source tile (bytes x rows), x origin = 2*ox-4; LDS row stride = PD_SW

__global__ __launch_bounds__(256) void pyr_down_lds_kernel(const PyrImage *__restrict__ imgs, int level)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_src[PD_SH * PD_SW];
    __shared__ __attribute__((aligned(16))) uint16_t s_h[PD_SH * PD_TW];

    const PyrImage &im = imgs[blockIdx.z];
    const int sw = im.w[level], sh = im.h[level], sstride = im.stride[level];
    const int dw = im.w[level + 1], dh = im.h[level + 1], dstride = im.stride[level + 1];
    const VO_GLOBAL uint8_t *__restrict__ src = (const VO_GLOBAL uint8_t *)im.lvl[level];
    VO_GLOBAL uint8_t *__restrict__ dst = (VO_GLOBAL uint8_t *)im.lvl[level + 1];
    const int ox = blockIdx.x * PD_TW, oy = blockIdx.y * PD_TH;
    if (ox >= dw || oy >= dh)
        return;
    const int tid = threadIdx.x;
    const int sx0 = 2 * ox - 4, sy0 = 2 * oy - 2;        // >= -4 / -2
    const int xmax = sstride - VO_BX;                    // first column outside the allocation

    // 35 rows x 9 x 16 bytes, coalesced along rows.  The source level's border is NOT read as data (it may not exist yet:
    // the borders of all levels are filled in one pass after the last pyr_down): rows outside the image are fetched from
    // their REFLECT_101 row, the up to two columns a valid output needs left / right of the image are patched in LDS below.
    // Bytes of border columns that do get loaded are unspecified and never used; columns past the allocation read as 0.
    for (int i = tid; i < PD_SH * (PD_SW / 16); i += 256) {
        const int r = i / (PD_SW / 16), c = i - r * (PD_SW / 16);
        const int x = sx0 + 16 * c, y = reflect101(sy0 + r, sh);
        const VO_GLOBAL uint8_t *g = src + (ptrdiff_t)y * sstride + x;
        U32x4 v = {0, 0, 0, 0};
        if (x + 16 <= xmax) {
            v = *(const VO_GLOBAL U32x4 *)g;
        } else {
            if (x + 4 <= xmax)
                v.a = *(const VO_GLOBAL uint32_t *)g;
            if (x + 8 <= xmax)
                v.b = *(const VO_GLOBAL uint32_t *)(g + 4);
            if (x + 12 <= xmax)
                v.c = *(const VO_GLOBAL uint32_t *)(g + 8);
        }
        *reinterpret_cast<uint4 *>(&s_src[r * PD_SW + 16 * c]) = make_uint4(v.a, v.b, v.c, v.d);
    }
    __syncthreads();
    // REFLECT_101 columns: outputs read source columns 2x - 2 .. 2x + 2 with x < dw = (sw + 1) / 2, i.e. -2 .. sw + 1 at most;
    // tile column = source column + 4 - 2 ox.  (sw - 2, sw - 3 lie inside the tile whenever sw or sw + 1 is needed: the
    // last tile has 2 ox <= sw - 1.)
    const bool left = ox == 0, right = sw + 4 - 2 * ox < PD_SW;
    if (left || right) {
        if (tid < PD_SH) {
            uint8_t *row = &s_src[tid * PD_SW];
            if (left) {
                row[2] = row[4 + reflect101(-2, sw)];
                row[3] = row[4 + reflect101(-1, sw)];
            }
            if (right) {
                const int c = sw + 4 - 2 * ox; // tile column of source column sw
                row[c] = row[c - sw + reflect101(sw, sw)];
                if (c + 1 < PD_SW)
                    row[c + 1] = row[c - sw + reflect101(sw + 1, sw)];
            }
        }
        __syncthreads();
    }

    // horizontal 5-tap, 4 outputs per thread: output column x reads source columns 2x-2 .. 2x+2 = tile columns 2x+2 .. 2x+6,
    // i.e. outputs x4 .. x4+3 read bytes 2+2k .. 6+2k (k = 0..3) of the 16 bytes at tile column 2*x4
    for (int i = tid; i < PD_SH * (PD_TW / 4); i += 256) {
        const int r = i / (PD_TW / 4), q = i - r * (PD_TW / 4);
        const uint2 lo = *reinterpret_cast<const uint2 *>(&s_src[r * PD_SW + 8 * q]);
        const uint2 hi = *reinterpret_cast<const uint2 *>(&s_src[r * PD_SW + 8 * q + 8]);
        const uint32_t w0 = lo.x, w1 = lo.y, w2 = hi.x, w3 = hi.y;
        const uint32_t taps = 0x04060401u; // weights of bytes 0..3 of the aligned group; the fifth tap is the next byte
        const uint32_t h0 = udot4(alignbyte(w1, w0, 2), taps, udot4(w1, 0x00010000u, 0));
        const uint32_t h1 = udot4(w1, taps, udot4(w2, 0x00000001u, 0));
        const uint32_t h2 = udot4(alignbyte(w2, w1, 2), taps, udot4(w2, 0x00010000u, 0));
        const uint32_t h3 = udot4(w2, taps, udot4(w3, 0x00000001u, 0));
        *reinterpret_cast<uint2 *>(&s_h[r * PD_TW + 4 * q]) = make_uint2(h0 | h1 << 16, h2 | h3 << 16);
    }
    __syncthreads();

    // vertical 5-tap; thread -> (row y, 4 adjacent columns) as two packed u16 pairs: 6 q2 + 4 (q1 + q3) + q0 + q4 + 128
    // <= 65408 fits 16 bits, the result is its high byte.  Columns >= dw land in the right border (stride - VO_BX - dw
    // >= VO_BY there) and are overwritten by border_fill_kernel afterwards
    const int y = tid >> 4, x4 = (tid & 15) * 4;
    if (oy + y < dh && ox + x4 < dw) {
        const uint16_t *q = &s_h[(2 * y) * PD_TW + x4];
        const uint2 q0 = *reinterpret_cast<const uint2 *>(q), q1 = *reinterpret_cast<const uint2 *>(q + PD_TW),
                    q2 = *reinterpret_cast<const uint2 *>(q + 2 * PD_TW), q3 = *reinterpret_cast<const uint2 *>(q + 3 * PD_TW),
                    q4 = *reinterpret_cast<const uint2 *>(q + 4 * PD_TW);
        const uint32_t va = pk_mad_u16(q2.x, 6, pk_mad_u16(pk_add_u16(q1.x, q3.x), 4, pk_add_u16(pk_add_u16(q0.x, q4.x), 0x00800080u)));
        const uint32_t vb = pk_mad_u16(q2.y, 6, pk_mad_u16(pk_add_u16(q1.y, q3.y), 4, pk_add_u16(pk_add_u16(q0.y, q4.y), 0x00800080u)));
        *(VO_GLOBAL uint32_t *)(dst + (ptrdiff_t)(oy + y) * dstride + ox + x4) = perm_b32(vb, va, 0x07050301u);
    }
}

// ---------------------------------------------------------------------------------------------------
// all levels of all images in one launch: blockIdx.y = image, blockIdx.x = tile of 512 x 4 pixels numbered
// level by level (every image of the table has the same geometry, so the per-level tile counts are launch
// constants: no workgroup is launched for tiles a smaller level does not have)
struct ScharrTiles {
    int first[VO_MAX_LEVELS + 1]; // first[l] = tiles of the levels before l
    int tiles_x[VO_MAX_LEVELS];
};

inline ScharrTiles scharr_tiles(int first_level, int n_levels, const int *lw, const int *lh)
{
    ScharrTiles st = {}; // levels below first_level get no tiles
    for (int l = 0; l < n_levels; l++) {
        st.tiles_x[l] = (lw[l] + 511) / 512;
        st.first[l + 1] = st.first[l] + (l < first_level ? 0 : st.tiles_x[l] * ((lh[l] + 3) / 4));
    }
    return st;
}

template <bool NT>
__device__ __forceinline__ void scharr_body(const PyrImage *__restrict__ imgs, int n_levels, const ScharrTiles &st)
{
    int level = 0;
    while (level + 1 < n_levels && (int)blockIdx.x >= st.first[level + 1])
        level++;
    const int tile = (int)blockIdx.x - st.first[level];
    const int ty = tile / st.tiles_x[level], tx = tile - ty * st.tiles_x[level];
    const PyrImage &im = imgs[blockIdx.y];
    const int w = im.w[level], h = im.h[level], stride = im.stride[level];
    const int x8 = (int)(tx * 64 + (threadIdx.x & 63)) * 8;
    const int y = (int)(ty * 4 + (threadIdx.x >> 6));
    if (x8 >= w || y >= h)
        return;
    // bytes j = 0..9 of the three rows = columns x8 - 1 + j (the reads end at column x8 + 10 <= w + 9: right border)
    const VO_GLOBAL uint8_t *__restrict__ p = (const VO_GLOBAL uint8_t *)im.lvl[level] + (ptrdiff_t)y * stride + x8 - 1;
    const U8x12 ra = *(const VO_GLOBAL U8x12 *)(p - stride);
    const U8x12 rb = *(const VO_GLOBAL U8x12 *)(p);
    const U8x12 rc = *(const VO_GLOBAL U8x12 *)(p + stride);
    // column pairs (j, j + 1), j = 0, 2, 4, 6, 8 as u16 lanes
    const uint32_t EVEN = 0x0c010c00u, ODD = 0x0c030c02u;
    uint32_t T[5], U[5]; // T = 4 t0 = 12 (above + below) + 40 row (<= 16320), U = t1 = below - above
#define VO_SCHARR_COLS(pi, word, sel)                                                              \
    {                                                                                              \
        const uint32_t a = perm_b32(0, ra.word, sel), b = perm_b32(0, rb.word, sel), c = perm_b32(0, rc.word, sel); \
        T[pi] = pk_mad_u16(b, 40, pk_mad_u16(pk_add_u16(a, c), 12, 0));                            \
        U[pi] = pk_sub_i16(c, a);                                                                  \
    }
    VO_SCHARR_COLS(0, a, EVEN) VO_SCHARR_COLS(1, a, ODD) VO_SCHARR_COLS(2, b, EVEN) VO_SCHARR_COLS(3, b, ODD)
    VO_SCHARR_COLS(4, c, EVEN)
#undef VO_SCHARR_COLS
    // pixel m = 0..7 has its centre in column j = m + 1:  4 Ix = T[j + 1] - T[j - 1],  4 Iy = 12 (U[j - 1] + U[j + 1]) + 40 U[j]
    uint32_t out[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t ix = pk_sub_i16(T[k + 1], T[k]);
        const uint32_t mid = alignbyte(U[k + 1], U[k], 2); // (U[2k + 1], U[2k + 2])
        const uint32_t iy = pk_mad_u16(mid, 40, pk_mad_u16(pk_add_u16(U[k], U[k + 1]), 12, 0));
        out[2 * k] = perm_b32(iy, ix, VO_SEL_LO16);
        out[2 * k + 1] = perm_b32(iy, ix, VO_SEL_HI16);
    }
    // pixels >= w of the last group fall into the (zero) right border: keep them zero
    if (x8 + 8 > w) {
#pragma unroll
        for (int k = 1; k < 8; k++)
            if (x8 + k >= w)
                out[k] = 0;
    }
    VO_GLOBAL uint4 *o = (VO_GLOBAL uint4 *)((VO_GLOBAL uint32_t *)im.der[level] + (ptrdiff_t)y * stride + x8);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VO_HOST_EMUL)
    if (NT) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 a = {out[0], out[1], out[2], out[3]}, b = {out[4], out[5], out[6], out[7]};
        __builtin_nontemporal_store(a, (VO_GLOBAL u32x4 *)o);
        __builtin_nontemporal_store(b, (VO_GLOBAL u32x4 *)o + 1);
        return;
    }
#endif
    o[0] = make_uint4(out[0], out[1], out[2], out[3]);
    o[1] = make_uint4(out[4], out[5], out[6], out[7]);
}

#if defined(VO_DEV_VARIANTS) || defined(VO_HOST_EMUL) // ordinary stores: the A/B partner of the product kernel below
__global__ __launch_bounds__(256) void scharr_kernel(const PyrImage *__restrict__ imgs, int n_levels, ScharrTiles st)
{
    scharr_body<false>(imgs, n_levels, st);
}
#endif

// the same with non-temporal stores: what launch_scharr uses
__global__ __launch_bounds__(256) void scharr_nt_kernel(const PyrImage *__restrict__ imgs, int n_levels, ScharrTiles st)
{
    scharr_body<true>(imgs, n_levels, st);
}


#endif // round-3 chain

// ---------------------------------------------------------------------------------------------------
// FUSED PYRAMID PASS (round 4).  One pass over a level reads it ONCE and emits everything that depends on it: its Scharr
// image, the next level (pyrDown) and its own REFLECT_101 border -- the three kernels above each fetched level 0 from memory
// (8 launches per pyramid build, FETCH_SIZE 1.6-3.1 x the algorithmic bytes in pyr_down).  No LDS, no
// barrier in the pass: a work item is 4 adjacent columns x PF_ROWS rows; the thread walks DOWN its rows with one 8-byte load
// per source row (columns x4 - 2 .. x4 + 5: the 6 columns of the 3 x 3 Scharr stencil and the 7 of the 5-tap pyrDown
// window of its 2 next-level outputs), keeps a 3-row window of lifted u16 pairs for Scharr and a 5-row window of horizontal
// pyrDown sums in registers, and stores 16 bytes of (4 Ix | 4 Iy << 16) per row plus 2 next-level pixels every second row.
// Rows / columns outside the image are taken through REFLECT_101 BY INDEX (rows by address, columns by a byte permute of the
// window), never from the level's border: the border is written by the same launch (border items in the grid -- the body of
// border_fill_kernel), and the next level's border by the next pass.  Arithmetic = scharr_body / pyr_hrow / pyr_column above,
// so the results are bit-identical to the three-kernel chain (emulator + GPU pyramid tests).
//   pyr_pass_kernel      one level of all images: one wavefront per workgroup, ids in XCD-aware order -> (image, tail | row block,
//                        wavefront of column groups);
//                        one launch per level (L instead of 2 L + 2 launches)
#ifndef VO_PF_ROWS
#define VO_PF_ROWS 8
#endif
constexpr int PF_ROWS = VO_PF_ROWS; // rows per work item (even)
static_assert(PF_ROWS % 2 == 0 && PF_ROWS <= 16, "a work item starts on an even row; one REFLECT_101 step covers its halo (levels have >= 22 rows)");

struct PassPlan {
    int ng[VO_MAX_LEVELS];     // 4-column groups of a level
    int nm[VO_MAX_LEVELS];     // of which MAIN: groups 0 .. nm - 1 find columns 4 g - 2 .. 4 g + 4 in their 8-byte window (group 0
                               // patches its two left neighbours in registers); the rest (one group at the right end) are EDGE
    int nb[VO_MAX_LEVELS];     // row blocks of PF_ROWS rows
    int nci[VO_MAX_LEVELS];    // wavefronts (= workgroups) that cover the main groups of one row block: a `row` of the image's workgroups
    int n_tail[VO_MAX_LEVELS]; // edge + border work items (the first rows of an image's workgroups)
    int gy[VO_MAX_LEVELS];     // rows of workgroups per image = the rows that hold the tail items + nb
    int wide;                  // border work items: 1 = four rows / all side chunks of a row per lane, 0 = one 16-byte chunk per lane
    uint64_t m_img[VO_MAX_LEVELS], m_row[VO_MAX_LEVELS]; // floor(2^64 / (nci gy)) + 1, floor(2^64 / nci) + 1: id -> (image, y, x) by
                                                         // multiply-high, exact for every 32-bit id (pass_div)
};
struct __attribute__((packed, aligned(2))) LkU2x { // an 8-byte row window at an even column
    uint32_t lo, hi;
};

// Border work items of the fused pass (round 5): a lane takes one 16-byte chunk column of FOUR rows above / below the image
// (neighbouring lanes neighbouring chunks: every store instruction of the wavefront writes one contiguous run of a row), or ALL
// side chunks of one image row (the two left ones and the two to four at the right end) -- 22 wavefronts per KITTI level-0
// image where one chunk per lane made 91 next to the 235 of the main items (tools/pass_ab.sh, profiles/r05_pyramid_ab.txt).
// That is the WIDE form, for launches over many images (PassPlan::wide).  A launch over a few images (the synchronous drop-in
// call: four) is a handful of wavefronts whose slowest lane is the launch's time: there every chunk stays a lane of its own
// (the THIN form; with the wide one the four passes of a call took 35 us instead of 28, gpurun_out/r5_18).
constexpr int BORDER_ROWS_PER_ITEM = 4; // outside rows per lane (wide form)
static_assert((2 * VO_BY) % BORDER_ROWS_PER_ITEM == 0, "row groups of the outside rows");
inline int border_items(int h, int w, int stride, bool wide)
{
    const int cpr = stride >> 4;
    if (wide)
        return 2 * VO_BY / BORDER_ROWS_PER_ITEM * cpr + h;
    return 2 * VO_BY * cpr + h * (VO_BX / 16 + ((stride - VO_BX - (w & ~15)) >> 4));
}

inline PassPlan pass_plan(int n_levels, const int *lw, const int *lh, const int *lstride, bool wide = true)
{
    PassPlan pp = {};
    pp.wide = wide ? 1 : 0;
    for (int l = 0; l < n_levels; l++) {
        pp.ng[l] = (lw[l] + 3) / 4;
        pp.nm[l] = ((lw[l] - 5) >> 2) + 1; // 4 g + 4 <= w - 1 (= ng - 1: every level is at least 22 columns, plan_levels)
        pp.nb[l] = (lh[l] + PF_ROWS - 1) / PF_ROWS;
        pp.nci[l] = pp.nm[l] > 0 ? (pp.nm[l] + 63) / 64 : 1;
        pp.n_tail[l] = pp.nb[l] * (pp.ng[l] - pp.nm[l]) + border_items(lh[l], lw[l], lstride[l], wide);
        pp.gy[l] = pp.nb[l] + (pp.n_tail[l] + 64 * pp.nci[l] - 1) / (64 * pp.nci[l]);
        pp.m_img[l] = ~0ull / (uint64_t)(pp.nci[l] * pp.gy[l]) + 1; // (floor((2^64 - 1) / d) = floor(2^64 / d) unless d is a power of two,
        pp.m_row[l] = ~0ull / (uint64_t)pp.nci[l] + 1;               //  where it is one less and the + 1 lands exactly on 2^64 / d: also exact;
                                                                     //  d = 1 wraps to 0 -- the dispatch does not divide by 1)
    }
    return pp;
}

// floor(n / d) for a 32-bit n through m = floor(2^64 / d) + 1 (or exactly 2^64 / d): floor(n m / 2^64).  Exact for every n < 2^32:
// n m / 2^64 = n / d + n e / 2^64 with 0 <= e <= 1, and n e / 2^64 < 2^-32 <= 1 / d cannot reach the next integer.  Scalar code in
// the kernel (two s_mul_i32 + two s_mul_hi_u32); an integer division is ~30 instructions.
#if defined(__HIPCC__) && !defined(VO_HOST_EMUL)
__host__ __device__
#endif
inline uint32_t pass_div(uint32_t n, uint64_t m)
{
    const uint64_t lo = (uint64_t)n * (uint32_t)m, hi = (uint64_t)n * (uint32_t)(m >> 32);
    return (uint32_t)((hi + (lo >> 32)) >> 32);
}

// workgroup id -> image z, workgroup `rem` of the image = row `by`, wavefront `bx`; false = a padding workgroup (checked id by id
// against plain division in tests/test_kernel_emulation.py)
#if defined(__HIPCC__) && !defined(VO_HOST_EMUL)
__host__ __device__
#endif
inline bool pass_decode(const PassPlan &pp, int level, uint32_t id, uint32_t n_images, int remap, uint32_t *z, uint32_t *rem, uint32_t *by,
                        uint32_t *bx)
{
    const uint32_t nci = (uint32_t)pp.nci[level], wpi = nci * (uint32_t)pp.gy[level];
    if (remap) { // workgroup b runs on XCD b % 8: slot b / 8 of the images that XCD owns (z % 8 == b % 8)
        const uint32_t slot = id >> 3, k = pass_div(slot, pp.m_img[level]);
        *z = k * 8 + (id & 7);
        *rem = slot - k * wpi;
        if (*z >= n_images)
            return false;
    } else {
        *z = pass_div(id, pp.m_img[level]);
        *rem = id - *z * wpi;
    }
    *by = nci == 1 ? *rem : pass_div(*rem, pp.m_row[level]);
    *bx = *rem - *by * nci;
    return true;
}

// workgroups of one launch over n images (XCD-aware order: padded to a multiple of 8 images)
inline uint32_t pass_grid(const PassPlan &pp, int l, int n, int remap)
{
    return (uint32_t)pp.nci[l] * (uint32_t)pp.gy[l] * (uint32_t)(remap ? (n + 7) / 8 * 8 : n);
}

// images per launch: at most 4096 (hundreds of thousands of workgroups: nothing left to amortise), fewer where the workgroup
// count would not fit 31 bits; a multiple of 8.  Further images go to the next launch (tests/test_gpu_round4.py runs 4104 images)
constexpr int PASS_MAX_IMAGES = 4096;
inline int pass_images_per_launch(const PassPlan &pp, int l)
{
    const uint64_t wpi = (uint64_t)pp.nci[l] * pp.gy[l];
    const uint64_t n = ((1ull << 31) - 1) / wpi;
    return (int)(n >= PASS_MAX_IMAGES + 8 ? PASS_MAX_IMAGES : n < 16 ? 8 : n / 8 * 8 - 8);
}

// one 16-byte chunk of a level's REFLECT_101 border: columns x0 .. x0 + 15 of bordered row y from the image row `src` it mirrors.
//   inside the image            the same 16 bytes
//   left of it / right of it    ONE 16-byte load at the mirrored position, bytes reversed (4 v_perm_b32) -- where a single
//                               reflection stays inside the row, i.e. everywhere but in levels narrower than the border
//   straddling the right edge, or a level narrower than the border: byte by byte through reflect101 (one chunk per row)
// (Round 4 gathered every side chunk byte by byte: 16 loads where one does.)
__device__ __forceinline__ void border_chunk(VO_GLOBAL uint8_t *__restrict__ p, const VO_GLOBAL uint8_t *__restrict__ src, int y, int x0, int w,
                                             int stride)
{
    uint32_t v[4];
    const int xm = x0 + 15 < 0 ? -(x0 + 15) : x0 >= w ? 2 * (w - 1) - (x0 + 15) : -1; // first column of the mirrored run
    if (x0 >= 0 && x0 + 15 < w) {
        const U32x4 t = *(const VO_GLOBAL U32x4 *)(src + x0);
        v[0] = t.a;
        v[1] = t.b;
        v[2] = t.c;
        v[3] = t.d;
    } else if (xm >= 0 && xm + 15 < w) {
        const U8x16 t = *(const VO_GLOBAL U8x16 *)(src + xm); // columns xm .. xm + 15 = the chunk's columns in reverse order
        v[0] = perm_b32(0, t.d, 0x00010203u);
        v[1] = perm_b32(0, t.c, 0x00010203u);
        v[2] = perm_b32(0, t.b, 0x00010203u);
        v[3] = perm_b32(0, t.a, 0x00010203u);
    } else if (x0 >= 0 && x0 < w && 2 * (w - 1) - (x0 + 15) >= 0) {
        // the chunk that straddles the right edge (widths that are no multiple of 16: every KITTI level): its first k = w - x0
        // bytes are the image's, the rest mirror columns w - 2, w - 3 ...: the 16 bytes at x0 as they are (the tail of that load
        // is border memory of this row, whatever it holds) blended with the reversed 16 bytes at 2 (w - 1) - (x0 + 15), one
        // v_perm_b32 per dword whose selector takes byte q from the image while 4 d + q < k.  (Byte by byte through reflect101
        // this one chunk made the lane that owns a row's side chunks the slowest of its wavefront: KITTI level 2 43 us against
        // 26 without any border work, gpurun_out/r5_14.)
        const int k = w - x0; // 1 .. 15
        const U32x4 a = *(const VO_GLOBAL U32x4 *)(src + x0);
        const U8x16 t = *(const VO_GLOBAL U8x16 *)(src + (2 * (w - 1) - (x0 + 15)));
        const uint32_t img[4] = {a.a, a.b, a.c, a.d};
        const uint32_t rev[4] = {perm_b32(0, t.d, 0x00010203u), perm_b32(0, t.c, 0x00010203u), perm_b32(0, t.b, 0x00010203u),
                                 perm_b32(0, t.a, 0x00010203u)};
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int n = k - 4 * d; // bytes of this dword that belong to the image
            const uint32_t mirrored = n >= 4 ? 0u : n <= 0 ? 0xffffffffu : 0xffffffffu << (8 * n);
            v[d] = perm_b32(rev[d], img[d], 0x03020100u + (0x04040404u & mirrored));
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int x = x0 + 4 * q;
            v[q] = (uint32_t)src[reflect101(x, w)] | (uint32_t)src[reflect101(x + 1, w)] << 8 |
                   (uint32_t)src[reflect101(x + 2, w)] << 16 | (uint32_t)src[reflect101(x + 3, w)] << 24;
        }
    }
    *(VO_GLOBAL uint4 *)(p + (ptrdiff_t)y * stride + x0) = make_uint4(v[0], v[1], v[2], v[3]);
}

// a border work item of the fused pass (border_items above): items 0 .. 2 VO_BY / 4 cpr - 1 are chunk column c of the four
// outside rows 4 rg .. 4 rg + 3 (rows 0 .. VO_BY - 1 above the image, the rest below), the next h items the side chunks of one
// image row each
__device__ __forceinline__ void border_item(const PyrImage &im, int level, int item, int wide)
{
    const int w = im.w[level], h = im.h[level], stride = im.stride[level];
    VO_GLOBAL uint8_t *__restrict__ p = (VO_GLOBAL uint8_t *)im.lvl[level];
    const int cpr = stride >> 4;
    if (!wide) { // one chunk: first the rows above / below the image chunk by chunk, then the side chunks of the image rows
        const int xr0 = w & ~15, nbc = VO_BX / 16 + ((stride - VO_BX - xr0) >> 4), n_rows = 2 * VO_BY * cpr;
        int y, x0;
        if (item < n_rows) {
            const int r = item / cpr;
            y = r < VO_BY ? r - VO_BY : h + (r - VO_BY);
            x0 = 16 * (item - r * cpr) - VO_BX;
        } else {
            item -= n_rows;
            const int r = item / nbc, k = item - r * nbc;
            if (r >= h)
                return;
            y = r;
            x0 = k < VO_BX / 16 ? 16 * k - VO_BX : xr0 + 16 * (k - VO_BX / 16);
        }
        border_chunk(p, p + (ptrdiff_t)reflect101(y, h) * stride, y, x0, w, stride);
        return;
    }
    const int n_out = 2 * VO_BY / BORDER_ROWS_PER_ITEM * cpr;
    if (item < n_out) {
        const int rg = item / cpr, c = item - rg * cpr;
#pragma unroll
        for (int k = 0; k < BORDER_ROWS_PER_ITEM; k++) {
            const int r = BORDER_ROWS_PER_ITEM * rg + k;
            const int y = r < VO_BY ? r - VO_BY : h + (r - VO_BY);
            border_chunk(p, p + (ptrdiff_t)reflect101(y, h) * stride, y, 16 * c - VO_BX, w, stride);
        }
        return;
    }
    const int y = item - n_out;
    if (y >= h)
        return;
    const VO_GLOBAL uint8_t *__restrict__ src = p + (ptrdiff_t)y * stride;
#pragma unroll
    for (int k = 0; k < VO_BX / 16; k++)
        border_chunk(p, src, y, 16 * k - VO_BX, w, stride);
    // from the chunk that holds pixel w - 1 (or starts at w) to the end of the row; a chunk that straddles the image edge
    // rewrites its interior bytes with the values they already have
    for (int x0 = w & ~15; x0 < stride - VO_BX; x0 += 16)
        border_chunk(p, src, y, x0, w, stride);
}

// main work item: column group g (columns 4 g .. 4 g + 3), row block b (rows PF_ROWS b ..)
//
// Why FOUR columns per lane: the Scharr image is 80 % of the stage's bytes and goes out with non-temporal stores.  A lane that
// owns 8 pixels writes 32 bytes per row as two 16-byte stores, i.e. every store instruction of the wavefront covers 64 x 16
// bytes at a 32-byte stride -- and that pattern runs at 2.4 TB/s on the MI355X, whatever the kernel around it
// (tools/ubench/store_rate.hip: 2.44 TB/s, against 6.26 TB/s for the same non-temporal stores when the 64 lanes of an
// instruction write one contiguous KB; round 3's scharr_nt_kernel and the first fused pass both sat at that rate).  With 4
// pixels per lane each store instruction is one contiguous KB.
// SM (developer build, VO_PYR_STORE): how the Scharr image is stored -- 0 non-temporal (product), 1 ordinary stores, 2 NOT AT
// ALL (timing experiment only: what the pass costs without its dominant traffic)
template <bool HAS_NEXT, bool EDGE, int SM = 0>
__device__ __forceinline__ void pass_item(const PyrImage &im, int level, int g, int b, bool first_wave)
{
    const int w = im.w[level], h = im.h[level], stride = im.stride[level];
    const VO_GLOBAL uint8_t *__restrict__ src = (const VO_GLOBAL uint8_t *)im.lvl[level];
    VO_GLOBAL uint32_t *__restrict__ der = (VO_GLOBAL uint32_t *)im.der[level];
    const int x4 = 4 * g, y0 = PF_ROWS * b, c0 = x4 - 2; // c0: column of byte 0 of the 8-byte window (bytes 0 .. 6 are used:
                                                         // columns x4 - 2 .. x4 + 4; an EDGE item's window starts at x4 - 4)
    int dw = 0, dh = 0, dstride = 0;
    VO_GLOBAL uint8_t *__restrict__ dst = nullptr;
    if (HAS_NEXT) {
        dw = im.w[level + 1];
        dh = im.h[level + 1];
        dstride = im.stride[level + 1];
        dst = (VO_GLOBAL uint8_t *)im.lvl[level + 1];
    }
    const int x2 = x4 >> 1, oy0 = y0 >> 1;
    // (output rows through byte pointers stepped per row: scalar for a main item; the lane adds a 32-bit offset)
    VO_GLOBAL uint8_t *der_row = (VO_GLOBAL uint8_t *)der + (ptrdiff_t)y0 * stride * 4;
    VO_GLOBAL uint8_t *dst_row = HAS_NEXT ? dst + (ptrdiff_t)oy0 * dstride : nullptr;
    uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0; // pyrDown: horizontal sums (two u16) of the last five source rows
    uint32_t A[3], B[3], C[3];                      // Scharr: lifted column pairs (j, j + 1), j = 0, 2, 4 of the last three rows
#pragma unroll
    for (int k = 0; k < 3; k++)
        A[k] = B[k] = C[k] = 0;
    // ALL source rows of the item are requested before the first one is used (interior items): with the load inside the row
    // loop the compiler waited for every row before touching it -- one exposed memory latency per row, ~11 per item, and the
    // level-0 pass sat at 0.48 ms whatever the store pattern (gpurun_out/r4_06; the ISA had `s_waitcnt vmcnt(0)` behind each
    // of the loads).  2 registers per row.  (Rows past the last one an output of this item depends on -- the last row block of
    // a level -- are loaded too: REFLECT_101 keeps the address inside the image.)
    uint32_t W0[PF_ROWS + 3], W1[PF_ROWS + 3];
    if (!EDGE) {
        // The row block of a main item is the same for the whole wavefront (blockIdx.y): row reflection and row addresses are
        // scalar code and a load is `scalar row address + 32-bit lane offset`.  (Per lane and row this was a 64-bit
        // multiply-add, a quarter-rate instruction -- 23 per item with the store addresses: 680 -> 340 vector instructions
        // per wavefront, level-0 pass 431 -> 322 us.)  A block none of whose rows reflects -- all but the first and the last
        // one or two of a level -- steps a row pointer by the stride instead of reflecting and multiplying per row (240 -> 150
        // scalar instructions per wavefront; this alone did not move the time, gpurun_out/r4_24).
        const VO_GLOBAL uint8_t *rows[PF_ROWS + 3]; // (addresses first, ONE load loop after them: with a load loop in each
                                                    // branch the compiler merged the branches' loads into 4-byte halves at
                                                    // 64-bit lane addresses)
        if (y0 >= 2 && y0 + PF_ROWS < h) {
            const VO_GLOBAL uint8_t *row = src + (ptrdiff_t)(y0 - 2) * stride;
#pragma unroll
            for (int r = 0; r < PF_ROWS + 3; r++) {
                rows[r] = row;
                row += stride;
            }
        } else {
            int hq = h;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VO_HOST_EMUL)
            asm volatile("" : "+s"(hq)); // (keeps this branch a branch: without it the compiler computes these addresses for
                                         // every block and then overrides them -- 9 scalar instructions per row on top of
                                         // the 2 of the branch above)
#endif
#pragma unroll
            for (int r = 0; r < PF_ROWS + 3; r++) {
                // (one reflection is enough: -2 <= row < h + PF_ROWS + 1 and every level is taller than PF_ROWS + 2 -- a
                // level is at least 22 rows, buildOpticalFlowPyramid's stop rule; reflect101's general loop costs ~30
                // instructions)
                const int p = y0 - 2 + r, sy = p < 0 ? -p : p >= hq ? 2 * hq - 2 - p : p;
                rows[r] = src + (ptrdiff_t)sy * stride;
            }
        }
#pragma unroll
        for (int r = 0; r < PF_ROWS + 3; r++) {
            const LkU2x v = *(const VO_GLOBAL LkU2x *)(rows[r] + (uint32_t)x4 - 2);
            W0[r] = v.lo;
            W1[r] = v.hi;
        }
        // group 0 (in the first wavefront of a row block): its window starts two bytes left of the row -- memory of the
        // level's border, which this launch is also writing; the two bytes are replaced here, whatever they hold: columns -2,
        // -1 = columns 2, 1 = bytes 4, 3 of the window
        if (first_wave) {
            const uint32_t left = g == 0 ? 0x03020304u : 0x03020100u;
#pragma unroll
            for (int r = 0; r < PF_ROWS + 3; r++)
                W0[r] = perm_b32(W1[r], W0[r], left);
        }
    } else {
        // an EDGE item (the group at the right end of a row block): its columns x4 - 2 .. x4 + 4 reflect into x4 - 4 .. x4 + 3
        // whatever w mod 4 is, so it loads the 8-byte window at x4 - 4 and picks its 7 bytes with two byte permutes whose
        // selectors hold the reflection -- all rows before the first use, like the main items.  (It gathered 7 single bytes
        // per row inside the row loop: 11 memory round trips per edge wavefront, which made the edge wavefronts of the last
        // images the tail of the launch.)
        const int xs = x4 - 4; // (>= 0: a level is at least 22 columns -- plan_levels -- so the edge group is never group 0)
        uint32_t sel[2] = {0, 0x0c000000u};
#pragma unroll
        for (int k = 0; k < 7; k++)
            sel[k >> 2] |= (uint32_t)(reflect101(c0 + k, w) - xs) << (8 * (k & 3));
#pragma unroll
        for (int r = 0; r < PF_ROWS + 3; r++) {
            const VO_GLOBAL uint8_t *__restrict__ row = src + (ptrdiff_t)reflect101(y0 - 2 + r, h) * stride;
            const LkU2x v = *(const VO_GLOBAL LkU2x *)(row + xs);
            W0[r] = perm_b32(v.hi, v.lo, sel[0]);
            W1[r] = perm_b32(v.hi, v.lo, sel[1]);
        }
    }
#pragma unroll
    for (int r = 0; r < PF_ROWS + 3; r++) {
        // source row y0 - 2 + r: Scharr row y0 + j reads r = j + 1 .. j + 3, next-level row oy0 + j reads r = 2 j .. 2 j + 4
        const uint32_t w0 = W0[r], w1 = W1[r];
        if (HAS_NEXT) { // horizontal [1 4 6 4 1] of the 2 next-level outputs: bytes 0 .. 4 and 2 .. 6
            const uint32_t taps = 0x04060401u;
            const uint32_t h0 = udot4(w0, taps, udot4(w1, 0x00000001u, 0));
            const uint32_t h1 = udot4(alignbyte(w1, w0, 2), taps, udot4(w1, 0x00010000u, 0));
            q0 = q1;
            q1 = q2;
            q2 = q3;
            q3 = q4;
            q4 = h0 | h1 << 16;
            if (r >= 4 && (r & 1) == 0) { // source row 2 (oy0 + j) + 2 has arrived: output row oy0 + j, j = (r - 4) / 2
                const int oy = oy0 + (r - 4) / 2;
#ifdef VO_PASS_X
                if (!(VO_PASS_X & 2))
#endif
                if (oy < dh && (!EDGE || x2 < dw)) { // (4 g <= w - 1: a main group always has x2 < dw)
                    // 6 q2 + 4 (q1 + q3) + q0 + q4 + 128 <= 65408 fits 16 bits; the result is its high byte.  A second column
                    // >= dw lands in the right border, which the next level's pass rewrites
                    const uint32_t v = pk_mad_u16(q2, 6, pk_mad_u16(pk_add_u16(q1, q3), 4, pk_add_u16(pk_add_u16(q0, q4), 0x00800080u)));
                    *(VO_GLOBAL uint16_t *)(dst_row + (uint32_t)x2) = (uint16_t)perm_b32(0, v, 0x0c0c0301u);
                }
                dst_row += dstride;
            }
        }
        // Scharr window: bytes 1 .. 6 of the 8 = columns x4 - 1 .. x4 + 4
#pragma unroll
        for (int k = 0; k < 3; k++) {
            A[k] = B[k];
            B[k] = C[k];
        }
        C[0] = perm_b32(w1, w0, 0x0c020c01u);
        C[1] = perm_b32(w1, w0, 0x0c040c03u);
        C[2] = perm_b32(w1, w0, 0x0c060c05u);
        if (r >= 3) { // rows y - 1, y, y + 1 of Scharr row y = y0 + r - 3 are in A, B, C
            const int y = y0 + r - 3;
            if (y < h) {
                uint32_t T[3], U[3]; // T = 4 t0 = 12 (above + below) + 40 row (<= 16320), U = t1 = below - above (scharr_body)
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    T[k] = pk_mad_u16(B[k], 40, pk_mad_u16(pk_add_u16(A[k], C[k]), 12, 0));
                    U[k] = pk_sub_i16(C[k], A[k]);
                }
                uint32_t out[4];
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const uint32_t ix = pk_sub_i16(T[k + 1], T[k]);
                    const uint32_t mid = alignbyte(U[k + 1], U[k], 2);
                    const uint32_t iy = pk_mad_u16(mid, 40, pk_mad_u16(pk_add_u16(U[k], U[k + 1]), 12, 0));
                    out[2 * k] = perm_b32(iy, ix, VO_SEL_LO16);
                    out[2 * k + 1] = perm_b32(iy, ix, VO_SEL_HI16);
                }
                if (EDGE && x4 + 4 > w) { // pixels >= w of the last group (always an edge group) fall into the (zero) right border
#pragma unroll
                    for (int k = 1; k < 4; k++)
                        if (x4 + k >= w)
                            out[k] = 0;
                }
                VO_GLOBAL uint4 *o = (VO_GLOBAL uint4 *)(der_row + (uint32_t)(16 * g));
                der_row += (ptrdiff_t)stride * 4;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VO_HOST_EMUL)
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 oa = {out[0], out[1], out[2], out[3]};
                if (SM == 0)
                    __builtin_nontemporal_store(oa, (VO_GLOBAL u32x4 *)o);
                else if (SM == 1)
                    *(VO_GLOBAL u32x4 *)o = oa;
                else if ((out[0] ^ out[1] ^ out[2] ^ out[3]) == 0x12345679u) // (keeps the arithmetic alive; practically never true)
                    *(VO_GLOBAL u32x4 *)o = oa;
#else
                o[0] = make_uint4(out[0], out[1], out[2], out[3]);
#endif
            }
        }
    }
}

// Grid of a level's pass: ONE dimension, nci x gy workgroups (of one wavefront) per image.  Within an image: x = wavefront of 64
// main column groups, y = (after the tail rows) row block -- the row block of a wavefront is a scalar, so row reflection, row
// addresses and the "row exists" predicates are scalar code and the lanes only add their column offset.  The FIRST rows hold
// the TAIL items, 64 per workgroup in x-then-y order: the EDGE group of every row block (the group at the right end, which
// needs columns beyond the image: its row pointers are per lane and its window is permuted through REFLECT_101) and then the
// border chunks.  (Edge items numbered among the main ones sat in 4 of every 10 wavefronts and the whole wavefront waited for
// them -- gpurun_out/r4_03; packed they are one wavefront per image.  First, not last: an edge wavefront lives several times
// longer than a main one, and behind the main items of the last image it was the tail of the launch.)
// XCD-aware order: the dispatcher places workgroup b on XCD b % 8, each XCD has its own L2, and neighbours in this grid share
// data -- row blocks y and y + 1 share 3 of their 11 source rows, wavefronts x and x + 1 a cache line per row, the border
// items re-read rows of the image.  In dispatch order those neighbours sit on DIFFERENT XCDs and every shared line came from
// memory once per XCD (FETCH_SIZE 0.90 GB per 514-image step against 0.32 GB of pixels, profiles/r04.md).  So workgroup b is
// given slot b / 8 of the images that XCD b % 8 owns: image z is processed entirely by XCD z % 8 (the grid is padded to 8
// images; tools/ubench/pass_bench.hip compares this with dispatch order and with coarser / finer chunks).  Placement is a
// speed matter only: nothing here depends on which XCD runs what.
template <int SM = 0>
__device__ __forceinline__ void pass_dispatch(const PyrImage *__restrict__ imgs, int level, int n_levels, const PassPlan &pp, uint32_t n_images,
                                              int remap)
{
    uint32_t z, rem, by, bx;
    if (!pass_decode(pp, level, blockIdx.x, n_images, remap, &z, &rem, &by, &bx))
        return;
    const PyrImage &im = imgs[z];
    const int nb = pp.nb[level], nm = pp.nm[level];
    const bool has_next = level + 1 < n_levels;
    const int ty = pp.gy[level] - nb; // rows 0 .. ty - 1 of an image's workgroups: the tail items; then one row per row block
    if ((int)by >= ty) {
        const int g = (int)(bx * 64 + threadIdx.x);
        if (g >= nm)
            return;
        if (has_next)
            pass_item<true, false, SM>(im, level, g, (int)by - ty, bx == 0);
        else
            pass_item<false, false, SM>(im, level, g, (int)by - ty, bx == 0);
        return;
    }
    int t = (int)(rem * 64 + threadIdx.x);
    if (t >= pp.n_tail[level])
        return;
#ifdef VO_PASS_X // (tools/ubench/pass_bench.hip: what a part of the pass costs -- 1 no border items, 2 no next-level stores, 4 no edge items)
    if (((VO_PASS_X & 1) && t >= nb * (pp.ng[level] - nm)) || ((VO_PASS_X & 4) && t < nb * (pp.ng[level] - nm)))
        return;
#endif
    const int ne = pp.ng[level] - nm, n_edge = nb * ne;
    if (t >= n_edge) {
        border_item(im, level, t - n_edge, pp.wide);
        return;
    }
    const int b = t / ne, g = nm + (t - b * ne);
    if (has_next)
        pass_item<true, true, SM>(im, level, g, b, false);
    else
        pass_item<false, true, SM>(im, level, g, b, false);
}

__global__ __launch_bounds__(64) void pyr_pass_kernel(const PyrImage *__restrict__ imgs, int level, int n_levels, PassPlan pp, uint32_t n_images, int remap)
{
    pass_dispatch<0>(imgs, level, n_levels, pp, n_images, remap);
}
#if defined(VO_DEV_VARIANTS) && !defined(VO_HOST_EMUL)
template <int SM>
__global__ __launch_bounds__(64) void pyr_pass_sm_kernel(const PyrImage *__restrict__ imgs, int level, int n_levels, PassPlan pp, uint32_t n_images, int remap)
{
    pass_dispatch<SM>(imgs, level, n_levels, pp, n_images, remap);
}
#endif

#ifndef VO_HOST_EMUL
#ifdef VO_DEV_VARIANTS
void launch_border_fill(const PyrImage *d_imgs, int n_images, int first_level, int n_levels, const int *lstride,
                        const int *lh, hipStream_t stream)
{
    if (first_level >= n_levels)
        return;
    const BorderBlocks bb = border_blocks(first_level, n_levels, lstride, lh);
    dim3 grid(bb.first[n_levels], n_images);
    hipLaunchKernelGGL(border_fill_kernel, grid, dim3(256), 0, stream, d_imgs, n_levels, bb);
}

void launch_pyr_down(const PyrImage *d_imgs, int n_images, int level, int dw, int dh, hipStream_t stream)
{
    // Which kernel: the column walk needs no LDS and therefore starts next to a large pose chain (whose EPnP workgroups fill
    // the CUs' LDS) -- that is what many frames per run look like; a few images have no such neighbour and want the tile
    // kernel's latency.  128 images = 64 stereo pairs: a pose chain of 64 frames occupies a quarter of the chip's LDS.
    bool lds = n_images < 128;
#ifdef VO_DEV_VARIANTS
    static const int forced = [] { const char *e = getenv("VO_PYR_LDS"); return e ? atoi(e) : -1; }();
    if (forced >= 0)
        lds = forced != 0;
#endif
    if (lds) {
        dim3 grid((dw + PD_TW - 1) / PD_TW, (dh + PD_TH - 1) / PD_TH, n_images);
        hipLaunchKernelGGL(pyr_down_lds_kernel, grid, dim3(256), 0, stream, d_imgs, level);
        return;
    }
    dim3 grid((dw + PN_TW - 1) / PN_TW, (dh + PN_TH - 1) / PN_TH, n_images);
    hipLaunchKernelGGL(pyr_down_kernel, grid, dim3(256), 0, stream, d_imgs, level);
}

void launch_scharr(const PyrImage *d_imgs, int n_images, int first_level, int n_levels, const int *lw, const int *lh,
                   hipStream_t stream)
{
    if (first_level >= n_levels)
        return;
    const ScharrTiles st = scharr_tiles(first_level, n_levels, lw, lh);
    dim3 grid(st.first[n_levels], n_images);
    // non-temporal stores by default: the 4 bytes per pixel written here are 80 % of the pyramid stage's traffic and LK reads
    // a few per cent of them much later -- measured 0.58 -> 0.47 ms per 512 KITTI images, +1 ... +3 % frames/s in every
    // configuration (VO_SCHARR_NT=0 restores ordinary stores)
#ifdef VO_DEV_VARIANTS
    static const bool nt = [] { const char *e = getenv("VO_SCHARR_NT"); return !(e && e[0] == '0'); }();
    if (!nt) {
        hipLaunchKernelGGL(scharr_kernel, grid, dim3(256), 0, stream, d_imgs, n_levels, st);
        return;
    }
#endif
    hipLaunchKernelGGL(scharr_nt_kernel, grid, dim3(256), 0, stream, d_imgs, n_levels, st);
}
#endif // VO_DEV_VARIANTS

// the whole pyramid build of a range of images: one launch of pyr_pass_kernel per level (more when a level's images exceed
// pass_images_per_launch) -- four launches for the reference's maxLevel 3 instead of round 3's eight of three kernels
void launch_pyramid_fused(const PyrImage *d_imgs, int n_images, int n_levels, const int *lw, const int *lh, const int *lstride,
                          hipStream_t stream)
{
    if (n_images <= 0 || n_levels <= 0)
        return;
    const PassPlan pp = pass_plan(n_levels, lw, lh, lstride, /*wide border items*/ n_images >= 16);
    // (Levels 1 .. L-1 of an image in ONE launch by one workgroup per image -- pyr_tail_kernel, pass / fence + barrier / pass
    // -- was measured first: 0.86 ms for the three small levels of 514 images against 0.46 ms for level 0, gpurun_out/r4_04: a
    // few hundred workgroups of serial phases do not fill the chip.)
#ifdef VO_DEV_VARIANTS
    static const int sm = [] { const char *e = getenv("VO_PYR_STORE"); return e ? atoi(e) : 0; }();
    static const int xcd_env = [] { const char *e = getenv("VO_PYR_XCD"); return e ? atoi(e) : 1; }(); // 0: workgroups in dispatch order
#endif
    for (int l = 0; l < n_levels; l++) {
        const int per = pass_images_per_launch(pp, l);
        for (int first = 0; first < n_images; first += per) {
            const int n = n_images - first < per ? n_images - first : per;
            // XCD-pinned workgroup order per LAUNCH (it was decided from the total, so the 8 images left over from
            // 4104 ran pinned): with fewer than two images per XCD pinning would leave XCDs without work
            int remap = n >= 16;
#ifdef VO_DEV_VARIANTS
            remap = remap && xcd_env;
#endif
            const uint32_t nwg = pass_grid(pp, l, n, remap);
#ifdef VO_DEV_VARIANTS
            if (sm == 1) {
                hipLaunchKernelGGL(pyr_pass_sm_kernel<1>, dim3(nwg), dim3(64), 0, stream, d_imgs + first, l, n_levels, pp, (uint32_t)n, remap);
                continue;
            }
            if (sm == 2) {
                hipLaunchKernelGGL(pyr_pass_sm_kernel<2>, dim3(nwg), dim3(64), 0, stream, d_imgs + first, l, n_levels, pp, (uint32_t)n, remap);
                continue;
            }
#endif
            hipLaunchKernelGGL(pyr_pass_kernel, dim3(nwg), dim3(64), 0, stream, d_imgs + first, l, n_levels, pp, (uint32_t)n, remap);
        }
    }
}

// One staged host image -> pixel (0, 0) ... of its level-0 rows, read by the GPU over PCIe straight from the page-locked
// staging slot (the slot already has the device's row pitch, slot and destination are 16-byte aligned): 16 bytes per lane.
// The synchronous drop-in calls use it instead of hipMemcpyAsync: four separate copies cost 81 us from the first enqueue to a
// kernel behind them (7-10 us of fixed cost each), one 1.97 MB copy or one such kernel 52 us -- the link's rate -- and a
// kernel per image starts moving image k while the host still repacks image k + 1 (tools/ubench/h2d_probe.hip, round 5).
// (the call's points and their count ride along with the last image: `extra` workgroups behind the image's copy n8 float2 from
// src2 to dst2, and one thread writes the count -- two copy kernels of their own were 9 us of every call)
__global__ __launch_bounds__(256) void pull_image_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t n16,
                                                         const uint2 *__restrict__ src2, uint2 *__restrict__ dst2, uint32_t n8,
                                                         int *__restrict__ count_dst, int count)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x, img_threads = (n16 + 255u) / 256u * 256u;
    if (i < n16)
        dst[i] = src[i];
    else if (i >= img_threads && i - img_threads < n8)
        dst2[i - img_threads] = src2[i - img_threads];
    if (i == 0 && count_dst)
        *count_dst = count;
}

void launch_pull_image(const void *src_pinned_dev, void *dst, size_t bytes, hipStream_t stream, const void *pts_pinned_dev,
                       void *pts_dst, int n_pts, int *count_dst)
{
    const uint32_t n16 = (uint32_t)((bytes + 15) / 16), n8 = pts_dst ? (uint32_t)n_pts : 0u;
    const uint32_t blocks = (n16 + 255) / 256 + (n8 + 255) / 256; // (bytes == 0: the points and their count alone)
    hipLaunchKernelGGL(pull_image_kernel, dim3(blocks ? blocks : 1u), dim3(256), 0, stream, (const uint4 *)src_pinned_dev,
                       (uint4 *)dst, n16, (const uint2 *)pts_pinned_dev, (uint2 *)pts_dst, n8, count_dst, n_pts);
}

#endif // VO_HOST_EMUL

} // namespace vo
