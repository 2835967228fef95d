// vo_dev.h -- device-side data layout shared by the HIP kernels and the C-ABI host code.
//
// HBM layout (see DESIGN.md "Data layout"):
//   * image table: every 8-bit image owns its whole pyramid.  Level l is stored WITH a border, the
//     way OpenCV's buildOpticalFlowPyramid keeps winSize extra pixels around each level: VO_BX
//     columns left, >= VO_BY columns right, VO_BY rows above and below, filled by REFLECT_101, so
//     that every 21 x 21 LK window the reference admits (top-left corner in [-21, w) x [-21, h))
//     reads real memory and no kernel needs a border path.  The row pitch `stride` (pixels) is a
//     multiple of 16 and pixel (0, 0) of every row is 16-byte aligned.  `lvl[l]` points at pixel (0, 0).
//   * Scharr image of every level (calcSharrDeriv of the reference's LK calls): one dword per pixel,
//     (4*Ix as int16) | (4*Iy as int16) << 16, same geometry as the level, border = 0 exactly like
//     OpenCV's BORDER_CONSTANT derivative buffer.  `der[l]` points at pixel (0, 0).
//   * a "frame" (one stereo pair at t0 and t1 = the unit of work of circularMatching(),
//     reference feature.h:61-65) is a Quad of four image-table indices (l0, r0, l1, r1).
//   * per frame SoA feature arrays with a fixed capacity `cap`: float2 points, u8 status.
//
// Reads stay inside their level (round 5).  A level's bordered allocation is rows -VO_BY .. h + VO_BY - 1
// of `stride` bytes (dwords for the Scharr image), each row holding columns -VO_BX .. stride - VO_BX - 1.  EVERY load of pixel
// or derivative memory by a kernel stays inside that rectangle -- no load runs over the end of a row into the next one, none
// leaves the level -- so a level may be the last thing in its buffer (a one-level pyramid, the last image of the table) and
// nothing depends on what follows it (another level, another image, vo_create's slack).  Who reads what:
//   pyr_pass_kernel   8-byte windows at columns 4 g - 2 .. 4 g + 5 (4 g <= w - 1; group 0's two border bytes are replaced),
//                     edge items at x4 - 4 .. x4 + 3; rows reflected into 0 .. h - 1                       (pyramid.hip)
//   lk_circular_*     I-window rows and the 48 x 40 search tile: tile origin clamped to [-VO_BX, stride - VO_BX - 48] x
//                     [-VO_BY, h + VO_BY - 40], 16-byte chunks; derivative dwords of the 21 x 21 window          (lk.hip)
//   fast_tile_*       16-byte chunks of rows y0 - 4 .. (clamped to h + VO_BY - 1), columns x0 - 4 ..; a chunk that would cross
//                     the row end reads the row's last 16 bytes instead (its bytes are never used)                  (fast.hip)
// Checked by the sanitizer tier: the CPU emulator (tests/host_check/kernel_emu.cpp) runs these kernel sources with every level
// in its own exactly-sized heap block under AddressSanitizer (tests/test_sanitize.py).
#pragma once

#include <stdint.h>

// Cross-lane primitives.  The CPU test-suite compiles the kernel sources against a coroutine SIMT
// emulator (tests/host_check/hip_emu.h defines VO_HOST_EMUL and the emu_* functions); the product
// build always takes the CDNA4 builtins.
#ifdef VO_HOST_EMUL
#define VO_READFIRSTLANE(v) emu_readfirstlane(v)
#define VO_READLANE(v, l) emu_readlane((v), (l))
#define VO_UPDATE_DPP(old, src, ctrl, rm, bm, bc) emu_update_dpp((old), (src), (ctrl), (rm), (bm), (bc))
#define VO_BALLOT(p) emu_ballot(p)
#define VO_POPCLL(m) __builtin_popcountll(m)
#define VO_MBCNT(m, acc, lane) ((acc) + (uint32_t)__builtin_popcountll((m) & ((1ull << (lane)) - 1ull)))
#define VO_PERMLANE32_SWAP(a, b) emu_permlane_swap((a), (b), 32)
#define VO_PERMLANE16_SWAP(a, b) emu_permlane_swap((a), (b), 16)
#else
#include <hip/hip_runtime.h>
#define VO_READFIRSTLANE(v) __builtin_amdgcn_readfirstlane(v)
#define VO_READLANE(v, l) __builtin_amdgcn_readlane((v), (l))
#define VO_UPDATE_DPP(old, src, ctrl, rm, bm, bc) __builtin_amdgcn_update_dpp((old), (src), (ctrl), (rm), (bm), (bc))
#define VO_BALLOT(p) __builtin_amdgcn_ballot_w64(p)
#define VO_POPCLL(m) __popcll(m)
// acc + the number of set bits of the 64-bit mask below this lane: v_mbcnt_lo_u32_b32 + v_mbcnt_hi_u32_b32
#define VO_MBCNT(m, acc, lane) __builtin_amdgcn_mbcnt_hi((uint32_t)((m) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(m), (acc)))
// gfx950 v_permlane32_swap_b32 a, b: lanes 32-63 of a <-> lanes 0-31 of b;
//        v_permlane16_swap_b32 a, b: odd 16-lane rows of a <-> even rows of b  (both operands are rewritten)
#define VO_PERMLANE32_SWAP(a, b)                                                                      \
    do {                                                                                              \
        auto r_ = __builtin_amdgcn_permlane32_swap((unsigned)(a), (unsigned)(b), false, false);       \
        (a) = (int)r_[0];                                                                             \
        (b) = (int)r_[1];                                                                             \
    } while (0)
#define VO_PERMLANE16_SWAP(a, b)                                                                      \
    do {                                                                                              \
        auto r_ = __builtin_amdgcn_permlane16_swap((unsigned)(a), (unsigned)(b), false, false);       \
        (a) = (int)r_[0];                                                                             \
        (b) = (int)r_[1];                                                                             \
    } while (0)
#endif

// Image pointers come out of the PyrImage table in memory, so the compiler can only treat them as generic
// (flat) addresses; VO_GLOBAL marks them as what they are -- global memory -- which turns flat_load into
// global_load with a scalar base (no lgkmcnt coupling with LDS traffic, no 64-bit per-lane address math).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VO_HOST_EMUL)
#define VO_GLOBAL __attribute__((address_space(1)))
#else
#define VO_GLOBAL /* host pass of hipcc, CPU emulator */
#endif

#define VO_MAX_LEVELS 5
#define VO_BX 32 /* left border columns (>= 21 + tile slack, keeps x = 0 16-byte aligned) */
#define VO_BY 24 /* top / bottom border rows and minimum right border columns (>= 21) */
static_assert(VO_BX % 16 == 0 && VO_BX >= 21 && VO_BY >= 21, "pixel (0, 0) of a row is 16-byte aligned; 21 x 21 windows at -21 read border, not the neighbour row");

namespace vo {

struct PyrImage {
    uint8_t *lvl[VO_MAX_LEVELS];
    uint32_t *der[VO_MAX_LEVELS];
    int w[VO_MAX_LEVELS], h[VO_MAX_LEVELS], stride[VO_MAX_LEVELS];
};

struct Quad {
    int l0, r0, l1, r1;
};

__device__ __forceinline__ int reflect101(int p, int len)
{
    // cv::borderInterpolate(p, len, BORDER_REFLECT_101); |p| excursions here are < len
    if (len == 1)
        return 0;
    while (p < 0 || p >= len)
        p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

// float -> int32 the way gfx950 converts (v_cvt_i32_f32 / v_cvt_i32_f64): truncation towards zero, SATURATING, NaN -> 0.
// x86 (the reference's CPU path, the CPU checker of tests/, and this source when g++ compiles it for the CPU emulator) converts with
// cvttss2si / cvttsd2si, which return INT_MIN for NaN and for every value outside int32 -- so a kernel that decides on a
// converted value can differ from the reference exactly on non-finite input (lk admitted NaN points).
// Every value-level float -> int conversion of the kernels goes through these two: the device takes the one instruction, the
// emulator spells the device's semantics out, so emulator == GPU on that input class and the CPU suite can hold the
// kernels to the reference's behaviour there (tests/test_kernel_emulation.py::test_lk_nonfinite_points).
__device__ __forceinline__ int vo_f2i(float v)
{
#ifdef VO_HOST_EMUL
    if (v != v)
        return 0;
    if (v >= 2147483648.f)
        return 2147483647;
    if (v <= -2147483648.f)
        return -2147483647 - 1;
#endif
    return (int)v;
}
__device__ __forceinline__ int vo_d2i(double v)
{
#ifdef VO_HOST_EMUL
    if (v != v)
        return 0;
    if (v >= 2147483647.0)
        return 2147483647;
    if (v <= -2147483648.0)
        return -2147483647 - 1;
#endif
    return (int)v;
}

__device__ __forceinline__ int uni(int v) { return VO_READFIRSTLANE(v); }
__device__ __forceinline__ float unif(float v)
{
    return __int_as_float(VO_READFIRSTLANE(__float_as_int(v)));
}

// DPP controls (CDNA3/4 ISA "DPP_CTRL"): quad_perm, row_half_mirror, row_mirror, row_bcast
#define VO_DPP_QUAD_XOR1 0xB1   /* quad_perm:[1,0,3,2] */
#define VO_DPP_QUAD_XOR2 0x4E   /* quad_perm:[2,3,0,1] */
#define VO_DPP_ROW_HALF_MIRROR 0x141
#define VO_DPP_ROW_MIRROR 0x140
#define VO_DPP_ROW_BCAST15 0x142
#define VO_DPP_ROW_BCAST31 0x143

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add(int v)
{
    return v + VO_UPDATE_DPP(0, v, CTRL, ROW_MASK, 0xf, true);
}

// Exact wave-wide sum of per-lane int32 partials, returned as a wave-uniform f32 rounded once from
// the exact integer (= (float)(int64 sum), what the CPU path computes).  Requires |v| < 2^28 so
// that the first three butterfly steps (8-lane sums) cannot overflow int32; the 8-lane sums are then
// split into a signed high and an unsigned low 16-bit half which are reduced separately (row_mirror,
// row_bcast15, row_bcast31: the total lands in lane 63), read with v_readlane and recombined by one
// v_fma_f32 (exact product and sum, one rounding).
// 9 v_add_u32_dpp + 2 split + 2 v_readlane + 2 cvt + 1 fma instead of 12 ds_bpermute round trips per sum.
// (Measured alternative, profiles/r01_v3: reading the eight 8-lane sums with v_readlane and adding
// them on the scalar unit saves 4 VALU instructions per sum but costs 22 SALU ones -- the CU's single
// scalar unit then becomes the bottleneck and the kernel runs 4 % slower.)
__device__ __forceinline__ float wave_sum_exact_f32(int v)
{
    v = dpp_add<VO_DPP_QUAD_XOR1, 0xf>(v);
    v = dpp_add<VO_DPP_QUAD_XOR2, 0xf>(v);
    v = dpp_add<VO_DPP_ROW_HALF_MIRROR, 0xf>(v);
    int lo = v & 0xffff, hi = v >> 16;
    lo = dpp_add<VO_DPP_ROW_MIRROR, 0xf>(lo);
    hi = dpp_add<VO_DPP_ROW_MIRROR, 0xf>(hi);
    lo = dpp_add<VO_DPP_ROW_BCAST15, 0xa>(lo);
    hi = dpp_add<VO_DPP_ROW_BCAST15, 0xa>(hi);
    lo = dpp_add<VO_DPP_ROW_BCAST31, 0xc>(lo);
    hi = dpp_add<VO_DPP_ROW_BCAST31, 0xc>(hi);
    // hi * 2^16 + lo rounded ONCE: both halves are exact in f32 (|hi| < 2^19, lo < 2^22) and a fused
    // multiply-add rounds the exact sum a single time = (float)(int64 total), the correctly rounded value
    return fmaf((float)VO_READLANE(hi, 63), 65536.f, (float)VO_READLANE(lo, 63));
}

// Two exact wave-wide sums for the price of (less than) one: the half-swap instructions of gfx950 fold two
// reduction trees into one register.
//   v_permlane32_swap(a, b); t = a + b      lanes 0-31: a[l] + a[l+32], lanes 32-63: the same of b
//   quad butterflies (2 DPP adds)           8-value sums, |t| < 2^31 for |v| < 2^28
//   split t into signed high / unsigned low 16 bits, v_permlane16_swap(hi, lo); u = hi + lo
//                                           rows 0 / 1 / 2 / 3: a's high, a's low, b's high, b's low partials
//   row_half_mirror, row_mirror (2 DPP adds) every lane of a row holds that row's total
//   convert to f32 (exact: |high| < 2^19, low < 2^22); rows 1 and 3 fetch the high total of the row before
//   them (row_bcast15) and fuse high * 2^16 + low with ONE rounding = (float)(int64 sum), what the CPU
//   path computes; two v_readlane return the results.
// 15 VALU instructions for two sums instead of 2 x 16 with wave_sum_exact_f32.
__device__ __forceinline__ void wave_sum2_exact_f32(int a, int b, float &fa, float &fb)
{
    VO_PERMLANE32_SWAP(a, b);
    int t = a + b;
    t = dpp_add<VO_DPP_QUAD_XOR1, 0xf>(t);
    t = dpp_add<VO_DPP_QUAD_XOR2, 0xf>(t);
    int hi = t >> 16, lo = t & 0xffff;
    VO_PERMLANE16_SWAP(hi, lo);
    int u = hi + lo;
    u = dpp_add<VO_DPP_ROW_HALF_MIRROR, 0xf>(u);
    u = dpp_add<VO_DPP_ROW_MIRROR, 0xf>(u);
    const float uf = (float)u;
    const float up = __int_as_float(VO_UPDATE_DPP(0, __float_as_int(uf), VO_DPP_ROW_BCAST15, 0xa, 0xf, true));
    const float res = fmaf(up, 65536.f, uf);
    fa = __int_as_float(VO_READLANE(__float_as_int(res), 31));
    fb = __int_as_float(VO_READLANE(__float_as_int(res), 63));
}

// Three exact sums (the structure tensor A11, A12, A22) in one tree: two half-swaps fold a|b and c|0 into two
// registers, the 16-lane swap folds those into one (rows: a, c, b, 0; 4-value sums < 2^30), one quad step gives
// 8-value sums < 2^31, then the signed-high / unsigned-low halves finish with three DPP steps each and one fma.
// 22 VALU instead of two wave_sum2 calls (32).
__device__ __forceinline__ void wave_sum3_exact_f32(int a, int b, int c, float &fa, float &fb, float &fc)
{
    int z = 0;
    VO_PERMLANE32_SWAP(a, b);
    int t1 = a + b;
    VO_PERMLANE32_SWAP(c, z);
    int t2 = c + z;
    VO_PERMLANE16_SWAP(t1, t2);
    int u = t1 + t2;
    u = dpp_add<VO_DPP_QUAD_XOR1, 0xf>(u);
    int hi = u >> 16, lo = u & 0xffff;
    hi = dpp_add<VO_DPP_QUAD_XOR2, 0xf>(hi);
    lo = dpp_add<VO_DPP_QUAD_XOR2, 0xf>(lo);
    hi = dpp_add<VO_DPP_ROW_HALF_MIRROR, 0xf>(hi);
    lo = dpp_add<VO_DPP_ROW_HALF_MIRROR, 0xf>(lo);
    hi = dpp_add<VO_DPP_ROW_MIRROR, 0xf>(hi);
    lo = dpp_add<VO_DPP_ROW_MIRROR, 0xf>(lo);
    const float r = fmaf((float)hi, 65536.f, (float)lo);
    fa = __int_as_float(VO_READLANE(__float_as_int(r), 0));
    fc = __int_as_float(VO_READLANE(__float_as_int(r), 16));
    fb = __int_as_float(VO_READLANE(__float_as_int(r), 32));
}

} // namespace vo
