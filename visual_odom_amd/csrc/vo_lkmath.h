// vo_lkmath.h -- packed integer pixel arithmetic of the Lucas-Kanade kernel (lk.hip).
//
// OpenCV's LKTrackerInvoker (the arithmetic behind the reference's cv::calcOpticalFlowPyrLK calls,
// feature.cpp:136-139) samples 21 x 21 windows with 14-bit fixed-point bilinear weights:
//     val = DESCALE(p00*iw00 + p01*iw01 + p10*iw10 + p11*iw11, n)        (n = 9 for pixels,
//                                                                         n = 14 for Scharr samples)
// On CDNA4 one lane owns a 7-pixel row segment of the window and evaluates it two pixels at a time
// with the packed 16-bit dot-product instructions:
//   * a pixel pair (p[k], p[k+1]) is lifted out of the 8 loaded bytes with one v_perm_b32 into the
//     HIGH byte of two u16 lanes (= 256 * p), so that v_dot2_u32_u16 against (iw00, iw01) and then
//     (iw10, iw11) yields 256 * S; with the rounding constant 1 << 16 as the initial accumulator the
//     top 16 bits are (S + 256) >> 8, and one packed shift right by 1 gives DESCALE(S, 9) exactly;
//   * the Scharr image is stored pre-multiplied by 4 (|4 d| <= 16320 fits int16), so two
//     v_dot2_i32_i16 with initial accumulator 1 << 15 leave DESCALE(S, 14) in the top 16 bits;
//   * results are re-packed two per register (v_perm_b32) so that diff = val - I is one v_pk_sub_i16
//     and b1 += diff*Ix, b2 += diff*Iy are one v_dot2_i32_i16 each per pixel pair.
// Every step is exact integer arithmetic, so the result is bit-identical to the scalar formula; the
// host build of this header (tests/host_check) proves that against the plain restatement.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#include <hip/hip_runtime.h>
#define VO_HD __host__ __device__ __forceinline__
#else
#define VO_HD static inline
#endif

namespace vo {

// ---- instruction wrappers (device: the CDNA4 instruction; host: its definition) -------------------
// v_perm_b32: result byte i = byte sel[i] of the 8-byte value {hi:lo} (0..3 -> lo, 4..7 -> hi), 0x0c -> 0
VO_HD uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    uint64_t v = ((uint64_t)hi << 32) | lo;
    uint32_t out = 0;
    for (int i = 0; i < 4; i++) {
        uint32_t s = (sel >> (8 * i)) & 0xff;
        uint32_t b = s <= 7 ? (uint32_t)((v >> (8 * s)) & 0xff) : 0u; // only selectors 0..7 and 0x0c are used
        out |= b << (8 * i);
    }
    return out;
#endif
}

// v_dot2_u32_u16: a.lo*b.lo + a.hi*b.hi + c (unsigned 16-bit lanes)
VO_HD uint32_t udot2(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), c, false);
#else
    return (a & 0xffff) * (b & 0xffff) + (a >> 16) * (b >> 16) + c;
#endif
}

// v_dot2_i32_i16: signed 16-bit lanes
VO_HD int32_t sdot2(uint32_t a, uint32_t b, int32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(i16x2, a), __builtin_bit_cast(i16x2, b), c, false);
#else
    return (int32_t)(int16_t)(a & 0xffff) * (int16_t)(b & 0xffff) + (int32_t)(int16_t)(a >> 16) * (int16_t)(b >> 16) + c;
#endif
}

// The same dot product as the first link of an accumulation chain.  v_dot2c_i32_i16 (what the compiler picks for
// sdot2) accumulates in place, so a chain that starts from a constant or from a value that must survive costs a
// v_mov per chain; the clamped variant only exists in the three-address VOP3P form, which takes the start value
// from any operand.  No chain here gets anywhere near the int32 range, so the clamp never acts.
VO_HD int32_t sdot2_first(uint32_t a, uint32_t b, int32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(i16x2, a), __builtin_bit_cast(i16x2, b), c, true);
#else
    return sdot2(a, b, c);
#endif
}

// v_pk_sub_i16 (wrapping) and v_pk_lshrrev_b16 by 1
VO_HD uint32_t pk_sub_i16(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, (i16x2)(__builtin_bit_cast(i16x2, a) - __builtin_bit_cast(i16x2, b)));
#else
    return (((a & 0xffff) - (b & 0xffff)) & 0xffff) | (((a >> 16) - (b >> 16)) << 16);
#endif
}

VO_HD uint32_t pk_lshr1_u16(uint32_t a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) >> (unsigned short)1));
#else
    return (a >> 1) & 0x7fff7fffu;
#endif
}

// v_dot4_u32_u8: sum of the four unsigned byte products + c
VO_HD uint32_t udot4(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(a, b, c, false);
#else
    uint32_t r = c;
    for (int i = 0; i < 4; i++)
        r += ((a >> (8 * i)) & 0xff) * ((b >> (8 * i)) & 0xff);
    return r;
#endif
}

// packed 16-bit lanes, wrapping: v_pk_add_u16, v_pk_mul_lo_u16, v_pk_mad_u16 (the low 16 bits of a product or sum do
// not depend on signedness, so the same instructions serve int16 lanes)
VO_HD uint32_t pk_add_u16(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b)));
#else
    return (((a & 0xffff) + (b & 0xffff)) & 0xffff) | (((a >> 16) + (b >> 16)) << 16);
#endif
}

// v_pk_sub_u16 clamp (saturating at 0) and v_pk_min_u16
VO_HD uint32_t pk_subsat_u16(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
#else
    const uint32_t al = a & 0xffff, bl = b & 0xffff, ah = a >> 16, bh = b >> 16;
    return (al > bl ? al - bl : 0) | (ah > bh ? ah - bh : 0) << 16;
#endif
}

VO_HD uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
#else
    const uint32_t al = a & 0xffff, bl = b & 0xffff, ah = a >> 16, bh = b >> 16;
    return (al < bl ? al : bl) | (ah < bh ? ah : bh) << 16;
#endif
}

VO_HD uint32_t pk_mad_u16(uint32_t a, uint32_t k /* both lanes */, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 kk = {(unsigned short)k, (unsigned short)k};
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) * kk + __builtin_bit_cast(u16x2, c)));
#else
    return (((a & 0xffff) * k + (c & 0xffff)) & 0xffff) | ((((a >> 16) * k + (c >> 16)) & 0xffff) << 16);
#endif
}

// v_alignbyte_b32 / v_alignbit_b32: the 8-byte value {hi:lo} shifted right by `bytes` bytes, low dword
VO_HD uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t bytes /* 0..3 */)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, bytes);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * bytes));
#endif
}

// ---- selectors ------------------------------------------------------------------------------------
// bytes k and k+1 of the 8 loaded bytes into the high byte of the two u16 lanes (256*p[k], 256*p[k+1])
#define VO_SEL_PIX(k) (0x0cu | ((uint32_t)(k) << 8) | (0x0cu << 16) | ((uint32_t)((k) + 1) << 24))
constexpr uint32_t VO_SEL_LO16 = 0x05040100u; // (lo.lo16, hi.lo16)
constexpr uint32_t VO_SEL_HI16 = 0x07060302u; // (lo.hi16, hi.hi16)

// two weights as int16 lanes (|w| <= 2^14, so v_cvt_pk_i16_i32's saturation never triggers)
VO_HD uint32_t pack_w(int w_lo, int w_hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(w_lo, w_hi));
#else
    return (uint32_t)(w_lo & 0xffff) | ((uint32_t)w_hi << 16);
#endif
}

// ---- one 7-pixel row segment ----------------------------------------------------------------------
// t = bytes x..x+7 of the upper image row, b = the same columns one row below.
// wt = (iw00, iw01), wb = (iw10, iw11) as packed int16 lanes.
// out[m] = (val[2m], val[2m+1]) as packed int16, out[3].hi = 0;
// val[k] = DESCALE(t[k]*iw00 + t[k+1]*iw01 + b[k]*iw10 + b[k+1]*iw11, 9).
// iw00, iw01, iw10 are rounded non-negative products, but iw11 = 2^14 - (their sum) is -1 when the
// three roundings add up to 2^14 + 1; an unsigned dot product cannot carry a negative weight, so
// that (rare, wave-uniform) case zeroes the weight and subtracts |iw11| * 256*p11 explicitly.
VO_HD void bilinear7_u8(uint32_t t_lo, uint32_t t_hi, uint32_t b_lo, uint32_t b_hi, uint32_t wt, uint32_t wb,
                        uint32_t out[4])
{
    uint32_t acc[8];
    if (!(wb & 0x80000000u)) {
#define VO_PIX_STEP(k)                                                                               \
    acc[k] = udot2(perm_b32(b_hi, b_lo, VO_SEL_PIX(k)), wb,                                          \
                   udot2(perm_b32(t_hi, t_lo, VO_SEL_PIX(k)), wt, 1u << 16));
        VO_PIX_STEP(0) VO_PIX_STEP(1) VO_PIX_STEP(2) VO_PIX_STEP(3) VO_PIX_STEP(4) VO_PIX_STEP(5) VO_PIX_STEP(6)
#undef VO_PIX_STEP
    } else {
        const uint32_t wb0 = wb & 0xffffu, kneg = (uint32_t)(-(int32_t)((int16_t)(wb >> 16)));
#define VO_PIX_STEP(k)                                                                               \
    {                                                                                                \
        const uint32_t pb = perm_b32(b_hi, b_lo, VO_SEL_PIX(k));                                     \
        acc[k] = udot2(pb, wb0, udot2(perm_b32(t_hi, t_lo, VO_SEL_PIX(k)), wt, 1u << 16)) -          \
                 kneg * (pb >> 16);                                                                  \
    }
        VO_PIX_STEP(0) VO_PIX_STEP(1) VO_PIX_STEP(2) VO_PIX_STEP(3) VO_PIX_STEP(4) VO_PIX_STEP(5) VO_PIX_STEP(6)
#undef VO_PIX_STEP
    }
    acc[7] = 0;
#pragma unroll
    for (int m = 0; m < 4; m++)
        out[m] = pk_lshr1_u16(perm_b32(acc[2 * m + 1], acc[2 * m], VO_SEL_HI16));
}

// The same in two steps, for callers that sample one pixel cell several times with different weights:
// lift7 does the weight-independent part (the 7 pixel pairs of one row as u16 lanes 256*p[k], 256*p[k+1]),
// blend7 the weighted sum.  blend7(lift7(top), lift7(bottom)) == bilinear7_u8.
VO_HD void lift7(uint32_t lo, uint32_t hi, uint32_t pair[7])
{
#define VO_LIFT(k) pair[k] = perm_b32(hi, lo, VO_SEL_PIX(k));
    VO_LIFT(0) VO_LIFT(1) VO_LIFT(2) VO_LIFT(3) VO_LIFT(4) VO_LIFT(5) VO_LIFT(6)
#undef VO_LIFT
}

VO_HD void blend7(const uint32_t pt[7], const uint32_t pb[7], uint32_t wt, uint32_t wb, uint32_t out[4])
{
    uint32_t acc[8];
    if (!(wb & 0x80000000u)) {
#pragma unroll
        for (int k = 0; k < 7; k++)
            acc[k] = udot2(pb[k], wb, udot2(pt[k], wt, 1u << 16));
    } else {
        const uint32_t wb0 = wb & 0xffffu, kneg = (uint32_t)(-(int32_t)((int16_t)(wb >> 16)));
#pragma unroll
        for (int k = 0; k < 7; k++)
            acc[k] = udot2(pb[k], wb0, udot2(pt[k], wt, 1u << 16)) - kneg * (pb[k] >> 16);
    }
    acc[7] = 0;
#pragma unroll
    for (int m = 0; m < 4; m++)
        out[m] = pk_lshr1_u16(perm_b32(acc[2 * m + 1], acc[2 * m], VO_SEL_HI16));
}

// Scharr samples: d[k] = (4*Ix | 4*Iy << 16) of pixel x+k, rows top / bottom, k = 0..7.
// ix[m] = (Ixval[2m], Ixval[2m+1]), iy likewise; *val[k] = DESCALE(sum d*iw, 14) of the true derivative.
VO_HD void bilinear7_deriv(const uint32_t dt[8], const uint32_t db[8], uint32_t wt, uint32_t wb, uint32_t ix[4],
                           uint32_t iy[4])
{
    int32_t ax[8], ay[8];
#pragma unroll
    for (int k = 0; k < 7; k++) {
        ax[k] = sdot2(perm_b32(db[k + 1], db[k], VO_SEL_LO16), wb,
                      sdot2_first(perm_b32(dt[k + 1], dt[k], VO_SEL_LO16), wt, 1 << 15));
        ay[k] = sdot2(perm_b32(db[k + 1], db[k], VO_SEL_HI16), wb,
                      sdot2_first(perm_b32(dt[k + 1], dt[k], VO_SEL_HI16), wt, 1 << 15));
    }
    ax[7] = ay[7] = 0;
#pragma unroll
    for (int m = 0; m < 4; m++) {
        ix[m] = perm_b32((uint32_t)ax[2 * m + 1], (uint32_t)ax[2 * m], VO_SEL_HI16);
        iy[m] = perm_b32((uint32_t)ay[2 * m + 1], (uint32_t)ay[2 * m], VO_SEL_HI16);
    }
}

// Scharr 3x3 (cv::calcSharrDeriv) of the centre pixel of a 3x3 patch, stored pre-multiplied by 4
VO_HD uint32_t scharr4_packed(int p00, int p01, int p02, int p10, int p12, int p20, int p21, int p22)
{
    // t0(x) = (row-1 + row+1)*3 + row*10 ; t1(x) = row+1 - row-1 ; Ix = t0(x+1) - t0(x-1) ;
    // Iy = (t1(x+1) + t1(x-1))*3 + t1(x)*10
    int ix = ((p02 + p22) * 3 + p12 * 10) - ((p00 + p20) * 3 + p10 * 10);
    int iy = ((p22 - p02) + (p20 - p00)) * 3 + (p21 - p01) * 10;
    return ((uint32_t)(ix * 4) & 0xffffu) | ((uint32_t)(iy * 4) << 16);
}

} // namespace vo
