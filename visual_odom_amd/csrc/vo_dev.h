// vo_dev.h -- device-side data layout shared by the HIP kernels and the C-ABI host code.
//
// HBM layout (see DESIGN.md "Data layout"):
//   * image table: every 8-bit image owns one allocation holding its whole pyramid, level l at a
//     16-byte aligned offset with a row stride padded to a multiple of 16 bytes (so tile loads are
//     aligned dword/dwordx4 loads); described by a PyrImage record.
//   * a "frame" (one stereo pair at t0 and t1 = the unit of work of circularMatching(),
//     reference feature.h:61-65) is a Quad of four image-table indices (l0, r0, l1, r1).
//   * per frame SoA feature arrays with a fixed capacity `cap`: float2 points, u8 status.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define VO_MAX_LEVELS 5

namespace vo {

struct PyrImage {
    uint8_t *lvl[VO_MAX_LEVELS];
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

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v)
{
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// exact wave-wide sum of per-lane int32 partials, returned to every lane as int64
__device__ __forceinline__ long long wave_sum_i64(int v)
{
    long long s = v;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        s += __shfl_xor(s, m, 64);
    return s;
}

} // namespace vo
