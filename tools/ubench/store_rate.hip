// store_rate.hip -- what does the Scharr image's write pattern cost?  (round 4, pyramid stage)
// The pyramid stage writes one dword per pyramid pixel (4 Ix | 4 Iy << 16): 1.27 GB per 514 KITTI images for level 0 alone,
// 80 % of the stage's traffic.  Both the three-kernel chain (scharr_nt_kernel) and the fused pass write it as 32 bytes per
// lane and row (two 16-byte non-temporal stores) and both run at ~2.7-3.3 TB/s.  Is that the chip or the pattern?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/store_rate.hip -o /tmp/store_rate && /tmp/store_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define G __attribute__((address_space(1)))

// V1: 32 B per lane: two adjacent 16-B stores (lane stride 32 B)
template <bool NT> __global__ __launch_bounds__(256) void v1(uint32_t *out, size_t n16)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 >= n16) return;
    const u32x4 a = {(uint32_t)i, 1, 2, 3}, b = {(uint32_t)i, 5, 6, 7};
    G u32x4 *o = (G u32x4 *)out + i;
    if (NT) { __builtin_nontemporal_store(a, o); __builtin_nontemporal_store(b, o + 1); }
    else { o[0] = a; o[1] = b; }
}
// V2: 16 B per lane, lanes contiguous (1 KB per wave instruction); each thread two stores 256 lanes apart
template <bool NT> __global__ __launch_bounds__(256) void v2(uint32_t *out, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
    if (i + 256 >= n16) return;
    const u32x4 a = {(uint32_t)i, 1, 2, 3}, b = {(uint32_t)i, 5, 6, 7};
    G u32x4 *o = (G u32x4 *)out + i;
    if (NT) { __builtin_nontemporal_store(a, o); __builtin_nontemporal_store(b, o + 256); }
    else { o[0] = a; o[256] = b; }
}
// V3: the walker: a thread writes 32 B per row for ROWS rows of pitch `pitch16` (16-byte units), lanes adjacent in a row
template <int ROWS> __global__ __launch_bounds__(256) void v3(uint32_t *out, int groups_per_row, int pitch16, int n_blocks_rows)
{
    const int item = blockIdx.x * 256 + threadIdx.x;
    const int b = item / groups_per_row, g = item - b * groups_per_row;
    if (b >= n_blocks_rows) return;
    G u32x4 *o = (G u32x4 *)out + ((size_t)blockIdx.y * n_blocks_rows * ROWS + (size_t)b * ROWS) * pitch16 + 2 * g;
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        const u32x4 a = {(uint32_t)item, 1, 2, (uint32_t)r}, c = {(uint32_t)item, 5, 6, 7};
        __builtin_nontemporal_store(a, o + (size_t)r * pitch16);
        __builtin_nontemporal_store(c, o + (size_t)r * pitch16 + 1);
    }
}
// read + write: 1 byte read per dword written (the pass's ratio), 16-B loads
__global__ __launch_bounds__(256) void v4(const uint32_t *in, uint32_t *out, size_t n16)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 >= n16) return;
    const uint64_t s = ((const G uint64_t *)in)[i / 2];
    const u32x4 a = {(uint32_t)s, 1, 2, 3}, b = {(uint32_t)(s >> 32), 5, 6, 7};
    G u32x4 *o = (G u32x4 *)out + i;
    __builtin_nontemporal_store(a, o);
    __builtin_nontemporal_store(b, o + 1);
}

template <typename F> static float timeit(F launch, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main()
{
    const int W = 1241, H = 376, NI = 514, pitch16 = 1312 / 4; // a KITTI level 0: 1312-pixel pitch = 328 16-byte units of dwords
    const size_t n16 = (size_t)NI * H * pitch16;                // 16-byte units
    uint32_t *out, *in;
    hipMalloc((void **)&out, n16 * 16 + (1 << 20));
    hipMalloc((void **)&in, n16 * 4 + (1 << 20));
    hipMemset(in, 1, n16 * 4);
    const double gb = n16 * 16 / 1e9;
    printf("bytes written per launch: %.3f GB (514 x 376 rows x 1312 dwords)\n", gb);
    float t;
    t = timeit([&] { hipLaunchKernelGGL(v1<true>, dim3((n16 / 2 + 255) / 256), dim3(256), 0, 0, out, n16); }, 20);
    printf("V1 nt   32 B per lane (lane stride 32 B)      %.3f ms  %.2f TB/s\n", t, gb / t);
    t = timeit([&] { hipLaunchKernelGGL(v1<false>, dim3((n16 / 2 + 255) / 256), dim3(256), 0, 0, out, n16); }, 20);
    printf("V1 plain                                       %.3f ms  %.2f TB/s\n", t, gb / t);
    t = timeit([&] { hipLaunchKernelGGL(v2<true>, dim3((n16 + 511) / 512), dim3(256), 0, 0, out, n16); }, 20);
    printf("V2 nt   16 B per lane, wave-contiguous 1 KB    %.3f ms  %.2f TB/s\n", t, gb / t);
    t = timeit([&] { hipLaunchKernelGGL(v2<false>, dim3((n16 + 511) / 512), dim3(256), 0, 0, out, n16); }, 20);
    printf("V2 plain                                       %.3f ms  %.2f TB/s\n", t, gb / t);
    {
        const int gpr = (W + 7) / 8;
        auto run = [&](auto kern, int rows) {
            const int nbr = H / rows;
            const double g2 = (double)NI * nbr * rows * gpr * 32 / 1e9;
            float tt = timeit([&] { hipLaunchKernelGGL(kern, dim3((nbr * gpr + 255) / 256, NI), dim3(256), 0, 0, out, gpr, pitch16, nbr); }, 20);
            printf("V3 walker %2d rows x 32 B per lane              %.3f ms  %.2f TB/s\n", rows, tt, g2 / tt);
        };
        run(v3<4>, 4); run(v3<8>, 8); run(v3<16>, 16);
    }
    t = timeit([&] { hipLaunchKernelGGL(v4, dim3((n16 / 2 + 255) / 256), dim3(256), 0, 0, in, out, n16); }, 20);
    printf("V4 nt + 8-byte load per 32 B written           %.3f ms  %.2f TB/s (stores only)\n", t, gb / t);
    t = timeit([&] { hipMemsetAsync(out, 0, n16 * 16, 0); }, 10);
    printf("hipMemsetAsync                                 %.3f ms  %.2f TB/s\n", t, gb / t);
    return 0;
}
