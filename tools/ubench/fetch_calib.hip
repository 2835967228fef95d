// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE on gfx950 for the access widths the LK kernel
// uses (MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of a wide coalesced 16 B/lane stream;
// other patterns must be calibrated on a known byte count).  Every kernel reads each byte of a 1 GiB
// buffer (4x the 256 MiB Infinity Cache) exactly once:
//   stream16   coalesced 16 B per lane                      (the guide's case)
//   stream8    coalesced 8 B per lane                       (scharr_kernel's loads)
//   rows16     LK-like gather: lane l of a wave reads 16 B of row (l / 3) of a 1304-byte-pitch image at
//              column 28 * (l % 3), the wave then moves on 16 B -- 21 rows x 3 segments per request
// run:  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib ; FETCH_SIZE (KB) vs 1048576 KB
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void stream16(const uint4 *__restrict__ p, size_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678)
        out[0] = acc;
}

__global__ __launch_bounds__(256) void stream8(const uint2 *__restrict__ p, size_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint2 v = p[i];
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678)
        out[0] = acc;
}

// image of `rows` rows, pitch 1344 bytes (84 x 16); one wave covers a 21-row x 84-byte band 16 B at a time
__global__ __launch_bounds__(64) void rows16(const uint8_t *__restrict__ p, int rows, uint32_t *out)
{
    const int lane = threadIdx.x, r = lane / 3, s = lane % 3;
    const int band = blockIdx.x; // 21 rows each
    uint32_t acc = 0;
    if (lane < 63 && band * 21 + r < rows) {
        const uint8_t *row = p + (size_t)(band * 21 + r) * 1344;
        for (int c = 0; c < 28; c++) { // 3 segments x 28 x 16 B = 1344 B per row
            uint4 v = *reinterpret_cast<const uint4 *>(row + (s * 28 + c) * 16);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678)
        out[0] = acc;
}

int main()
{
    const size_t bytes = 1ull << 30;
    uint8_t *d;
    uint32_t *o;
    hipMalloc(&d, bytes);
    hipMalloc(&o, 4);
    hipMemset(d, 1, bytes);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, (const uint4 *)d, bytes / 16, o);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(stream8, dim3(4096), dim3(256), 0, 0, (const uint2 *)d, bytes / 8, o);
    hipDeviceSynchronize();
    const int rows = (int)(bytes / 1344);
    hipLaunchKernelGGL(rows16, dim3((rows + 20) / 21), dim3(64), 0, 0, d, rows, o);
    hipDeviceSynchronize();
    printf("bytes read by each kernel: %zu (rows16: %zu)\n", bytes, (size_t)rows * 1344);
    return 0;
}
