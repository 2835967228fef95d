// How long do the four image uploads of a synchronous call take, and what shortens them?  (round 5)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/h2d_probe tools/ubench/h2d_probe.hip && /tmp/h2d_probe
// Cases (4 x 493 KB pinned -> device, time from the first enqueue to completion, median of 200):
//   A one stream, four hipMemcpyAsync            B two streams (0,1 | 2,3) + event join
//   C four streams + joins                       D one kernel reading the pinned buffers directly (zero-copy, 16 B / lane)
//   E one hipMemcpyAsync of all four (1.97 MB)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void pull_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}
__global__ void tiny_kernel(int *p) { if (p) *p = 1; }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t img = 1296 * 375 + 1241; // pitched KITTI level 0 as upload_image sends it
    const size_t img16 = (img + 15) / 16 * 16;
    uint8_t *h, *d, *hd;
    int *flag;
    CK(hipHostMalloc((void **)&h, img16 * 4, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&hd, h, 0));
    CK(hipMalloc((void **)&d, img16 * 4));
    CK(hipMalloc((void **)&flag, 4));
    for (size_t i = 0; i < img16 * 4; i++) h[i] = (uint8_t)(i * 7);
    hipStream_t s[4];
    hipEvent_t ev[4];
    for (int i = 0; i < 4; i++) { CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
    auto run = [&](int mode) -> double {
        std::vector<double> t;
        for (int it = 0; it < 220; it++) {
            CK(hipDeviceSynchronize());
            const double t0 = now_us();
            if (mode == 0) {
                for (int i = 0; i < 4; i++) CK(hipMemcpyAsync(d + i * img16, h + i * img16, img, hipMemcpyHostToDevice, s[0]));
            } else if (mode == 1) {
                for (int i = 0; i < 4; i++) CK(hipMemcpyAsync(d + i * img16, h + i * img16, img, hipMemcpyHostToDevice, s[i / 2]));
                CK(hipEventRecord(ev[1], s[1]));
                CK(hipStreamWaitEvent(s[0], ev[1], 0));
            } else if (mode == 2) {
                for (int i = 0; i < 4; i++) CK(hipMemcpyAsync(d + i * img16, h + i * img16, img, hipMemcpyHostToDevice, s[i]));
                for (int i = 1; i < 4; i++) { CK(hipEventRecord(ev[i], s[i])); CK(hipStreamWaitEvent(s[0], ev[i], 0)); }
            } else if (mode == 3) {
                hipLaunchKernelGGL(pull_kernel, dim3(256), dim3(256), 0, s[0], (const uint4 *)hd, (uint4 *)d, img16 * 4 / 16);
            } else if (mode == 4) {
                CK(hipMemcpyAsync(d, h, img16 * 4, hipMemcpyHostToDevice, s[0]));
            } else if (mode == 5) {
                hipLaunchKernelGGL(pull_kernel, dim3(1024), dim3(256), 0, s[0], (const uint4 *)hd, (uint4 *)d, img16 * 4 / 16);
            }
            hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s[0], flag); // what the call's first kernel would see
            CK(hipStreamSynchronize(s[0]));
            if (it >= 20) t.push_back(now_us() - t0);
        }
        std::sort(t.begin(), t.end());
        return t[t.size() / 2];
        return 0;
    };
    const char *names[] = {"A one stream, 4 copies", "B two streams + join", "C four streams + joins", "D zero-copy pull kernel (256 x 256)",
                           "E one 1.97 MB copy", "F zero-copy pull kernel (1024 x 256)"};
    for (int m = 0; m < 6; m++)
        printf("%-40s %7.1f us (enqueue -> a kernel behind it has run; host side included)\n", names[m], run(m));
    // reference: the tiny kernel alone
    {
        std::vector<double> t;
        for (int it = 0; it < 220; it++) {
            CK(hipDeviceSynchronize());
            const double t0 = now_us();
            hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s[0], flag);
            CK(hipStreamSynchronize(s[0]));
            if (it >= 20) t.push_back(now_us() - t0);
        }
        std::sort(t.begin(), t.end());
        printf("%-40s %7.1f us\n", "(launch + sync of the tiny kernel alone)", t[t.size() / 2]);
    }
    return 0;
}
