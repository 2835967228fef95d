// What slows the lock-step loop's host ingest when it runs under LK?  (round 6)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/ingest_under_load tools/ubench/ingest_under_load.hip && /tmp/ingest_under_load
// 256 sequences at the 2 000-point load: 13.9 ms per step with page-locked host pairs against 10.0 with resident ones --
// the 239 MB of a step cross the link at 17 GB/s where the 340-point load (LK 2 ms) gets 39 GB/s, whatever the stream
// priority and whatever the ingest kernel's shape (gpurun_out/r6_ingab).  This probe separates the candidates:
//   load  = a stand-in for LK: one-wave workgroups, ~71 VGPRs, ~94 SGPRs (7 waves per SIMD), pure VALU, ~10 ms
//   copy  = the ingest kernel's access pattern: one wave per 1241-byte row, 8 B per lane, page-locked host memory -> HBM
// Cases: copy alone | under load, equal priority | under load, copy at the highest priority | CU masks: load on all but R
// CUs, copy on those R CUs (R = 8, 16, 32) | copy through the copy ENGINE instead (one hipMemcpy2DAsync per image) under load
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(64, 7) void load_kernel(float *out, int iters)
{
    // 64 live accumulators keep ~70 VGPRs busy; wave-uniform scalars inflate the SGPR count
    float a[64];
#pragma unroll
    for (int k = 0; k < 64; k++)
        a[k] = (float)(threadIdx.x + k);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++)
            a[k] = a[k] * 1.0001f + a[(k + 1) & 63];
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 64; k++)
        s += a[k];
    if (s == 123.456f)
        out[blockIdx.x] = s;
}

// the same with LK's memory behaviour: every few hundred VALU instructions the wave fetches a fresh 2 KB window (16 B per
// lane, two requests) from a 256 MB buffer at a pseudo-random place and waits for it -- latency-sensitive loads through the L2
__global__ __launch_bounds__(64, 7) void load_mem_kernel(float *out, const uint4 *__restrict__ buf, uint32_t n16, int iters)
{
    float a[48];
#pragma unroll
    for (int k = 0; k < 48; k++)
        a[k] = (float)(threadIdx.x + k);
    uint32_t pos = blockIdx.x * 2654435761u;
    for (int i = 0; i < iters; i++) {
        pos = pos * 1664525u + 1013904223u;
        const uint32_t base = (pos % (n16 - 256)) & ~63u;
        const uint4 u = buf[base + threadIdx.x], v = buf[base + 64 + threadIdx.x];
        const float f = (float)((u.x ^ v.y) & 1023u) * 1e-9f;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int k = 0; k < 48; k++)
                a[k] = a[k] * 1.0001f + a[(k + 1) % 48] + f;
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 48; k++)
        s += a[k];
    if (s == 123.456f)
        out[blockIdx.x] = s;
}

struct __attribute__((packed, aligned(1))) U2 {
    uint32_t lo, hi;
};
__global__ __launch_bounds__(64) void copy_rows(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int h, int sstride,
                                                int pitch, size_t simg, size_t dimg)
{
    const uint8_t *s = src + (size_t)blockIdx.y * simg + (size_t)blockIdx.x * sstride;
    uint8_t *d = dst + (size_t)blockIdx.y * dimg + (size_t)blockIdx.x * pitch;
    const int last = w - 8;
    for (int x0 = 0; x0 < w; x0 += 512) {
        int x = x0 + (int)threadIdx.x * 8;
        if (x < w) {
            x = x < last ? x : last;
            *reinterpret_cast<U2 *>(d + x) = *reinterpret_cast<const U2 *>(s + x);
        }
    }
}

// persistent form: G single-wave workgroups walk over all rows (row = blockIdx.x, + gridDim.x, ...): at most G copy waves are
// ever resident, however long each row's PCIe round trips queue
__global__ __launch_bounds__(64) void copy_rows_persistent(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int h,
                                                           int sstride, int pitch, size_t simg, size_t dimg, int n_rows)
{
    const int last = w - 8;
    for (int r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const int img = r / h, row = r - img * h;
        const uint8_t *s = src + (size_t)img * simg + (size_t)row * sstride;
        uint8_t *d = dst + (size_t)img * dimg + (size_t)row * pitch;
        for (int x0 = 0; x0 < w; x0 += 512) {
            int x = x0 + (int)threadIdx.x * 8;
            if (x < w) {
                x = x < last ? x : last;
                *reinterpret_cast<U2 *>(d + x) = *reinterpret_cast<const U2 *>(s + x);
            }
        }
    }
}

int main()
{
    const int w = 1241, h = 376, pitch = 1312, n_img = 512;
    const size_t simg = (size_t)w * h, dimg = (size_t)pitch * h;
    uint8_t *hsrc, *ddst;
    float *dout;
    CK(hipHostMalloc((void **)&hsrc, simg * n_img, hipHostMallocDefault));
    for (size_t i = 0; i < simg * n_img; i += 4096)
        hsrc[i] = (uint8_t)i;
    CK(hipMalloc((void **)&ddst, dimg * n_img));
    CK(hipMalloc((void **)&dout, 4 << 20));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t s_load, s_copy, s_copy_hi;
    CK(hipStreamCreateWithFlags(&s_load, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_copy, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&s_copy_hi, hipStreamNonBlocking, greatest));
    hipEvent_t e0, e1, l0, l1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreate(&l0));
    CK(hipEventCreate(&l1));
    const int load_blocks = 522240, load_iters = 40; // tuned below to ~10 ms
    uint4 *dbuf;
    const uint32_t n16 = 16u << 20; // 256 MB
    CK(hipMalloc((void **)&dbuf, (size_t)n16 * 16));
    CK(hipMemset(dbuf, 1, (size_t)n16 * 16));
    bool mem_load = false;
    auto launch_load = [&](hipStream_t st, int iters) {
        if (mem_load)
            hipLaunchKernelGGL(load_mem_kernel, dim3(load_blocks), dim3(64), 0, st, dout, dbuf, n16, iters);
        else
            hipLaunchKernelGGL(load_kernel, dim3(load_blocks), dim3(64), 0, st, dout, iters);
    };
    auto launch_copy = [&](hipStream_t st) {
        hipLaunchKernelGGL(copy_rows, dim3(h, n_img), dim3(64), 0, st, hsrc, ddst, w, h, w, pitch, simg, dimg);
    };
    int iters = 0;
    uint8_t *dstage;
    CK(hipMalloc((void **)&dstage, simg * n_img));
    for (int pass = 0; pass < 2; pass++) {
    mem_load = pass == 1;
    // calibrate the load to ~10 ms
    iters = load_iters;
    for (int tries = 0; tries < 6; tries++) {
        CK(hipEventRecord(l0, s_load));
        launch_load(s_load, iters);
        CK(hipEventRecord(l1, s_load));
        CK(hipStreamSynchronize(s_load));
        float ms;
        CK(hipEventElapsedTime(&ms, l0, l1));
        if (tries == 5)
            printf("\n=== load kernel %s: %d iterations = %.2f ms alone (%d CUs)\n", mem_load ? "WITH window fetches through the L2" : "pure VALU", iters, ms, n_cu);
        iters = std::max(1, (int)(iters * 10.0f / ms));
    }
    int persistent = 0; // > 0: the persistent copy kernel with that many workgroups
    auto measure = [&](const char *name, hipStream_t sl, hipStream_t sc, bool with_load, bool engine) -> int {
        std::vector<float> tc, tl;
        for (int it = 0; it < 6; it++) {
            if (with_load) {
                CK(hipEventRecord(l0, sl));
                launch_load(sl, iters);
                CK(hipEventRecord(l1, sl));
            }
            CK(hipEventRecord(e0, sc));
            if (engine) {
                for (int i = 0; i < n_img; i++)
                    CK(hipMemcpy2DAsync(ddst + (size_t)i * dimg, pitch, hsrc + (size_t)i * simg, w, w, h, hipMemcpyHostToDevice, sc));
            } else if (persistent > 0) {
                hipLaunchKernelGGL(copy_rows_persistent, dim3(persistent), dim3(64), 0, sc, hsrc, ddst, w, h, w, pitch, simg, dimg, h * n_img);
            } else {
                launch_copy(sc);
            }
            CK(hipEventRecord(e1, sc));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            tc.push_back(ms);
            if (with_load) {
                CK(hipEventElapsedTime(&ms, l0, l1));
                tl.push_back(ms);
            }
        }
        std::sort(tc.begin(), tc.end());
        std::sort(tl.begin(), tl.end());
        const double mb = (double)simg * n_img / 1e6;
        printf("%-58s copy %6.2f ms = %5.1f GB/s", name, tc[tc.size() / 2], mb / tc[tc.size() / 2]);
        if (with_load)
            printf("   load %6.2f ms", tl[tl.size() / 2]);
        printf("\n");
        return 0;
    };
    measure("copy alone", s_load, s_copy, false, false);
    measure("under load, equal priority", s_load, s_copy, true, false);
    measure("under load, copy at the highest priority", s_load, s_copy_hi, true, false);
    for (int G : {256, 1024, 4096}) {
        persistent = G;
        char name[96];
        snprintf(name, sizeof(name), "persistent copy, %d waves, alone", G);
        measure(name, s_load, s_copy_hi, false, false);
        snprintf(name, sizeof(name), "persistent copy, %d waves, under load, highest priority", G);
        measure(name, s_load, s_copy_hi, true, false);
        snprintf(name, sizeof(name), "persistent copy, %d waves, under load, equal priority", G);
        measure(name, s_load, s_copy, true, false);
    }
    persistent = 0;
    {   // the copy ENGINE: one linear hipMemcpyAsync per image into an unpitched staging buffer (a kernel would re-pitch it D2D)
        std::vector<float> tc, tl, th;
        for (int it = 0; it < 6; it++) {
            CK(hipEventRecord(l0, s_load));
            launch_load(s_load, iters);
            CK(hipEventRecord(l1, s_load));
            CK(hipEventRecord(e0, s_copy));
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n_img; i++)
                CK(hipMemcpyAsync(dstage + (size_t)i * simg, hsrc + (size_t)i * simg, simg, hipMemcpyHostToDevice, s_copy));
            th.push_back(std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count());
            CK(hipEventRecord(e1, s_copy));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            tc.push_back(ms);
            CK(hipEventElapsedTime(&ms, l0, l1));
            tl.push_back(ms);
        }
        std::sort(tc.begin(), tc.end());
        std::sort(tl.begin(), tl.end());
        std::sort(th.begin(), th.end());
        printf("%-58s copy %6.2f ms = %5.1f GB/s   load %6.2f ms   (host: %.2f ms to enqueue)\n", "under load, copy engine: 512 linear hipMemcpyAsync",
               tc[3], (double)simg * n_img / 1e6 / tc[3], tl[3], th[3]);
        for (int ns : {2, 4, 8}) { // the 512 linear copies spread over ns streams: does the runtime use several copy engines?
            static hipStream_t cs[8] = {};
            static hipEvent_t ce[8] = {};
            for (int i = 0; i < ns; i++)
                if (!cs[i]) {
                    CK(hipStreamCreateWithFlags(&cs[i], hipStreamNonBlocking));
                    CK(hipEventCreateWithFlags(&ce[i], hipEventDisableTiming));
                }
            for (int it = 0; it < 6; it++) {
                CK(hipEventRecord(l0, s_load));
                launch_load(s_load, iters);
                CK(hipEventRecord(l1, s_load));
                CK(hipEventRecord(e0, s_copy));
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < ns; i++)
                    CK(hipStreamWaitEvent(cs[i], e0, 0));
                for (int i = 0; i < n_img; i++)
                    CK(hipMemcpyAsync(dstage + (size_t)i * simg, hsrc + (size_t)i * simg, simg, hipMemcpyHostToDevice, cs[i % ns]));
                for (int i = 0; i < ns; i++) {
                    CK(hipEventRecord(ce[i], cs[i]));
                    CK(hipStreamWaitEvent(s_copy, ce[i], 0));
                }
                th[it] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
                CK(hipEventRecord(e1, s_copy));
                CK(hipDeviceSynchronize());
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                tc[it] = ms;
                CK(hipEventElapsedTime(&ms, l0, l1));
                tl[it] = ms;
            }
            std::sort(tc.begin(), tc.end());
            std::sort(tl.begin(), tl.end());
            std::sort(th.begin(), th.end());
            char name[96];
            snprintf(name, sizeof(name), "under load, copy engine: 512 linear copies on %d streams", ns);
            printf("%-58s copy %6.2f ms = %5.1f GB/s   load %6.2f ms   (host: %.2f ms to enqueue)\n", name, tc[3],
                   (double)simg * n_img / 1e6 / tc[3], tl[3], th[3]);
        }
        {   // ONE hipMemcpyBatchAsync of the 512 images
            std::vector<void *> dsts(n_img), srcs(n_img);
            std::vector<size_t> sizes(n_img, simg);
            for (int i = 0; i < n_img; i++) {
                dsts[i] = dstage + (size_t)i * simg;
                srcs[i] = hsrc + (size_t)i * simg;
            }
            bool ok = true;
            for (int it = 0; it < 6 && ok; it++) {
                CK(hipEventRecord(l0, s_load));
                launch_load(s_load, iters);
                CK(hipEventRecord(l1, s_load));
                CK(hipEventRecord(e0, s_copy));
                size_t fail = 0;
                const auto t0 = std::chrono::steady_clock::now();
                hipError_t e = hipMemcpyBatchAsync(dsts.data(), srcs.data(), sizes.data(), (size_t)n_img, nullptr, nullptr, 0, &fail, s_copy);
                th[it] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (e != hipSuccess) {
                    printf("hipMemcpyBatchAsync: %s (fail index %zu)\n", hipGetErrorString(e), fail);
                    (void)hipGetLastError();
                    ok = false;
                }
                CK(hipEventRecord(e1, s_copy));
                CK(hipDeviceSynchronize());
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                tc[it] = ms;
                CK(hipEventElapsedTime(&ms, l0, l1));
                tl[it] = ms;
            }
            if (ok) {
                std::sort(tc.begin(), tc.end());
                std::sort(tl.begin(), tl.end());
                std::sort(th.begin(), th.end());
                printf("%-58s copy %6.2f ms = %5.1f GB/s   load %6.2f ms   (host: %.2f ms to enqueue)\n", "under load, copy engine: ONE hipMemcpyBatchAsync x 512",
                       tc[3], (double)simg * n_img / 1e6 / tc[3], tl[3], th[3]);
            }
        }
        for (int it = 0; it < 6; it++) {
            CK(hipEventRecord(l0, s_load));
            launch_load(s_load, iters);
            CK(hipEventRecord(l1, s_load));
            CK(hipEventRecord(e0, s_copy));
            CK(hipMemcpyAsync(dstage, hsrc, simg * n_img, hipMemcpyHostToDevice, s_copy));
            CK(hipEventRecord(e1, s_copy));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            tc[it] = ms;
            CK(hipEventElapsedTime(&ms, l0, l1));
            tl[it] = ms;
        }
        std::sort(tc.begin(), tc.end());
        std::sort(tl.begin(), tl.end());
        printf("%-58s copy %6.2f ms = %5.1f GB/s   load %6.2f ms\n", "under load, copy engine: ONE 239 MB hipMemcpyAsync", tc[3],
               (double)simg * n_img / 1e6 / tc[3], tl[3]);
    }
    for (int R : {8}) {
        const size_t words = (size_t)(n_cu + 31) / 32;
        std::vector<uint32_t> mload(words, 0u), mcopy(words, 0u);
        for (int cu = 0; cu < n_cu; cu++) // every (n_cu / R)-th CU goes to the copy side: spread over the XCDs whatever the numbering
            ((cu % (n_cu / R)) == 0 ? mcopy : mload)[(size_t)cu >> 5] |= 1u << (cu & 31);
        hipStream_t ml, mc;
        CK(hipExtStreamCreateWithCUMask(&ml, (uint32_t)words, mload.data()));
        CK(hipExtStreamCreateWithCUMask(&mc, (uint32_t)words, mcopy.data()));
        char name[96];
        snprintf(name, sizeof(name), "under load, CU masks: copy on %d CUs, load on %d", R, n_cu - R);
        measure(name, ml, mc, true, false);
        snprintf(name, sizeof(name), "  (copy alone on %d CUs)", R);
        measure(name, ml, mc, false, false);
        CK(hipStreamDestroy(ml));
        CK(hipStreamDestroy(mc));
    }
    }
    return 0;
}
