// pass_bench.hip -- the product's fused pyramid pass (pyramid.hip, included as is) on a 514-image KITTI table, level by level,
// outside the library: no tracking kernel, no pose chain, no Python.  Build variants knock parts of the pass out
// (-DVO_PASS_X=bits: 1 no border items, 2 no next-level stores, 4 no edge items; results are wrong then, the time is the point)
// to see what each part costs (profiles/r04_experiments.md section 5).
//   hipcc --offload-arch=gfx950 -O3 -DVO_DEV_VARIANTS -Iinclude -Ivisual_odom_amd/csrc tools/ubench/pass_bench.hip -o /tmp/pass_bench
#include "../../visual_odom_amd/csrc/pyramid.hip"
#include <stdio.h>
#include <vector>
#include <string.h>

using namespace vo;
#ifndef PASS_BENCH_REPS
#define PASS_BENCH_REPS 20 // launches per figure (3 under a counter pass: tools/pass_pmc.sh)
#endif

template <typename F> static float timeit(F launch, int reps)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; i++) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 1241, H = argc > 2 ? atoi(argv[2]) : 376, NI = argc > 3 ? atoi(argv[3]) : 514;
    int lw[VO_MAX_LEVELS], lh[VO_MAX_LEVELS], ls[VO_MAX_LEVELS];
    size_t loff[VO_MAX_LEVELS], off = 0;
    int L = 0;
    for (int cw = W, ch = H;; L++) { // (plan_levels, capi.hip)
        lw[L] = cw; lh[L] = ch; ls[L] = (VO_BX + cw + VO_BY + 15) / 16 * 16; loff[L] = off;
        off += (size_t)ls[L] * (ch + 2 * VO_BY);
        off = (off + 255) / 256 * 256;
        const int nw = (cw + 1) / 2, nh = (ch + 1) / 2;
        if (L == 3 || nw <= 21 || nh <= 21) break;
        cw = nw; ch = nh;
    }
    L++;
    const size_t img_bytes = off + (argc > 4 ? (size_t)atol(argv[4]) : 0); // (4th argument: padding between images, bytes -- channel-aliasing experiment)
    uint8_t *pix; uint32_t *der; PyrImage *d_imgs;
    (void)hipMalloc((void **)&pix, img_bytes * NI);
    (void)hipMalloc((void **)&der, img_bytes * NI * 4);
    (void)hipMalloc((void **)&d_imgs, sizeof(PyrImage) * NI);
    std::vector<uint8_t> host(img_bytes * NI);
    uint32_t s = 12345;
    for (auto &b : host) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
    (void)hipMemcpy(pix, host.data(), host.size(), hipMemcpyHostToDevice);
    std::vector<PyrImage> tab(NI);
    for (int i = 0; i < NI; i++) {
        tab[i] = PyrImage{};
        for (int l = 0; l < L; l++) {
            const size_t org = (size_t)i * img_bytes + loff[l] + (size_t)VO_BY * ls[l] + VO_BX;
            tab[i].lvl[l] = pix + org; tab[i].der[l] = der + org;
            tab[i].w[l] = lw[l]; tab[i].h[l] = lh[l]; tab[i].stride[l] = ls[l];
        }
    }
    (void)hipMemcpy(d_imgs, tab.data(), sizeof(PyrImage) * NI, hipMemcpyHostToDevice);
    const PassPlan pp = pass_plan(L, lw, lh, ls);
#ifndef VO_PASS_X
#define VO_PASS_X 0
#endif
    printf("%d images %d x %d, %d levels, VO_PASS_X=%d, rows per item %d, image pitch %zu B (x 4 for the Scharr images)\n", NI, W, H, L, VO_PASS_X, PF_ROWS, img_bytes);
    float tot[4] = {0, 0, 0, 0};
    for (int l = 0; l < L; l++) {
        const uint32_t g1 = pass_grid(pp, l, NI, 1), g0 = pass_grid(pp, l, NI, 0);
        const float t0 = timeit([&] { hipLaunchKernelGGL(pyr_pass_kernel, dim3(g1), dim3(64), 0, 0, d_imgs, l, L, pp, (uint32_t)NI, 1); }, PASS_BENCH_REPS);
        const float t1 = timeit([&] { hipLaunchKernelGGL(pyr_pass_sm_kernel<1>, dim3(g1), dim3(64), 0, 0, d_imgs, l, L, pp, (uint32_t)NI, 1); }, PASS_BENCH_REPS);
        const float t2 = timeit([&] { hipLaunchKernelGGL(pyr_pass_sm_kernel<2>, dim3(g1), dim3(64), 0, 0, d_imgs, l, L, pp, (uint32_t)NI, 1); }, PASS_BENCH_REPS);
        const float t3 = timeit([&] { hipLaunchKernelGGL(pyr_pass_kernel, dim3(g0), dim3(64), 0, 0, d_imgs, l, L, pp, (uint32_t)NI, 0); }, PASS_BENCH_REPS);
        printf("  level %d (%4d x %3d): %d x %d workgroups per image  non-temporal %6.1f us   ordinary stores %6.1f   no Scharr stores %6.1f   dispatch order (image not pinned to an XCD) %6.1f\n", l, lw[l], lh[l],
               pp.nci[l], pp.gy[l], t0 * 1e3, t1 * 1e3, t2 * 1e3, t3 * 1e3);
        tot[0] += t0; tot[1] += t1; tot[2] += t2; tot[3] += t3;
    }
    printf("  all levels: %.1f / %.1f / %.1f / %.1f us\n", tot[0] * 1e3, tot[1] * 1e3, tot[2] * 1e3, tot[3] * 1e3);
    return 0;
}
