// wave_sums_gpu.hip -- TEST ONLY: runs the exact wave reductions of vo_dev.h (v_permlane32/16_swap + DPP trees) on
// the GPU with per-lane partials at the documented bound and compares with (float)(int64 sum) computed on the host.
// Built and run by tests/test_gpu_parity.py::test_exact_wave_sums_on_the_gpu with hipcc on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../visual_odom_amd/csrc/vo_dev.h"

__global__ void sums_kernel(const int *a, const int *b, const int *c, float *out, int n_cases)
{
    const int k = blockIdx.x, l = threadIdx.x;
    if (k >= n_cases)
        return;
    float s0, s1, t0, t1, t2;
    vo::wave_sum2_exact_f32(a[k * 64 + l], b[k * 64 + l], s0, s1);
    vo::wave_sum3_exact_f32(a[k * 64 + l], b[k * 64 + l], c[k * 64 + l], t0, t1, t2);
    if (l == 0) {
        out[k * 5 + 0] = s0;
        out[k * 5 + 1] = s1;
        out[k * 5 + 2] = t0;
        out[k * 5 + 3] = t1;
        out[k * 5 + 4] = t2;
    }
}

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}

int main()
{
    const int N = 4096, B = (1 << 28) - 1;
    int *ha = (int *)malloc(sizeof(int) * 64 * N), *hb = (int *)malloc(sizeof(int) * 64 * N),
        *hc = (int *)malloc(sizeof(int) * 64 * N);
    for (int k = 0; k < N; k++)
        for (int l = 0; l < 64; l++) {
            int v[3];
            for (int j = 0; j < 3; j++) {
                const int mode = (k + j) % 5;
                if (mode == 0)
                    v[j] = B;
                else if (mode == 1)
                    v[j] = -B;
                else if (mode == 2)
                    v[j] = (l & 1) ? B : -B + (int)(rnd() % 3);
                else
                    v[j] = (int)(rnd() % (2u * B + 1u)) - B;
            }
            ha[k * 64 + l] = v[0];
            hb[k * 64 + l] = v[1];
            hc[k * 64 + l] = v[2];
        }
    int *da, *db, *dc;
    float *dout, *hout = (float *)malloc(sizeof(float) * 5 * N);
    if (hipMalloc(&da, sizeof(int) * 64 * N) != hipSuccess || hipMalloc(&db, sizeof(int) * 64 * N) != hipSuccess ||
        hipMalloc(&dc, sizeof(int) * 64 * N) != hipSuccess || hipMalloc(&dout, sizeof(float) * 5 * N) != hipSuccess) {
        printf("FAIL: no device memory\n");
        return 2;
    }
    hipMemcpy(da, ha, sizeof(int) * 64 * N, hipMemcpyHostToDevice);
    hipMemcpy(db, hb, sizeof(int) * 64 * N, hipMemcpyHostToDevice);
    hipMemcpy(dc, hc, sizeof(int) * 64 * N, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sums_kernel, dim3(N), dim3(64), 0, 0, da, db, dc, dout, N);
    if (hipMemcpy(hout, dout, sizeof(float) * 5 * N, hipMemcpyDeviceToHost) != hipSuccess) {
        printf("FAIL: kernel\n");
        return 2;
    }
    int bad = 0;
    for (int k = 0; k < N; k++) {
        long long sa = 0, sb = 0, sc = 0;
        for (int l = 0; l < 64; l++) {
            sa += ha[k * 64 + l];
            sb += hb[k * 64 + l];
            sc += hc[k * 64 + l];
        }
        const float ref[5] = {(float)sa, (float)sb, (float)sa, (float)sb, (float)sc};
        for (int j = 0; j < 5; j++)
            if (memcmp(&ref[j], &hout[k * 5 + j], 4) != 0) {
                if (bad < 5)
                    printf("case %d out %d: got %.9g want %.9g\n", k, j, hout[k * 5 + j], ref[j]);
                bad++;
            }
    }
    printf(bad ? "FAIL: %d mismatches\n" : "OK %d\n", bad ? bad : 5 * N);
    return bad ? 1 : 0;
}
