// mfma_reduce.hip -- paper gate of VERDICT r04 item 7, measured: the two exact wave sums of the LK iteration
// (wave_sum2_exact_f32, vo_dev.h: permlane32_swap / quad DPP / permlane16_swap / row DPP / readlane, 15 VALU, ~70 issue cycles
// of the iteration's ~324) against the same two sums on the MATRIX pipe:
//     v_cvt_f64_i32 x 2  ->  per sum: v_mfma_f64_4x4x4 (x ones), v_mfma_f64_4x4x4 (ones x), v_mfma_f64_16x16x4 (x ones)
//     -> v_cvt_f32_f64 x 2          (exact: integer-valued f64 sums below 2^53; the f32 rounding is the single one of
//                                    (float)(int64 sum), what the tree returns)
// inside a loop whose other ~63 VALU instructions have the iteration's opcode mix (profiles/r02_lk_issue_bound.md), at the
// LK kernel's occupancy.  Prints: correctness of the MFMA sums on random partials, registers per variant, ns per iteration.
//   hipcc --offload-arch=gfx950 -O3 mfma_reduce.hip -o mfma_reduce && ./mfma_reduce
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef double v4f64 __attribute__((ext_vector_type(4)));

#define DPP_QUAD_XOR1 0xB1
#define DPP_QUAD_XOR2 0x4E
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140
#define DPP_ROW_BCAST15 0x142

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add(int v) { return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true); }

// the product's tree (vo_dev.h)
__device__ __forceinline__ void tree_sum2(int a, int b, float &fa, float &fb)
{
    auto r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
    a = (int)r[0];
    b = (int)r[1];
    int t = a + b;
    t = dpp_add<DPP_QUAD_XOR1, 0xf>(t);
    t = dpp_add<DPP_QUAD_XOR2, 0xf>(t);
    int hi = t >> 16, lo = t & 0xffff;
    auto q = __builtin_amdgcn_permlane16_swap((unsigned)hi, (unsigned)lo, false, false);
    hi = (int)q[0];
    lo = (int)q[1];
    int u = hi + lo;
    u = dpp_add<DPP_ROW_HALF_MIRROR, 0xf>(u);
    u = dpp_add<DPP_ROW_MIRROR, 0xf>(u);
    const float uf = (float)u;
    const float up = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(uf), DPP_ROW_BCAST15, 0xa, 0xf, true));
    const float res = fmaf(up, 65536.f, uf);
    fa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(res), 31));
    fb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(res), 63));
}

// ORDER 0: (x . ones) then (ones . r); ORDER 1: (ones . x) then (r . ones) -- which one sums a 16-lane block depends on the
// operand layouts; main() checks both
template <int ORDER>
__device__ __forceinline__ double mfma_sum(double x)
{
    const double one = 1.0;
    double r1 = ORDER == 0 ? __builtin_amdgcn_mfma_f64_4x4x4f64(x, one, 0.0, 0, 0, 0) : __builtin_amdgcn_mfma_f64_4x4x4f64(one, x, 0.0, 0, 0, 0);
    double r2 = ORDER == 0 ? __builtin_amdgcn_mfma_f64_4x4x4f64(one, r1, 0.0, 0, 0, 0) : __builtin_amdgcn_mfma_f64_4x4x4f64(r1, one, 0.0, 0, 0, 0);
    const v4f64 z = {0, 0, 0, 0};
    v4f64 r3 = __builtin_amdgcn_mfma_f64_16x16x4f64(r2, one, z, 0, 0, 0); // lanes l, l + 16, l + 32, l + 48
    return r3[0];
}
template <int ORDER>
__device__ __forceinline__ void mfma_sum2(int a, int b, float &fa, float &fb)
{
    fa = (float)mfma_sum<ORDER>((double)a);
    fb = (float)mfma_sum<ORDER>((double)b);
}

template <int ORDER>
__global__ void check_kernel(const int *a, const int *b, float *out /* [waves][4][64] */)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    float fa, fb, ta, tb;
    mfma_sum2<ORDER>(a[i], b[i], fa, fb);
    tree_sum2(a[i], b[i], ta, tb);
    out[(blockIdx.x * 4 + 0) * 64 + threadIdx.x] = fa;
    out[(blockIdx.x * 4 + 1) * 64 + threadIdx.x] = fb;
    out[(blockIdx.x * 4 + 2) * 64 + threadIdx.x] = ta;
    out[(blockIdx.x * 4 + 3) * 64 + threadIdx.x] = tb;
}

// ~63 VALU of the iteration's mix around the reduction; everything depends on the loop-carried state so nothing folds
#define FILLER                                                                                                          \
    asm volatile(                                                                                                       \
        "v_dot2_u32_u16 %0, %4, %5, %0\n v_dot2_u32_u16 %1, %4, %6, %1\n v_dot2_u32_u16 %2, %5, %6, %2\n v_dot2_u32_u16 %3, %4, %5, %3\n" \
        "v_dot2_u32_u16 %0, %6, %5, %0\n v_dot2_u32_u16 %1, %5, %6, %1\n v_dot2_u32_u16 %2, %4, %6, %2\n v_dot2_u32_u16 %3, %6, %5, %3\n" \
        "v_dot2_u32_u16 %0, %4, %5, %0\n v_dot2_u32_u16 %1, %4, %6, %1\n v_dot2_u32_u16 %2, %5, %6, %2\n v_dot2_u32_u16 %3, %4, %5, %3\n" \
        "v_dot2_u32_u16 %0, %6, %5, %0\n v_dot2_u32_u16 %1, %5, %6, %1\n"                                                  \
        "v_perm_b32 %0, %0, %1, %6\n v_perm_b32 %1, %1, %2, %6\n v_perm_b32 %2, %2, %3, %6\n v_perm_b32 %3, %3, %0, %6\n"  \
        "v_perm_b32 %0, %0, %2, %6\n v_perm_b32 %1, %1, %3, %6\n"                                                          \
        "v_pk_lshrrev_b16 %0, 1, %0 op_sel_hi:[0,1]\n v_pk_lshrrev_b16 %1, 1, %1 op_sel_hi:[0,1]\n"                        \
        "v_pk_lshrrev_b16 %2, 1, %2 op_sel_hi:[0,1]\n v_pk_lshrrev_b16 %3, 1, %3 op_sel_hi:[0,1]\n"                        \
        "v_dot2c_i32_i16 %7, %0, %4\n v_dot2c_i32_i16 %8, %0, %5\n v_dot2c_i32_i16 %7, %1, %4\n v_dot2c_i32_i16 %8, %1, %5\n" \
        "v_dot2c_i32_i16 %7, %2, %4\n v_dot2c_i32_i16 %8, %2, %5\n v_dot2_i32_i16 %7, %3, %4, %7\n v_dot2_i32_i16 %8, %3, %5, %8\n" \
        "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %4\n"                      \
        "v_add_u32 %0, %0, %5\n v_add_u32 %1, %1, %6\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %5\n"                      \
        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(b1), "+v"(b2))

#define SOLVE                                                                                                           \
    asm volatile(                                                                                                       \
        "v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0\n"          \
        "v_pk_add_f32 %0, %0, %1\n v_pk_mul_f32 %1, %0, %2\n v_pk_add_f32 %0, %1, %0\n v_pk_mul_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %1\n" \
        "v_sub_f32 %3, %3, %4\n v_sub_f32 %4, %4, %3\n v_sub_f32 %3, %3, %4\n v_sub_f32 %4, %4, %3\n v_sub_f32 %3, %3, %4\n v_sub_f32 %4, %4, %3\n" \
        "v_add_f32 %3, %3, %4\n v_add_f32 %4, %4, %3\n v_add_f32 %3, %3, %4\n v_mul_f32 %3, %3, %4\n v_mul_f32 %4, %4, %3\n" \
        "v_fmac_f32 %3, %4, %4\n v_fmac_f32 %4, %3, %3\n"                                                                 \
        : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(s0), "+v"(s1))

template <int MODE>
__global__ __launch_bounds__(64) void loop_kernel(float *out, int iters, uint32_t seed)
{
    uint32_t p0 = seed * (threadIdx.x + 1), p1 = p0 + 977, p2 = p0 ^ 0x5555, p3 = p0 * 3;
    uint32_t w0 = 0x00030001u + threadIdx.x, w1 = 0x00010002u, w2 = 0x07060100u;
    int b1 = 0, b2 = 0;
    double q0 = 1.0, q1 = 1.0000001, q2 = 0.9999999; // (register pairs for the packed-f32 ops)
    float s0 = 1.f, s1 = 0.5f;
    for (int it = 0; it < iters; it++) {
        FILLER;
        float fa, fb;
        const int a = (b1 & 0x0fffffff) - 0x07ffffff, b = (b2 & 0x0fffffff) - 0x07ffffff; // |.| < 2^28, the tree's contract
        if (MODE == 0)
            tree_sum2(a, b, fa, fb);
        else if (MODE == 1)
            mfma_sum2<0>(a, b, fa, fb);
        else {
            fa = (float)a;
            fb = (float)b;
        }
        s0 += fa * 1e-12f; // the sums feed the next iteration like the solve's update does
        s1 += fb * 1e-12f;
        SOLVE;
        w0 ^= __float_as_uint(s0) & 1u;
    }
    out[blockIdx.x * 64 + threadIdx.x] = s0 + s1 + (float)(p0 ^ p1 ^ p2 ^ p3) + (float)q0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int waves = 256, n = waves * 64;
    std::vector<int> a(n), b(n);
    srand(5);
    for (int i = 0; i < n; i++) {
        a[i] = (int)((((long long)rand() << 16) ^ rand()) % (1 << 28)) * (rand() & 1 ? 1 : -1);
        b[i] = (int)((((long long)rand() << 16) ^ rand()) % (1 << 28)) * (rand() & 1 ? 1 : -1);
        if (i < 64) { a[i] = (1 << 28) - 1; b[i] = -((1 << 28) - 1); }        // the extremes
    }
    int *da, *db;
    float *dout;
    CK(hipMalloc(&da, n * 4));
    CK(hipMalloc(&db, n * 4));
    CK(hipMalloc(&dout, (size_t)waves * 4 * 64 * 4));
    CK(hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice));
    std::vector<float> out((size_t)waves * 4 * 64);
    int good_order = -1;
    for (int order = 0; order < 2; order++) {
        if (order == 0)
            hipLaunchKernelGGL(check_kernel<0>, dim3(waves), dim3(64), 0, 0, da, db, dout);
        else
            hipLaunchKernelGGL(check_kernel<1>, dim3(waves), dim3(64), 0, 0, da, db, dout);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
        long bad_m = 0, bad_t = 0;
        for (int w = 0; w < waves; w++) {
            long long sa = 0, sb = 0;
            for (int l = 0; l < 64; l++) { sa += a[w * 64 + l]; sb += b[w * 64 + l]; }
            const float ea = (float)sa, eb = (float)sb;
            for (int l = 0; l < 64; l++) {
                bad_m += out[(w * 4 + 0) * 64 + l] != ea || out[(w * 4 + 1) * 64 + l] != eb;
                bad_t += out[(w * 4 + 2) * 64 + l] != ea || out[(w * 4 + 3) * 64 + l] != eb;
            }
        }
        printf("order %d: MFMA sums wrong in %ld of %d lanes (tree: %ld)\n", order, bad_m, n, bad_t);
        if (bad_m == 0 && good_order < 0)
            good_order = order;
    }
    hipFuncAttributes fa;
    const void *fn[3] = {(const void *)loop_kernel<0>, (const void *)loop_kernel<1>, (const void *)loop_kernel<2>};
    const char *names[3] = {"tree (product)", "f64 MFMA", "no reduction"};
    for (int m = 0; m < 3; m++) {
        CK(hipFuncGetAttributes(&fa, fn[m]));
        printf("%-16s %3d VGPR, %d B scratch\n", names[m], fa.numRegs, (int)fa.localSizeBytes);
    }
    if (good_order != 0)
        printf("NOTE: the timing loop uses order 0; the layouts want order %d\n", good_order);
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    float *dres;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int wps = 5; wps <= 8; wps++) { // waves per SIMD resident (one-wave workgroups; the grid is exactly one residency)
        const int grid = 256 * 4 * wps;
        CK(hipMalloc(&dres, (size_t)grid * 64 * 4));
        for (int m = 0; m < 3; m++) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0, 0));
                if (m == 0) hipLaunchKernelGGL(loop_kernel<0>, dim3(grid), dim3(64), 0, 0, dres, iters, 12345u);
                if (m == 1) hipLaunchKernelGGL(loop_kernel<1>, dim3(grid), dim3(64), 0, 0, dres, iters, 12345u);
                if (m == 2) hipLaunchKernelGGL(loop_kernel<2>, dim3(grid), dim3(64), 0, 0, dres, iters, 12345u);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            // per SIMD: wps waves x iters iterations in `best` ms
            printf("%d waves/SIMD  %-16s %8.3f ms  = %6.1f ns per iteration per SIMD-slot (%.1f SIMD cycles at 2.4 GHz per wave-iteration)\n",
                   wps, names[m], best, 1e6 * best / iters / wps, 2.4e3 * 1e3 * best / iters / wps);
        }
        CK(hipFree(dres));
    }
    return 0;
}
