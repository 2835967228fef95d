// valu_rate.hip -- developer micro-benchmark: issue rate of the VALU instructions the LK kernel is
// made of (v_perm_b32, v_dot2_u32_u16, v_dot2_i32_i16, v_pk_sub_i16, v_add_u32 DPP, v_fma_f32) on
// gfx950, in wave-instructions per cycle per SIMD, from a kernel that keeps 8 independent chains per
// lane and 8 waves per SIMD busy.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t v[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
        v[i] = seed * (threadIdx.x + 1 + i * 977) + blockIdx.x;
    const uint32_t w = seed | 0x00030001u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0)
                    v[i] = __builtin_amdgcn_perm(v[i], w, 0x0c010c00u + i);
                else if (OP == 1)
                    v[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, v[i]), __builtin_bit_cast(u16x2, w), v[i], false);
                else if (OP == 2)
                    v[i] = (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(i16x2, v[i]), __builtin_bit_cast(i16x2, w), (int)v[i], false);
                else if (OP == 3)
                    v[i] = __builtin_bit_cast(uint32_t, (i16x2)(__builtin_bit_cast(i16x2, v[i]) - __builtin_bit_cast(i16x2, w)));
                else if (OP == 4)
                    v[i] = v[i] + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[i], 0xB1, 0xf, 0xf, true);
                else if (OP == 5)
                    v[i] = __float_as_uint(fmaf(__uint_as_float(v[i]), 1.0000001f, 0.5f));
                else if (OP == 6)
                    v[i] = v[i] * 3u + w;  // v_mad_u32_u24 / v_mul_lo
                else
                    v[i] = v[i] + w;       // v_add_u32
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
        s ^= v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
double run(const char *name, uint32_t *d_out, int blocks, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 12345u);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 12345u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double insts = (double)blocks * 4 /*waves*/ * iters * 64.0; // wave-instructions
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double simds = p.multiProcessorCount * 4.0, clk = p.clockRate * 1e3;
    printf("%-18s %8.3f ms  %7.2f G wave-inst/s  = %.3f inst/clk/SIMD at %.0f MHz nominal\n", name, ms,
           insts / ms / 1e6, insts / (ms * 1e-3) / simds / clk, clk / 1e6);
    return ms;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 8; // 8 x 256 threads per CU = 8 waves per SIMD
    uint32_t *d_out;
    hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    const int iters = 4000;
    printf("CUs %d, clock %d kHz\n", p.multiProcessorCount, p.clockRate);
    run<0>("v_perm_b32", d_out, blocks, iters);
    run<1>("v_dot2_u32_u16", d_out, blocks, iters);
    run<2>("v_dot2_i32_i16", d_out, blocks, iters);
    run<3>("v_pk_sub_i16", d_out, blocks, iters);
    run<4>("v_add_u32 dpp", d_out, blocks, iters);
    run<5>("v_fma_f32", d_out, blocks, iters);
    run<6>("v_mad/mul u32", d_out, blocks, iters);
    run<7>("v_add_u32", d_out, blocks, iters);
    hipFree(d_out);
    return 0;
}
