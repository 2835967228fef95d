// valu_rate.hip -- developer micro-benchmark: issue rate of the VALU instructions the LK kernel is (or
// could be) made of on gfx950, in wave-instructions per cycle per SIMD, from a kernel that keeps 8
// independent dependency chains per lane and 8 waves per SIMD busy.  Operands are data dependent so the
// compiler cannot fold a chain.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { PERM, UDOT2, SDOT2, PKSUB, PKLSHR, DPPADD, FMA, MULF, ADDU, ANDB, LSHL, MAD24, MULLO, CVTUB, FLOORF, CVTI, CNDMASK, PKFMA,
       DOT4, ADDF64, CVTF64, XORB, NOPS };
static const char *NAMES[] = {"v_perm_b32", "v_dot2_u32_u16", "v_dot2_i32_i16", "v_pk_sub_i16", "v_pk_lshrrev_b16", "v_add_u32 dpp",
                              "v_fma_f32", "v_mul_f32", "v_add_u32", "v_and_b32", "v_lshlrev_b32", "v_mad_u32_u24", "v_mul_lo_u32",
                              "v_cvt_f32_ubyte0", "v_floor_f32", "v_cvt_i32_f32", "v_cndmask_b32", "v_pk_fma_f32", "v_dot4_u32_u8",
                              "v_add_f64", "v_cvt_f64_i32", "v_xor_b32"};

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t v[8];
    f32x2 p2[4];
    double d[4];
#pragma unroll
    for (int i = 0; i < 8; i++)
        v[i] = seed * (threadIdx.x + 1 + i * 977) + blockIdx.x;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        p2[i] = f32x2{(float)v[i], (float)v[i + 4]};
        d[i] = (double)v[i];
    }
    uint32_t w = seed | 0x00030001u;
    const float fw = __uint_as_float(0x3f800001u + (seed & 7));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == PERM)
                    v[i] = __builtin_amdgcn_perm(v[i], w, 0x0c010c00u + i);
                else if (OP == UDOT2)
                    v[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, v[i]), __builtin_bit_cast(u16x2, w), v[i], false);
                else if (OP == SDOT2)
                    v[i] = (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(i16x2, v[i]), __builtin_bit_cast(i16x2, w), (int)v[i], false);
                else if (OP == PKSUB)
                    v[i] = __builtin_bit_cast(uint32_t, (i16x2)(__builtin_bit_cast(i16x2, w) - __builtin_bit_cast(i16x2, v[i])));
                else if (OP == PKLSHR)
                    v[i] = __builtin_bit_cast(uint32_t, (u16x2)((__builtin_bit_cast(u16x2, v[i]) >> (unsigned short)1))) | w;
                else if (OP == DPPADD)
                    v[i] = v[i] + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[i], 0xB1, 0xf, 0xf, true);
                else if (OP == FMA)
                    v[i] = __float_as_uint(fmaf(__uint_as_float(v[i]), fw, 0.5f));
                else if (OP == MULF)
                    v[i] = __float_as_uint(__uint_as_float(v[i]) * fw);
                else if (OP == ADDU)
                    v[i] = (w - v[i]) ^ 0;           // v_sub_u32 (data dependent, not foldable)
                else if (OP == ANDB)
                    v[i] = (v[i] & w) | 0x10000u;
                else if (OP == LSHL)
                    v[i] = (v[i] << 1) | 1u;
                else if (OP == MAD24)
                    v[i] = __umul24(v[i], w) + v[i];
                else if (OP == MULLO)
                    v[i] = v[i] * w + 1u;
                else if (OP == CVTUB)
                    v[i] = __float_as_uint((float)(v[i] & 0xffu)) + w;
                else if (OP == FLOORF)
                    v[i] = __float_as_uint(floorf(__uint_as_float(v[i])) + fw);
                else if (OP == CVTI)
                    v[i] = (uint32_t)(int)__uint_as_float(v[i] | 0x3f000000u);
                else if (OP == CNDMASK)
                    v[i] = (v[i] & 1u) ? w : v[i] + 1u;
                else if (OP == DOT4)
                    v[i] = __builtin_amdgcn_udot4(v[i], w, v[i], false);
                else if (OP == XORB)
                    v[i] = v[i] ^ (w + i);
            }
            if (OP == PKFMA) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    p2[i] = __builtin_elementwise_fma(p2[i], f32x2{fw, fw}, f32x2{0.5f, 0.25f});
                    p2[i] = __builtin_elementwise_fma(p2[i], f32x2{fw, fw}, f32x2{0.5f, 0.25f});
                }
            }
            if (OP == ADDF64) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    d[i] = d[i] + (double)fw;
                    d[i] = d[i] + (double)fw;
                }
            }
            if (OP == CVTF64) {
#pragma unroll
                for (int i = 0; i < 8; i++)
                    v[i] = (uint32_t)(__double_as_longlong((double)(int)v[i]) >> 20) + w;
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
        s ^= v[i];
#pragma unroll
    for (int i = 0; i < 4; i++)
        s ^= __float_as_uint(p2[i].x + p2[i].y) ^ (uint32_t)__double_as_longlong(d[i]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(uint32_t *d_out, int blocks, int iters, double simds, double clk)
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
    const double insts = (double)blocks * 4 /*waves*/ * iters * 64.0; // wave-instructions of the measured opcode
    printf("%-18s %8.3f ms  %8.1f G wave-inst/s  = %.3f of that opcode /clk/SIMD at %.0f MHz nominal (chain body may hold 1-2 helper ops)\n",
           NAMES[OP], ms, insts / ms / 1e6, insts / (ms * 1e-3) / simds / clk, clk / 1e6);
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 8; // 8 x 256 threads per CU = 8 waves per SIMD
    uint32_t *d_out;
    (void)hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    const int iters = 2000;
    const double simds = p.multiProcessorCount * 4.0, clk = p.clockRate * 1e3;
    printf("CUs %d, clock %d kHz\n", p.multiProcessorCount, p.clockRate);
    run<PERM>(d_out, blocks, iters, simds, clk);
    run<UDOT2>(d_out, blocks, iters, simds, clk);
    run<SDOT2>(d_out, blocks, iters, simds, clk);
    run<DOT4>(d_out, blocks, iters, simds, clk);
    run<PKSUB>(d_out, blocks, iters, simds, clk);
    run<PKLSHR>(d_out, blocks, iters, simds, clk);
    run<DPPADD>(d_out, blocks, iters, simds, clk);
    run<FMA>(d_out, blocks, iters, simds, clk);
    run<MULF>(d_out, blocks, iters, simds, clk);
    run<PKFMA>(d_out, blocks, iters, simds, clk);
    run<ADDU>(d_out, blocks, iters, simds, clk);
    run<XORB>(d_out, blocks, iters, simds, clk);
    run<ANDB>(d_out, blocks, iters, simds, clk);
    run<LSHL>(d_out, blocks, iters, simds, clk);
    run<MAD24>(d_out, blocks, iters, simds, clk);
    run<MULLO>(d_out, blocks, iters, simds, clk);
    run<CVTUB>(d_out, blocks, iters, simds, clk);
    run<FLOORF>(d_out, blocks, iters, simds, clk);
    run<CVTI>(d_out, blocks, iters, simds, clk);
    run<CNDMASK>(d_out, blocks, iters, simds, clk);
    run<ADDF64>(d_out, blocks, iters, simds, clk);
    run<CVTF64>(d_out, blocks, iters, simds, clk);
    (void)hipFree(d_out);
    return 0;
}
