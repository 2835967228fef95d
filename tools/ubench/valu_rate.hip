// valu_rate.hip -- developer micro-benchmark: ISSUE COST of the VALU instructions the LK kernel is (or could be)
// made of on gfx950, in SIMD cycles per wave64 instruction.
//
// Round-1 version wrote the chains in C and let the compiler fold several of them ((w - v) ^ 0 is an involution ...):
// its plain-integer rows were meaningless.  This version emits every measured instruction
// through `asm volatile`, so the instruction count is exact by construction: each kernel executes
//     iters x 8 (unroll) x 8 (independent chains)  wave-instructions of ONE opcode per wave,
// with 8 waves per SIMD resident (8 x 256-thread workgroups per CU), i.e. the figure is an issue cost, not a latency.
// `--check` prints the expected instruction count so that a rocprofv3 --pmc SQ_INSTS_VALU pass can confirm it:
//     cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d out -- ./valu_rate
// Cycles come from hipEvent time x the SCLK the device reports; the PMC pass (GRBM_GUI_ACTIVE) gives them directly.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define X8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define REGS32 "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)

// kernel with 8 VGPR chains d = op(d, w [, w2])
#define K_VGPR(NAME, OP)                                                                                   \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, int iters, uint32_t seed)               \
    {                                                                                                      \
        uint32_t v0 = seed * (threadIdx.x + 1), v1 = v0 + 977, v2 = v0 ^ 0x5555, v3 = v0 * 3, v4 = v0 + 5,  \
                 v5 = v0 ^ 77, v6 = v0 * 7, v7 = v0 + 11;                                                  \
        uint32_t w = (seed | 0x00030001u) + threadIdx.x, w2 = 0x3f800001u + (seed & 7);                   \
        for (int it = 0; it < iters; it++) {                                                               \
            _Pragma("unroll") for (int r = 0; r < 8; r++)                                                  \
                asm volatile(X8(OP) : REGS32 : "v"(w), "v"(w2));                                           \
        }                                                                                                  \
        out[blockIdx.x * 256 + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;                       \
    }

#define OP_MOV(d) "v_mov_b32 %" #d ", %8\n"
#define OP_ADD(d) "v_add_u32 %" #d ", %" #d ", %8\n"
#define OP_SUB(d) "v_sub_u32 %" #d ", %8, %" #d "\n"
#define OP_ADD3(d) "v_add3_u32 %" #d ", %" #d ", %8, %9\n"
#define OP_AND(d) "v_and_b32 %" #d ", %" #d ", %8\n"
#define OP_XOR(d) "v_xor_b32 %" #d ", %" #d ", %8\n"
#define OP_LSHL(d) "v_lshlrev_b32 %" #d ", 1, %" #d "\n"
#define OP_ASHR(d) "v_ashrrev_i32 %" #d ", 16, %" #d "\n"
#define OP_LSHLOR(d) "v_lshl_or_b32 %" #d ", %" #d ", 1, %8\n"
#define OP_CNDMASK(d) "v_cndmask_b32 %" #d ", %" #d ", %8, vcc\n"
#define OP_PERM(d) "v_perm_b32 %" #d ", %" #d ", %8, %9\n"
#define OP_ALIGNBYTE(d) "v_alignbyte_b32 %" #d ", %" #d ", %8, 1\n"
#define OP_UDOT2(d) "v_dot2_u32_u16 %" #d ", %" #d ", %8, %" #d "\n"
#define OP_SDOT2(d) "v_dot2_i32_i16 %" #d ", %" #d ", %8, %" #d "\n"
#define OP_SDOT2C(d) "v_dot2c_i32_i16 %" #d ", %8, %9\n"
#define OP_DOT4(d) "v_dot4_u32_u8 %" #d ", %" #d ", %8, %" #d "\n"
#define OP_PKSUB(d) "v_pk_sub_i16 %" #d ", %8, %" #d "\n"
#define OP_PKLSHR(d) "v_pk_lshrrev_b16 %" #d ", 1, %" #d " op_sel_hi:[0,1]\n"
#define OP_PKADDU16(d) "v_pk_add_u16 %" #d ", %" #d ", %8\n"
#define OP_DPPADD(d) "v_add_u32_dpp %" #d ", %" #d ", %" #d " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP_DPPMOV(d) "v_mov_b32_dpp %" #d ", %" #d " row_mirror row_mask:0xf bank_mask:0xf\n"
#define OP_MAD24(d) "v_mad_u32_u24 %" #d ", %" #d ", %8, %" #d "\n"
#define OP_MULLO(d) "v_mul_lo_u32 %" #d ", %" #d ", %8\n"
#define OP_FMA(d) "v_fma_f32 %" #d ", %" #d ", %9, %9\n"
#define OP_FMAC(d) "v_fmac_f32 %" #d ", %9, %9\n"
#define OP_MULF(d) "v_mul_f32 %" #d ", %" #d ", %9\n"
#define OP_ADDF(d) "v_add_f32 %" #d ", %" #d ", %9\n"
#define OP_SUBF(d) "v_sub_f32 %" #d ", %" #d ", %9\n"
#define OP_FLOOR(d) "v_floor_f32 %" #d ", %" #d "\n"
#define OP_FRACT(d) "v_fract_f32 %" #d ", %" #d "\n"
#define OP_CVTFI(d) "v_cvt_f32_i32 %" #d ", %" #d "\n"
#define OP_CVTIF(d) "v_cvt_i32_f32 %" #d ", %" #d "\n"
#define OP_CVTUB(d) "v_cvt_f32_ubyte0 %" #d ", %" #d "\n"
#define OP_CVTPKI16(d) "v_cvt_pk_i16_i32 %" #d ", %" #d ", %8\n"
#define OP_RCP(d) "v_rcp_f32 %" #d ", %" #d "\n"
#define OP_SQRT(d) "v_sqrt_f32 %" #d ", %" #d "\n"
#define OP_MAXF(d) "v_max_f32 %" #d ", %" #d ", %9\n"
#define OP_MED3(d) "v_med3_i32 %" #d ", %" #d ", %8, %9\n"

K_VGPR(mov, OP_MOV)
K_VGPR(add_u32, OP_ADD)
K_VGPR(sub_u32, OP_SUB)
K_VGPR(add3_u32, OP_ADD3)
K_VGPR(and_b32, OP_AND)
K_VGPR(xor_b32, OP_XOR)
K_VGPR(lshlrev_b32, OP_LSHL)
K_VGPR(ashrrev_i32, OP_ASHR)
K_VGPR(lshl_or_b32, OP_LSHLOR)
K_VGPR(cndmask_b32, OP_CNDMASK)
K_VGPR(perm_b32, OP_PERM)
K_VGPR(alignbyte_b32, OP_ALIGNBYTE)
K_VGPR(dot2_u32_u16, OP_UDOT2)
K_VGPR(dot2_i32_i16, OP_SDOT2)
K_VGPR(dot2c_i32_i16, OP_SDOT2C)
K_VGPR(dot4_u32_u8, OP_DOT4)
K_VGPR(pk_sub_i16, OP_PKSUB)
K_VGPR(pk_lshrrev_b16, OP_PKLSHR)
K_VGPR(pk_add_u16, OP_PKADDU16)
K_VGPR(add_u32_dpp, OP_DPPADD)
K_VGPR(mov_b32_dpp, OP_DPPMOV)
K_VGPR(mad_u32_u24, OP_MAD24)
K_VGPR(mul_lo_u32, OP_MULLO)
K_VGPR(fma_f32, OP_FMA)
K_VGPR(fmac_f32, OP_FMAC)
K_VGPR(mul_f32, OP_MULF)
K_VGPR(add_f32, OP_ADDF)
K_VGPR(sub_f32, OP_SUBF)
K_VGPR(floor_f32, OP_FLOOR)
K_VGPR(fract_f32, OP_FRACT)
K_VGPR(cvt_f32_i32, OP_CVTFI)
K_VGPR(cvt_i32_f32, OP_CVTIF)
K_VGPR(cvt_f32_ubyte0, OP_CVTUB)
K_VGPR(cvt_pk_i16_i32, OP_CVTPKI16)
K_VGPR(rcp_f32, OP_RCP)
K_VGPR(sqrt_f32, OP_SQRT)
K_VGPR(max_f32, OP_MAXF)
K_VGPR(med3_i32, OP_MED3)

// 64-bit chains (packed f32 pairs / f64): 8 VGPR pairs
#define REGS64 "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
#define K_VGPR64(NAME, OP)                                                                                 \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, int iters, uint32_t seed)               \
    {                                                                                                      \
        double d0 = seed * (threadIdx.x + 1.0), d1 = d0 + 977, d2 = d0 * 0.5, d3 = d0 * 3, d4 = d0 + 5,     \
               d5 = d0 - 77, d6 = d0 * 7, d7 = d0 + 11;                                                    \
        double w = 1.0000001 + 1e-9 * (seed & 7);                                                          \
        for (int it = 0; it < iters; it++) {                                                               \
            _Pragma("unroll") for (int r = 0; r < 8; r++)                                                  \
                asm volatile(X8(OP) : REGS64 : "v"(w));                                                    \
        }                                                                                                  \
        out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)__double_as_longlong(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7); \
    }
#define OP_PKFMA(d) "v_pk_fma_f32 %" #d ", %" #d ", %8, %8\n"
#define OP_PKMUL(d) "v_pk_mul_f32 %" #d ", %" #d ", %8\n"
#define OP_PKADD(d) "v_pk_add_f32 %" #d ", %" #d ", %8\n"
#define OP_ADDF64(d) "v_add_f64 %" #d ", %" #d ", %8\n"
#define OP_MULF64(d) "v_mul_f64 %" #d ", %" #d ", %8\n"
#define OP_FMAF64(d) "v_fma_f64 %" #d ", %" #d ", %8, %8\n"
K_VGPR64(pk_fma_f32, OP_PKFMA)
K_VGPR64(pk_mul_f32, OP_PKMUL)
K_VGPR64(pk_add_f32, OP_PKADD)
K_VGPR64(add_f64, OP_ADDF64)
K_VGPR64(mul_f64, OP_MULF64)
K_VGPR64(fma_f64, OP_FMAF64)

// half-swaps of gfx950: two VGPR operands rewritten (4 pairs per statement, 16 statements = 64 instructions)
__global__ __launch_bounds__(256) void k_permlane32_swap(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t v0 = seed * (threadIdx.x + 1), v1 = v0 + 977, v2 = v0 ^ 0x5555, v3 = v0 * 3, v4 = v0 + 5, v5 = v0 ^ 77,
             v6 = v0 * 7, v7 = v0 + 11;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
            asm volatile("v_permlane32_swap_b32 %0, %1\nv_permlane32_swap_b32 %2, %3\nv_permlane32_swap_b32 %4, %5\n"
                         "v_permlane32_swap_b32 %6, %7\n"
                         : REGS32);
    }
    out[blockIdx.x * 256 + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
}
__global__ __launch_bounds__(256) void k_permlane16_swap(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t v0 = seed * (threadIdx.x + 1), v1 = v0 + 977, v2 = v0 ^ 0x5555, v3 = v0 * 3, v4 = v0 + 5, v5 = v0 ^ 77,
             v6 = v0 * 7, v7 = v0 + 11;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
            asm volatile("v_permlane16_swap_b32 %0, %1\nv_permlane16_swap_b32 %2, %3\nv_permlane16_swap_b32 %4, %5\n"
                         "v_permlane16_swap_b32 %6, %7\n"
                         : REGS32);
    }
    out[blockIdx.x * 256 + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
}

// VALU -> SGPR: v_readlane_b32 / v_readfirstlane_b32 / v_cmp (64 per outer iteration)
__global__ __launch_bounds__(256) void k_readlane(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t v0 = seed * (threadIdx.x + 1), acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint32_t s0, s1, s2, s3, s4, s5, s6, s7;
            asm volatile("v_readlane_b32 %0, %8, 1\nv_readlane_b32 %1, %8, 2\nv_readlane_b32 %2, %8, 3\nv_readlane_b32 %3, %8, 4\n"
                         "v_readlane_b32 %4, %8, 5\nv_readlane_b32 %5, %8, 6\nv_readlane_b32 %6, %8, 7\nv_readlane_b32 %7, %8, 8\n"
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7)
                         : "v"(v0));
            acc ^= s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7; // scalar xors (SALU), not counted
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_cmp_lt_f32(uint32_t *out, int iters, uint32_t seed)
{
    float v0 = (float)(seed * (threadIdx.x + 1)), w = 3.5f + (seed & 3);
    unsigned long long acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            unsigned long long s0, s1, s2, s3, s4, s5, s6, s7;
            asm volatile("v_cmp_lt_f32_e64 %0, %8, %9\nv_cmp_lt_f32_e64 %1, %8, %9\nv_cmp_lt_f32_e64 %2, %8, %9\n"
                         "v_cmp_lt_f32_e64 %3, %8, %9\nv_cmp_lt_f32_e64 %4, %8, %9\nv_cmp_lt_f32_e64 %5, %8, %9\n"
                         "v_cmp_lt_f32_e64 %6, %8, %9\nv_cmp_lt_f32_e64 %7, %8, %9\n"
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7)
                         : "v"(v0), "v"(w));
            acc ^= s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)acc;
}

typedef void (*kern_t)(uint32_t *, int, uint32_t);
struct Entry {
    const char *name;
    kern_t fn;
};
#define E(NAME) {#NAME, k_##NAME}
static const Entry TABLE[] = {
    E(mov), E(add_u32), E(sub_u32), E(add3_u32), E(and_b32), E(xor_b32), E(lshlrev_b32), E(ashrrev_i32), E(lshl_or_b32),
    E(cndmask_b32), E(perm_b32), E(alignbyte_b32), E(dot2_u32_u16), E(dot2_i32_i16), E(dot2c_i32_i16), E(dot4_u32_u8),
    E(pk_sub_i16), E(pk_lshrrev_b16), E(pk_add_u16), E(add_u32_dpp), E(mov_b32_dpp), E(permlane32_swap), E(permlane16_swap),
    E(readlane), E(cmp_lt_f32), E(mad_u32_u24), E(mul_lo_u32), E(fma_f32), E(fmac_f32), E(mul_f32), E(add_f32), E(sub_f32),
    E(max_f32), E(med3_i32), E(floor_f32), E(fract_f32), E(cvt_f32_i32), E(cvt_i32_f32), E(cvt_f32_ubyte0), E(cvt_pk_i16_i32),
    E(rcp_f32), E(sqrt_f32), E(pk_fma_f32), E(pk_mul_f32), E(pk_add_f32), E(add_f64), E(mul_f64), E(fma_f64),
};

int main(int argc, char **argv)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 8; // 8 x 256 threads per CU = 8 waves per SIMD
    uint32_t *d_out;
    (void)hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    const int iters = argc > 1 && !strcmp(argv[1], "--quick") ? 200 : 2000;
    const double simds = p.multiProcessorCount * 4.0, clk = p.clockRate * 1e3;
    const double insts_per_wave = (double)iters * 64.0;
    printf("CUs %d, SCLK %d kHz, %d workgroups x 4 waves, %d x 64 measured instructions per wave (exact: asm volatile)\n",
           p.multiProcessorCount, p.clockRate, blocks, iters);
    printf("%-20s %9s %14s %22s\n", "opcode", "ms", "G wave-inst/s", "SIMD cycles per inst");
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (const Entry &e : TABLE) {
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d_out, iters, 12345u); // warm-up
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d_out, iters, 12345u);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
        }
        const double insts = (double)blocks * 4 * insts_per_wave;
        // every SIMD holds 8 waves; cycles per instruction per SIMD = time x clock / (instructions per SIMD)
        printf("v_%-18s %9.3f %14.1f %22.2f\n", e.name, best, insts / best / 1e6, best * 1e-3 * clk / (insts / simds));
    }
    (void)hipFree(d_out);
    return 0;
}
