// load_rate.hip -- what bounds the fused pyramid pass besides its stores?  (round 4)
// The level-0 pass of 514 KITTI images takes 320 us with its Scharr stores and 160 us without them, although it issues only
// ~95 us of vector instructions and reads 0.24 GB.  Its loads are one 8-byte window per lane and row at a 2-byte-aligned
// address, lanes 4 bytes apart (every byte is requested twice).  This program runs the pass's load / store skeleton -- same
// grid, same addresses, no arithmetic -- with different load shapes:
//   A  8 bytes per lane at x4 - 2 (the pass)          B  4 bytes per lane at x4 (aligned; neighbours would come from DPP)
//   C  8 bytes per lane at x4 (4-byte aligned)        each also with the pass's 8 x 1 KB non-temporal stores per wavefront
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/load_rate.hip -o /tmp/load_rate && /tmp/load_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define G __attribute__((address_space(1)))
struct __attribute__((packed, aligned(2))) U2x { uint32_t lo, hi; };
struct __attribute__((packed, aligned(4))) U2a { uint32_t lo, hi; };

constexpr int ROWS = 8, LOADS = ROWS + 3;

template <int SHAPE, bool STORE>
__global__ __launch_bounds__(64) void skel(const uint8_t *src, uint32_t *der, int w, int h, int stride, size_t img_bytes, size_t der_dwords, int nm)
{
    const int g = blockIdx.x * 64 + threadIdx.x;
    if (g >= nm)
        return;
    const int y0 = ROWS * blockIdx.y;
    const uint8_t *base = src + (size_t)blockIdx.z * img_bytes;
    uint32_t acc[LOADS];
#pragma unroll
    for (int r = 0; r < LOADS; r++) {
        int p = y0 - 2 + r;
        p = p < 0 ? -p : p >= h ? 2 * h - 2 - p : p;
        const uint8_t *row = base + (size_t)p * stride + 64; // (64: a left border)
        if (SHAPE == 0) {
            const U2x v = *(const U2x *)(row + (uint32_t)(4 * g) - 2);
            acc[r] = v.lo ^ v.hi;
        } else if (SHAPE == 1) {
            acc[r] = *(const uint32_t *)(row + (uint32_t)(4 * g));
        } else {
            const U2a v = *(const U2a *)(row + (uint32_t)(4 * g));
            acc[r] = v.lo ^ v.hi;
        }
    }
    G uint32_t *d = (G uint32_t *)der + (size_t)blockIdx.z * der_dwords;
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        const int y = y0 + r;
        if (y >= h)
            break;
        const u32x4 o = {acc[r], acc[r + 1], acc[r + 2], acc[r + 3]};
        G u32x4 *out = (G u32x4 *)(d + (size_t)y * stride) + g;
        if (STORE)
            __builtin_nontemporal_store(o, out);
        else if ((o.x ^ o.y ^ o.z ^ o.w) == 0x12345679u)
            *out = o;
    }
}


// ---- second question: how do the pass's three phases (loads, ~340 vector instructions, stores) combine? ----
// FILL vector instructions (v_perm_b32, full rate) between the loads and the stores; WPB wavefronts per workgroup (each its own
// row block); PIPE: a wavefront does two row blocks and requests the second one's rows before it computes the first.
template <int FILL, bool STORE, int WPB, bool PIPE>
__global__ __launch_bounds__(64 * WPB) void phases(const uint8_t *src, uint32_t *der, int w, int h, int stride, size_t img_bytes, size_t der_dwords, int nm, uint32_t sel)
{
    const int g = blockIdx.x * 64 + (threadIdx.x & 63);
    if (g >= nm)
        return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nblk = PIPE ? 2 : 1;
    const int b0 = (blockIdx.y * WPB + wave) * nblk;
    const uint8_t *base = src + (size_t)blockIdx.z * img_bytes;
    G uint32_t *d = (G uint32_t *)der + (size_t)blockIdx.z * der_dwords;
    uint32_t acc[2][LOADS];
    auto load = [&](int k) {
        const int y0 = ROWS * (b0 + k);
#pragma unroll
        for (int r = 0; r < LOADS; r++) {
            int p = y0 - 2 + r;
            p = p < 0 ? -p : p >= h ? 2 * h - 2 - p : p;
            const uint8_t *row = base + (size_t)p * stride + 64;
            const U2x v = *(const U2x *)(row + (uint32_t)(4 * g) - 2);
            acc[k][r] = v.lo ^ v.hi;
        }
    };
    auto work = [&](int k) {
        const int y0 = ROWS * (b0 + k);
#pragma unroll
        for (int i = 0; i < FILL / LOADS; i++)
#pragma unroll
            for (int r = 0; r < LOADS; r++)
                acc[k][r] = __builtin_amdgcn_perm(acc[k][r], acc[k][(r + 1) % LOADS], sel);
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const int y = y0 + r;
            if (y >= h)
                break;
            const u32x4 o = {acc[k][r], acc[k][r + 1], acc[k][r + 2], acc[k][r + 3]};
            G u32x4 *out = (G u32x4 *)(d + (size_t)y * stride) + g;
            if (STORE)
                __builtin_nontemporal_store(o, out);
            else if ((o.x ^ o.y ^ o.z ^ o.w) == 0x12345679u)
                *out = o;
        }
    };
    if (ROWS * b0 >= h)
        return;
    load(0);
    if (PIPE) {
        load(1);
        work(0);
        if (ROWS * (b0 + 1) < h)
            work(1);
    } else {
        work(0);
    }
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
    const int W = 1241, H = 376, NI = 514, stride = 1312 + 64;
    const size_t img_bytes = (size_t)stride * (H + 4), der_dwords = (size_t)stride * H;
    uint8_t *src; uint32_t *der;
    hipMalloc((void **)&src, img_bytes * NI + (1 << 20));
    hipMalloc((void **)&der, der_dwords * 4 * NI + (1 << 20));
    hipMemset(src, 7, img_bytes * NI);
    const int nm = (W + 3) / 4;
    const dim3 grid((nm + 63) / 64, (H + ROWS - 1) / ROWS, NI);
    printf("grid %d x %d x %d wavefronts, %d loads + %d stores each\n", grid.x, grid.y, grid.z, LOADS, ROWS);
#define RUN(S, ST, name) do { float ms = timeit([&] { hipLaunchKernelGGL((skel<S, ST>), grid, dim3(64), 0, 0, src, der, W, H, stride, img_bytes, der_dwords, nm); }, 20); \
        printf("  %-58s %7.1f us\n", name, ms * 1e3); } while (0)
    RUN(0, false, "A  8 B / lane at x4 - 2 (the pass), no stores");
    RUN(1, false, "B  4 B / lane at x4, no stores");
    RUN(2, false, "C  8 B / lane at x4, no stores");
    RUN(0, true, "A  + 8 x 1 KB non-temporal stores per wavefront");
    RUN(1, true, "B  + stores");
    RUN(2, true, "C  + stores");

    const int nb = (H + ROWS - 1) / ROWS;
#define RUNP(FILL, ST, WPB, PIPE, name) do { const dim3 gr((nm + 63) / 64, (nb + WPB * (PIPE ? 2 : 1) - 1) / (WPB * (PIPE ? 2 : 1)), NI); \
        float ms = timeit([&] { hipLaunchKernelGGL((phases<FILL, ST, WPB, PIPE>), gr, dim3(64 * WPB), 0, 0, src, der, W, H, stride, img_bytes, der_dwords, nm, 0x07020500u); }, 20); \
        printf("  %-58s %7.1f us\n", name, ms * 1e3); } while (0)
    RUNP(0, true, 1, false, "loads + stores");
    RUNP(341, false, 1, false, "loads + 341 VALU");
    RUNP(341, true, 1, false, "loads + 341 VALU + stores");
    RUNP(682, true, 1, false, "loads + 682 VALU + stores");
    RUNP(341, true, 4, false, "loads + 341 VALU + stores, 4 wavefronts / workgroup");
    RUNP(341, true, 1, true, "loads + 341 VALU + stores, 2 row blocks pipelined");
    RUNP(341, false, 1, true, "loads + 341 VALU, 2 row blocks pipelined");
    return 0;
}
