// hip_emu.h -- TEST ONLY.  A tiny single-workgroup SIMT emulator so the CPU test-suite can execute
// the *real* kernel sources (lk.hip, pyramid.hip) without a GPU: every work-item of one workgroup
// is a ucontext coroutine; a cross-lane operation (DPP, readlane, readfirstlane) or a barrier
// yields to a round-robin scheduler, so all lanes arrive before any proceeds -- which is the
// lock-step the hardware provides inside a wavefront.  Control flow must be wave-uniform around
// cross-lane operations (it is in these kernels).  Semantics of the DPP controls follow the CDNA3/4
// ISA manual ("DPP_CTRL": quad_perm, row_half_mirror, row_mirror, row_bcast:15, row_bcast:31 with
// row_mask / bank_mask / bound_ctrl).  Not a product path: libvo_hip never includes this.
#pragma once
#define VO_HOST_EMUL 1

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <algorithm>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

struct float2 {
    float x, y;
};
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
struct double2 {
    double x, y;
};
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
struct uint4 {
    uint32_t x, y, z, w;
};
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct uint2 {
    uint32_t x, y;
};
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
struct EmuIdx {
    unsigned x, y, z;
};
typedef void *hipStream_t;

namespace emu {

struct Block {
    int n = 0;
    ucontext_t main;
    std::vector<ucontext_t> ctx;
    char *stacks = nullptr;
    std::vector<char> done;
    std::vector<uint64_t> xbuf;
    int cur = 0;
    int bar_count = 0, bar_gen = 0; // emu::barrier()
    int wbar_count[16] = {}, wbar_gen[16] = {}; // emu::wave_barrier(), per wavefront (blocks of up to 1024 threads)
    std::function<void()> body;
};
inline Block *&current()
{
    static Block *b = nullptr;
    return b;
}
inline EmuIdx &tidx()
{
    static EmuIdx t{0, 0, 0};
    return t;
}
inline EmuIdx &bidx()
{
    static EmuIdx t{0, 0, 0};
    return t;
}
inline EmuIdx &bdim()
{
    static EmuIdx t{1, 1, 1};
    return t;
}

// dynamic LDS of the block being run (set by the harness before the launch)
inline void *&dyn_shared()
{
    static void *p = nullptr;
    return p;
}

inline void yield()
{
    Block *b = current();
    swapcontext(&b->ctx[b->cur], &b->main);
}

// A real barrier over the work-items that are still running (yield() alone only works while every lane passes the same
// number of yields -- not the case when DPP rows of one wavefront take different branches between two meeting points).
inline void barrier()
{
    Block *b = current();
    const int gen = b->bar_gen;
    b->bar_count++;
    for (;;) {
        if (b->bar_gen != gen)
            return;
        int live = 0;
        for (int i = 0; i < b->n; i++)
            live += !b->done[i];
        if (b->bar_count >= live) { // the last one in (or the last one left) opens it
            b->bar_count = 0;
            b->bar_gen++;
            return;
        }
        yield();
    }
}

// the same for the 64 lanes of the caller's wavefront only (code that one wavefront of a block runs on its own)
inline void wave_barrier()
{
    Block *b = current();
    const int w = b->cur >> 6, lo = w * 64, hi = lo + 64 < b->n ? lo + 64 : b->n;
    const int gen = b->wbar_gen[w];
    b->wbar_count[w]++;
    for (;;) {
        if (b->wbar_gen[w] != gen)
            return;
        int live = 0;
        for (int i = lo; i < hi; i++)
            live += !b->done[i];
        if (b->wbar_count[w] >= live) {
            b->wbar_count[w] = 0;
            b->wbar_gen[w]++;
            return;
        }
        yield();
    }
}

inline void trampoline()
{
    Block *b = current();
    b->body();
    b->done[b->cur] = 1;
    swapcontext(&b->ctx[b->cur], &b->main);
}

// run `body` once per work-item of a workgroup of n threads (threadIdx.x = 0 .. n-1)
inline void run_block(int n, unsigned bx, unsigned by, unsigned bz, const std::function<void()> &body)
{
    static const size_t STACK = 64 * 1024;
    static std::vector<char> pool; // reused across blocks (no re-zeroing)
    if (pool.size() < (size_t)n * STACK)
        pool.resize((size_t)n * STACK);
    Block blk;
    blk.n = n;
    blk.ctx.resize(n);
    blk.stacks = pool.data();
    blk.done.assign(n, 0);
    blk.xbuf.assign(n, 0);
    blk.body = body;
    current() = &blk;
    bidx() = EmuIdx{bx, by, bz};
    bdim() = EmuIdx{(unsigned)n, 1, 1};
    for (int i = 0; i < n; i++) {
        getcontext(&blk.ctx[i]);
        blk.ctx[i].uc_stack.ss_sp = blk.stacks + (size_t)i * STACK;
        blk.ctx[i].uc_stack.ss_size = STACK;
        blk.ctx[i].uc_link = &blk.main;
        makecontext(&blk.ctx[i], (void (*)())trampoline, 0);
    }
    for (;;) {
        bool any = false;
        for (int i = 0; i < n; i++) {
            if (blk.done[i])
                continue;
            any = true;
            blk.cur = i;
            tidx() = EmuIdx{(unsigned)i, 0, 0};
            swapcontext(&blk.main, &blk.ctx[i]);
        }
        if (!any)
            break;
    }
    current() = nullptr;
}

inline int lane_id() { return current()->cur; }

// every lane publishes `v`, then reads the value lane `src` published (src < 0: returns `fallback`)
inline uint32_t exchange(uint32_t v, int src, uint32_t fallback)
{
    Block *b = current();
    b->xbuf[b->cur] = v;
    yield();
    uint32_t r = src >= 0 ? (uint32_t)b->xbuf[src] : fallback;
    yield();
    return r;
}

inline int dpp_source(int lane, int ctrl)
{
    const int row = lane & ~15, l16 = lane & 15;
    if (ctrl >= 0 && ctrl <= 0xff)
        return (lane & ~3) + ((ctrl >> (2 * (lane & 3))) & 3);
    if (ctrl == 0x141)
        return (lane & ~7) + (7 - (lane & 7));
    if (ctrl == 0x140)
        return row + (15 - l16);
    if (ctrl == 0x142)
        return row >= 16 ? row - 1 : -1;
    if (ctrl == 0x143)
        return (lane & 63) >= 32 ? (lane & ~63) + 31 : -1;
    abort(); // control not modelled
}

} // namespace emu

#define threadIdx (emu::tidx())
#define blockIdx (emu::bidx())
#define blockDim (emu::bdim())

static inline void __threadfence() {} // (one host thread runs every work item: stores are visible at once)
static inline void __syncthreads() { emu::barrier(); } // a real barrier: wavefronts of a block may pass different numbers of yields
static inline int __float2int_rn(float v) { return (int)lrintf(v); }
static inline int __float_as_int(float v)
{
    int i;
    memcpy(&i, &v, 4);
    return i;
}
static inline float __int_as_float(int i)
{
    float v;
    memcpy(&v, &i, 4);
    return v;
}

static inline int emu_readfirstlane(int v)
{
    const int wave0 = emu::lane_id() & ~63;
    return (int)emu::exchange((uint32_t)v, wave0, 0);
}
static inline int emu_readlane(int v, int lane)
{
    const int wave0 = emu::lane_id() & ~63;
    return (int)emu::exchange((uint32_t)v, wave0 + lane, 0);
}
static inline int emu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    const int lane = emu::lane_id();
    const int wl = lane & 63;
    int s = emu::dpp_source(wl, ctrl);
    const bool enabled = ((row_mask >> (wl >> 4)) & 1) && ((bank_mask >> ((wl & 15) >> 2)) & 1);
    const uint32_t got = emu::exchange((uint32_t)src, s < 0 ? -1 : (lane & ~63) + s, 0);
    if (!enabled)
        return old;
    if (s < 0)
        return bound_ctrl ? 0 : old;
    return (int)got;
}
// v_permlane32_swap (width 32): lanes 32-63 of a <-> lanes 0-31 of b;
// v_permlane16_swap (width 16): odd 16-lane rows of a <-> even rows of b
static inline void emu_permlane_swap(int &a, int &b, int width)
{
    const int lane = emu::lane_id(), partner = lane ^ width;
    const bool upper = (lane & width) != 0;
    const int ga = (int)emu::exchange((uint32_t)a, partner, 0);
    const int gb = (int)emu::exchange((uint32_t)b, partner, 0);
    if (upper)
        a = gb; // this lane's a came from the partner's b
    else
        b = ga;
}
// wave-wide ballot: bit l = predicate of lane l of the caller's wavefront
static inline unsigned long long emu_ballot(bool pred)
{
    emu::Block *b = emu::current();
    const int lane = emu::lane_id(), wave0 = lane & ~63;
    b->xbuf[lane] = pred ? 1 : 0;
    emu::yield();
    unsigned long long m = 0;
    for (int l = 0; l < 64 && wave0 + l < b->n; l++)
        if (!b->done[wave0 + l] && b->xbuf[wave0 + l])
            m |= 1ull << l;
    emu::yield();
    return m;
}
static inline bool __any(bool pred) { return emu_ballot(pred) != 0; }
static inline unsigned long long __ballot(bool pred) { return emu_ballot(pred); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline uint64_t __umul64hi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }
// __shfl family (width 64) for 4- and 8-byte types: lane `src` of the caller's wavefront, own value if out of range
template <typename T>
static inline T emu_shfl_abs(T v, int src_lane)
{
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4- or 8-byte types");
    const int lane = emu::lane_id(), wave0 = lane & ~63;
    const int src = (src_lane >= 0 && src_lane < 64) ? wave0 + src_lane : lane;
    uint64_t u = 0;
    memcpy(&u, &v, sizeof(T));
    const uint32_t lo = emu::exchange((uint32_t)u, src, 0);
    uint32_t hi = 0;
    if (sizeof(T) == 8)
        hi = emu::exchange((uint32_t)(u >> 32), src, 0);
    u = ((uint64_t)hi << 32) | lo;
    T r;
    memcpy(&r, &u, sizeof(T));
    return r;
}
template <typename T>
static inline T __shfl(T v, int src_lane, int = 64) { return emu_shfl_abs(v, src_lane); }
template <typename T>
static inline T __shfl_up(T v, unsigned delta, int = 64) { return emu_shfl_abs(v, (emu::lane_id() & 63) - (int)delta); }
template <typename T>
static inline T __shfl_xor(T v, int mask, int = 64) { return emu_shfl_abs(v, (emu::lane_id() & 63) ^ mask); }
// lanes run one at a time between yields, so plain read-modify-write is atomic here
static inline int atomicAdd(int *p, int v)
{
    int o = *p;
    *p = o + v;
    return o;
}
static inline unsigned long long atomicOr(unsigned long long *p, unsigned long long v)
{
    const unsigned long long o = *p;
    *p = o | v;
    return o;
}
static inline int atomicMax(int *p, int v)
{
    int o = *p;
    *p = o > v ? o : v;
    return o;
}
static inline int atomicMin(int *p, int v)
{
    int o = *p;
    *p = o < v ? o : v;
    return o;
}
using std::max;
using std::min;
static inline uint32_t emu_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh)
{
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    return (uint32_t)(v >> (8 * (sh & 3)));
}
