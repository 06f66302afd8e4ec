// flood_layouts.hip -- the Ant reach flood (39 % of the movegen kernel's SIMD time) in two lane layouts, as a measurement:
//   quad   the shipped layout (csrc/hive_bb.hpp): one board = 4 lanes (3 used), 2 words per lane, 16 boards per wave
//   pair   one board = 2 lanes, 3 words per lane (rows 0-5 / 6-11), 32 boards per wave
// Same algorithm (csrc/hive_bb.hpp: occupancy views -> slide context -> x |= slide_step(x) twice per convergence test,
// wave-level exit), same inputs (occupancy words + the cell of the lifted piece), outputs compared word for word by
// tools/flood_layouts.py.  VERDICT round 2 item 6(b): "build it for the saturated regime at least, or show with
// SQ_INSTS_VALU why not".
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../hive-alphazero_amd/csrc/hive_bb.hpp"

using namespace hive;

extern "C" __global__ void __launch_bounds__(256)
flood_quad(const uint32_t *__restrict__ occ_in, const uint8_t *__restrict__ start, int n, uint32_t *__restrict__ out,
           int *__restrict__ trips)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long b = t >> 2;
    const bool valid = b < n;
    const long long bb = valid ? b : 0;
    BB occ = bb_load(occ_in + bb * 6);
    const unsigned c0 = start[bb];
    occ = bb_andn(occ, bb_bit(c0));                 // the mover lifted off
    BB S[6];
    occupancy_views(occ, S);
    const SlideCtx ctx = make_slide_ctx(occ, S);
    BB x = bb_bit(c0);
    if (!valid) x = bb_zero();
    int it = 0;
    for (;;) {
        BB y = bb_or(x, slide_step(ctx, x));
        y = bb_or(y, slide_step(ctx, y));
        ++it;
        const bool same = bb_eq(x, y);
        x = y;
        if (!__any(!same)) break;
    }
    x = bb_andn(x, bb_bit(c0));
    if (valid) bb_store(out + b * 6, x);
    if (trips && (threadIdx.x & 63) == 0) atomicAdd(trips, it);
}

// ---------------------------------------------------------------- pair layout
namespace pr {
struct B3 {
    uint32_t w0, w1, w2;
};
constexpr int kSwap = 1 | (0 << 2) | (3 << 4) | (2 << 6);      // quad_perm [1,0,3,2]: the other lane of the pair
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ int half() { return (int)(threadIdx.x & 1u); }
__device__ __forceinline__ B3 zero() { return B3{0u, 0u, 0u}; }
__device__ __forceinline__ B3 band(B3 a, B3 b) { return B3{a.w0 & b.w0, a.w1 & b.w1, a.w2 & b.w2}; }
__device__ __forceinline__ B3 bor(B3 a, B3 b) { return B3{a.w0 | b.w0, a.w1 | b.w1, a.w2 | b.w2}; }
__device__ __forceinline__ B3 bxor(B3 a, B3 b) { return B3{a.w0 ^ b.w0, a.w1 ^ b.w1, a.w2 ^ b.w2}; }
__device__ __forceinline__ B3 bandn(B3 a, B3 b) { return B3{a.w0 & ~b.w0, a.w1 & ~b.w1, a.w2 & ~b.w2}; }
__device__ __forceinline__ B3 bit(unsigned cell)
{
    const unsigned row = cell / 12u, col = cell - row * 12u;
    const unsigned r6 = row >= 6u ? row - 6u : row;
    const uint32_t m = (cell < 144u && (int)(row >= 6u) == half()) ? (1u << (((r6 & 1u) << 4) | col)) : 0u;
    const unsigned wi = r6 >> 1;
    return B3{wi == 0u ? m : 0u, wi == 1u ? m : 0u, wi == 2u ? m : 0u};
}
__device__ __forceinline__ bool eq(B3 a, B3 b)
{
    uint32_t d = (a.w0 ^ b.w0) | (a.w1 ^ b.w1) | (a.w2 ^ b.w2);
    d |= dpp<kSwap>(d);
    return d == 0u;
}
// row j -> j+1 (11 -> 0)
__device__ __forceinline__ B3 up(B3 x)
{
    const uint32_t other2 = dpp<kSwap>(x.w2);
    return B3{__builtin_amdgcn_alignbit(x.w0, other2, 16), __builtin_amdgcn_alignbit(x.w1, x.w0, 16),
              __builtin_amdgcn_alignbit(x.w2, x.w1, 16)};
}
// row j -> j-1
__device__ __forceinline__ B3 down(B3 x)
{
    const uint32_t other0 = dpp<kSwap>(x.w0);
    return B3{__builtin_amdgcn_alignbit(x.w1, x.w0, 16), __builtin_amdgcn_alignbit(x.w2, x.w1, 16),
              __builtin_amdgcn_alignbit(other0, x.w2, 16)};
}
__device__ __forceinline__ B3 right(B3 x) { return B3{col_right(x.w0), col_right(x.w1), col_right(x.w2)}; }
__device__ __forceinline__ B3 left(B3 x) { return B3{col_left(x.w0), col_left(x.w1), col_left(x.w2)}; }
__device__ __forceinline__ B3 shift_dirs(B3 a0, B3 a1, B3 a2, B3 a3, B3 a4, B3 a5)
{
    const B3 u2 = up(a2), d5 = down(a5);
    const B3 r = right(bor(a0, up(a1)));
    const B3 l = left(bor(a3, down(a4)));
    return B3{u2.w0 | d5.w0 | r.w0 | l.w0, u2.w1 | d5.w1 | r.w1 | l.w1, u2.w2 | d5.w2 | r.w2 | l.w2};
}
struct Ctx {
    B3 cs[6], allowed;
};
__device__ __forceinline__ Ctx make_ctx(B3 occ)
{
    const B3 u = up(occ), d = down(occ);
    B3 S[6];
    S[5] = u; S[2] = d; S[0] = left(occ); S[3] = right(occ); S[1] = left(d); S[4] = right(u);
    Ctx c;
    c.cs[0] = bxor(S[5], S[1]); c.cs[1] = bxor(S[0], S[2]); c.cs[2] = bxor(S[1], S[3]);
    c.cs[3] = bxor(S[2], S[4]); c.cs[4] = bxor(S[3], S[5]); c.cs[5] = bxor(S[4], S[0]);
    const B3 nocc = bor(bor(bor(S[0], S[1]), bor(S[2], S[3])), bor(S[4], S[5]));
    c.allowed = bandn(nocc, occ);
    return c;
}
__device__ __forceinline__ B3 slide_step(const Ctx &c, B3 x)
{
    return band(shift_dirs(band(x, c.cs[0]), band(x, c.cs[1]), band(x, c.cs[2]), band(x, c.cs[3]), band(x, c.cs[4]),
                           band(x, c.cs[5])), c.allowed);
}
}  // namespace pr

extern "C" __global__ void __launch_bounds__(256)
flood_pair(const uint32_t *__restrict__ occ_in, const uint8_t *__restrict__ start, int n, uint32_t *__restrict__ out,
           int *__restrict__ trips)
{
    using namespace pr;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long b = t >> 1;
    const bool valid = b < n;
    const long long bb = valid ? b : 0;
    const uint32_t *p = occ_in + bb * 6 + 3 * half();
    B3 occ{p[0], p[1], p[2]};
    const unsigned c0 = start[bb];
    occ = bandn(occ, bit(c0));
    const Ctx ctx = make_ctx(occ);
    B3 x = bit(c0);
    if (!valid) x = zero();
    int it = 0;
    for (;;) {
        B3 y = bor(x, slide_step(ctx, x));
        y = bor(y, slide_step(ctx, y));
        ++it;
        const bool same = eq(x, y);
        x = y;
        if (!__any(!same)) break;
    }
    x = bandn(x, bit(c0));
    if (valid) {
        uint32_t *o = out + b * 6 + 3 * half();
        o[0] = x.w0; o[1] = x.w1; o[2] = x.w2;
    }
    if (trips && (threadIdx.x & 63) == 0) atomicAdd(trips, it);
}

extern "C" int flood_launch(int layout, const uint32_t *occ, const uint8_t *start, int n, uint32_t *out, int *trips, void *stream)
{
    const long long threads = (long long)n * (layout == 0 ? 4 : 2);
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (layout == 0) hipLaunchKernelGGL(flood_quad, dim3(grid), dim3(256), 0, (hipStream_t)stream, occ, start, n, out, trips);
    else hipLaunchKernelGGL(flood_pair, dim3(grid), dim3(256), 0, (hipStream_t)stream, occ, start, n, out, trips);
    return (int)hipGetLastError();
}
