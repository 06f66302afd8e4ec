// hive_bb.hpp -- 144-cell hex-torus bitboards for gfx950 (CDNA4) wavefronts.
//
// Layout: six 32-bit VGPRs; word r holds board rows 2r (bits 0-11) and 2r+1 (bits 16-27).
// A row step is one v_alignbit_b32 per word, a column step is two packed 16-bit shifts
// (v_pk_lshlrev_b16 / v_pk_lshrrev_b16) plus one v_and_or_b32 per word; bits 12-15 and
// 28-31 stay zero under every operation here.
//
// Geometry (reference tile.py:111-123): cell (j, c) touches (j+-1, c), (j, c+-1),
// (j-1, c-1), (j+1, c+1), all mod 12.  Ring order used for the slide ("gate") test:
//   d0 = (0,+1) R   d1 = (+1,+1) UR   d2 = (+1,0) U   d3 = (0,-1) L   d4 = (-1,-1) DL   d5 = (-1,0) D
// consecutive directions are mutually adjacent, so the two cells flanking a step along
// d_i are c + d_(i-1) and c + d_(i+1) (reference move_checker.py:193-198 "overlap_tiles").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hive {

struct BB {
    uint32_t w[6];
};

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

#define HIVE_UNROLL _Pragma("unroll")

__device__ __forceinline__ BB bb_zero()
{
    BB r;
    HIVE_UNROLL for (int i = 0; i < 6; ++i) r.w[i] = 0u;
    return r;
}
__device__ __forceinline__ BB bb_and(BB a, BB b)
{
    BB r;
    HIVE_UNROLL for (int i = 0; i < 6; ++i) r.w[i] = a.w[i] & b.w[i];
    return r;
}
__device__ __forceinline__ BB bb_or(BB a, BB b)
{
    BB r;
    HIVE_UNROLL for (int i = 0; i < 6; ++i) r.w[i] = a.w[i] | b.w[i];
    return r;
}
__device__ __forceinline__ BB bb_xor(BB a, BB b)
{
    BB r;
    HIVE_UNROLL for (int i = 0; i < 6; ++i) r.w[i] = a.w[i] ^ b.w[i];
    return r;
}
__device__ __forceinline__ BB bb_andn(BB a, BB b)   // a & ~b
{
    BB r;
    HIVE_UNROLL for (int i = 0; i < 6; ++i) r.w[i] = a.w[i] & ~b.w[i];
    return r;
}
__device__ __forceinline__ BB bb_not(BB a)          // complement inside the 144 valid bits
{
    BB r;
    HIVE_UNROLL for (int i = 0; i < 6; ++i) r.w[i] = ~a.w[i] & 0x0FFF0FFFu;
    return r;
}
__device__ __forceinline__ bool bb_any(BB a)
{
    return (a.w[0] | a.w[1] | a.w[2] | a.w[3] | a.w[4] | a.w[5]) != 0u;
}
__device__ __forceinline__ bool bb_eq(BB a, BB b)
{
    return ((a.w[0] ^ b.w[0]) | (a.w[1] ^ b.w[1]) | (a.w[2] ^ b.w[2]) | (a.w[3] ^ b.w[3]) |
            (a.w[4] ^ b.w[4]) | (a.w[5] ^ b.w[5])) == 0u;
}
__device__ __forceinline__ BB bb_select(bool c, BB a, BB b)
{
    BB r;
    HIVE_UNROLL for (int i = 0; i < 6; ++i) r.w[i] = c ? a.w[i] : b.w[i];
    return r;
}
// single-cell board; cell >= 144 gives the empty board
__device__ __forceinline__ BB bb_bit(unsigned cell)
{
    unsigned row = cell / 12u, col = cell - row * 12u;
    unsigned wi = row >> 1, bit = ((row & 1u) << 4) | col;
    uint32_t m = cell < 144u ? (1u << bit) : 0u;
    BB r;
    HIVE_UNROLL for (int i = 0; i < 6; ++i) r.w[i] = (wi == (unsigned)i) ? m : 0u;
    return r;
}
__device__ __forceinline__ void cell_word_bit(unsigned cell, unsigned &wi, unsigned &bit)
{
    unsigned row = cell / 12u, col = cell - row * 12u;
    wi = row >> 1;
    bit = ((row & 1u) << 4) | col;
}

// row j -> j+1 (wraps 11 -> 0)
__device__ __forceinline__ BB bb_up(BB x)
{
    BB r;
    r.w[0] = __builtin_amdgcn_alignbit(x.w[0], x.w[5], 16);
    r.w[1] = __builtin_amdgcn_alignbit(x.w[1], x.w[0], 16);
    r.w[2] = __builtin_amdgcn_alignbit(x.w[2], x.w[1], 16);
    r.w[3] = __builtin_amdgcn_alignbit(x.w[3], x.w[2], 16);
    r.w[4] = __builtin_amdgcn_alignbit(x.w[4], x.w[3], 16);
    r.w[5] = __builtin_amdgcn_alignbit(x.w[5], x.w[4], 16);
    return r;
}
// row j -> j-1
__device__ __forceinline__ BB bb_down(BB x)
{
    BB r;
    r.w[0] = __builtin_amdgcn_alignbit(x.w[1], x.w[0], 16);
    r.w[1] = __builtin_amdgcn_alignbit(x.w[2], x.w[1], 16);
    r.w[2] = __builtin_amdgcn_alignbit(x.w[3], x.w[2], 16);
    r.w[3] = __builtin_amdgcn_alignbit(x.w[4], x.w[3], 16);
    r.w[4] = __builtin_amdgcn_alignbit(x.w[5], x.w[4], 16);
    r.w[5] = __builtin_amdgcn_alignbit(x.w[0], x.w[5], 16);
    return r;
}
__device__ __forceinline__ uint32_t col_right(uint32_t x)   // col c -> c+1 (11 -> 0)
{
    u16x2 v = __builtin_bit_cast(u16x2, x);
    u16x2 a = v << (unsigned short)1;
    u16x2 b = v >> (unsigned short)11;
    return (__builtin_bit_cast(uint32_t, a) & 0x0FFF0FFFu) | __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ uint32_t col_left(uint32_t x)    // col c -> c-1 (0 -> 11)
{
    u16x2 v = __builtin_bit_cast(u16x2, x);
    u16x2 a = v >> (unsigned short)1;
    u16x2 b = v << (unsigned short)11;
    return __builtin_bit_cast(uint32_t, a) | (__builtin_bit_cast(uint32_t, b) & 0x08000800u);
}
__device__ __forceinline__ BB bb_right(BB x)
{
    BB r;
    HIVE_UNROLL for (int i = 0; i < 6; ++i) r.w[i] = col_right(x.w[i]);
    return r;
}
__device__ __forceinline__ BB bb_left(BB x)
{
    BB r;
    HIVE_UNROLL for (int i = 0; i < 6; ++i) r.w[i] = col_left(x.w[i]);
    return r;
}

// union of the six neighbours of every set cell
__device__ __forceinline__ BB bb_neighbours(BB x)
{
    BB u = bb_up(x), d = bb_down(x);
    BB r = bb_right(bb_or(x, u));   // R and UR
    BB l = bb_left(bb_or(x, d));    // L and DL
    BB o;
    HIVE_UNROLL for (int i = 0; i < 6; ++i) o.w[i] = u.w[i] | d.w[i] | r.w[i] | l.w[i];
    return o;
}

// Everything a sliding piece needs about the board with the mover lifted off.
struct SlideCtx {
    BB cs[6];      // cs[i] bit c: exactly one of the two cells flanking the step c -> c+d_i is occupied
    BB allowed;    // empty and touching the hive
    BB nocc;       // cells touching the hive (occupied or not)
};

// S_i bit c = occ[c + d_i]
__device__ __forceinline__ void occupancy_views(BB occ, BB S[6])
{
    BB u = bb_up(occ), d = bb_down(occ);
    S[5] = u;              // occ[c + D]  = occ moved up
    S[2] = d;              // occ[c + U]  = occ moved down
    S[0] = bb_left(occ);   // occ[c + R]
    S[3] = bb_right(occ);  // occ[c + L]
    S[1] = bb_left(d);     // occ[c + UR]
    S[4] = bb_right(u);    // occ[c + DL]
}

__device__ __forceinline__ SlideCtx make_slide_ctx(BB occ)
{
    BB S[6];
    occupancy_views(occ, S);
    SlideCtx c;
    c.cs[0] = bb_xor(S[5], S[1]);
    c.cs[1] = bb_xor(S[0], S[2]);
    c.cs[2] = bb_xor(S[1], S[3]);
    c.cs[3] = bb_xor(S[2], S[4]);
    c.cs[4] = bb_xor(S[3], S[5]);
    c.cs[5] = bb_xor(S[4], S[0]);
    HIVE_UNROLL for (int i = 0; i < 6; ++i)
        c.nocc.w[i] = S[0].w[i] | S[1].w[i] | S[2].w[i] | S[3].w[i] | S[4].w[i] | S[5].w[i];
    c.allowed = bb_andn(c.nocc, occ);
    return c;
}

// one slide step of every cell in x: k == 1 gate rule (reference move_checker.py:189-214)
__device__ __forceinline__ BB slide_raw(const SlideCtx &c, BB x)
{
    BB a0 = bb_and(x, c.cs[0]), a1 = bb_and(x, c.cs[1]), a2 = bb_and(x, c.cs[2]);
    BB a3 = bb_and(x, c.cs[3]), a4 = bb_and(x, c.cs[4]), a5 = bb_and(x, c.cs[5]);
    BB u2 = bb_up(a2), d5 = bb_down(a5);
    BB r = bb_right(bb_or(a0, bb_up(a1)));
    BB l = bb_left(bb_or(a3, bb_down(a4)));
    BB o;
    HIVE_UNROLL for (int i = 0; i < 6; ++i) o.w[i] = u2.w[i] | d5.w[i] | r.w[i] | l.w[i];
    return o;
}
__device__ __forceinline__ BB slide_step(const SlideCtx &c, BB x)
{
    return bb_and(slide_raw(c, x), c.allowed);
}

// lowest set cell of a non-empty board, as a single-cell board
__device__ __forceinline__ BB bb_lowest(BB x)
{
    BB r;
    bool found = false;
    HIVE_UNROLL for (int i = 0; i < 6; ++i) {
        uint32_t low = x.w[i] & (0u - x.w[i]);
        r.w[i] = found ? 0u : low;
        found = found || (x.w[i] != 0u);
    }
    return r;
}

}  // namespace hive
