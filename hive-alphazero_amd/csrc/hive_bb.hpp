// hive_bb.hpp -- 144-cell hex-torus bitboards spread over the lanes of a gfx950 wavefront.
//
// One board = one DPP quad (4 consecutive lanes).  Lane l = 0..2 of the quad holds board rows
// 4l..4l+3 as four 16-bit fields in two VGPRs (lo = rows 4l | 4l+1 << 16, hi = rows 4l+2 | 4l+3 << 16,
// column c = bit c of the field); lane 3 of the quad always holds zero.  In memory a board is six
// 32-bit words (word r = rows 2r, 2r+1), i.e. lane l owns words 2l and 2l+1.
//
//   row step   : 1 v_mov_b32_dpp quad_perm (neighbour lane's edge rows) + 2 v_alignbit_b32
//   column step: v_pk_lshlrev_b16 + v_pk_lshrrev_b16 + v_and_or_b32 per VGPR
//   any / equal: 2 v_or_b32_dpp quad_perm reductions
//
// A wavefront therefore carries 16 boards; a bitboard costs 2 VGPRs per lane (not 6), which
// keeps the kernels at 8 waves/SIMD, and the critical path of one flood iteration is ~3x
// shorter than with one board per lane.
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
    uint32_t lo, hi;
};

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

#define HIVE_UNROLL _Pragma("unroll")

// quad_perm selectors: lane i of every quad reads lane sel[i]
constexpr int kQuadPrev = 2 | (0 << 2) | (1 << 4) | (3 << 6);   // lanes 0,1,2 read 2,0,1 (cyclic previous)
constexpr int kQuadNext = 1 | (2 << 2) | (0 << 4) | (3 << 6);   // lanes 0,1,2 read 1,2,0 (cyclic next)
constexpr int kQuadSwap1 = 1 | (0 << 2) | (3 << 4) | (2 << 6);
constexpr int kQuadSwap2 = 2 | (3 << 2) | (0 << 4) | (1 << 6);
constexpr int kQuadB0 = 0, kQuadB1 = 0x55;                     // broadcast lane 0 / 1

template <int CTRL>
__device__ __forceinline__ uint32_t quad(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);   // bound_ctrl: no "old" operand to set up
}

__device__ __forceinline__ int quad_lane() { return (int)(threadIdx.x & 3u); }

__device__ __forceinline__ BB bb_zero() { return BB{0u, 0u}; }
__device__ __forceinline__ BB bb_and(BB a, BB b) { return BB{a.lo & b.lo, a.hi & b.hi}; }
__device__ __forceinline__ BB bb_or(BB a, BB b) { return BB{a.lo | b.lo, a.hi | b.hi}; }
__device__ __forceinline__ BB bb_or3(BB a, BB b, BB c) { return BB{a.lo | b.lo | c.lo, a.hi | b.hi | c.hi}; }
__device__ __forceinline__ BB bb_xor(BB a, BB b) { return BB{a.lo ^ b.lo, a.hi ^ b.hi}; }
__device__ __forceinline__ BB bb_andn(BB a, BB b) { return BB{a.lo & ~b.lo, a.hi & ~b.hi}; }
// all 144 cells (zero in lane 3)
__device__ __forceinline__ BB bb_full()
{
    uint32_t m = quad_lane() < 3 ? 0x0FFF0FFFu : 0u;
    return BB{m, m};
}
__device__ __forceinline__ BB bb_not(BB a) { return bb_andn(bb_full(), a); }

// OR over the quad, result in every lane
__device__ __forceinline__ uint32_t quad_or(uint32_t t)
{
    t |= quad<kQuadSwap1>(t);
    t |= quad<kQuadSwap2>(t);
    return t;
}
__device__ __forceinline__ bool bb_any(BB a) { return quad_or(a.lo | a.hi) != 0u; }
__device__ __forceinline__ bool bb_eq(BB a, BB b) { return quad_or((a.lo ^ b.lo) | (a.hi ^ b.hi)) == 0u; }

// single-cell board; cell >= 144 gives the empty board
__device__ __forceinline__ BB bb_bit(unsigned cell)
{
    unsigned row = cell / 12u, col = cell - row * 12u;
    unsigned f = row & 3u;
    uint32_t m = (cell < 144u && (row >> 2) == (unsigned)quad_lane()) ? (1u << (((f & 1u) << 4) | col)) : 0u;
    return BB{(f < 2u) ? m : 0u, (f < 2u) ? 0u : m};
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
    uint32_t prev_hi = quad<kQuadPrev>(x.hi);
    return BB{__builtin_amdgcn_alignbit(x.lo, prev_hi, 16), __builtin_amdgcn_alignbit(x.hi, x.lo, 16)};
}
// row j -> j-1
__device__ __forceinline__ BB bb_down(BB x)
{
    uint32_t next_lo = quad<kQuadNext>(x.lo);
    return BB{__builtin_amdgcn_alignbit(x.hi, x.lo, 16), __builtin_amdgcn_alignbit(next_lo, x.hi, 16)};
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
__device__ __forceinline__ BB bb_right(BB x) { return BB{col_right(x.lo), col_right(x.hi)}; }
__device__ __forceinline__ BB bb_left(BB x) { return BB{col_left(x.lo), col_left(x.hi)}; }

// union of the six neighbours of every set cell
__device__ __forceinline__ BB bb_neighbours(BB x)
{
    BB u = bb_up(x), d = bb_down(x);
    BB r = bb_right(bb_or(x, u));   // R and UR
    BB l = bb_left(bb_or(x, d));    // L and DL
    return BB{u.lo | d.lo | r.lo | l.lo, u.hi | d.hi | r.hi | l.hi};
}

// per-direction source boards -> union of the shifted boards (d0..d5 = R UR U L DL D)
__device__ __forceinline__ BB shift_dirs(BB a0, BB a1, BB a2, BB a3, BB a4, BB a5)
{
    BB u2 = bb_up(a2), d5 = bb_down(a5);
    BB r = bb_right(bb_or(a0, bb_up(a1)));
    BB l = bb_left(bb_or(a3, bb_down(a4)));
    return BB{u2.lo | d5.lo | r.lo | l.lo, u2.hi | d5.hi | r.hi | l.hi};
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

__device__ __forceinline__ SlideCtx make_slide_ctx(BB occ, const BB S[6])
{
    SlideCtx c;
    c.cs[0] = bb_xor(S[5], S[1]);
    c.cs[1] = bb_xor(S[0], S[2]);
    c.cs[2] = bb_xor(S[1], S[3]);
    c.cs[3] = bb_xor(S[2], S[4]);
    c.cs[4] = bb_xor(S[3], S[5]);
    c.cs[5] = bb_xor(S[4], S[0]);
    c.nocc = bb_or(bb_or3(S[0], S[1], S[2]), bb_or3(S[3], S[4], S[5]));
    c.allowed = bb_andn(c.nocc, occ);
    return c;
}

// one slide step of every cell in x: k == 1 gate rule (reference move_checker.py:189-214)
__device__ __forceinline__ BB slide_raw(const SlideCtx &c, BB x)
{
    return shift_dirs(bb_and(x, c.cs[0]), bb_and(x, c.cs[1]), bb_and(x, c.cs[2]), bb_and(x, c.cs[3]),
                      bb_and(x, c.cs[4]), bb_and(x, c.cs[5]));
}
__device__ __forceinline__ BB slide_step(const SlideCtx &c, BB x) { return bb_and(slide_raw(c, x), c.allowed); }

// lowest set cell of a board (row-major order), as a single-cell board; empty -> empty
__device__ __forceinline__ BB bb_lowest(BB x)
{
    uint32_t llo = x.lo & (0u - x.lo);
    uint32_t lhi = x.lo ? 0u : (x.hi & (0u - x.hi));
    uint32_t any = x.lo | x.hi;
    uint32_t a0 = quad<kQuadB0>(any), a1 = quad<kQuadB1>(any);
    int l = quad_lane();
    bool keep = (l == 0) || (l == 1 && a0 == 0u) || (l == 2 && (a0 | a1) == 0u);
    return BB{keep ? llo : 0u, keep ? lhi : 0u};
}

// board <-> six-word memory image (lane l owns words 2l, 2l+1; lane 3 nothing)
__device__ __forceinline__ BB bb_load(const uint32_t *p)
{
    int l = quad_lane();
    if (l < 3) {
        uint2 v = *reinterpret_cast<const uint2 *>(p + 2 * l);
        return BB{v.x, v.y};
    }
    return BB{0u, 0u};
}
__device__ __forceinline__ void bb_store(uint32_t *p, BB v)
{
    int l = quad_lane();
    if (l < 3) {
        p[2 * l] = v.lo;
        p[2 * l + 1] = v.hi;
    }
}

}  // namespace hive
