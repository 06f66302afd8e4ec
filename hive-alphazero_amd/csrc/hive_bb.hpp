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

// ---------------------------------------------------------------- the pair layout
// One board = TWO consecutive lanes, three VGPRs per lane: lane 0 of the pair holds memory words 0-2 (rows 0-5), lane 1
// words 3-5 (rows 6-11).  A wave carries 32 boards and no lane idles (the quad layout leaves every fourth lane empty), so a
// board operation costs 3 instructions per 32 boards instead of 2 per 16; the price is a third more registers per board
// value and a longer wave-level flood loop (the slowest of 32 boards, not of 16).  Used by the movegen kernel for launches
// large enough to fill the card either way (hive_env.hip: hive_piece_kernel<false, false, PairLay>); measured first on
// the Ant flood alone (tools/dev/flood_layouts.hip, profiles/r03_flood_layouts.md).
struct B3 {
    uint32_t w0, w1, w2;
};
constexpr int kPairB0 = 0 | (0 << 2) | (2 << 4) | (2 << 6);     // both lanes of a pair read its lane 0

__device__ __forceinline__ int pair_lane() { return (int)(threadIdx.x & 1u); }
__device__ __forceinline__ B3 bb_and(B3 a, B3 b) { return B3{a.w0 & b.w0, a.w1 & b.w1, a.w2 & b.w2}; }
__device__ __forceinline__ B3 bb_or(B3 a, B3 b) { return B3{a.w0 | b.w0, a.w1 | b.w1, a.w2 | b.w2}; }
__device__ __forceinline__ B3 bb_or3(B3 a, B3 b, B3 c) { return B3{a.w0 | b.w0 | c.w0, a.w1 | b.w1 | c.w1, a.w2 | b.w2 | c.w2}; }
__device__ __forceinline__ B3 bb_xor(B3 a, B3 b) { return B3{a.w0 ^ b.w0, a.w1 ^ b.w1, a.w2 ^ b.w2}; }
__device__ __forceinline__ B3 bb_andn(B3 a, B3 b) { return B3{a.w0 & ~b.w0, a.w1 & ~b.w1, a.w2 & ~b.w2}; }
__device__ __forceinline__ uint32_t pair_or(uint32_t t) { return t | quad<kQuadSwap1>(t); }
__device__ __forceinline__ bool bb_any(B3 a) { return pair_or(a.w0 | a.w1 | a.w2) != 0u; }
__device__ __forceinline__ bool bb_eq(B3 a, B3 b) { return pair_or((a.w0 ^ b.w0) | (a.w1 ^ b.w1) | (a.w2 ^ b.w2)) == 0u; }
__device__ __forceinline__ B3 bb_up(B3 x)
{
    uint32_t other2 = quad<kQuadSwap1>(x.w2);
    return B3{__builtin_amdgcn_alignbit(x.w0, other2, 16), __builtin_amdgcn_alignbit(x.w1, x.w0, 16),
              __builtin_amdgcn_alignbit(x.w2, x.w1, 16)};
}
__device__ __forceinline__ B3 bb_down(B3 x)
{
    uint32_t other0 = quad<kQuadSwap1>(x.w0);
    return B3{__builtin_amdgcn_alignbit(x.w1, x.w0, 16), __builtin_amdgcn_alignbit(x.w2, x.w1, 16),
              __builtin_amdgcn_alignbit(other0, x.w2, 16)};
}
__device__ __forceinline__ B3 bb_right(B3 x) { return B3{col_right(x.w0), col_right(x.w1), col_right(x.w2)}; }
__device__ __forceinline__ B3 bb_left(B3 x) { return B3{col_left(x.w0), col_left(x.w1), col_left(x.w2)}; }
__device__ __forceinline__ B3 bb_neighbours(B3 x)
{
    B3 u = bb_up(x), d = bb_down(x);
    B3 r = bb_right(bb_or(x, u));
    B3 l = bb_left(bb_or(x, d));
    return B3{u.w0 | d.w0 | r.w0 | l.w0, u.w1 | d.w1 | r.w1 | l.w1, u.w2 | d.w2 | r.w2 | l.w2};
}
__device__ __forceinline__ B3 shift_dirs(B3 a0, B3 a1, B3 a2, B3 a3, B3 a4, B3 a5)
{
    B3 u2 = bb_up(a2), d5 = bb_down(a5);
    B3 r = bb_right(bb_or(a0, bb_up(a1)));
    B3 l = bb_left(bb_or(a3, bb_down(a4)));
    return B3{u2.w0 | d5.w0 | r.w0 | l.w0, u2.w1 | d5.w1 | r.w1 | l.w1, u2.w2 | d5.w2 | r.w2 | l.w2};
}
__device__ __forceinline__ B3 bb_lowest(B3 x)
{
    uint32_t l0 = x.w0 & (0u - x.w0);
    uint32_t l1 = x.w0 ? 0u : (x.w1 & (0u - x.w1));
    uint32_t l2 = (x.w0 | x.w1) ? 0u : (x.w2 & (0u - x.w2));
    uint32_t any = x.w0 | x.w1 | x.w2;
    uint32_t a0 = quad<kPairB0>(any);
    bool keep = pair_lane() == 0 || a0 == 0u;
    return B3{keep ? l0 : 0u, keep ? l1 : 0u, keep ? l2 : 0u};
}
__device__ __forceinline__ int bb_popc_lane(BB x) { return __popc(x.lo) + __popc(x.hi); }      // this lane's share
__device__ __forceinline__ int bb_popc_lane(B3 x) { return __popc(x.w0) + __popc(x.w1) + __popc(x.w2); }

// Everything a sliding piece needs about the board with the mover lifted off.
template <class B>
struct SlideCtxT {
    B cs[6];      // cs[i] bit c: exactly one of the two cells flanking the step c -> c+d_i is occupied
    B allowed;    // empty and touching the hive
    B nocc;       // cells touching the hive (occupied or not)
};
using SlideCtx = SlideCtxT<BB>;

// S_i bit c = occ[c + d_i]
template <class B>
__device__ __forceinline__ void occupancy_views(B occ, B S[6])
{
    B u = bb_up(occ), d = bb_down(occ);
    S[5] = u;              // occ[c + D]  = occ moved up
    S[2] = d;              // occ[c + U]  = occ moved down
    S[0] = bb_left(occ);   // occ[c + R]
    S[3] = bb_right(occ);  // occ[c + L]
    S[1] = bb_left(d);     // occ[c + UR]
    S[4] = bb_right(u);    // occ[c + DL]
}

template <class B>
__device__ __forceinline__ SlideCtxT<B> make_slide_ctx(B occ, const B S[6])
{
    SlideCtxT<B> c;
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
template <class B>
__device__ __forceinline__ B slide_raw(const SlideCtxT<B> &c, B x)
{
    return shift_dirs(bb_and(x, c.cs[0]), bb_and(x, c.cs[1]), bb_and(x, c.cs[2]), bb_and(x, c.cs[3]),
                      bb_and(x, c.cs[4]), bb_and(x, c.cs[5]));
}
template <class B>
__device__ __forceinline__ B slide_step(const SlideCtxT<B> &c, B x) { return bb_and(slide_raw(c, x), c.allowed); }

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


// The two layouts behind one set of names, for code that is generic over them (hive_env.hip: piece_dests, placement_board).
struct QuadLay {
    using B = BB;
    static constexpr int kLanes = 4;           // lanes per board
    static constexpr int kShift = 2;
    static __device__ __forceinline__ int sub() { return quad_lane(); }
    static __device__ __forceinline__ bool holds() { return quad_lane() < 3; }       // this lane carries words of the board
    static __device__ __forceinline__ int word0() { return 2 * quad_lane(); }        // its first memory word
    static __device__ __forceinline__ B zero() { return bb_zero(); }
    static __device__ __forceinline__ B full() { return bb_full(); }
    static __device__ __forceinline__ B bit(unsigned cell) { return bb_bit(cell); }
    static __device__ __forceinline__ B load(const uint32_t *p) { return bb_load(p); }
    static __device__ __forceinline__ void store(uint32_t *p, B v) { bb_store(p, v); }
};
struct PairLay {
    using B = B3;
    static constexpr int kLanes = 2;
    static constexpr int kShift = 1;
    static __device__ __forceinline__ int sub() { return pair_lane(); }
    static __device__ __forceinline__ bool holds() { return true; }
    static __device__ __forceinline__ int word0() { return 3 * pair_lane(); }
    static __device__ __forceinline__ B zero() { return B3{0u, 0u, 0u}; }
    static __device__ __forceinline__ B full() { return B3{0x0FFF0FFFu, 0x0FFF0FFFu, 0x0FFF0FFFu}; }
    static __device__ __forceinline__ B bit(unsigned cell)
    {
        unsigned row = cell / 12u, col = cell - row * 12u;
        unsigned r6 = row >= 6u ? row - 6u : row;
        uint32_t m = (cell < 144u && (int)(row >= 6u) == pair_lane()) ? (1u << (((r6 & 1u) << 4) | col)) : 0u;
        unsigned wi = r6 >> 1;
        return B3{wi == 0u ? m : 0u, wi == 1u ? m : 0u, wi == 2u ? m : 0u};
    }
    static __device__ __forceinline__ B load(const uint32_t *p)
    {
        const uint32_t *q = p + 3 * pair_lane();
        return B3{q[0], q[1], q[2]};
    }
    static __device__ __forceinline__ void store(uint32_t *p, B v)
    {
        uint32_t *q = p + 3 * pair_lane();
        q[0] = v.w0; q[1] = v.w1; q[2] = v.w2;
    }
};

}  // namespace hive
