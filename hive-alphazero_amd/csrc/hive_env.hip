// hive_env.hip -- batched Hive env kernels for gfx950 (MI355X) + the C ABI of include/hive_abi.h.
//
// Work decomposition: a (board, piece) pair is one DPP quad -- the 144-bit
// boards are spread over three lanes (hive_bb.hpp) -- so a wavefront carries 16 pairs.  A
// workgroup is 11 waves, each working on one piece slot, hence every wave runs exactly one piece
// type (no type divergence); it owns 16 boards (movegen) or 8 boards x both colours (planes).
// The packed records, the occupancy / top-colour boards and the per-cell piece sets are staged in
// LDS; three of the waves first settle the one-hive question for all pieces of the workgroup on the
// 22-node piece graph (one lane per piece) and build the per-board placement / next_move_tiles
// boards; every (board, slot) quad writes its destination board straight to HBM -- the legal set in
// the reference's pre_actions form; the same launch (LIST) or hive_list_kernel turns it into the ascending id list with
// wave ballots and mbcnt prefix sums when a caller asks for it.  Movegen launches of 16,384 boards and more run the same
// kernel body in the PAIR layout (hive_bb.hpp: a board on two lanes, 32 boards per workgroup; template parameter L).
//
// Reference semantics implemented here (paths relative to the reference root):
//   env_hive.py:196-304   pre_actions / get_actions / encode_action
//   env_hive.py:320-485   make_state_value / mini_black_actions (56 planes)
//   env_hive.py:99-171    move
//   move_checker.py:9-265 is_valid_move and the rule helpers
//   pieces.py:35-158      per-piece move_is_valid
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <new>
#include <type_traits>
#include <cstring>
#include <string>

#include "../../include/hive_abi.h"
#include "hive_bb.hpp"
#include "hive_tables.hpp"
#include "hive_step.hpp"

namespace hive {

__device__ const Tables d_tables = kTables;
#ifdef HIVE_DBG_ITERS
__device__ unsigned long long d_iters[32];
__device__ unsigned long long d_hist[6][32];     // trips-per-wave histogram by piece type
__device__ unsigned long long d_stamps[11][8];
#define HIVE_STAMP(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x % 16 == 0) atomicAdd(&d_stamps[wv][i], (unsigned long long)(clock64() - t_start)); } while (0)
#else
#define HIVE_STAMP(i) do { } while (0)
#endif
#ifdef HIVE_DBG_ITERS
#define HIVE_COUNT_ITER() do { ++dbg_it; } while (0)
#else
#define HIVE_COUNT_ITER() do { } while (0)
#endif

constexpr int NW = 11;                 // waves per workgroup = piece slots per colour
#ifndef HIVE_HIVE_STEPS
#define HIVE_HIVE_STEPS 2              // flood expansions per convergence test: one-hive flood
#endif
#ifndef HIVE_ANT_STEPS
#define HIVE_ANT_STEPS 2               // ... Ant reach flood
#endif
#ifndef HIVE_PIECE_WPE
#define HIVE_PIECE_WPE 6                // waves per SIMD asked of the register allocator (two 11-wave workgroups per CU)
#endif
#ifndef HIVE_PAIR_ANT_STEPS
#define HIVE_PAIR_ANT_STEPS HIVE_ANT_STEPS
#endif
#ifndef HIVE_PAIR_WPE
#define HIVE_PAIR_WPE 6                 // ... of the pair-layout variant (three registers per board value instead of two)
#endif
constexpr unsigned kHand = 255u;
constexpr int kPinWaves = 3;           // waves 0-2 run the one-hive test for all 176 (board, piece) items of a workgroup

enum PieceType { T_QUEEN = 0, T_BEETLE = 1, T_SPIDER = 2, T_GRASS = 3, T_ANT = 4 };
__device__ __forceinline__ int slot_type(int slot)
{
    // Q B B S S G G G A A A
    return slot == 0 ? T_QUEEN : slot < 3 ? T_BEETLE : slot < 5 ? T_SPIDER : slot < 8 ? T_GRASS : T_ANT;
}
__device__ __forceinline__ int slot_group_start(int slot)
{
    return slot == 0 ? 0 : slot < 3 ? 1 : slot < 5 ? 3 : slot < 8 ? 5 : 8;
}

// ------------------------------------------------------------------ LDS image
// One workgroup = 11 waves (one per piece slot, so every wave runs a single piece type) x 16
// quads.  FULL = false: 16 boards per workgroup, quad = board, wave = slot of the side to move.
// FULL = true : 8 boards per workgroup, quads 0-7 = white piece, 8-15 = black piece of slot
// `wave` (both colours are needed by the planes).
// Pair layout (movegen only, hive_bb.hpp): 32 boards per workgroup, pair = board.
template <bool FULL, int GB = (FULL ? 8 : 16)>
struct alignas(16) Smem {
    static constexpr int G = GB;
    uint32_t state[G][16];       // HiveBoard records
    uint32_t occ[G][6];          // cells with at least one piece
    uint32_t topw[G][6];         // cells whose top piece is white
    uint8_t pinfo[G][24];        // per piece: stack height | stack index << 4 (0 = in hand)
    int32_t nlegal[G];           // legal-move count being accumulated
    alignas(16) uint32_t cellmask[G][kCells];   // bit q: piece q stands on this cell
    alignas(16) uint32_t adj[G][24];       // piece graph: bit j of row q = pieces q and j share a cell or touch
    uint32_t pinmask[G];         // bit q: lifting piece q would split the hive (or it is the only piece)
    alignas(8) uint32_t place[G][12];  // per board: [0..5] where the side to move may place a piece from its hand (before the
                                       // turn gate), [6..11] next_move_tiles
    int pin_done;                // pin waves that have published their pinmask bits
    int adj_done;                // pin waves that have written their share of the adjacency rows
    unsigned long long feat[FULL ? G : 1][FULL ? kCells : 1];   // 56 feature bits per cell
    int done;                    // waves that have delivered their destinations
};

__device__ __forceinline__ unsigned state_byte(const uint32_t *st, unsigned idx)
{
    return reinterpret_cast<const uint8_t *>(st)[idx];
}
// move_checker.py:106-137
__device__ __forceinline__ bool obeys_queen_by_4(unsigned turn, int nq, int first_color, bool mover_queen,
                                                 int mover_color)
{
    if (nq == 2) return true;
    if (nq == 0)
        return (turn == 7 && mover_queen && mover_color == 0) || (turn == 8 && mover_queen && mover_color == 1);
    return (first_color == 0 && turn == 7) || (first_color == 1 && turn == 7 && mover_queen) ||
           (first_color == 1 && turn == 8) || (first_color == 0 && turn == 8 && mover_queen);
}

template <class B>
struct PieceInfoT {
    B D;             // final destination set
    unsigned cell;   // kHand if in hand
    unsigned lvl, h;
    bool on_board, on_top, pinned;
};

// The per-(board, piece) scalars piece_dests works with, all derived from the staged record and the phase-0 piece info.
struct PieceScal {
    unsigned turn, c, h, lv;
    int stm, color, slot, nq, first_color;
    bool in_hand, on_board, on_top, stacked, stm_queen;
};
__device__ __forceinline__ PieceScal piece_scalars(const uint32_t *st, const uint8_t *pinfo, int q, bool valid)
{
    PieceScal s;
    s.turn = state_byte(st, 33);
    s.stm = (s.turn & 1u) ? 0 : 1;
    s.color = q >= 11 ? 1 : 0;
    s.slot = q - 11 * s.color;
    s.c = state_byte(st, (unsigned)q);
    s.in_hand = s.c >= (unsigned)kCells;
    // stack height of the mover's cell and the mover's index in it (env_hive.py:213), from phase 0
    const unsigned pi = pinfo[q];
    s.h = pi & 15u;
    s.lv = pi >> 4;
    s.on_board = valid && !s.in_hand;
    s.on_top = s.on_board && (s.lv + 1u == s.h);
    s.stacked = s.h > 1u;
    const bool wq = state_byte(st, 0u) < (unsigned)kCells;         // white queen placed
    const bool bq = state_byte(st, 11u) < (unsigned)kCells;        // black queen (piece 11) placed
    s.nq = (wq ? 1 : 0) + (bq ? 1 : 0);
    s.first_color = wq ? 0 : 1;
    s.stm_queen = s.stm == 0 ? wq : bq;
    return s;
}

template <class L>
__device__ __forceinline__ PieceInfoT<typename L::B> piece_finish(const PieceScal s, const uint32_t *st, const uint32_t *occ_p,
                                                                  const uint32_t *place_p, int q, int type, bool own, bool valid,
                                                                  bool pinned, typename L::B rule, typename L::B occ_in,
                                                                  typename L::B srcbit_in, typename L::B nsrc_in);

// Everything one (board, piece) quad computes; scalars are replicated over the quad's lanes.
// `type` and the loops' trip tests are wave-uniform.
// own == true : get_actions semantics (env_hive.py:207-285)
// own == false: mini_black_actions semantics for an enemy piece (env_hive.py:449-485)
// What piece_dests reads of its (board, piece): LDS addresses of the board's staged data, the piece index, validity.
struct PieceArgs {
    const uint32_t *st, *occ_p;
    const uint8_t *pinfo;
    const uint32_t *pinmask_p, *place_p;
    int q;
    bool valid;
};

// late(): the same PieceArgs built AGAIN from scratch (pair layout only; see piece_finish)
template <int ANT_STEPS, class L = QuadLay, class Late>
__device__ __forceinline__ PieceInfoT<typename L::B> piece_dests(const PieceArgs a, const int *pin_done_p, int type, bool own, Late late)
{
    using BB = typename L::B;
    const uint32_t *st = a.st, *occ_p = a.occ_p, *pinmask_p = a.pinmask_p, *place_p = a.place_p;
    const uint8_t *pinfo = a.pinfo;
    const int q = a.q;
    const bool valid = a.valid;
    PieceInfoT<BB> out;
    const PieceScal s0 = piece_scalars(st, pinfo, q, valid);
    const unsigned c = s0.c;
    const bool on_board = s0.on_board, on_top = s0.on_top, stacked = s0.stacked;

    const BB occ = L::load(occ_p);
    const BB srcbit = L::bit(on_board ? c : 255u);
    const BB occp = (on_board && !stacked) ? bb_xor(occ, srcbit) : occ;
    // the six neighbours of src.  Pair layout: three registers per board value and the same 80-register budget, so what
    // is only needed behind the loops (nsrc, occ) is fetched again there instead of being carried (or spilled) through them
    constexpr bool kLate = L::kLanes == 2;
    auto load_nsrc = [&]() { return on_board ? L::load(d_tables.nmask[c]) : L::zero(); };
    const BB nsrc = kLate ? L::zero() : load_nsrc();

    // The one-hive test (move_checker.py:58-83 / env_hive.py:509-530) is not made here: the pin waves
    // (pin_phase below) decide it for every piece of the workgroup on the 22-node piece graph while
    // this wave floods; `pinned` is picked up from LDS once the piece's own rule is done.  The piece
    // floods (Ant reach, Grasshopper line flood, Spider steps) run speculatively for every top piece.
    int dbg_it = 0; (void)dbg_it;

    // ---- piece rule (pieces.py) on the board with the mover lifted
    BB rule = L::zero();
    if (type == T_GRASS) {
        // pieces.py:128-158: flood from src through occupied cells on a "line" from src
        BB Ln = L::load(d_tables.line[on_board ? c : 0u]);
        BB Lo = bb_and(Ln, occ);
        BB V = srcbit;
        bool ga = on_top;
        while (__any(ga)) {
            HIVE_COUNT_ITER();
            if (ga) {
                BB n1 = bb_or(V, bb_and(bb_neighbours(V), Lo));
                BB nx = bb_or(n1, bb_and(bb_neighbours(n1), Lo));
                if (bb_eq(nx, n1)) ga = false;              // the last expansion added nothing: closed
                V = nx;
            }
        }
        if constexpr (kLate) rule = bb_andn(bb_and(bb_andn(bb_neighbours(V), L::load(occ_p)), L::load(d_tables.line[on_board ? c : 0u])), load_nsrc());
        else rule = bb_andn(bb_and(bb_andn(bb_neighbours(V), occ), Ln), nsrc);
    } else {
        BB S[6];
        occupancy_views(occp, S);
        SlideCtxT<BB> ctx = make_slide_ctx(occp, S);
        if (type == T_QUEEN) {
            rule = slide_step(ctx, srcbit);                       // pieces.py:35-44
        } else if (type == T_ANT) {
            BB R = srcbit;                                        // pieces.py:59-63, move_checker.py:217-246
            bool aa = on_top;
            while (__any(aa)) {
                HIVE_COUNT_ITER();
                if (aa) {
                    BB nx = R, before_last = R;
                    HIVE_UNROLL for (int u = 0; u < ANT_STEPS; ++u) {
                        before_last = nx;
                        nx = bb_or(nx, slide_step(ctx, nx));
                    }
                    if (bb_eq(nx, before_last)) aa = false;     // the last expansion added nothing: closed
                    R = nx;
                }
            }
            rule = bb_andn(R, srcbit);
        } else {
            // both flanks occupied (k == 2) per direction
            BB b0 = bb_and(S[5], S[1]), b1 = bb_and(S[0], S[2]), b2 = bb_and(S[1], S[3]);
            BB b3 = bb_and(S[2], S[4]), b4 = bb_and(S[3], S[5]), b5 = bb_and(S[4], S[0]);
            if (type == T_SPIDER) {
                // pieces.py:78-85: simple path of exactly three k==1 steps, then the direct-hop veto.
                // For each first step a: {c} = slide(slide(a) \ src) \ {src, a}; c != b holds by itself.
                BB A = slide_step(ctx, srcbit);
                BB acc = L::zero();
                bool sa = on_top && bb_any(A);
                while (__any(sa)) {
                    HIVE_COUNT_ITER();
                    if (sa) {
                        BB a = bb_lowest(A);
                        A = bb_andn(A, a);
                        BB Bs = bb_andn(slide_step(ctx, a), srcbit);
                        BB Cs = bb_andn(bb_andn(slide_step(ctx, Bs), srcbit), a);
                        acc = bb_or(acc, Cs);
                        sa = bb_any(A);
                    }
                }
                BB veto = shift_dirs(bb_and(srcbit, b0), bb_and(srcbit, b1), bb_and(srcbit, b2),
                                     bb_and(srcbit, b3), bb_and(srcbit, b4), bb_and(srcbit, b5));
                rule = bb_andn(acc, veto);
            } else {
                // Beetle, pieces.py:100-113 + move_checker.py:201-209
                BB Q1 = slide_raw(ctx, srcbit);
                BB Q0 = shift_dirs(bb_andn(bb_andn(srcbit, ctx.cs[0]), b0), bb_andn(bb_andn(srcbit, ctx.cs[1]), b1),
                                   bb_andn(bb_andn(srcbit, ctx.cs[2]), b2), bb_andn(bb_andn(srcbit, ctx.cs[3]), b3),
                                   bb_andn(bb_andn(srcbit, ctx.cs[4]), b4), bb_andn(bb_andn(srcbit, ctx.cs[5]), b5));
                const BB nb = kLate ? load_nsrc() : nsrc;
                rule = bb_or3(bb_and(nb, occ), Q1, bb_and(Q0, ctx.nocc));
                if (stacked) rule = bb_or(rule, nb);
            }
        }
    }
#if defined(HIVE_DBG_ITERS) && !defined(HIVE_DBG_STAMPS_ONLY)
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&d_iters[type], (unsigned long long)dbg_it);
        atomicAdd(&d_iters[8 + type], 1ull);
        atomicMax(&d_iters[16 + type], (unsigned long long)dbg_it);
        if (dbg_it < 32) atomicAdd(&d_hist[type][dbg_it], 1ull);
    }
#endif
    // the verdict of the pin waves (they run concurrently with the loops above and are much shorter)
    while (__hip_atomic_load(pin_done_p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < kPinWaves)
        __builtin_amdgcn_s_sleep(1);
    if constexpr (kLate) {
        const PieceArgs b = late();
        const bool pinned = ((*b.pinmask_p >> b.q) & 1u) != 0u;
        return piece_finish<L>(piece_scalars(b.st, b.pinfo, b.q, b.valid), b.st, b.occ_p, b.place_p, b.q, type, own, b.valid, pinned,
                               rule, L::zero(), L::zero(), L::zero());
    } else {
        const bool pinned = ((*pinmask_p >> q) & 1u) != 0u;
        return piece_finish<L>(s0, st, occ_p, place_p, q, type, own, valid, pinned, rule, occ, srcbit, nsrc);
    }
}

// The tail of piece_dests: the piece's rule result -> its destination board (turn gates, placement, domains).  Pair layout:
// called with the scalars derived AGAIN behind the pin wait (an acquire: the LDS loads are made again), so that none of them
// is carried -- in this layout: spilled -- through the flood loops; occ, srcbit and nsrc are rebuilt likewise.
template <class L>
__device__ __forceinline__ PieceInfoT<typename L::B> piece_finish(const PieceScal s, const uint32_t *st, const uint32_t *occ_p,
                                                                  const uint32_t *place_p, int q, int type, bool own, bool valid,
                                                                  bool pinned, typename L::B rule, typename L::B occ_in,
                                                                  typename L::B srcbit_in, typename L::B nsrc_in)
{
    using BB = typename L::B;
    constexpr bool kLate = L::kLanes == 2;
    PieceInfoT<BB> out;
    const unsigned turn = s.turn, c = s.c, h = s.h, lv = s.lv;
    const int color = s.color, slot = s.slot, nq = s.nq, first_color = s.first_color;
    const bool in_hand = s.in_hand, on_board = s.on_board, on_top = s.on_top, stm_queen = s.stm_queen;
    const bool movable = on_top && !pinned;
    auto load_nsrc = [&]() { return on_board ? L::load(d_tables.nmask[c]) : L::zero(); };
    const BB srcbit = kLate ? L::bit(on_board ? c : 255u) : srcbit_in;
    const BB nsrc = nsrc_in, occ = occ_in;

    // ---- turn gating (move_checker.py:38-55)
    bool gate = true;
    if (turn >= 3u && turn <= 6u) gate = in_hand || stm_queen;                      // queen_is_on_board
    else if (turn == 7u || turn == 8u) gate = obeys_queen_by_4(turn, nq, first_color, type == T_QUEEN, color);

    BB D = L::zero();
    if (valid && in_hand) {
        if (own) {
            // only the first in-hand piece of each type emits placements (env_hive.py:218-219)
            const int g0 = slot_group_start(slot);
            bool first = true;      // every earlier slot of the same type is already on the board
            if (slot > g0) first = state_byte(st, (unsigned)(color * 11 + g0)) < (unsigned)kCells;
            if (slot > g0 + 1) first = first && state_byte(st, (unsigned)(color * 11 + g0 + 1)) < (unsigned)kCells;
            // the destinations are the same for every piece in the mover's hand: placement_board() built them once
            if (first && (turn <= 2u || gate)) D = L::load(place_p);
        }
    } else if (movable) {
        const bool adjacent_domain = (type == T_QUEEN || type == T_BEETLE);
        BB domain;
        if (!own) domain = L::full();
        else if (adjacent_domain) domain = kLate ? load_nsrc() : nsrc;
        else domain = L::load(place_p + 6);       // next_move_tiles (env_hive.py:66-69,150-161), shared per board
        if (turn <= 2u) {
            const BB occ2 = kLate ? L::load(occ_p) : occ;
            BB base = bb_andn(type == T_BEETLE ? L::full() : bb_andn(L::full(), occ2), srcbit);
            D = bb_and(domain, base);
            D = bb_and(D, turn == 1u ? L::bit((unsigned)kStartCell) : bb_neighbours(occ2));
        } else if (gate) {
            D = bb_and(rule, domain);
        }
    }
    out.D = D;
    out.cell = on_board ? c : kHand;
    out.lvl = lv;
    out.h = h;
    out.on_board = on_board;
    out.on_top = on_top;
    out.pinned = pinned;
    return out;
}

// Where the side to move may put a piece from its hand (env_hive.py:218-232 with move_checker.py:9-55,168-179), one
// quad per board: turn 1 the start tile, turn 2 any empty tile of next_move_tiles next to the hive (colour rule waived),
// later empty hive-adjacent tiles none of whose neighbours is topped by the other colour.  The turn gate (queen rules)
// is per piece type and applied by the reader.
template <class L = QuadLay>
__device__ __forceinline__ typename L::B placement_board(const uint32_t *st, const uint32_t *occ_p, const uint32_t *topw_p,
                                                         typename L::B &nmt)
{
    using BB = typename L::B;
    const unsigned turn = state_byte(st, 33);
    const unsigned mode = state_byte(st, 34) & 3u;
    const int stm = (turn & 1u) ? 0 : 1;
    const BB occ = L::load(occ_p);
    const BB nocc = bb_neighbours(occ);
    const BB empty_adj = bb_andn(nocc, occ);
    nmt = empty_adj;                            // next_move_tiles (env_hive.py:66-69,150-161)
    if (mode == 1u) nmt = L::bit((unsigned)kStartCell);
    else if (mode == 2u) nmt = bb_and(empty_adj, L::bit((unsigned)kTurn2Cell));
    const BB base = bb_andn(nmt, occ);
    if (turn == 1u) return bb_and(base, L::bit((unsigned)kStartCell));
    if (turn == 2u) return bb_and(base, nocc);
    const BB topw = L::load(topw_p);
    const BB top_enemy = stm == 0 ? bb_andn(occ, topw) : topw;
    return bb_andn(bb_and(base, nocc), bb_neighbours(top_enemy));
}

// ------------------------------------------------------------------ one-hive test on the piece graph
// move_does_not_break_hive (move_checker.py:58-83, env_hive.py:509-530): lift the top piece of src,
// BFS over the occupied cells, connected?  Only an unstacked top piece can change the occupancy, and
// cells are connected exactly when the pieces standing on them are (pieces sharing a cell or on
// touching cells are neighbours), so the test runs on the <= 22-node piece graph instead of 144-cell
// boards: ONE LANE per (board, piece) -- 64 items per wave instead of 16 quads -- holds the 22
// adjacency rows of its board in registers and floods a 22-bit reach mask with alternating
// ascending / descending Gauss-Seidel sweeps (two instructions per row) until every neighbour of
// the lifted piece is reached (free) or nothing grows (pinned).
// 32 boards per workgroup (the pair layout) are two passes of the three pin waves over the 352 items.
template <bool FULL, class SM>
__device__ __forceinline__ void pin_phase(SM &sm, int wave, int lane)
{
    constexpr int G = SM::G;
    // piece graph first: row q = the pieces on q's cell and on its six neighbour cells (G x 22 rows, two per lane);
    // the other eight waves are already at their piece work and never wait for this
    for (int r = wave * 64 + lane; r < G * 22; r += kPinWaves * 64) {
        const int rb = r / 22, rq = r - rb * 22;
        const unsigned c = state_byte(sm.state[rb], (unsigned)rq);
        uint32_t row = 0u;
        if (c < (unsigned)kCells) {
            const uint2 nb = *reinterpret_cast<const uint2 *>(d_tables.nbr[c]);
            row = sm.cellmask[rb][c];
            HIVE_UNROLL for (int k = 0; k < 6; ++k)
                row |= sm.cellmask[rb][((k < 4 ? nb.x : nb.y) >> ((k & 3) * 8)) & 0xFFu];
            row &= ~(1u << rq);
        }
        sm.adj[rb][rq] = row;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(&sm.adj_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(&sm.adj_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < kPinWaves)
        __builtin_amdgcn_s_sleep(1);

    constexpr int per = FULL ? 22 : 11;
    constexpr int kPasses = (G * per + kPinWaves * 64 - 1) / (kPinWaves * 64);
#pragma unroll 1
    for (int pass = 0; pass < kPasses; ++pass) {
    const int item = pass * (kPinWaves * 64) + wave * 64 + lane;      // 176 items: G boards x (FULL ? 22 : 11) pieces
    int b = item / per;
    const bool ok = b < G;
    if (!ok) b = G - 1;
    int q = item - (item / per) * per;
    if (!FULL) q += (state_byte(sm.state[b], 33) & 1u) ? 0 : 11;      // pieces of the side to move
    const unsigned pi = sm.pinfo[b][q];
    const unsigned h = pi & 15u, lv = pi >> 4;
    const bool lifts = ok && h == 1u && lv == 0u;      // on the board, alone on its cell
    const uint32_t notq = ~(1u << q);
    uint32_t a[24];
    HIVE_UNROLL for (int i = 0; i < 6; ++i) {
        const uint4 r = reinterpret_cast<const uint4 *>(sm.adj[b])[i];
        a[4 * i] = r.x & notq; a[4 * i + 1] = r.y & notq; a[4 * i + 2] = r.z & notq; a[4 * i + 3] = r.w & notq;
    }                                                  // nothing floods through the lifted piece
    const uint32_t target = sm.adj[b][q];              // its neighbours
    bool pinned = lifts && target == 0u;               // the only piece on the board: "empty board => False"
    bool act = lifts && target != 0u;
    uint32_t reach = target & (0u - target);
#ifdef HIVE_ABL_NOPINLOOP
    act = false;
#endif
    int pin_trips = 0; (void)pin_trips;
    while (__any(act)) {
#ifdef HIVE_DBG_ITERS
        ++pin_trips;
#endif
        if (act) {
            uint32_t up = reach;
            HIVE_UNROLL for (int i = 0; i < 22; ++i) up |= a[i] & (uint32_t)(((int)(up << (31 - i))) >> 31);
            uint32_t nx = up;
            HIVE_UNROLL for (int i = 21; i >= 0; --i) nx |= a[i] & (uint32_t)(((int)(nx << (31 - i))) >> 31);
            const bool covered = (target & ~nx) == 0u;
            // a descending sweep that adds nothing has looked at every reached piece and found all its neighbours
            // reached: the set is closed (no need for another round trip to see that nothing grows)
            const bool fixed = nx == up;
            reach = nx;
            if (covered) act = false;
            else if (fixed) { act = false; pinned = true; }
        }
    }
#if defined(HIVE_DBG_ITERS) && !defined(HIVE_DBG_STAMPS_ONLY)
    if (lane == 0) {      // pin waves are booked as "type 5" of the trip statistics
        atomicAdd(&d_iters[5], (unsigned long long)pin_trips);
        atomicAdd(&d_iters[8 + 5], 1ull);
        atomicMax(&d_iters[16 + 5], (unsigned long long)pin_trips);
        if (pin_trips < 32) atomicAdd(&d_hist[5][pin_trips], 1ull);
    }
#endif
    if (pinned) atomicOr(&sm.pinmask[b], 1u << q);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(&sm.pin_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// cell test on a lane-distributed board (result replicated over the quad)
__device__ __forceinline__ bool bb_test(BB x, unsigned cell) { return bb_any(bb_and(x, bb_bit(cell))); }

// GamePlay.encode_action (env_hive.py:287-304): {slot: destination board} -> ascending action ids.  Action id =
// cell * 11 + slot: the list is the 11 x 144 slot-major bit matrix read CELL-major.  A wave builds one board's list from
// the board's destination image in LDS ([slot][6 words], the layout of the legal mask) and the OR of its eleven slot boards:
//   1. the cells that are anybody's destination (~35 of 144) are compacted, in ascending order, into an LDS byte list
//      (three passes of 64 cells: one bit test, one ballot, one v_mbcnt each);
//   2. lane k takes the k-th such cell and gathers its bit from the eleven slot boards (eleven independent ds_reads at
//      immediate offsets -- no per-bit atomics, no dependent LDS round trips); four ballots + v_mbcnt over the cells'
//      popcounts give every cell its place in the list, its <= 11 ids are laid down in an LDS row;
//   3. the row leaves as one coalesced 8-byte store per lane (-1 padded).
// Cells ascend with the lanes, slots ascend inside a cell: the list is sorted.
// inclusive prefix sum over the 64 lanes of a wave: Hillis-Steele inside the rows of 16 (row_shr 1, 2, 4, 8, zero fill),
// then lane 15 of row 0 / 2 into row 1 / 3 (row_bcast:15) and lane 31 into rows 2 and 3 (row_bcast:31)
__device__ __forceinline__ int wave_inclusive_sum(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
    return x;
}

__device__ __forceinline__ void emit_id_list(const uint32_t *img, const uint32_t *any, uint8_t *cells, int16_t *rowbuf,
                                             int16_t *out, int lane)
{
    reinterpret_cast<unsigned long long *>(rowbuf)[lane] = 0xFFFFFFFFFFFFFFFFull;      // four -1 entries per lane
    int ncell = 0;
    HIVE_UNROLL for (int pass = 0; pass < 3; ++pass) {
        const unsigned c = (unsigned)(pass * 64 + lane);
        const unsigned cc = c < (unsigned)kCells ? c : 0u;
        const unsigned row = cc / 12u, col = cc - row * 12u;
        const bool has = c < (unsigned)kCells && ((any[row >> 1] >> (((row & 1u) << 4) | col)) & 1u);
        const unsigned long long bal = __ballot(has);
        if (has)
            cells[ncell + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = (uint8_t)c;
        ncell += (int)__popcll(bal);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // (same-wave LDS traffic only: program order suffices)
    __builtin_amdgcn_wave_barrier();
    int base = 0;
    for (int k0 = 0; k0 < ncell; k0 += 64) {                    // one trip unless more than 64 cells are destinations
        const bool act = k0 + lane < ncell;
        const unsigned c = act ? (unsigned)cells[k0 + lane] : 0u;
        const unsigned row = c / 12u, col = c - row * 12u;
        const uint32_t *w = img + (row >> 1);
        const unsigned sh = ((row & 1u) << 4) | col;
        unsigned v = 0u;
        HIVE_UNROLL for (int s = 10; s >= 0; --s) v = (v << 1) | __builtin_amdgcn_ubfe(w[s * 6], sh, 1u);      // v_bfe_u32 + v_lshl_or_b32 per slot
        if (!act) v = 0u;
        const int cnt = __popc(v);                              // <= 11 ids per cell
        const int upto = wave_inclusive_sum(cnt);               // six DPP adds (four ballots + v_mbcnt took twenty instructions)
        int pos = base + upto - cnt;
        base += __builtin_amdgcn_readlane(upto, 63);
        while (v) {                                             // this cell's ids in ascending slot order
            const unsigned sl = (unsigned)__builtin_ctz(v);
            v &= v - 1u;
            if (pos < HIVE_LIST_CAP) rowbuf[pos] = (int16_t)(c * 11u + sl);
            ++pos;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    reinterpret_cast<unsigned long long *>(out)[lane] = reinterpret_cast<const unsigned long long *>(rowbuf)[lane];
    __builtin_amdgcn_wave_barrier();                            // (the row and cell buffers may be reused by this wave)
}

// LDS of the fused list variant: the destination images of the workgroup's 16 boards and one list row per wave
template <bool LIST, int G> struct ListMem { uint32_t pad; };
template <int G> struct alignas(16) ListMem<true, G> {
    uint32_t dest[G][11][6];
    uint32_t any[G][6];                    // OR of a board's eleven destination boards
    uint8_t cells[NW][kCells];             // per list-building wave: the cells that are somebody's destination
    alignas(8) int16_t rowbuf[NW][HIVE_LIST_CAP];
};

// The piece kernel.  FULL = false: legal mask/count of the side to move (GamePlay.actions).
// FULL = true: both colours; additionally the 56 feature bits per cell (planes 0-35, 44-55 of
// GamePlay.make_state_value; history planes 36-43 and plane 31 are added by hive_expand_kernel).
// There is no barrier after the piece work: each wave ORs its results into LDS and leaves; the
// last wave to finish writes the workgroup's boards out.
// LIST = true (movegen only): the sorted id list (GamePlay.actions() itself) leaves the same launch: the destination
// boards are also kept in LDS, the waves meet at one barrier and build the lists of the workgroup's 16 boards between
// them -- no second launch, no re-read of the mask.
// L = PairLay (movegen without the list only): one board = two lanes, 32 boards per workgroup -- a quarter fewer VALU
// instructions per board, for launches that fill the card in either layout (hive_movegen_launch picks it from
// kPairFromBoards on; a 4096-board launch is one quad workgroup per CU and would leave half the CUs idle in pairs).
template <bool FULL, bool LIST = false, class L = QuadLay>
__global__ void __launch_bounds__(NW * 64, FULL ? 5 : (L::kLanes == 2 ? HIVE_PAIR_WPE : HIVE_PIECE_WPE))
hive_piece_kernel(const HiveBoard *__restrict__ boards, int n, uint32_t *__restrict__ mask,
                  int32_t *__restrict__ count, unsigned long long *__restrict__ feat, int16_t *__restrict__ list)
{
    static_assert(!(FULL && LIST), "the fused list belongs to the movegen variant");
    static_assert(L::kLanes == 4 || !FULL, "the pair layout is built for the movegen variant");
    using BB = typename L::B;
    constexpr int G = FULL ? 8 : 64 / L::kLanes;
    __shared__ Smem<FULL, G> sm;
    __shared__ ListMem<LIST, G> lm;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int item = lane >> L::kShift;              // quad (pair) index inside the wave
    const int bl = FULL ? (item & 7) : item;         // board of this quad inside the workgroup
    // wave -> piece slot.  Waves land on SIMD (wave id mod 4) and start in id order: waves 0-2 are the Queen and the
    // Beetles (they run the pin phase first), then the three Ants start early on three different SIMDs, the Spiders
    // follow, the Grasshoppers fill up: SIMD 0 = Q A0 G2, 1 = B0 A1 G0, 2 = B1 S1 G1, 3 = A2 S0.  Measured against the
    // identity map: -7 % at 1 M boards, and as fast at 4096 boards as s_setprio-raised Ant/Spider waves were.
#ifndef HIVE_WAVE_SLOTS
#define HIVE_WAVE_SLOTS 0x6573498A210ull      /* hex digit w (lowest = wave 0) = piece slot of wave w */
#endif
#ifndef HIVE_PIN_IDS
#define HIVE_PIN_IDS 0xFFFFFFFF210ull         /* hex digit w = share (0..2) of the one-hive test wave w takes first, F = none */
#endif
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifndef HIVE_PAIR_WAVE_SLOTS
#define HIVE_PAIR_WAVE_SLOTS HIVE_WAVE_SLOTS
#endif
    const int wv = (int)(((L::kLanes == 2 ? HIVE_PAIR_WAVE_SLOTS : HIVE_WAVE_SLOTS) >> (4 * wave_id)) & 15ull);      // the piece slot this wave works on
    const int pin_id = (int)((HIVE_PIN_IDS >> (4 * wave_id)) & 15ull);
    const int nthreads = NW * 64;
    const long long gbase = (long long)blockIdx.x * G;
#ifdef HIVE_DBG_ITERS
    const long long t_start = clock64();
#endif

    // ---------------- phase 0: stage records, clear accumulators
    // the record load is issued first and lands while the accumulators are being cleared (G * 4 <= 64: one trip)
    uint4 rec_part = make_uint4(0u, 0u, 0u, 0u);
    if (tid < G * 4) {
        const int b = tid >> 2, part = tid & 3;
        if (gbase + b < n) rec_part = reinterpret_cast<const uint4 *>(boards)[(gbase + b) * 4 + part];
        else if (part == 0) rec_part = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        else if (part == 1) rec_part = make_uint4(0xFFFFFFFFu, 0x0000FFFFu, 0u, 0u);
        else if (part == 2) rec_part = make_uint4(0x00010100u, 0u, 0u, 0u);
    }
    for (int i = tid; i < G * 6; i += nthreads) {
        (&sm.occ[0][0])[i] = 0u;
        (&sm.topw[0][0])[i] = 0u;
    }
    for (int i = tid; i < G * 6; i += nthreads) reinterpret_cast<uint32_t *>(&sm.pinfo[0][0])[i] = 0u;
    if (FULL)
        for (int i = tid; i < G * kCells; i += nthreads) (&sm.feat[0][0])[i] = 0ull;
    for (int i = tid; i < G * kCells / 4; i += nthreads)
        reinterpret_cast<uint4 *>(&sm.cellmask[0][0])[i] = make_uint4(0u, 0u, 0u, 0u);
    if constexpr (LIST)
        if (tid < G * 6) (&lm.any[0][0])[tid] = 0u;
    if (tid == 0) { sm.done = 0; sm.pin_done = 0; sm.adj_done = 0; }
    if (tid < G) { sm.nlegal[tid] = 0; sm.pinmask[tid] = 0u; }
    if (tid < G * 4) reinterpret_cast<uint4 *>(sm.state[tid >> 2])[tid & 3] = rec_part;
    __syncthreads();

    // occupancy and top-colour boards: every (board, piece) pair ORs its bit in
    for (int pi = tid; pi < G * 22; pi += nthreads) {
        int b = pi % G, q = pi / G;
        const uint32_t *st = sm.state[b];
        unsigned c = state_byte(st, (unsigned)q);
        if (c < (unsigned)kCells) {
            unsigned lb = state_byte(st, 22u + ((unsigned)q >> 1));
            unsigned lv = (q & 1) ? (lb >> 4) : (lb & 15u);
            // stack height = pieces standing on cell c: four position bytes per step (exact zero-byte test of w ^ cccc)
            unsigned h = 0;
            const uint32_t cccc = c * 0x01010101u;
            HIVE_UNROLL for (int wq = 0; wq < 6; ++wq) {
                uint32_t x = st[wq] ^ cccc;
                if (wq == 5) x |= 0xFFFF0000u;                       // bytes 22, 23 are stack indices, not positions
                const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);      // 0x80 where the byte was 0
                h += (unsigned)__popc(z);
            }
            unsigned wi, bit;
            cell_word_bit(c, wi, bit);
            atomicOr(&sm.occ[b][wi], 1u << bit);
            if (lv + 1u == h && q < 11) atomicOr(&sm.topw[b][wi], 1u << bit);
            atomicOr(&sm.cellmask[b][c], 1u << q);
            sm.pinfo[b][q] = (uint8_t)(h | (lv << 4));
        }
    }
    __syncthreads();

    HIVE_STAMP(0);
    // ---------------- phase 1: one quad per (board, piece), one piece slot per wave
    const bool valid = gbase + bl < n;
    const uint32_t *st = sm.state[bl];
    const unsigned turn = state_byte(st, 33);
    const int stm = (turn & 1u) ? 0 : 1;
    const int type = slot_type(wv);
    if (pin_id == kPinWaves - 1) {       // published together with this wave's pin results (pin_done)
        BB nmt;
        const BB pl = placement_board<L>(sm.state[bl], sm.occ[bl], sm.topw[bl], nmt);
        L::store(sm.place[bl], pl);
        L::store(sm.place[bl] + 6, nmt);
    }
    if (pin_id < kPinWaves) {
        // the three lightest waves (Queen, Beetles) settle the one-hive question for the whole workgroup first
        pin_phase<FULL>(sm, pin_id, lane);
        HIVE_STAMP(4);
    }
    int q;
    bool own;
    if (!FULL) { q = stm * 11 + wv; own = true; }
    else { int col = item >> 3; q = col * 11 + wv; own = (col == stm); }
    const PieceArgs pa{st, sm.occ[bl], sm.pinfo[bl], &sm.pinmask[bl], sm.place[bl], q, valid};
    // (pair layout: the tail of the piece work builds these again from the thread id instead of carrying -- spilling -- them
    // through the flood loops; the empty asm keeps the compiler from recognising the value it already has)
    auto late_args = [&]() -> PieceArgs {
        int b2 = (int)(threadIdx.x & 63u) >> L::kShift;
        asm volatile("" : "+v"(b2));
        const uint32_t *st2 = sm.state[b2];
        const int stm2 = (state_byte(st2, 33) & 1u) ? 0 : 1;
        return PieceArgs{st2, sm.occ[b2], sm.pinfo[b2], &sm.pinmask[b2], sm.place[b2], stm2 * 11 + wv, gbase + b2 < n};
    };
    PieceInfoT<BB> pc = piece_dests<(L::kLanes == 2 ? HIVE_PAIR_ANT_STEPS : HIVE_ANT_STEPS), L>(pa, &sm.pin_done, type, own, late_args);
    HIVE_STAMP(1);
    if (own && (mask != nullptr || count != nullptr)) {
        // the legal set leaves as it is: slot wv's destination board, two words per lane, straight to HBM
        if constexpr (L::kLanes == 4) {
        if (valid && mask != nullptr && (lane & 3) < 3)
            *reinterpret_cast<uint2 *>(mask + (gbase + bl) * HIVE_MASK_WORDS + wv * 6 + 2 * (lane & 3)) =
                make_uint2(pc.D.lo, pc.D.hi);
        } else {
            int b3 = (int)(threadIdx.x & 63u) >> L::kShift;      // (built again, like late_args above)
            asm volatile("" : "+v"(b3));
            if (gbase + b3 < n && mask != nullptr) {      // three words per lane (12-byte pieces, 4-byte aligned)
                uint32_t *mp = mask + (gbase + b3) * HIVE_MASK_WORDS + wv * 6 + 3 * (int)(threadIdx.x & 1u);
                mp[0] = pc.D.w0; mp[1] = pc.D.w1; mp[2] = pc.D.w2;
            }
        }
        if constexpr (LIST) {     // ... and stays in LDS for the list builders (every (board, slot) quad writes: no clearing)
            if constexpr (L::kLanes == 4) {
            if ((lane & 3) < 3) {
                *reinterpret_cast<uint2 *>(&lm.dest[bl][wv][2 * (lane & 3)]) = make_uint2(pc.D.lo, pc.D.hi);
                if (pc.D.lo) atomicOr(&lm.any[bl][2 * (lane & 3)], pc.D.lo);
                if (pc.D.hi) atomicOr(&lm.any[bl][2 * (lane & 3) + 1], pc.D.hi);
            }
            } else {
                int b5 = (int)(threadIdx.x & 63u) >> L::kShift;
                asm volatile("" : "+v"(b5));
                uint32_t *dp = &lm.dest[b5][wv][3 * (int)(threadIdx.x & 1u)], *ap = &lm.any[b5][3 * (int)(threadIdx.x & 1u)];
                dp[0] = pc.D.w0; dp[1] = pc.D.w1; dp[2] = pc.D.w2;
                if (pc.D.w0) atomicOr(ap, pc.D.w0);
                if (pc.D.w1) atomicOr(ap + 1, pc.D.w1);
                if (pc.D.w2) atomicOr(ap + 2, pc.D.w2);
            }
        }
        const int nd = bb_popc_lane(pc.D);                      // destination sets of different pieces are disjoint
        if constexpr (L::kLanes == 4) {
            if (nd) atomicAdd(&sm.nlegal[bl], nd);
        } else {
            int b4 = (int)(threadIdx.x & 63u) >> L::kShift;
            asm volatile("" : "+v"(b4));
            if (nd) atomicAdd(&sm.nlegal[b4], nd);
        }
    }
    HIVE_STAMP(2);

    if constexpr (FULL) {
        // ---------------- per-piece feature bits (env_hive.py:352-429)
        const BB occ = bb_load(sm.occ[bl]);
        const bool d_any = bb_any(pc.D);
        unsigned long long bits = 0ull;
        if (own) {
            if (!pc.on_top || !d_any) bits |= 1ull << 34;
            unsigned eq = state_byte(st, (unsigned)((1 - stm) * 11));      // enemy queen
            if (pc.on_top && eq < (unsigned)kCells)
                for (int i = 0; i < 6; ++i) {
                    unsigned adj = d_tables.nbr[eq][i];
                    if (!bb_test(occ, adj) && bb_test(pc.D, adj)) bits |= 1ull << (50 + i);
                }
        } else {
            if (!pc.on_top || pc.pinned) bits |= 1ull << 35;
            unsigned oq = state_byte(st, (unsigned)(stm * 11));            // own queen
            if (pc.on_top && !pc.pinned && oq < (unsigned)kCells)
                for (int i = 0; i < 6; ++i) {
                    unsigned adj = d_tables.nbr[oq][i];
                    if (!bb_test(occ, adj) && bb_test(pc.D, adj)) bits |= 1ull << (44 + i);
                }
        }
        unsigned qnb = 0u;     // occupied neighbours of a queen, by adjacent_tiles index
        if (type == T_QUEEN && pc.on_board)
            for (int i = 0; i < 6; ++i) qnb |= bb_test(occ, d_tables.nbr[pc.cell][i]) ? (1u << i) : 0u;
        if (pc.on_board && (lane & 3) == 0) {
            const int slot = wv;
            bits |= 1ull << ((own ? 0u : 12u) + (unsigned)slot);
            bits |= 1ull << (own ? 11u : 23u);
            bits |= 1ull << 30;
            if (type == T_BEETLE) {
                const unsigned bb = own ? 24u : 27u;
                if (pc.lvl == 2u) bits |= 1ull << bb;
                if (pc.lvl == 3u) bits |= 1ull << (bb + 1u);
                if (pc.lvl == 4u && pc.h == 5u) bits |= 1ull << (bb + 2u);
            }
            atomicOr(&sm.feat[bl][pc.cell], bits);
            if (type == T_QUEEN) {
                const unsigned long long qb = 1ull << (own ? 32 : 33);
                for (int i = 0; i < 6; ++i)
                    if ((qnb >> i) & 1u) atomicOr(&sm.feat[bl][d_tables.nbr[pc.cell][i]], qb);
            }
        }
    }

    if constexpr (LIST) {
        // ---------------- fused encode_action: every wave stays, the 16 boards' lists are shared out between the 11 waves
        // (a raw barrier behind an LDS-only wait: __syncthreads() would also wait for the mask stores above to be
        // acknowledged by memory -- a microsecond or two nobody needs; only the LDS image must be complete)
        HIVE_STAMP(5);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        HIVE_STAMP(6);
        const int nbl = (int)((n - gbase) < G ? (n - gbase) : G);
        if (wave_id == 0 && count != nullptr && lane < nbl) count[gbase + lane] = sm.nlegal[lane];
        // 16 boards on 11 waves, four per SIMD (waves land on SIMD wave_id mod 4: SIMD 3 holds two waves, the others three):
        // every wave builds board wave_id, waves 3 and 7 (SIMD 3) and one wave of each other SIMD (4, 5, 6) a second one
        static_assert(G == 16 || G == 32 || !LIST, "the list builders are shared out for 16 or 32 boards on 11 waves");
        // (two boards in lock step inside one wave, and 16 boards as 16 equal shares, were built and measured: 9.24 us
        // against 9.27 -- the eleven waves together are short of VALU issue slots here, not of latency hiding)
        if constexpr (G == 32) {
            // 32 boards on 11 waves: three rounds (wave w builds boards w, w + 11, w + 22; 33 turns for 32 boards)
#pragma unroll 1
            for (int b = wave_id; b < nbl; b += NW)
                emit_id_list(&lm.dest[b][0][0], lm.any[b], lm.cells[wave_id], lm.rowbuf[wave_id],
                             list + (gbase + b) * HIVE_LIST_CAP, lane);
            HIVE_STAMP(7);
            return;
        }
        if (wave_id < nbl)
            emit_id_list(&lm.dest[wave_id][0][0], lm.any[wave_id], lm.cells[wave_id], lm.rowbuf[wave_id],
                         list + (gbase + wave_id) * HIVE_LIST_CAP, lane);
        const int second = wave_id == 3 ? 11 : wave_id == 7 ? 12 : (wave_id >= 4 && wave_id <= 6) ? 9 + wave_id : -1;
        if (second >= 0 && second < nbl)
            emit_id_list(&lm.dest[second][0][0], lm.any[second], lm.cells[wave_id], lm.rowbuf[wave_id],
                         list + (gbase + second) * HIVE_LIST_CAP, lane);
        HIVE_STAMP(7);
        return;
    }
    // ---------------- tail: the last wave to arrive writes the workgroup's results
    int prior = 0;
    if (lane == 0) prior = __hip_atomic_fetch_add(&sm.done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    prior = __builtin_amdgcn_readfirstlane(prior);
    if (prior != NW - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    const int nb = (int)((n - gbase) < G ? (n - gbase) : G);
    if (count != nullptr && lane < nb) count[gbase + lane] = sm.nlegal[lane];
    HIVE_STAMP(3);
    if (FULL && feat != nullptr) {
        const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(&sm.feat[0][0]);
        ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(feat + gbase * kCells);
        for (int i = lane; i < nb * (kCells / 2); i += 64) dst[i] = src[i];
    }
}

// GamePlay.encode_action over an existing mask (one wave per board; emit_id_list above).
__global__ void __launch_bounds__(256)
hive_list_kernel(const uint32_t *__restrict__ mask, int n, int16_t *__restrict__ list)
{
    __shared__ uint32_t dest[4][HIVE_MASK_WORDS + 2];
    __shared__ uint32_t any[4][6];
    __shared__ uint8_t cells[4][kCells];
    __shared__ __attribute__((aligned(8))) int16_t rowbuf[4][HIVE_LIST_CAP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long b = (long long)blockIdx.x * 4 + wv;
    if (b >= n) return;
    const uint32_t *m = mask + b * HIVE_MASK_WORDS;
    dest[wv][lane] = m[lane];
    if (lane < HIVE_MASK_WORDS - 64) dest[wv][64 + lane] = m[64 + lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // same-wave LDS traffic only: program order suffices
    __builtin_amdgcn_wave_barrier();
    if (lane < 6) {
        uint32_t o = 0u;
        HIVE_UNROLL for (int sl = 0; sl < 11; ++sl) o |= dest[wv][sl * 6 + lane];
        any[wv][lane] = o;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    emit_id_list(dest[wv], any[wv], cells[wv], rowbuf[wv], list + b * HIVE_LIST_CAP, lane);
}

// value encoders for the plane writer
template <typename T> struct PlaneVal;
template <> struct PlaneVal<float> {
    static __device__ __forceinline__ uint32_t one() { return 0x3F800000u; }
    static __device__ __forceinline__ uint32_t of(unsigned v) { return __float_as_uint((float)v); }
};
struct half_tag {};
struct bf16_tag {};
template <> struct PlaneVal<half_tag> {
    static __device__ __forceinline__ uint32_t one() { return 0x3C00u; }
    static __device__ __forceinline__ uint32_t of(unsigned v)
    {
        _Float16 hval = (_Float16)(float)v;
        return (uint32_t) __builtin_bit_cast(unsigned short, hval);
    }
};
template <> struct PlaneVal<bf16_tag> {
    static __device__ __forceinline__ uint32_t one() { return 0x3F80u; }
    static __device__ __forceinline__ uint32_t of(unsigned v) { return __float_as_uint((float)v) >> 16; }   // v <= 255: exact
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// NT: streaming (nontemporal) stores for launches whose output is far larger than the caches -- 5.3 instead of 4.2 TB/s at
// 65,536 boards (1 GB of planes); leaf batches of a search (<= 4096 boards, read back at once by the network) keep plain stores
template <bool NT>
__device__ __forceinline__ void plane_store(u32x4 val, u32x4 *ptr)
{
    if (NT) __builtin_nontemporal_store(val, ptr);
    else *ptr = val;
}
constexpr int kExpandStreamBoards = 8192;              // from this many boards per launch (127 MB of bf16 planes) on: NT stores
constexpr int kExpandAhead = 4;                        // boards whose inputs are in flight per workgroup
constexpr long long kExpandMaxBlocks = 256 * 8;       // all resident at once: 8 workgroups of 4 waves per CU

// Packed features (56 bits per cell) + history boards + turn -> the plane tensor (GamePlay.encode_board's output,
// env_hive.py:320-447); pure streaming-store kernel.  A workgroup takes whole boards (grid-stride).  Per board: 144
// threads complete the cells' 56-bit words in LDS (packed features | the eight history bits of planes 36..43, read from
// the history bitboards: env_hive.py:431-434) -- one barrier, double-buffered --, then thread t writes the 16-byte (f16 / bf16; 32-byte f32)
// items t, t + 256, t + 512, t + 768 (< 1008) of the board: consecutive lanes write consecutive, line-aligned pieces, nothing is divided by a runtime value, and no global load sits between a thread and its stores.
// (One workgroup per 256 items with per-item history loads ran at 2.0 TB/s of 6.9 the card stores: profiles/r03_encode.md.)
template <int DT, int LAYOUT, bool NT>
__global__ void __launch_bounds__(256)
hive_expand_kernel(const HiveBoard *__restrict__ boards, const HiveHistory *__restrict__ hist,
                   const unsigned long long *__restrict__ feat, int n, void *__restrict__ planes)
{
    using V = typename std::conditional<DT == 0, float, typename std::conditional<DT == 1, half_tag, bf16_tag>::type>::type;
    constexpr int kItems = kCells * HIVE_PLANES / 8;      // 1008 items of 8 elements per board = 4 x 252
    __shared__ unsigned long long full[2][kCells];
    const uint32_t one = PlaneVal<V>::one();
    const int tid = threadIdx.x;
    int buf = 0;
    // the record bytes and packed words of the next kExpandAhead boards are in flight: loads and stores retire through ONE
    // in-order counter, so a load requested d boards ahead only waits for stores that are d boards old
    uint32_t meta_q[kExpandAhead];
    unsigned long long word_q[kExpandAhead];
    HIVE_UNROLL for (int d = 0; d < kExpandAhead; ++d) {
        const long long bq = (long long)blockIdx.x + (long long)d * gridDim.x;
        meta_q[d] = 0u;
        word_q[d] = 0ull;
        if (bq < n) {
            meta_q[d] = reinterpret_cast<const uint32_t *>(&boards[bq])[8];    // bytes 32..35
            if (tid < kCells) word_q[d] = feat[bq * kCells + tid];
        }
    }
    for (long long b = blockIdx.x; b < n; b += gridDim.x, buf ^= 1) {
        const uint32_t meta = meta_q[0];
        unsigned long long w = word_q[0];
        HIVE_UNROLL for (int d = 0; d + 1 < kExpandAhead; ++d) { meta_q[d] = meta_q[d + 1]; word_q[d] = word_q[d + 1]; }
        const long long bn = b + (long long)kExpandAhead * gridDim.x;
        if (bn < n) {
            meta_q[kExpandAhead - 1] = reinterpret_cast<const uint32_t *>(&boards[bn])[8];
            if (tid < kCells) word_q[kExpandAhead - 1] = feat[bn * kCells + tid];
        }
        const unsigned turn = (meta >> 8) & 0xFFu, hl = meta >> 24;
        const int persp = (turn & 1u) ? 0 : 1;
        const unsigned hlen = persp == 0 ? (hl & 15u) : (hl >> 4);
        const uint32_t tv = PlaneVal<V>::of(turn);
        if (tid < kCells) {
            if (hist != nullptr) {
                unsigned wi, bit;
                cell_word_bit((unsigned)tid, wi, bit);
                // history planes 36..43 = (own, enemy) occupancy 1..4 encodes ago
                for (unsigned age = 0; age < hlen && age < 4u; ++age) {
                    w |= (unsigned long long)((hist[b].m[persp][age][0][wi] >> bit) & 1u) << (36u + 2u * age);
                    w |= (unsigned long long)((hist[b].m[persp][age][1][wi] >> bit) & 1u) << (37u + 2u * age);
                }
            }
            full[buf][tid] = w;
        }
        // (the other buffer is rewritten only after the NEXT barrier: no second one needed; a raw barrier behind an
        // LDS-only wait, because __syncthreads() would also wait for the previous board's stores to be acknowledged)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
            // items tid, tid + 256, tid + 512, tid + 768 (< 1008): every wave's store is one 128-byte-aligned KiB
            HIVE_UNROLL for (int j = 0; j < 4; ++j) {
                if (j == 3 && tid >= kItems - 768) break;
                const int e0 = (tid + j * 256) * 8;
                uint32_t v[8];
                if (LAYOUT == HIVE_HWC) {
                    const int cell = e0 / HIVE_PLANES, p0 = e0 - cell * HIVE_PLANES;
                    const unsigned bits8 = (unsigned)(full[buf][cell] >> p0) & 0xFFu;
                    HIVE_UNROLL for (int k = 0; k < 8; ++k) v[k] = ((bits8 >> k) & 1u) ? one : 0u;
                    if (p0 == 24) v[7] = tv;      // plane 31 = raw turn number (env_hive.py:331)
                } else {
                    const int pl = e0 / kCells, c0 = e0 - pl * kCells;
                    HIVE_UNROLL for (int k = 0; k < 8; ++k) v[k] = ((full[buf][c0 + k] >> pl) & 1ull) ? one : 0u;
                    if (pl == 31) { HIVE_UNROLL for (int k = 0; k < 8; ++k) v[k] = tv; }
                }
                const long long eoff = b * (long long)(kCells * HIVE_PLANES) + e0;
                if (DT == 0) {
                    u32x4 *dst = reinterpret_cast<u32x4 *>(reinterpret_cast<float *>(planes) + eoff);
                    plane_store<NT>(u32x4{v[0], v[1], v[2], v[3]}, dst);
                    plane_store<NT>(u32x4{v[4], v[5], v[6], v[7]}, dst + 1);
                } else {
                    u32x4 *dst = reinterpret_cast<u32x4 *>(reinterpret_cast<uint16_t *>(planes) + eoff);
                    plane_store<NT>(u32x4{v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)}, dst);
                }
            }
        }
    }
}

// ------------------------------------------------------------------ step / reset / terminal
// GamePlay.move (env_hive.py:99-171); one lane per board.
__global__ void hive_step_kernel(HiveBoard *__restrict__ boards, HiveHistory *__restrict__ hist, int n,
                                 const int32_t *__restrict__ actions, const uint32_t *__restrict__ legal_mask,
                                 unsigned long long *__restrict__ illegal_count)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    int a = actions[b];
    if (a == -2) return;
    if (a < -2 || a >= HIVE_ACTIONS) { if (illegal_count) atomicAdd(illegal_count, 1ull); return; }
    if (a >= 0 && legal_mask != nullptr) {
        if (!HIVE_MASK_TEST(legal_mask + (long long)b * HIVE_MASK_WORDS, a)) {
            if (illegal_count) atomicAdd(illegal_count, 1ull);
            return;
        }
    }
    apply_action(&boards[b], hist ? &hist[b] : nullptr, a);
}

__global__ void hive_reset_kernel(HiveBoard *__restrict__ boards, HiveHistory *__restrict__ hist, int n,
                                  const int32_t *__restrict__ idx, int k)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    int b = idx ? idx[i] : i;
    if (b < 0 || b >= n) return;
    uint4 *p = reinterpret_cast<uint4 *>(&boards[b]);
    p[0] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    p[1] = make_uint4(0xFFFFFFFFu, 0x0000FFFFu, 0u, 0u);
    p[2] = make_uint4(0x00050100u, 0u, 0u, 0u);     // lvl[10]=0, turn=1, flags = nmt_mode 1 | history bit, hist_len=0
    p[3] = make_uint4(0u, 0u, 0u, 0u);
    if (hist) {
        uint4 *hp = reinterpret_cast<uint4 *>(&hist[b]);
        for (int j = 0; j < (int)(sizeof(HiveHistory) / 16); ++j) hp[j] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// move_checker.py:140-165
__global__ void hive_terminal_kernel(const HiveBoard *__restrict__ boards, int n, int8_t *__restrict__ over,
                                     int8_t *__restrict__ winner)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    const uint4 *rec = reinterpret_cast<const uint4 *>(&boards[b]);
    uint4 r0 = rec[0], r1 = rec[1];
    const uint32_t pw[6] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y};      // pos[0..23] (22 used)
    bool surrounded[2] = {false, false};
    HIVE_UNROLL for (int col = 0; col < 2; ++col) {
        unsigned c = col == 0 ? (pw[0] & 0xFFu) : (pw[2] >> 24);
        if (c < (unsigned)kCells) {
            const uint2 nb = *reinterpret_cast<const uint2 *>(d_tables.nbr[c]);
            int cnt = 0;
            HIVE_UNROLL for (int i = 0; i < 6; ++i) {
                unsigned adj = ((i < 4 ? nb.x : nb.y) >> ((i & 3) * 8)) & 0xFFu;
                bool occd = false;
                HIVE_UNROLL for (int r = 0; r < 22; ++r) occd = occd || (((pw[r >> 2] >> ((r & 3) * 8)) & 0xFFu) == adj);
                cnt += occd ? 1 : 0;
            }
            surrounded[col] = cnt == 6;
        }
    }
    bool ov = surrounded[0] || surrounded[1];
    int w = 0;
    if (surrounded[0] && !surrounded[1]) w = 2;       // white queen surrounded -> black wins
    else if (surrounded[1] && !surrounded[0]) w = 1;
    if (over) over[b] = ov ? 1 : 0;
    if (winner) winner[b] = (int8_t)w;
}

// debug/test hook: copy the compile-time tables out (tests/test_tables.py)
__global__ void hive_tables_kernel(uint32_t *line, uint8_t *nbr)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < kCells * 6) line[i] = (&d_tables.line[0][0])[i];
    if (i < kCells * 8) nbr[i] = (&d_tables.nbr[0][0])[i];
}


// ---------------------------------------------------------------------------------------------
// hive_leaf_dedup_launch: equal leaves of one batch.  The planes are a function of (HiveBoard, HiveHistory) alone, so two
// rows with the same 448 bytes get the same prediction: lock-step games from the opening position ask the network about the
// same few positions a thousand times (plies 0-2: 97 / 90 / 48 % of the rows, tools/leaf_dups.py).
//   leaf_key_kernel    one wave per row: a 64-bit mix of the row's 56 eight-byte words
//   leaf_dedup_kernel  ONE workgroup: an open-addressing table of the keys in LDS; every needed row finds its key's slot and
//                      takes the minimum row index stored there as its representative (atomicMin: the outcome does not
//                      depend on the order the threads arrive in), then compares its 448 bytes with the representative's --
//                      a hash collision keeps the row as its own representative.
constexpr int kDedupMaxRows = 4096, kDedupSlots = 8192;

__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

__global__ void __launch_bounds__(256)
leaf_key_kernel(const HiveBoard *__restrict__ boards, const HiveHistory *__restrict__ hist, int n,
                unsigned long long *__restrict__ keys)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    unsigned long long v = 0ull;
    if (lane < 8) v = reinterpret_cast<const unsigned long long *>(boards + row)[lane];
    else if (lane < 56) v = reinterpret_cast<const unsigned long long *>(hist + row)[lane - 8];
    unsigned long long h = mix64(v + 0x9e3779b97f4a7c15ull * (unsigned long long)(lane + 1));
    HIVE_UNROLL for (int d = 32; d >= 1; d >>= 1) h += __shfl_xor(h, d, 64);       // lanes >= 56 add a constant
    h = mix64(h);
    if (lane == 0) keys[row] = h ? h : 1ull;                                       // 0 marks an empty table slot
}

__global__ void __launch_bounds__(1024)
leaf_dedup_kernel(const HiveBoard *__restrict__ boards, const HiveHistory *__restrict__ hist, int n,
                  const unsigned long long *__restrict__ keys, int8_t *__restrict__ need, int32_t *__restrict__ rep,
                  unsigned long long *__restrict__ total)
{
    // 96 KB of LDS: this library is built for gfx950 only (160 KB per CU; a 64 KB-LDS target could not launch this kernel)
    static_assert(kDedupSlots * (sizeof(unsigned long long) + sizeof(int)) + 64 <= 160 * 1024, "leaf_dedup_kernel: table exceeds gfx950's LDS");
    __shared__ unsigned long long tkey[kDedupSlots];
    __shared__ int tmin[kDedupSlots];
    __shared__ int dropped;
    const int tid = threadIdx.x;
    for (int i = tid; i < kDedupSlots; i += 1024) { tkey[i] = 0ull; tmin[i] = 0x7fffffff; }
    if (tid == 0) dropped = 0;
    __syncthreads();
    int slot_of[kDedupMaxRows / 1024];
    HIVE_UNROLL for (int j = 0; j < kDedupMaxRows / 1024; ++j) {
        const int i = tid + j * 1024;
        slot_of[j] = -1;
        if (i < n && need[i]) {
            const unsigned long long k = keys[i];
            unsigned slot = (unsigned)(k >> 20) & (kDedupSlots - 1);
            for (int probe = 0; probe < kDedupSlots; ++probe) {                    // at most 4096 keys in 8192 slots: ends
                const unsigned long long prev = atomicCAS(&tkey[slot], 0ull, k);
                if (prev == 0ull || prev == k) { atomicMin(&tmin[slot], i); slot_of[j] = (int)slot; break; }
                slot = (slot + 1u) & (kDedupSlots - 1);
            }
        }
    }
    __syncthreads();
    int mine = 0;
    HIVE_UNROLL for (int j = 0; j < kDedupMaxRows / 1024; ++j) {
        const int i = tid + j * 1024;
        if (i >= n) continue;
        int r = i;
        if (slot_of[j] >= 0) {
            const int first = tmin[slot_of[j]];
            if (first != i) {
                const unsigned long long *a = reinterpret_cast<const unsigned long long *>(boards + i);
                const unsigned long long *b = reinterpret_cast<const unsigned long long *>(boards + first);
                const unsigned long long *c = reinterpret_cast<const unsigned long long *>(hist + i);
                const unsigned long long *d = reinterpret_cast<const unsigned long long *>(hist + first);
                unsigned long long diff = 0ull;
                for (int w = 0; w < (int)(sizeof(HiveBoard) / 8); ++w) diff |= a[w] ^ b[w];
                for (int w = 0; w < (int)(sizeof(HiveHistory) / 8); ++w) diff |= c[w] ^ d[w];
                if (diff == 0ull) { r = first; need[i] = 0; ++mine; }
            }
        }
        rep[i] = r;
    }
    if (mine) atomicAdd(&dropped, mine);
    __syncthreads();
    if (tid == 0 && total && dropped) atomicAdd(total, 0ull - (unsigned long long)dropped);
}

}  // namespace hive

// ====================================================================== host side / C ABI
using namespace hive;

static thread_local std::string g_err;
static int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
namespace hive {
int set_error(int code, const std::string &msg) { return fail(code, msg); }
}
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(HIVE_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

struct HiveBatch {
    int n = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    HiveBoard *boards = nullptr;
    HiveHistory *hist = nullptr;
    uint32_t *legal = nullptr;              // cached legal mask of the current positions
    int32_t *legal_count = nullptr;
    int16_t *legal_list = nullptr;
    bool legal_valid = false;               // legal / legal_count describe the current positions
    bool list_valid = false;                // ... and so does legal_list (built only when somebody asks for it)
    unsigned long long *feat = nullptr;     // packed-feature workspace of the encoder
    unsigned long long *illegal = nullptr;  // device counter
};

extern "C" {

const char *hive_last_error(void) { return g_err.c_str(); }
const char *hive_version(void) { return "hive-hip 0.1 (gfx950)"; }

int hive_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return c;
}

// Launches of at least this many boards (mask / count, no id list) run in the pair layout.  16,384 boards = 512 pair
// workgroups = two per CU, the residency the kernel is built for; below that the quad layout's 16-board workgroups
// spread the same boards over twice as many CUs.
constexpr int kPairFromBoards = 16384;
static int g_pair_from = kPairFromBoards;

static int launch_pieces(const HiveBoard *boards, int n, uint32_t *mask, int32_t *count, int16_t *list,
                         hipStream_t stream)
{
    if (n <= 0 || boards == nullptr) return fail(HIVE_E_ARG, "movegen: n <= 0 or boards == NULL");
    if (list != nullptr && mask == nullptr) return fail(HIVE_E_ARG, "movegen: the id list needs the mask buffer too");
    if (list != nullptr && ((uintptr_t)list & 7u) != 0)
        return fail(HIVE_E_ARG, "movegen: the id list must be 8-byte aligned (rows leave as 8-byte stores)");
    if (list != nullptr && n >= g_pair_from)
        hipLaunchKernelGGL((hive_piece_kernel<false, true, PairLay>), dim3((unsigned)((n + 31) / 32)), dim3(NW * 64), 0, stream,
                           boards, n, mask, count, (unsigned long long *)nullptr, list);
    else if (list != nullptr) // mask, count and the sorted id lists in ONE launch (hive_piece_kernel<false, true>)
        hipLaunchKernelGGL((hive_piece_kernel<false, true>), dim3((unsigned)((n + 15) / 16)), dim3(NW * 64), 0, stream, boards, n,
                           mask, count, (unsigned long long *)nullptr, list);
    else if (n >= g_pair_from)      // enough boards to fill the card with 32-board workgroups: the pair layout
        hipLaunchKernelGGL((hive_piece_kernel<false, false, PairLay>), dim3((unsigned)((n + 31) / 32)), dim3(NW * 64), 0, stream,
                           boards, n, mask, count, (unsigned long long *)nullptr, (int16_t *)nullptr);
    else
        hipLaunchKernelGGL((hive_piece_kernel<false, false>), dim3((unsigned)((n + 15) / 16)), dim3(NW * 64), 0, stream, boards, n,
                           mask, count, (unsigned long long *)nullptr, (int16_t *)nullptr);
    HIP_TRY(hipGetLastError());
    return HIVE_OK;
}

static int launch_encode(const HiveBoard *boards, const HiveHistory *hist, int n, void *planes, int dtype, int layout,
                         unsigned long long *feat, hipStream_t stream, uint32_t *mask = nullptr, int32_t *count = nullptr)
{
    if (n <= 0 || boards == nullptr || planes == nullptr || feat == nullptr)
        return fail(HIVE_E_ARG, "encode: n <= 0 or a NULL buffer");
    if (dtype < 0 || dtype > 2 || layout < 0 || layout > 1) return fail(HIVE_E_ARG, "encode: unknown dtype/layout");
    hipLaunchKernelGGL((hive_piece_kernel<true, false>), dim3((unsigned)((n + 7) / 8)), dim3(NW * 64), 0, stream, boards, n, mask,
                       count, feat, (int16_t *)nullptr);
    HIP_TRY(hipGetLastError());
    dim3 grid((unsigned)std::min<long long>(n, kExpandMaxBlocks));
#define HIVE_EXP_CASE(DT, LY)                                                                                  \
    if (dtype == DT && layout == LY) {                                                                         \
        if (n >= kExpandStreamBoards)                                                                          \
            hipLaunchKernelGGL((hive_expand_kernel<DT, LY, true>), grid, dim3(256), 0, stream, boards, hist, feat, n, planes); \
        else                                                                                                   \
            hipLaunchKernelGGL((hive_expand_kernel<DT, LY, false>), grid, dim3(256), 0, stream, boards, hist, feat, n, planes); \
    }
    HIVE_EXP_CASE(0, 0) HIVE_EXP_CASE(0, 1) HIVE_EXP_CASE(1, 0) HIVE_EXP_CASE(1, 1) HIVE_EXP_CASE(2, 0) HIVE_EXP_CASE(2, 1)
#undef HIVE_EXP_CASE
    HIP_TRY(hipGetLastError());
    return HIVE_OK;
}

int hive_movegen_launch(const HiveBoard *boards, int n, uint32_t *mask, int32_t *count, int16_t *list, void *stream)
{
    return launch_pieces(boards, n, mask, count, list, (hipStream_t)stream);
}

int hive_movegen_pair_threshold(int boards)
{
    const int prev = g_pair_from;
    if (boards > 0) g_pair_from = boards;
    else if (boards == 0) g_pair_from = kPairFromBoards;
    return prev;
}

int hive_encode_launch(const HiveBoard *boards, const HiveHistory *hist, int n, void *planes, HiveDType dtype,
                       HiveLayout layout, void *workspace, void *stream)
{
    return launch_encode(boards, hist, n, planes, (int)dtype, (int)layout, (unsigned long long *)workspace,
                         (hipStream_t)stream);
}

int hive_expand_launch(const HiveBoard *boards, const HiveHistory *hist, const void *features, int n, void *planes,
                       HiveDType dtype, HiveLayout layout, void *stream)
{
    if (n <= 0 || boards == nullptr || features == nullptr || planes == nullptr)
        return fail(HIVE_E_ARG, "expand: n <= 0 or a NULL buffer");
    const int dt = (int)dtype, ly = (int)layout;
    if (dt < 0 || dt > 2 || ly < 0 || ly > 1) return fail(HIVE_E_ARG, "expand: unknown dtype/layout");
    dim3 grid((unsigned)std::min<long long>(n, kExpandMaxBlocks));
    const unsigned long long *feat = (const unsigned long long *)features;
#define HIVE_EXP_CASE(DT, LY)                                                                                  \
    if (dt == DT && ly == LY) {                                                                                \
        if (n >= kExpandStreamBoards)                                                                          \
            hipLaunchKernelGGL((hive_expand_kernel<DT, LY, true>), grid, dim3(256), 0, (hipStream_t)stream, boards, hist, feat, n, planes); \
        else                                                                                                   \
            hipLaunchKernelGGL((hive_expand_kernel<DT, LY, false>), grid, dim3(256), 0, (hipStream_t)stream, boards, hist, feat, n, planes); \
    }
    HIVE_EXP_CASE(0, 0) HIVE_EXP_CASE(0, 1) HIVE_EXP_CASE(1, 0) HIVE_EXP_CASE(1, 1) HIVE_EXP_CASE(2, 0) HIVE_EXP_CASE(2, 1)
#undef HIVE_EXP_CASE
    HIP_TRY(hipGetLastError());
    return HIVE_OK;
}

int hive_terminal_launch(const HiveBoard *boards, int n, int8_t *over, int8_t *winner, void *stream)
{
    if (n <= 0 || boards == nullptr) return fail(HIVE_E_ARG, "terminal: n <= 0 or boards == NULL");
    hipLaunchKernelGGL(hive_terminal_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boards, n,
                       over, winner);
    HIP_TRY(hipGetLastError());
    return HIVE_OK;
}

int hive_step_launch(HiveBoard *boards, HiveHistory *hist, int n, const int32_t *actions, const uint32_t *legal_mask,
                     void *stream)
{
    if (n <= 0 || boards == nullptr || actions == nullptr) return fail(HIVE_E_ARG, "step: bad argument");
    hipLaunchKernelGGL(hive_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boards, hist,
                       n, actions, legal_mask, (unsigned long long *)nullptr);
    HIP_TRY(hipGetLastError());
    return HIVE_OK;
}

int hive_step_launch_counted(HiveBoard *boards, HiveHistory *hist, int n, const int32_t *actions, const uint32_t *legal_mask,
                             unsigned long long *illegal_count, void *stream)
{
    if (n <= 0 || boards == nullptr || actions == nullptr) return fail(HIVE_E_ARG, "step: bad argument");
    if (legal_mask == nullptr && illegal_count != nullptr)
        return fail(HIVE_E_ARG, "step: counting refused actions needs the legal mask they are tested against");
    hipLaunchKernelGGL(hive_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boards, hist,
                       n, actions, legal_mask, illegal_count);
    HIP_TRY(hipGetLastError());
    return HIVE_OK;
}

int hive_leaf_launch(const HiveBoard *boards, const HiveHistory *hist, int n, void *planes, HiveDType dtype,
                     HiveLayout layout, void *workspace, uint32_t *mask, int32_t *count, int8_t *over, int8_t *winner,
                     void *stream)
{
    int rc = launch_encode(boards, hist, n, planes, (int)dtype, (int)layout, (unsigned long long *)workspace,
                           (hipStream_t)stream, mask, count);
    if (rc != HIVE_OK) return rc;
    if (over != nullptr || winner != nullptr) return hive_terminal_launch(boards, n, over, winner, stream);
    return HIVE_OK;
}

int hive_leaf_dedup_launch(const HiveBoard *boards, const HiveHistory *hist, int n, int8_t *need, int32_t *rep,
                           uint64_t *keys, uint64_t *total, void *stream)
{
    static_assert(sizeof(HiveBoard) == 64 && sizeof(HiveHistory) == 384, "leaf_key_kernel reads 8 + 48 eight-byte words");
    if (n <= 0 || n > kDedupMaxRows || boards == nullptr || hist == nullptr || need == nullptr || rep == nullptr || keys == nullptr)
        return fail(HIVE_E_ARG, "hive_leaf_dedup_launch: bad argument (1 <= n <= 4096)");
    hipLaunchKernelGGL(leaf_key_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, boards, hist, n,
                       reinterpret_cast<unsigned long long *>(keys));
    hipLaunchKernelGGL(leaf_dedup_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, boards, hist, n,
                       reinterpret_cast<const unsigned long long *>(keys), need, rep, reinterpret_cast<unsigned long long *>(total));
    HIP_TRY(hipGetLastError());
    return HIVE_OK;
}

static int batch_alloc(HiveBatch *h)
{
    const size_t n = (size_t)h->n;
    HIP_TRY(hipMalloc(&h->boards, sizeof(HiveBoard) * n));
    HIP_TRY(hipMalloc(&h->hist, sizeof(HiveHistory) * n));
    HIP_TRY(hipMalloc(&h->legal, sizeof(uint32_t) * HIVE_MASK_WORDS * n));
    HIP_TRY(hipMalloc(&h->legal_count, sizeof(int32_t) * n));
    HIP_TRY(hipMalloc(&h->legal_list, sizeof(int16_t) * HIVE_LIST_CAP * n));
    HIP_TRY(hipMalloc(&h->feat, sizeof(unsigned long long) * kCells * n));
    HIP_TRY(hipMalloc(&h->illegal, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(h->illegal, 0, sizeof(unsigned long long)));
    return hive_batch_reset(h, nullptr, h->n);
}

int hive_batch_create(int n, int device, HiveBatch **out)
{
    if (n <= 0 || out == nullptr) return fail(HIVE_E_ARG, "hive_batch_create: n <= 0 or out == NULL");
    *out = nullptr;
    int cnt = hive_device_count();
    if (cnt <= 0) return fail(HIVE_E_DEVICE, "hive_batch_create: no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= cnt) return fail(HIVE_E_ARG, "hive_batch_create: bad device ordinal");
    HIP_TRY(hipSetDevice(device));
    HiveBatch *h = new (std::nothrow) HiveBatch();
    if (h == nullptr) return fail(HIVE_E_DEVICE, "hive_batch_create: out of host memory");
    h->n = n;
    h->device = device;
    int rc = batch_alloc(h);
    if (rc != HIVE_OK) {            // a partly built handle never leaves the library
        hive_batch_destroy(h);
        return rc;
    }
    *out = h;
    return HIVE_OK;
}

int hive_batch_destroy(HiveBatch *h)
{
    if (!h) return HIVE_OK;
    (void)hipSetDevice(h->device);
    (void)hipFree(h->boards);
    (void)hipFree(h->hist);
    (void)hipFree(h->legal);
    (void)hipFree(h->legal_count);
    (void)hipFree(h->legal_list);
    (void)hipFree(h->feat);
    (void)hipFree(h->illegal);
    delete h;
    return HIVE_OK;
}

int hive_batch_size(const HiveBatch *h) { return h ? h->n : 0; }

int hive_batch_set_stream(HiveBatch *h, void *stream)
{
    if (!h) return fail(HIVE_E_ARG, "null handle");
    h->stream = (hipStream_t)stream;
    return HIVE_OK;
}

int hive_batch_reset(HiveBatch *h, const int32_t *idx, int k)
{
    if (!h) return fail(HIVE_E_ARG, "null handle");
    if (idx == nullptr) k = h->n;
    if (k <= 0) return HIVE_OK;
    HIP_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(hive_reset_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, h->stream, h->boards,
                       h->hist, h->n, idx, k);
    HIP_TRY(hipGetLastError());
    h->legal_valid = false;
    return HIVE_OK;
}

static int ensure_legal(HiveBatch *h, bool want_list)
{
    if (!h->legal_valid) {
        int rc = launch_pieces(h->boards, h->n, h->legal, h->legal_count, want_list ? h->legal_list : nullptr, h->stream);
        if (rc != HIVE_OK) return rc;
        h->legal_valid = true;
        h->list_valid = want_list;
    }
    if (want_list && !h->list_valid) {
        hipLaunchKernelGGL(hive_list_kernel, dim3((unsigned)((h->n + 3) / 4)), dim3(256), 0, h->stream, h->legal, h->n,
                           h->legal_list);
        HIP_TRY(hipGetLastError());
        h->list_valid = true;
    }
    return HIVE_OK;
}

int hive_batch_legal(HiveBatch *h, uint32_t *mask, int32_t *count, int16_t *list)
{
    if (!h) return fail(HIVE_E_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    int rc = ensure_legal(h, list != nullptr);
    if (rc != HIVE_OK) return rc;
    size_t n = (size_t)h->n;
    if (mask) HIP_TRY(hipMemcpyAsync(mask, h->legal, sizeof(uint32_t) * HIVE_MASK_WORDS * n, hipMemcpyDeviceToDevice, h->stream));
    if (count) HIP_TRY(hipMemcpyAsync(count, h->legal_count, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, h->stream));
    if (list) HIP_TRY(hipMemcpyAsync(list, h->legal_list, sizeof(int16_t) * HIVE_LIST_CAP * n, hipMemcpyDeviceToDevice, h->stream));
    return HIVE_OK;
}

int hive_batch_illegal_count(HiveBatch *h, int64_t *count)
{
    if (!h || !count) return fail(HIVE_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    unsigned long long v = 0;
    HIP_TRY(hipMemcpyAsync(&v, h->illegal, sizeof v, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *count = (int64_t)v;
    return HIVE_OK;
}

int hive_batch_step(HiveBatch *h, const int32_t *actions, int sync)
{
    if (!h || !actions) return fail(HIVE_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    int rc = ensure_legal(h, false);       // the step only checks the action against the destination boards
    if (rc != HIVE_OK) return rc;
    int64_t before = 0;
    if (sync) { rc = hive_batch_illegal_count(h, &before); if (rc != HIVE_OK) return rc; }
    hipLaunchKernelGGL(hive_step_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, h->stream, h->boards,
                       h->hist, h->n, actions, h->legal, h->illegal);
    HIP_TRY(hipGetLastError());
    h->legal_valid = false;
    if (sync) {
        int64_t after = 0;
        rc = hive_batch_illegal_count(h, &after);
        if (rc != HIVE_OK) return rc;
        if (after != before) {
            char buf[96];
            snprintf(buf, sizeof buf, "hive_batch_step: %lld board(s) refused an illegal action", (long long)(after - before));
            return fail(HIVE_E_ILLEGAL, buf);
        }
    }
    return HIVE_OK;
}

int hive_batch_encode(HiveBatch *h, void *planes, HiveDType dtype, HiveLayout layout)
{
    if (!h || !planes) return fail(HIVE_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    return launch_encode(h->boards, h->hist, h->n, planes, (int)dtype, (int)layout, h->feat, h->stream);
}

int hive_batch_terminal(HiveBatch *h, int8_t *over, int8_t *winner)
{
    if (!h) return fail(HIVE_E_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(hive_terminal_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, h->stream, h->boards,
                       h->n, over, winner);
    HIP_TRY(hipGetLastError());
    return HIVE_OK;
}

int hive_batch_export(HiveBatch *h, HiveBoard *boards, HiveHistory *hist)
{
    if (!h) return fail(HIVE_E_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    if (boards) HIP_TRY(hipMemcpyAsync(boards, h->boards, sizeof(HiveBoard) * (size_t)h->n, hipMemcpyDeviceToDevice, h->stream));
    if (hist) HIP_TRY(hipMemcpyAsync(hist, h->hist, sizeof(HiveHistory) * (size_t)h->n, hipMemcpyDeviceToDevice, h->stream));
    return HIVE_OK;
}

int hive_batch_import(HiveBatch *h, const HiveBoard *boards, const HiveHistory *hist)
{
    if (!h || !boards) return fail(HIVE_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(h->boards, boards, sizeof(HiveBoard) * (size_t)h->n, hipMemcpyDeviceToDevice, h->stream));
    if (hist) HIP_TRY(hipMemcpyAsync(h->hist, hist, sizeof(HiveHistory) * (size_t)h->n, hipMemcpyDeviceToDevice, h->stream));
    else HIP_TRY(hipMemsetAsync(h->hist, 0, sizeof(HiveHistory) * (size_t)h->n, h->stream));
    h->legal_valid = false;
    return HIVE_OK;
}

// ------------------------------------------------------------------ single-position calls with HOST buffers
// GamePlay's one-game surface (env_hive.py:99-171,196-304,306-485) without a round trip per question: one call copies
// the 448-byte position in, runs step -> legal set -> game over as one launch chain on the handle's own stream, copies
// the answers out and synchronises once.  Device block: [0,64) record, [64,448) history, [448,452) action,
// [512,776) legal set of the position BEFORE the move (to refuse an illegal action), [1024,1288) legal set after,
// [1288,1292) count, [1292] over, [1293] winner, [1296,1304) refused-action counter, [2048,3200) packed features,
// [4096,36352) planes.
struct HiveSingle {
    int device = 0;
    hipStream_t stream = nullptr;
    unsigned char *dev = nullptr;
    unsigned char *pin = nullptr;       // pinned host mirror of the device block
};
static constexpr size_t kSingleBytes = 36352;

int hive_single_destroy(HiveSingle *h);

int hive_single_create(int device, HiveSingle **out)
{
    if (out == nullptr) return fail(HIVE_E_ARG, "hive_single_create: out == NULL");
    *out = nullptr;
    int cnt = hive_device_count();
    if (cnt <= 0) return fail(HIVE_E_DEVICE, "hive_single_create: no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= cnt) return fail(HIVE_E_ARG, "hive_single_create: bad device ordinal");
    HIP_TRY(hipSetDevice(device));
    HiveSingle *h = new (std::nothrow) HiveSingle();
    if (h == nullptr) return fail(HIVE_E_DEVICE, "hive_single_create: out of host memory");
    h->device = device;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&h->dev), kSingleBytes);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&h->pin), kSingleBytes, hipHostMallocDefault);
    if (e == hipSuccess) memset(h->pin, 0, kSingleBytes);
    if (e == hipSuccess) e = hipMemsetAsync(h->dev, 0, kSingleBytes, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        hive_single_destroy(h);
        return fail(HIVE_E_DEVICE, std::string("hive_single_create: ") + hipGetErrorString(e));
    }
    *out = h;
    return HIVE_OK;
}

int hive_single_destroy(HiveSingle *h)
{
    if (!h) return HIVE_OK;
    (void)hipSetDevice(h->device);
    if (h->dev) (void)hipFree(h->dev);
    if (h->pin) (void)hipHostFree(h->pin);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return HIVE_OK;
}

int hive_single_advance(HiveSingle *h, const HiveBoard *rec, const HiveHistory *hist, int action, const uint32_t *legal_before,
                        HiveBoard *rec_out, HiveHistory *hist_out, uint32_t *legal_after, int32_t *count, int8_t *over,
                        int8_t *winner)
{
    if (!h || !rec_out || !hist_out || !legal_after) return fail(HIVE_E_ARG, "hive_single_advance: null argument");
    if (action != -3 && (!rec || !hist)) return fail(HIVE_E_ARG, "hive_single_advance: a position is needed unless action == -3");
    if (action < -3 || action >= HIVE_ACTIONS) return fail(HIVE_E_ARG, "hive_single_advance: action out of range");
    HIP_TRY(hipSetDevice(h->device));
    HiveBoard *dboard = reinterpret_cast<HiveBoard *>(h->dev);
    HiveHistory *dhist = reinterpret_cast<HiveHistory *>(h->dev + 64);
    int32_t *dact = reinterpret_cast<int32_t *>(h->dev + 448);
    uint32_t *dmask_in = reinterpret_cast<uint32_t *>(h->dev + 512), *dmask = reinterpret_cast<uint32_t *>(h->dev + 1024);
    unsigned long long *dill = reinterpret_cast<unsigned long long *>(h->dev + 1296);
    unsigned long long before = 0;
    memcpy(&before, h->pin + 1296, sizeof before);          // the counter as of the previous call's copy-out
    if (action == -3) {
        hipLaunchKernelGGL(hive_reset_kernel, dim3(1), dim3(64), 0, h->stream, dboard, dhist, 1, (const int32_t *)nullptr, 1);
    } else {
        memcpy(h->pin, rec, 64);
        memcpy(h->pin + 64, hist, 384);
        memcpy(h->pin + 448, &action, 4);
        size_t in_bytes = 452;
        if (action >= 0 && legal_before != nullptr) { memcpy(h->pin + 512, legal_before, 4 * HIVE_MASK_WORDS); in_bytes = 776; }
        HIP_TRY(hipMemcpyAsync(h->dev, h->pin, in_bytes, hipMemcpyHostToDevice, h->stream));
        if (action >= 0 && legal_before == nullptr) {       // nobody vouches for the action: derive the legal set first
            int rc = launch_pieces(dboard, 1, dmask_in, nullptr, nullptr, h->stream);
            if (rc != HIVE_OK) return rc;
        }
        if (action >= -1)
            hipLaunchKernelGGL(hive_step_kernel, dim3(1), dim3(64), 0, h->stream, dboard, dhist, 1, dact, dmask_in, dill);
    }
    HIP_TRY(hipGetLastError());
    int rc = launch_pieces(dboard, 1, dmask, reinterpret_cast<int32_t *>(h->dev + 1288), nullptr, h->stream);
    if (rc != HIVE_OK) return rc;
    hipLaunchKernelGGL(hive_terminal_kernel, dim3(1), dim3(64), 0, h->stream, dboard, 1, reinterpret_cast<int8_t *>(h->dev + 1292),
                       reinterpret_cast<int8_t *>(h->dev + 1293));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h->pin, h->dev, 1304, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    unsigned long long after = 0;
    memcpy(&after, h->pin + 1296, sizeof after);
    if (after != before) return fail(HIVE_E_ILLEGAL, "hive_single_advance: the action is not in the legal set; position unchanged");
    memcpy(rec_out, h->pin, 64);
    memcpy(hist_out, h->pin + 64, 384);
    memcpy(legal_after, h->pin + 1024, 4 * HIVE_MASK_WORDS);
    if (count) memcpy(count, h->pin + 1288, 4);
    if (over) *over = (int8_t)h->pin[1292];
    if (winner) *winner = (int8_t)h->pin[1293];
    return HIVE_OK;
}

int hive_single_encode(HiveSingle *h, const HiveBoard *rec, const HiveHistory *hist, float *planes)
{
    if (!h || !rec || !hist || !planes) return fail(HIVE_E_ARG, "hive_single_encode: null argument");
    HIP_TRY(hipSetDevice(h->device));
    memcpy(h->pin, rec, 64);
    memcpy(h->pin + 64, hist, 384);
    HIP_TRY(hipMemcpyAsync(h->dev, h->pin, 448, hipMemcpyHostToDevice, h->stream));
    int rc = launch_encode(reinterpret_cast<HiveBoard *>(h->dev), reinterpret_cast<HiveHistory *>(h->dev + 64), 1, h->dev + 4096,
                           HIVE_F32, HIVE_HWC, reinterpret_cast<unsigned long long *>(h->dev + 2048), h->stream);
    if (rc != HIVE_OK) return rc;
    HIP_TRY(hipMemcpyAsync(h->pin + 4096, h->dev + 4096, sizeof(float) * kCells * HIVE_PLANES, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    memcpy(planes, h->pin + 4096, sizeof(float) * kCells * HIVE_PLANES);
    return HIVE_OK;
}

#ifdef HIVE_DBG_ITERS
int hive_debug_iters(unsigned long long *host16)
{
    HIP_TRY(hipMemcpyFromSymbol(host16, HIP_SYMBOL(hive::d_iters), sizeof(unsigned long long) * 32));
    HIP_TRY(hipMemcpyFromSymbol(host16 + 32, HIP_SYMBOL(hive::d_hist), sizeof(unsigned long long) * 192));
    return HIVE_OK;
}
int hive_debug_stamps(unsigned long long *host88)
{
    HIP_TRY(hipMemcpyFromSymbol(host88, HIP_SYMBOL(hive::d_stamps), sizeof(unsigned long long) * 88));
    return HIVE_OK;
}
#endif

// test hook (not in hive_abi.h's product surface): copy the device tables to device buffers
int hive_debug_tables(uint32_t *line /* [144*6] */, uint8_t *nbr /* [144*8] */, void *stream)
{
    hipLaunchKernelGGL(hive_tables_kernel, dim3(5), dim3(256), 0, (hipStream_t)stream, line, nbr);
    HIP_TRY(hipGetLastError());
    return HIVE_OK;
}

}  // extern "C"
