// hive_search.hip -- GPU-resident PUCT tree search over thousands of concurrent Hive games.
//
// Flat SoA node pool in HBM, one tree per game, one wavefront per game per kernel: lanes stride
// over a node's edges, so the per-edge arrays (prior, visits, value sum, child) are read with
// coalesced 256-byte wave loads; argmax / sums are wave reductions.  See include/hive_search.h
// for the reference functions each entry point replaces.
#include <hip/hip_runtime.h>

#include <new>
#include <string>

#include "../../include/hive_search.h"
#include "hive_step.hpp"

namespace hive {

constexpr int EC = HIVE_EDGE_CAP;
enum LeafKind : int8_t { LEAF_NONE = 0, LEAF_ROOT = 1, LEAF_EXPAND = 2, LEAF_TERMINAL = 3, LEAF_COLLISION = 4, LEAF_CAPDRAW = 5 };
constexpr float kDrawSentinel = 5.0f;      // solo_play.py:180,183

struct SearchDev {
    int G, MN, L;
    HiveSearchParams prm;
    unsigned long long seed;
    HiveBoard *node_board;
    HiveHistory *node_hist;
    int32_t *node_nedge, *node_sum_n;
    int8_t *node_term;
    float *node_tv;
    int16_t *e_act;
    float *e_p, *e_w;
    int32_t *e_n, *e_child;
    int32_t *n_nodes;
    int8_t *active, *root_pending;
    HiveBoard *root_board;
    HiveHistory *root_hist;
    int32_t *path_node, *path_edge, *path_len;     // [L][G][MN], [L][G]
    int8_t *leaf_kind;                             // [L][G]
    int32_t *leaf_node;                            // [L][G]: terminal node reached / parent node of the expansion
    int32_t *leaf_edge;                            // [L][G]: parent edge of the expansion
    int32_t *tt;                                   // [G][TT] open-addressing table position -> node (-1 empty)
    int32_t *tt_hits;                              // [G] descents that continued through a transposition (statistics)
    int TT, merge;                                 // table size (power of two), merging on/off
    long long *game_id;                            // [G] global game index: the only per-game key of the noise streams
    int32_t *kind_hist;                            // [G][8] leaves by LeafKind since create (statistics)
};

// ------------------------------------------------------------------ small device helpers
__device__ __forceinline__ unsigned long long splitmix(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
struct Rng {
    unsigned long long s;
    __device__ Rng(unsigned long long seed, unsigned long long a, unsigned long long b, unsigned long long c)
    {
        s = splitmix(seed ^ splitmix(a * 0x100000001B3ull + splitmix(b * 0x9E3779B1ull + c)));
    }
    __device__ float uniform()     // (0, 1)
    {
        s = splitmix(s);
        return ((float)(s >> 40) + 0.5f) * (1.0f / 16777216.0f);
    }
    __device__ float normal()
    {
        float u1 = uniform(), u2 = uniform();
        return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2);
    }
    // Marsaglia-Tsang, alpha < 1 through the alpha+1 boost
    __device__ float gamma(float alpha)
    {
        float a = alpha + 1.0f, d = a - 1.0f / 3.0f, c = rsqrtf(9.0f * d), out = d;
        for (int it = 0; it < 8; ++it) {
            float x = normal(), v = 1.0f + c * x;
            if (v <= 0.0f) continue;
            v = v * v * v;
            float u = uniform();
            if (__logf(u) < 0.5f * x * x + d - d * v + d * __logf(v)) { out = d * v; break; }
        }
        return out * __powf(uniform(), 1.0f / alpha);
    }
};

// noise key of one draw: (search seed, global game id, turn of the root position, simulation of this search, in-flight
// slot, edge) -- nothing that depends on where in a batch, on which GPU or after how many other searches the game runs
__device__ __forceinline__ Rng noise_rng(unsigned long long seed, long long game, unsigned turn, unsigned long long sim,
                                         int slot, int e)
{
    return Rng(seed, (unsigned long long)game, ((unsigned long long)turn << 40) | (sim * 8ull + (unsigned long long)slot),
               (unsigned long long)e);
}

__device__ __forceinline__ float wave_sum(float x)
{
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}
__device__ __forceinline__ float wave_max(float x)
{
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
    return x;
}
__device__ __forceinline__ int wave_sum_i(int x)
{
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}
// (score, index) argmax; ties go to the lower index (first max wins, solo_play.py:332-334)
__device__ __forceinline__ void wave_argmax(float &score, int &idx)
{
    for (int o = 32; o > 0; o >>= 1) {
        float s2 = __shfl_xor(score, o);
        int i2 = __shfl_xor(idx, o);
        if (s2 > score || (s2 == score && i2 < idx)) { score = s2; idx = i2; }
    }
}
__device__ __forceinline__ void wave_copy(void *dst, const void *src, int bytes, int lane)
{
    const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
    uint32_t *d = reinterpret_cast<uint32_t *>(dst);
    for (int i = lane; i < bytes / 4; i += 64) d[i] = s[i];
}

// eta ~ Dirichlet(alpha) over the ne edges of one node, edge e = lane + 64 k in noise[k] (np.random.dirichlet,
// solo_play.py:322-323): independent Gamma(alpha, 1) draws divided by their sum
__device__ __forceinline__ void wave_dirichlet(float noise[4], int ne, float alpha, unsigned long long seed, long long game,
                                               unsigned turn, unsigned long long sim, int slot, int lane)
{
    float tot = 0.f;
    for (int k = 0; k < 4; ++k) {
        const int e = lane + 64 * k;
        noise[k] = 0.f;
        if (e < ne) {
            Rng r = noise_rng(seed, game, turn, sim, slot, e);
            noise[k] = r.gamma(alpha);
            tot += noise[k];
        }
    }
    tot = wave_sum(tot);
    const float inv = tot > 0.f ? 1.0f / tot : 0.f;
    for (int k = 0; k < 4; ++k) noise[k] *= inv;
}

// child_Q() + child_U() of alpha_zero/MCTS_chess.py:52-57 for one edge, in the reference's fp32 operation order
// (numpy float32 arrays; math.sqrt(number_visits) enters as a float32 scalar): W/(1+N) + sqrtN * (|P|/(1+N)).
// No fused multiply-add: the sum must round like numpy's separate multiply and add.
__device__ __forceinline__ float uct_score(float w, float n, float p, float sqrt_visits)
{
#pragma clang fp contract(off)
    const float d = 1.0f + n;
    const float q = w / d;
    const float u = sqrt_visits * (fabsf(p) / d);
    return q + u;
}

// The reference keeps its tree in a dict keyed by GamePlay.state_key (solo_play.py:167-197, env_hive.py:70-94,151-168):
// cells in board order with the piece keys bottom->top plus the side digit -- turn number and history excluded.
// The same information is bytes 0..32 of the HiveBoard record (pos[22], 4-bit stack indices) and the turn parity.
__device__ __forceinline__ uint32_t key_word(const uint32_t *rec, int i)      // i = 0..8
{
    uint32_t w = rec[i];
    if (i == 8) w = (w & 0xFFu) | ((w >> 8) & 1u) << 8;       // lvl[10] and the parity of the turn byte
    return w;
}
__device__ __forceinline__ uint32_t key_hash(const uint32_t *rec, int lane)
{
    unsigned long long h = 0ull;
    if (lane < 9) h = splitmix(((unsigned long long)(lane + 1) << 32) | key_word(rec, lane));
    for (int o = 8; o > 0; o >>= 1) h ^= __shfl_xor(h, o);      // lanes 0..15 hold the xor of lanes 0..8 (others add 0)
    return (uint32_t)(__shfl(h, 0) >> 17);
}
__device__ __forceinline__ bool key_equal(const uint32_t *a, const uint32_t *b, int lane)
{
    bool ne = lane < 9 && key_word(a, lane) != key_word(b, lane);
    return __ballot(ne) == 0ull;
}
// node holding the position `rec` of game g, or -1
__device__ __forceinline__ int tt_find(const SearchDev &S, int g, const uint32_t *rec, int lane)
{
    const uint32_t mask = (uint32_t)S.TT - 1u;
    uint32_t slot = key_hash(rec, lane) & mask;
    for (int probe = 0; probe < S.TT; ++probe, slot = (slot + 1u) & mask) {
        const int id = S.tt[(long long)g * S.TT + slot];
        if (id < 0) return -1;
        if (key_equal(rec, reinterpret_cast<const uint32_t *>(&S.node_board[(long long)g * S.MN + id]), lane)) return id;
    }
    return -1;
}
__device__ __forceinline__ void tt_insert(const SearchDev &S, int g, const uint32_t *rec, int id, int lane)
{
    const uint32_t mask = (uint32_t)S.TT - 1u;
    uint32_t slot = key_hash(rec, lane) & mask;
    for (int probe = 0; probe < S.TT; ++probe, slot = (slot + 1u) & mask)
        if (S.tt[(long long)g * S.TT + slot] < 0) {
            if (lane == 0) S.tt[(long long)g * S.TT + slot] = id;
            return;
        }
}

// ------------------------------------------------------------------ kernels
__global__ void __launch_bounds__(256)
search_reset_kernel(SearchDev S, const HiveBoard *__restrict__ boards, const HiveHistory *__restrict__ hist,
                    const int8_t *__restrict__ active)
{
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= S.G) return;
    wave_copy(&S.root_board[g], &boards[g], sizeof(HiveBoard), lane);
    wave_copy(&S.root_hist[g], &hist[g], sizeof(HiveHistory), lane);
    for (int i = lane; i < S.TT; i += 64) S.tt[(long long)g * S.TT + i] = -1;
    if (lane == 0) {
        S.n_nodes[g] = 0;
        S.root_pending[g] = 0;
        S.active[g] = active ? active[g] : 1;
    }
}

// solo_play.py:167-215 (descent), :294-335 (PUCT with per-simulation root noise)
__global__ void __launch_bounds__(256)
search_select_kernel(SearchDev S, int slot, unsigned long long sim, HiveBoard *__restrict__ leaf_boards,
                     HiveHistory *__restrict__ leaf_hist)
{
    __shared__ uint32_t stage[4][112];
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= S.G) return;
    const long long sg = (long long)slot * S.G + g;
    int32_t *pnode = S.path_node + sg * S.MN, *pedge = S.path_edge + sg * S.MN;
    if (!S.active[g]) { if (lane == 0) { S.leaf_kind[sg] = LEAF_NONE; S.path_len[sg] = 0; } return; }

    if (S.n_nodes[g] == 0) {
        // first simulation of a search only evaluates the root (solo_play.py:188-197)
        int kind = S.root_pending[g] ? LEAF_COLLISION : LEAF_ROOT;
        if (kind == LEAF_ROOT) {
            wave_copy(&leaf_boards[g], &S.root_board[g], sizeof(HiveBoard), lane);
            wave_copy(&leaf_hist[g], &S.root_hist[g], sizeof(HiveHistory), lane);
        }
        if (lane == 0) { S.leaf_kind[sg] = (int8_t)kind; S.path_len[sg] = 0; S.root_pending[g] = 1; }
        return;
    }

    const long long nbase = (long long)g * S.MN;
    // The env of this simulation (deepcopy(env) of solo_play.py:158) lives in LDS and takes every move of the descent
    // (env.move, solo_play.py:213): turn number and history planes are those of THIS path even when the nodes it runs
    // through were first reached another way.
    uint32_t *st = stage[threadIdx.x >> 6];
    wave_copy(st, &S.node_board[nbase], sizeof(HiveBoard), lane);
    wave_copy(st + 16, &S.node_hist[nbase], sizeof(HiveHistory), lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    int node = 0, depth = 0, kind = LEAF_NONE, leafnode = 0, leafedge = 0;
    const unsigned root_turn = reinterpret_cast<const uint8_t *>(st)[33];
    while (true) {
        if (S.node_term[nbase + node]) { kind = LEAF_TERMINAL; leafnode = node; break; }          // game over (solo_play.py:169-180)
        if ((int)reinterpret_cast<const uint8_t *>(st)[33] >= S.prm.max_game_length) { kind = LEAF_CAPDRAW; break; }   // :181-183
        const int ne = S.node_nedge[nbase + node];
        const long long eb = (nbase + node) * EC;
        const bool uct = S.prm.mode == HIVE_SEARCH_UCT;
        // PUCT: sqrt(sum_n + 1) (solo_play.py:316).  UCT: sqrt(number_visits) of this node (MCTS_chess.py:55-57) -- every
        // backup through a node passes one of its edges except the one that expanded it, so number_visits = sum_n + 1 too
        const float xx = sqrtf((float)S.node_sum_n[nbase + node] + 1.0f);
        // fresh Dirichlet noise on the root priors for every simulation (solo_play.py:322-323)
        float noise[4] = {0.f, 0.f, 0.f, 0.f};
        const bool noisy = depth == 0 && !uct && S.prm.noise_eps > 0.f;
        if (noisy)
            wave_dirichlet(noise, ne, S.prm.dirichlet_alpha, S.seed, S.game_id[g], root_turn, sim, slot, lane);
        float best = -3.0e38f;
        int bidx = 0x7FFFFFFF;
        for (int k = 0; k < 4; ++k) {
            int e = lane + 64 * k;
            if (e < ne) {
                float p = S.e_p[eb + e], w = S.e_w[eb + e];
                int n = S.e_n[eb + e];
                float b;
                if (uct) {
                    b = uct_score(w, (float)n, p, xx);
                } else {
                    float q = n > 0 ? w / (float)n : 0.f;
                    if (noisy) p = (1.0f - S.prm.noise_eps) * p + S.prm.noise_eps * noise[k];
                    b = q + S.prm.c_puct * p * xx / (1.0f + (float)n);
                }
                if (b > best) { best = b; bidx = e; }
            }
        }
        wave_argmax(best, bidx);
        int child = S.e_child[eb + bidx];
        if (child == -1 && S.n_nodes[g] + S.L >= S.MN) child = -2;    // node pool exhausted: treat as a collision
        if (lane == 0) {
            // virtual loss (solo_play.py:205-208)
            S.node_sum_n[nbase + node] += 1;
            S.e_n[eb + bidx] += 1;
            S.e_w[eb + bidx] -= S.prm.virtual_loss;
            pnode[depth] = node;
            pedge[depth] = bidx;
            apply_action(reinterpret_cast<HiveBoard *>(st), reinterpret_cast<HiveHistory *>(st + 16), S.e_act[eb + bidx]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        depth++;
        if (child == -1 && S.merge) {
            // `state in self.tree` (solo_play.py:188): the position may already have a node, reached by another move order
            const int known = tt_find(S, g, st, lane);
            if (known >= 0) {
                child = known;
                if (lane == 0) { S.e_child[eb + bidx] = known; S.tt_hits[g] += 1; }
            }
        }
        if (child == -1) {
            if (lane == 0) S.e_child[eb + bidx] = -2;                 // expansion in flight
            kind = LEAF_EXPAND; leafnode = node; leafedge = bidx;
            break;
        }
        if (child == -2 || depth >= S.MN - 1) { kind = LEAF_COLLISION; break; }
        node = child;
    }
    if (kind == LEAF_EXPAND) {
        wave_copy(&leaf_boards[g], st, sizeof(HiveBoard), lane);
        wave_copy(&leaf_hist[g], st + 16, sizeof(HiveHistory), lane);
    }
    if (lane == 0) {
        S.leaf_kind[sg] = (int8_t)kind;
        S.leaf_node[sg] = leafnode;
        S.leaf_edge[sg] = leafedge;
        S.path_len[sg] = depth;
    }
}

// expansion (solo_play.py:188-197,304-313) + backup (solo_play.py:217-247)
__global__ void __launch_bounds__(256)
search_backup_kernel(SearchDev S, int slot, const HiveBoard *__restrict__ leaf_boards,
                     const HiveHistory *__restrict__ leaf_hist, const uint32_t *__restrict__ leaf_mask,
                     const int8_t *__restrict__ over, const int8_t *__restrict__ winner, const float *__restrict__ p,
                     const float *__restrict__ v)
{
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= S.G) return;
    const long long sg = (long long)slot * S.G + g;
    const int kind = S.leaf_kind[sg];
    if (kind == LEAF_NONE) return;
    const long long nbase = (long long)g * S.MN;
    const int32_t *pnode = S.path_node + sg * S.MN, *pedge = S.path_edge + sg * S.MN;
    const int plen = S.path_len[sg];

    if (kind == LEAF_COLLISION) {
        // two in-flight selections met: take the virtual loss back, nothing to learn
        if (lane == 0)
            for (int d = 0; d < plen; ++d) {
                long long eb = (nbase + pnode[d]) * EC + pedge[d];
                S.node_sum_n[nbase + pnode[d]] -= 1;
                S.e_n[eb] -= 1;
                S.e_w[eb] += S.prm.virtual_loss;
            }
        if (lane == 0) S.kind_hist[g * 8 + LEAF_COLLISION] += 1;
        return;
    }
    const bool uct = S.prm.mode == HIVE_SEARCH_UCT;

    float ret;
    int known = -1;
    const bool capped = kind == LEAF_CAPDRAW ||
                        (kind == LEAF_EXPAND && !over[g] && (int)leaf_boards[g].turn >= S.prm.max_game_length);
    int stat = kind;                                        // hive_search_leaf_histogram bucket
    if (kind == LEAF_EXPAND && !capped && S.merge)          // another in-flight slot may have created this position meanwhile
        known = tt_find(S, g, reinterpret_cast<const uint32_t *>(&leaf_boards[g]), lane);
    if (kind == LEAF_TERMINAL) {
        ret = S.node_tv[nbase + S.leaf_node[sg]];
    } else if (capped) {
        // length cap (solo_play.py:181-183): a property of this path's turn count, not of the position -- no node
        if (kind == LEAF_EXPAND && lane == 0) S.e_child[(nbase + S.leaf_node[sg]) * EC + S.leaf_edge[sg]] = -1;
        ret = uct ? 0.f : kDrawSentinel;                    // (UCT has no cap; 250 plies only guard the 8-bit turn counter)
        if (kind == LEAF_EXPAND) stat = 6;
    } else if (known >= 0) {
        if (lane == 0) S.e_child[(nbase + S.leaf_node[sg]) * EC + S.leaf_edge[sg]] = known;
        ret = S.node_term[nbase + known] ? S.node_tv[nbase + known] : v[g];
        stat = 7;
    } else {
        const int id = (kind == LEAF_ROOT) ? 0 : S.n_nodes[g];
        wave_copy(&S.node_board[nbase + id], &leaf_boards[g], sizeof(HiveBoard), lane);
        wave_copy(&S.node_hist[nbase + id], &leaf_hist[g], sizeof(HiveHistory), lane);
        if (S.merge) tt_insert(S, g, reinterpret_cast<const uint32_t *>(&leaf_boards[g]), id, lane);
        const unsigned turn = leaf_boards[g].turn;
        const int stm = (turn & 1u) ? 0 : 1;
        bool term = false;
        float tv = 0.f;
        if (over[g] && uct) {
            // MCTS_chess.py:144-146: a finished game is never expanded; every visit backs up the network's value of
            // that same position -- the evaluator is a pure function of the planes, so the value is kept with the node
            term = true;
            tv = v[g];
        } else if (over[g]) {                               // solo_play.py:169-180
            term = true;
            int w = winner[g];
            tv = (w == 0) ? kDrawSentinel : ((w - 1) == stm ? 1.0f : -1.0f);
            stat = 6;
        } else if ((int)turn >= S.prm.max_game_length) {    // solo_play.py:181-183
            term = true;
            tv = kDrawSentinel;
        }
        const long long eb = (nbase + id) * EC;
        int ne = 0;
        if (!term) {
            // the leaf's legal set arrives as 11 destination boards (hive_abi.h); lane l of pass t owns action 64 t + l,
            // so the ballots below are the id-ordered mask words and the edges come out in ascending action order
            const uint32_t *m = leaf_mask + (long long)g * HIVE_MASK_WORDS;
            const float *pg = p + (long long)g * HIVE_ACTIONS;
            unsigned legal_bits = 0u;                       // bit t: action 64 t + lane is legal
            float tot = 0.f;
            for (int t = 0; t < 25; ++t) {
                const int a = t * 64 + lane;
                if (a < HIVE_ACTIONS && HIVE_MASK_TEST(m, a)) { legal_bits |= 1u << t; tot += pg[a]; }
            }
            tot = wave_sum(tot) + 1e-8f;                    // solo_play.py:305-312
            const float inv = uct ? 1.0f : 1.0f / tot;      // MCTS_chess.py:89-94: illegal priors zeroed, no renormalisation
            int total = 0;
            for (int t = 0; t < 25; ++t) {
                const bool mine = (legal_bits >> t) & 1u;
                const unsigned long long w = __ballot(mine);
                if (mine) {
                    int pos = total + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(w >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)w, 0u));
                    if (pos < EC) {
                        int a = t * 64 + lane;
                        S.e_act[eb + pos] = (int16_t)a;
                        S.e_p[eb + pos] = pg[a] * inv;
                        S.e_n[eb + pos] = 0;
                        S.e_w[eb + pos] = 0.f;
                        S.e_child[eb + pos] = -1;
                    }
                }
                total += __popcll(w);
            }
            ne = total < EC ? total : EC;
            if (ne == 0 && uct) {
                // MCTS_chess.py:87-88: is_expanded stays False, the node is evaluated again on every visit (same value)
                term = true;
                tv = v[g];
            } else if (ne == 0) {                           // no legal move: the pass edge (solo_play.py:299-300)
                if (lane == 0) { S.e_act[eb] = -1; S.e_p[eb] = 0.f; S.e_n[eb] = 0; S.e_w[eb] = 0.f; S.e_child[eb] = -1; }
                ne = 1;
            }
        }
        if (lane == 0) {
            S.node_nedge[nbase + id] = ne;
            S.node_sum_n[nbase + id] = 0;
            S.node_term[nbase + id] = term ? 1 : 0;
            S.node_tv[nbase + id] = tv;
            S.n_nodes[g] = id + 1;
            if (kind == LEAF_EXPAND) S.e_child[(nbase + S.leaf_node[sg]) * EC + S.leaf_edge[sg]] = id;
        }
        ret = term ? tv : v[g];
    }
    if (lane == 0) {
        S.kind_hist[g * 8 + stat] += 1;
        const float vl = S.prm.virtual_loss;
        if (uct) {
            // MCTS_chess.py:111-119: every node from the leaf up to the root gets number_visits += 1 (done at selection)
            // and total_value += v if black is to move AT that node, -v if white is -- i.e. the edge leaving a position
            // with white to move (odd turn) collects +v
            for (int d = plen - 1; d >= 0; --d) {
                const long long eidx = (nbase + pnode[d]) * EC + pedge[d];
                const float sv = (S.node_board[nbase + pnode[d]].turn & 1u) ? ret : -ret;
                S.e_w[eidx] = vl != 0.f ? S.e_w[eidx] + (vl + sv) : S.e_w[eidx] + sv;
            }
        } else {
            float val = ret;
            for (int d = plen - 1; d >= 0; --d) {
                long long eidx = (nbase + pnode[d]) * EC + pedge[d];
                bool reach_max = (val == kDrawSentinel);    // solo_play.py:219-223
                float leaf_v = reach_max ? -1.0f : -val;
                S.e_w[eidx] += vl + leaf_v;                 // virtual_loss + leaf_v; n and sum_n: -vl + 1 = 0
                val = reach_max ? kDrawSentinel : leaf_v;   // solo_play.py:244-247
            }
        }
    }
}

// solo_play.py:337-374 + self_play.py:139-157
__global__ void __launch_bounds__(256)
search_policy_kernel(SearchDev S, float *__restrict__ policy, int32_t *__restrict__ action,
                     int32_t *__restrict__ sum_n_out, int selfplay)
{
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= S.G) return;
    float *pol = policy ? policy + (long long)g * HIVE_ACTIONS : nullptr;
    if (pol) for (int i = lane; i < HIVE_ACTIONS; i += 64) pol[i] = 0.f;
    if (!S.active[g] || S.n_nodes[g] == 0 || S.node_term[(long long)g * S.MN]) {
        if (lane == 0) { action[g] = -2; if (sum_n_out) sum_n_out[g] = 0; }
        return;
    }
    const long long nbase = (long long)g * S.MN, eb = nbase * EC;
    const int ne = S.node_nedge[nbase];
    int nsum = 0;
    float wmax = -3.0e38f;
    for (int k = 0; k < 4; ++k) {
        int e = lane + 64 * k;
        if (e < ne) { nsum += S.e_n[eb + e]; wmax = fmaxf(wmax, S.e_w[eb + e]); }
    }
    nsum = wave_sum_i(nsum);
    wmax = wave_max(wmax);
    // solo_play.py:372-373 (HivePlayer falls back to the priors when every W is negative); MCTS_chess.py:157-161
    // (get_policy) has no such rule -- before the first backed-up visit its policy is all zero
    const bool uct = S.prm.mode == HIVE_SEARCH_UCT;
    const bool use_prior = uct ? false : ((wmax < 0.f) || nsum == 0);
    float pi[4];
    float best = -1.f;
    int bidx = 0x7FFFFFFF;
    for (int k = 0; k < 4; ++k) {
        int e = lane + 64 * k;
        pi[k] = 0.f;
        if (e < ne) {
            pi[k] = use_prior ? S.e_p[eb + e] : (nsum > 0 ? (float)S.e_n[eb + e] / (float)nsum : 0.f);
            int a = S.e_act[eb + e];
            if (pol && a >= 0) pol[a] = pi[k];
            if (pi[k] > best) { best = pi[k]; bidx = e; }
        }
    }
    wave_argmax(best, bidx);                                // tau < 0.1 always => argmax (solo_play.py:338-345)
    int chosen = bidx;
    const unsigned turn = S.node_board[nbase].turn;
    if (selfplay && S.e_act[eb] >= 0) {
        // self_play.py:143-157: e = 0.7 - int(turn+1)/2 * 0.15; resample while e >= 0.1 (turns 1..6)
        float err = 0.7f - 0.5f * (float)(turn + 1u) * 0.15f;
        if (turn <= 6u) {
            float nz[4], tot = 0.f;
            for (int k = 0; k < 4; ++k) {
                int e = lane + 64 * k;
                nz[k] = 0.f;
                if (e < ne) {
                    Rng r = noise_rng(S.seed ^ 0xD1B54A32D192ED03ull, S.game_id[g], turn, 0ull, 0, e);
                    nz[k] = r.gamma(0.5f);
                    tot += nz[k];
                }
            }
            tot = wave_sum(tot);
            float inv = tot > 0.f ? 1.0f / tot : 0.f, psum = 0.f;
            for (int k = 0; k < 4; ++k) { nz[k] = (1.0f - err) * pi[k] + err * nz[k] * inv; psum += nz[k]; }
            psum = wave_sum(psum);
            Rng r = noise_rng(S.seed ^ 0xA24BAED4963EE407ull, S.game_id[g], turn, 0ull, 0, 7777);
            float target = r.uniform() * psum;
            // inclusive prefix over edges in index order: lanes hold e = lane + 64k, so scan k-major
            float acc = 0.f;
            int pick = ne - 1;
            bool done = false;
            for (int k = 0; k < 4 && !done; ++k) {
                float x = nz[k], incl = x;
                for (int o = 1; o < 64; o <<= 1) {
                    float y = __shfl_up(incl, o);
                    if (lane >= o) incl += y;
                }
                float row_total = __shfl(incl, 63);
                bool hit = (lane + 64 * k < ne) && (acc + incl >= target);
                unsigned long long hm = __ballot(hit);
                if (hm) { pick = 64 * k + (int)__builtin_ctzll(hm); done = true; }
                acc += row_total;
            }
            chosen = pick;
        }
    }
    if (lane == 0) {
        action[g] = S.e_act[eb + chosen];
        if (sum_n_out) sum_n_out[g] = nsum;
    }
}

// hive_search_leaf_need: which of the selected leaves will search_backup_kernel ask the network about?  Exactly the
// conditions of that kernel: a finished game (PUCT), a leaf at the length cap, a revisited terminal node, a collision and an
// idle tree take their value from elsewhere -- the evaluator may skip their rows.
__global__ void __launch_bounds__(256)
search_need_kernel(SearchDev S, int slots, const HiveBoard *__restrict__ leaf_boards, const int8_t *__restrict__ over,
                   int8_t *__restrict__ need, unsigned long long *__restrict__ total)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)slots * S.G;
    bool want = false;
    if (i < n) {
        const int kind = S.leaf_kind[i];
        want = kind == LEAF_ROOT || kind == LEAF_EXPAND;
        if (want && S.prm.mode != HIVE_SEARCH_UCT)
            want = !over[i] && (int)leaf_boards[i].turn < S.prm.max_game_length;
        need[i] = want ? 1 : 0;
    }
    if (total) {
        const unsigned long long w = __ballot(want);
        if ((threadIdx.x & 63) == 0 && w) atomicAdd(total, (unsigned long long)__popcll(w));
    }
}

// UCTNode.child_number_visits / child_total_value / child_priors of the root (MCTS_chess.py:33-35) as dense action-indexed rows
__global__ void __launch_bounds__(256)
search_root_stats_kernel(SearchDev S, float *__restrict__ visits, float *__restrict__ total_value, float *__restrict__ priors)
{
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= S.G) return;
    const long long row = (long long)g * HIVE_ACTIONS;
    for (int i = lane; i < HIVE_ACTIONS; i += 64) {
        if (visits) visits[row + i] = 0.f;
        if (total_value) total_value[row + i] = 0.f;
        if (priors) priors[row + i] = 0.f;
    }
    if (!S.active[g] || S.n_nodes[g] == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const long long eb = (long long)g * S.MN * EC;
    const int ne = S.node_term[(long long)g * S.MN] ? 0 : S.node_nedge[(long long)g * S.MN];
    for (int e = lane; e < ne; e += 64) {
        const int a = S.e_act[eb + e];
        if (a < 0) continue;
        if (visits) visits[row + a] = (float)S.e_n[eb + e];
        if (total_value) total_value[row + a] = S.e_w[eb + e];
        if (priors) priors[row + a] = S.e_p[eb + e];
    }
}

// hive_search_sample_noise: one wave per draw, the same wave_dirichlet / mixing arithmetic as search_select_kernel
__global__ void __launch_bounds__(256)
search_noise_kernel(unsigned long long seed, long long first_game, unsigned turn, float alpha, int k, int draws,
                    const float *__restrict__ prior, float eps, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int d = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (d >= draws) return;
    float noise[4];
    wave_dirichlet(noise, k, alpha, seed, first_game + d, turn, 0ull, 0, lane);
    for (int j = 0; j < 4; ++j) {
        const int e = lane + 64 * j;
        if (e < k) out[(long long)d * k + e] = prior ? (1.0f - eps) * prior[e] + eps * noise[j] : noise[j];
    }
}

}  // namespace hive

// ====================================================================== host side / C ABI
using namespace hive;

extern "C" const char *hive_last_error(void);
namespace hive { int set_error(int code, const std::string &msg); }

#define S_TRY(expr)                                                                                 \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return hive::set_error(HIVE_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct HiveSearch {
    SearchDev d{};
    int device = 0;
    hipStream_t stream = nullptr;
    unsigned long long sim = 0;
    void *pool[40];
    int npool = 0;
};

template <typename T>
static hipError_t alloc(HiveSearch *s, T **p, size_t count)
{
    hipError_t e = hipMalloc(reinterpret_cast<void **>(p), sizeof(T) * count);
    if (e == hipSuccess) s->pool[s->npool++] = *p;
    return e;
}

extern "C" {

static int search_alloc(HiveSearch *s, int games, int max_nodes, int slots)
{
    SearchDev &d = s->d;
    const size_t GN = (size_t)games * max_nodes, GE = GN * EC, LG = (size_t)slots * games;
    S_TRY(alloc(s, &d.node_board, GN));
    S_TRY(alloc(s, &d.node_hist, GN));
    S_TRY(alloc(s, &d.node_nedge, GN));
    S_TRY(alloc(s, &d.node_sum_n, GN));
    S_TRY(alloc(s, &d.node_term, GN));
    S_TRY(alloc(s, &d.node_tv, GN));
    S_TRY(alloc(s, &d.e_act, GE));
    S_TRY(alloc(s, &d.e_p, GE));
    S_TRY(alloc(s, &d.e_w, GE));
    S_TRY(alloc(s, &d.e_n, GE));
    S_TRY(alloc(s, &d.e_child, GE));
    S_TRY(alloc(s, &d.n_nodes, (size_t)games));
    S_TRY(alloc(s, &d.active, (size_t)games));
    S_TRY(alloc(s, &d.root_pending, (size_t)games));
    S_TRY(alloc(s, &d.root_board, (size_t)games));
    S_TRY(alloc(s, &d.root_hist, (size_t)games));
    S_TRY(alloc(s, &d.path_node, LG * max_nodes));
    S_TRY(alloc(s, &d.path_edge, LG * max_nodes));
    S_TRY(alloc(s, &d.path_len, LG));
    S_TRY(alloc(s, &d.leaf_kind, LG));
    S_TRY(alloc(s, &d.leaf_node, LG));
    S_TRY(alloc(s, &d.leaf_edge, LG));
    d.TT = 64;
    while (d.TT < 2 * max_nodes) d.TT *= 2;
    d.merge = 1;
    S_TRY(alloc(s, &d.tt, (size_t)games * d.TT));
    S_TRY(alloc(s, &d.tt_hits, (size_t)games));
    S_TRY(alloc(s, &d.game_id, (size_t)games));
    S_TRY(alloc(s, &d.kind_hist, (size_t)games * 8));
    S_TRY(hipMemset(d.kind_hist, 0, sizeof(int32_t) * (size_t)games * 8));
    {
        long long *ids = new (std::nothrow) long long[games];
        if (ids == nullptr) return hive::set_error(HIVE_E_DEVICE, "hive_search_create: out of host memory");
        for (int i = 0; i < games; ++i) ids[i] = i;
        hipError_t e = hipMemcpy(d.game_id, ids, sizeof(long long) * (size_t)games, hipMemcpyHostToDevice);
        delete[] ids;
        S_TRY(e);
    }
    S_TRY(hipMemset(d.tt, 0xFF, sizeof(int32_t) * (size_t)games * d.TT));
    S_TRY(hipMemset(d.tt_hits, 0, sizeof(int32_t) * games));
    S_TRY(hipMemset(d.n_nodes, 0, sizeof(int32_t) * games));
    S_TRY(hipMemset(d.active, 0, games));
    S_TRY(hipMemset(d.root_pending, 0, games));
    S_TRY(hipMemset(d.leaf_kind, 0, LG));
    S_TRY(hipMemset(d.path_len, 0, sizeof(int32_t) * LG));
    S_TRY(hipMemset(d.node_term, 0, GN));
    return HIVE_OK;
}

int hive_search_destroy(HiveSearch *s);

int hive_search_create(int games, int max_nodes, int slots, int device, uint64_t seed, HiveSearch **out)
{
    if (games <= 0 || max_nodes < 2 || slots < 1 || slots > HIVE_MAX_SLOTS || !out)
        return hive::set_error(HIVE_E_ARG, "hive_search_create: bad argument");
    *out = nullptr;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0)
        return hive::set_error(HIVE_E_DEVICE, "hive_search_create: no HIP device visible (no CPU path)");
    if (device < 0 || device >= cnt) return hive::set_error(HIVE_E_ARG, "hive_search_create: bad device ordinal");
    S_TRY(hipSetDevice(device));
    HiveSearch *s = new (std::nothrow) HiveSearch();
    if (s == nullptr) return hive::set_error(HIVE_E_DEVICE, "hive_search_create: out of host memory");
    s->device = device;
    s->d.G = games; s->d.MN = max_nodes; s->d.L = slots; s->d.seed = seed;
    s->d.prm = HiveSearchParams{0.7f, 0.25f, 0.3f, 55, HIVE_SEARCH_PUCT, 1.0f};
    int rc = search_alloc(s, games, max_nodes, slots);
    if (rc != HIVE_OK) {            // free whatever was allocated; the caller never sees a partial handle
        hive_search_destroy(s);
        return rc;
    }
    *out = s;
    return HIVE_OK;
}

int hive_search_destroy(HiveSearch *s)
{
    if (!s) return HIVE_OK;
    (void)hipSetDevice(s->device);
    for (int i = 0; i < s->npool; ++i) (void)hipFree(s->pool[i]);
    delete s;
    return HIVE_OK;
}

int hive_search_set_stream(HiveSearch *s, void *stream)
{
    if (!s) return hive::set_error(HIVE_E_ARG, "null handle");
    s->stream = (hipStream_t)stream;
    return HIVE_OK;
}

int hive_search_set_params(HiveSearch *s, const HiveSearchParams *p)
{
    if (!s || !p) return hive::set_error(HIVE_E_ARG, "null argument");
    if ((p->mode != HIVE_SEARCH_PUCT && p->mode != HIVE_SEARCH_UCT) || !(p->virtual_loss >= 0.f) || p->max_game_length < 1 ||
        p->max_game_length > 250)
        return hive::set_error(HIVE_E_ARG, "hive_search_set_params: unknown mode, negative virtual loss or length cap outside 1..250");
    s->d.prm = *p;
    return HIVE_OK;
}

static dim3 wave_grid(int games) { return dim3((unsigned)((games + 3) / 4)); }

int hive_search_set_roots(HiveSearch *s, const HiveBoard *boards, const HiveHistory *hist, const int8_t *active)
{
    if (!s || !boards || !hist) return hive::set_error(HIVE_E_ARG, "null argument");
    S_TRY(hipSetDevice(s->device));
    hipLaunchKernelGGL(search_reset_kernel, wave_grid(s->d.G), dim3(256), 0, s->stream, s->d, boards, hist, active);
    S_TRY(hipGetLastError());
    s->sim = 0;                 // simulations are numbered inside one search: part of the noise key (hive_search_set_game_ids)
    return HIVE_OK;
}

int hive_search_select(HiveSearch *s, int slot, HiveBoard *leaf_boards, HiveHistory *leaf_hist)
{
    if (!s || !leaf_boards || !leaf_hist || slot < 0 || slot >= s->d.L) return hive::set_error(HIVE_E_ARG, "bad argument");
    S_TRY(hipSetDevice(s->device));
    hipLaunchKernelGGL(search_select_kernel, wave_grid(s->d.G), dim3(256), 0, s->stream, s->d, slot, s->sim++, leaf_boards,
                       leaf_hist);
    S_TRY(hipGetLastError());
    return HIVE_OK;
}

int hive_search_backup(HiveSearch *s, int slot, const HiveBoard *leaf_boards, const HiveHistory *leaf_hist,
                       const uint32_t *leaf_mask, const int8_t *over, const int8_t *winner, const float *p, const float *v)
{
    if (!s || !leaf_boards || !leaf_hist || !leaf_mask || !over || !winner || !p || !v || slot < 0 || slot >= s->d.L)
        return hive::set_error(HIVE_E_ARG, "bad argument");
    S_TRY(hipSetDevice(s->device));
    hipLaunchKernelGGL(search_backup_kernel, wave_grid(s->d.G), dim3(256), 0, s->stream, s->d, slot, leaf_boards, leaf_hist,
                       leaf_mask, over, winner, p, v);
    S_TRY(hipGetLastError());
    return HIVE_OK;
}

int hive_search_leaf_need(HiveSearch *s, int slots, const HiveBoard *leaf_boards, const int8_t *over, int8_t *need,
                          uint64_t *total)
{
    if (!s || !leaf_boards || !over || !need || slots < 1 || slots > s->d.L) return hive::set_error(HIVE_E_ARG, "bad argument");
    S_TRY(hipSetDevice(s->device));
    const long long n = (long long)slots * s->d.G;
    hipLaunchKernelGGL(search_need_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, s->d, slots, leaf_boards,
                       over, need, reinterpret_cast<unsigned long long *>(total));
    S_TRY(hipGetLastError());
    return HIVE_OK;
}

int hive_search_policy(HiveSearch *s, float *policy, int32_t *action, int32_t *sum_n, int selfplay)
{
    if (!s || !action) return hive::set_error(HIVE_E_ARG, "null argument");
    S_TRY(hipSetDevice(s->device));
    hipLaunchKernelGGL(search_policy_kernel, wave_grid(s->d.G), dim3(256), 0, s->stream, s->d, policy, action, sum_n, selfplay);
    S_TRY(hipGetLastError());
    return HIVE_OK;
}

int hive_search_set_transpositions(HiveSearch *s, int merge)
{
    if (!s) return hive::set_error(HIVE_E_ARG, "null handle");
    s->d.merge = merge ? 1 : 0;
    return HIVE_OK;
}

int hive_search_transposition_hits(HiveSearch *s, int32_t *hits)
{
    if (!s || !hits) return hive::set_error(HIVE_E_ARG, "null argument");
    S_TRY(hipSetDevice(s->device));
    S_TRY(hipMemcpyAsync(hits, s->d.tt_hits, sizeof(int32_t) * s->d.G, hipMemcpyDeviceToDevice, s->stream));
    return HIVE_OK;
}

int hive_search_node_counts(HiveSearch *s, int32_t *counts)
{
    if (!s || !counts) return hive::set_error(HIVE_E_ARG, "null argument");
    S_TRY(hipSetDevice(s->device));
    S_TRY(hipMemcpyAsync(counts, s->d.n_nodes, sizeof(int32_t) * s->d.G, hipMemcpyDeviceToDevice, s->stream));
    return HIVE_OK;
}

int hive_search_set_game_ids(HiveSearch *s, const int64_t *ids)
{
    if (!s || !ids) return hive::set_error(HIVE_E_ARG, "null argument");
    S_TRY(hipSetDevice(s->device));
    S_TRY(hipMemcpyAsync(s->d.game_id, ids, sizeof(long long) * (size_t)s->d.G, hipMemcpyDeviceToDevice, s->stream));
    return HIVE_OK;
}

int hive_search_root_stats(HiveSearch *s, float *visits, float *total_value, float *priors)
{
    if (!s) return hive::set_error(HIVE_E_ARG, "null handle");
    S_TRY(hipSetDevice(s->device));
    hipLaunchKernelGGL(search_root_stats_kernel, wave_grid(s->d.G), dim3(256), 0, s->stream, s->d, visits, total_value, priors);
    S_TRY(hipGetLastError());
    return HIVE_OK;
}

int hive_search_leaf_histogram(HiveSearch *s, int32_t *hist)
{
    if (!s || !hist) return hive::set_error(HIVE_E_ARG, "null argument");
    S_TRY(hipSetDevice(s->device));
    S_TRY(hipMemcpyAsync(hist, s->d.kind_hist, sizeof(int32_t) * (size_t)s->d.G * 8, hipMemcpyDeviceToDevice, s->stream));
    return HIVE_OK;
}

int hive_search_sample_noise(uint64_t seed, int64_t first_game, int turn, float alpha, int k, int draws, const float *prior,
                             float eps, float *out, void *stream)
{
    if (!out || k < 1 || k > HIVE_EDGE_CAP || draws < 1 || !(alpha > 0.f) || turn < 0 || turn > 255)
        return hive::set_error(HIVE_E_ARG, "hive_search_sample_noise: bad argument");
    hipLaunchKernelGGL(search_noise_kernel, wave_grid(draws), dim3(256), 0, (hipStream_t)stream, (unsigned long long)seed,
                       (long long)first_game, (unsigned)turn, alpha, k, draws, prior, eps, out);
    S_TRY(hipGetLastError());
    return HIVE_OK;
}

}  // extern "C"
