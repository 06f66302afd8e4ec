/*
 * hive_search.h -- C ABI of the GPU-resident tree search (libhive_hip.so).
 *
 * Replaces the reference's per-process MCTS for the throughput path:
 *   woker/solo_play.py::HivePlayer (:69-384)   search_my_move / select_action_q_and_u / calc_policy
 *   woker/self_play.py::self_play_buffer (:116-193)   move selection rules of the self-play loop
 * One tree per game lives in a flat SoA node pool in HBM; G games are searched in lock step, one
 * leaf per game (and per in-flight slot) per simulation, so the network sees batches of G leaves
 * (woker/api_hive.py:47-74 batched whatever arrived within 1 ms).
 *
 * Division of labour per simulation:
 *   hive_search_select   PUCT descent + virtual loss (solo_play.py:199-208,294-335); writes the leaf
 *                        positions (parent record + apply_action) into caller buffers
 *   -- caller: hive_encode_launch / hive_movegen_launch / terminal on the leaves, then the network --
 *   hive_search_backup   expansion (priors masked + renormalised, solo_play.py:304-313) and the
 *                        value backup with the reference's draw sentinel (solo_play.py:217-247)
 *
 * Differences from the reference, by design: an explicit node pool per game instead of a
 * dict keyed by state_key (transpositions are merged through a hash table, see below); fp32 statistics; the
 * Dirichlet noise comes from a counter-based generator, not numpy's stream.  The reference-exact sequential
 * search is hive-alphazero_amd/solo_play.py.
 *
 * All pointers are DEVICE pointers; the caller allocates leaf/policy buffers, the handle owns the
 * node pool.  Return codes and hive_last_error() as in hive_abi.h.
 */
#ifndef HIVE_SEARCH_H
#define HIVE_SEARCH_H

#include <stdint.h>

#include "hive_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HIVE_EDGE_CAP 256      /* edges stored per node (legal ids beyond it are dropped; max seen: 131) */
#define HIVE_MAX_SLOTS 8       /* in-flight leaves per tree */

typedef struct HiveSearch HiveSearch;

/* Which of the reference's two searches the kernels run. */
typedef enum {
    HIVE_SEARCH_PUCT = 0,  /* woker/solo_play.py::HivePlayer (:167-374): q = w/n, u = c_puct p sqrt(sum_n + 1)/(1 + n), priors masked
                              and renormalised, root Dirichlet noise per simulation, draw sentinel, length cap, state_key merging */
    HIVE_SEARCH_UCT = 1    /* alpha_zero/MCTS_chess.py::UCTNode/UCT_search (:52-151): Q = W/(1 + N), U = sqrt(N_node) |P|/(1 + N)
                              (c = 1), illegal priors zeroed WITHOUT renormalisation, no noise, plain tree, the backup adds
                              +v to edges played by white and -v to edges played by black (:111-119), a finished game or a
                              position without legal moves is re-evaluated by the network on every visit (:144-146, :87-88),
                              no length cap.  All score arithmetic in fp32 with the reference's operation order. */
} HiveSearchMode;

typedef struct HiveSearchParams {
    float c_puct;          /* solo_play.py:26   0.7  */
    float noise_eps;       /* solo_play.py:29   0.25 */
    float dirichlet_alpha; /* solo_play.py:28   0.3  */
    int32_t max_game_length; /* hive_engine/config.py:22  55 */
    int32_t mode;          /* HiveSearchMode */
    float virtual_loss;    /* solo_play.py:30   1 (n += 1, w -= virtual_loss while a simulation is in flight).  UCT with one
                              slot: 0, which keeps W bit-identical to the reference's sequential sums */
} HiveSearchParams;

int hive_search_create(int games, int max_nodes, int slots, int device, uint64_t seed, HiveSearch **out);
int hive_search_destroy(HiveSearch *s);
int hive_search_set_stream(HiveSearch *s, void *stream);
int hive_search_set_params(HiveSearch *s, const HiveSearchParams *p);

/* Start a new search from these positions (HivePlayer.action -> reset, solo_play.py:110-116).
 * active = int8[games] or NULL: games with active == 0 are skipped by every later call. */
int hive_search_set_roots(HiveSearch *s, const HiveBoard *boards, const HiveHistory *hist, const int8_t *active);

/* Selection for in-flight slot `slot`: leaf_boards / leaf_hist = [games] records of the positions to
 * evaluate (untouched for games whose path ended in a known terminal node or a collision). */
int hive_search_select(HiveSearch *s, int slot, HiveBoard *leaf_boards, HiveHistory *leaf_hist);

/* Expansion + backup for slot `slot`.  leaf_mask = uint32[games][HIVE_MASK_WORDS] legal sets of the leaves (destination boards, hive_abi.h),
 * over / winner as hive_batch_terminal, p = float[games][1584] (softmax output), v = float[games]. */
int hive_search_backup(HiveSearch *s, int slot, const HiveBoard *leaf_boards, const HiveHistory *leaf_hist,
                       const uint32_t *leaf_mask, const int8_t *over, const int8_t *winner, const float *p,
                       const float *v);

/* Which leaves of the last hive_search_select calls (slots 0 .. slots-1, after hive_leaf_launch filled `over`) will
 * hive_search_backup ask the evaluator about?  need = int8[slots * games] (1 = the network's p / v of that row are read;
 * 0 = finished game, length cap, revisited terminal node, collision or idle tree: solo_play.py:169-183 take those values
 * without a prediction, and the reference never calls its model for them -- api_hive.py:62-69 is reached only from
 * solo_play.py:188-197).  leaf_boards / over are the full [slots * games] arrays.  total (may be NULL): uint64 device
 * counter, the number of rows flagged 1 is ADDED to it.  Feed `need` to hive_nn_conv3x3_sel / hive_nn_resblock_sel. */
int hive_search_leaf_need(HiveSearch *s, int slots, const HiveBoard *leaf_boards, const int8_t *over, int8_t *need,
                          uint64_t *total);

/* HivePlayer.calc_policy + apply_temperature (solo_play.py:337-374) at the roots:
 * policy = float[games][1584] visit distribution (may be NULL), action = int32[games] (argmax, -1 = pass),
 * sum_n = int32[games] (may be NULL).  selfplay != 0 adds self_play.py:139-157: for turn <= 6 the
 * move is resampled from (1 - e) * policy + e * Dirichlet(0.5) over the legal moves, e = 0.7 - 0.15 * int(turn+1)/2. */
int hive_search_policy(HiveSearch *s, float *policy, int32_t *action, int32_t *sum_n, int selfplay);

/* The reference's tree is a dict keyed by GamePlay.state_key (solo_play.py:167-197; the key holds the pieces of every
 * cell bottom->top and the side to move, not the turn number or the history): a position reached by two move orders is
 * ONE entry with shared statistics.  merge != 0 (the default) gives the GPU tree the same semantics through a per-game
 * hash table over bytes 0..32 + turn parity of the HiveBoard record; merge == 0 keeps a plain tree.
 * hive_search_transposition_hits: int32[games], descents that continued through such a shared node since create. */
int hive_search_set_transpositions(HiveSearch *s, int merge);
int hive_search_transposition_hits(HiveSearch *s, int32_t *hits);

/* Statistics for tests: nodes allocated per tree (int32[games]). */
int hive_search_node_counts(HiveSearch *s, int32_t *counts);

/* Global game index of every tree (device int64[games]; default: 0 .. games-1).  The Dirichlet / resampling streams of
 * game i are keyed on (seed, ids[i], turn of the root position, simulation number inside the search, slot, edge) and on
 * nothing else, so a self-play game comes out the same whichever batch slot, process or GPU runs it
 * (seed = base + global_game_index; woker/self_play.py:37-75 seeds nothing and is not reproducible). */
int hive_search_set_game_ids(HiveSearch *s, const int64_t *ids);

/* Root statistics as the reference's UCTNode arrays (alpha_zero/MCTS_chess.py:33-35): visits / total_value / priors =
 * float[games][1584] each (any may be NULL), zero outside the root's edges.  PUCT mode: n, w, p of the root entry
 * (solo_play.py:51-66). */
int hive_search_root_stats(HiveSearch *s, float *visits, float *total_value, float *priors);

/* Leaves by kind since create: int32[games][8], every simulation of an active game adds one: [1] root evaluated,
 * [2] new position evaluated (1 + 2 = the network evaluations the search NEEDED), [3] descent ended in a known finished
 * game, [4] collision between in-flight slots, [5] descent ended at the length cap, [6] new position that is a finished
 * game or sits at the length cap (PUCT: its value does not come from the network), [7] new position that another
 * in-flight slot had already created.  [0] unused. */
int hive_search_leaf_histogram(HiveSearch *s, int32_t *hist);

/* The search's own noise generator, exposed for statistical tests: draw d (= 0 .. draws-1) is what the root of game id
 * `first_game + d` would get at (turn, sim 0, slot 0): eta ~ Dirichlet(alpha) over k <= HIVE_EDGE_CAP edges
 * (solo_play.py:322-323, np.random.dirichlet).  out = float[draws][k]; prior = float[k] or NULL:
 * out = (1 - eps) prior + eps eta (solo_play.py:323), or eta itself when prior is NULL. */
int hive_search_sample_noise(uint64_t seed, int64_t first_game, int turn, float alpha, int k, int draws, const float *prior,
                             float eps, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HIVE_SEARCH_H */
