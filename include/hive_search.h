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

typedef struct HiveSearchParams {
    float c_puct;          /* solo_play.py:26   0.7  */
    float noise_eps;       /* solo_play.py:29   0.25 */
    float dirichlet_alpha; /* solo_play.py:28   0.3  */
    int32_t max_game_length; /* hive_engine/config.py:22  55 */
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

#ifdef __cplusplus
}
#endif
#endif /* HIVE_SEARCH_H */
