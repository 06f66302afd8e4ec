/*
 * hive_abi.h -- C ABI of libhive_hip.so, the MI355X (gfx950) Hive env engine.
 *
 * The reference (HaiDangDang/hive-Alphazero) is pure Python and has no FFI layer; its
 * boundary is the duck-typed GamePlay object (hive_engine/env_hive.py:24-507).  Each
 * entry point below names the reference interface it replaces (paths relative to the
 * reference root).  The Python mirror of GamePlay that a reference caller
 * (woker/self_play.py, woker/solo_play.py, alpha_zero/MCTS_chess.py) imports instead is
 * hive-alphazero_amd/env_hive.py; INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - cell   = index_xy[0]*12 + index_xy[1]               (tile.py:190-196)
 *   - piece  = colour*11 + slot; colour 0 = white; slot order Q,B,B,S,S,G,G,G,A,A,A
 *              (inventory_frame.py:47-99, env_hive.py:66-87)
 *   - action = cell*11 + slot, pass = -1                   (env_hive.py:100-114,287-304)
 *   - All array arguments are DEVICE pointers (HBM) unless a parameter says "host".
 *     The caller allocates every output buffer; the library owns only the handle.
 *   - Every call returns 0 on success or a negative HIVE_E_* code; hive_last_error()
 *     gives the message for the calling thread.  Nothing throws across the ABI.
 *   - A handle is not thread-safe; distinct handles are independent.  All work of a
 *     handle is enqueued on its HIP stream (hive_batch_set_stream); calls are
 *     asynchronous unless stated otherwise.
 *   - Positions must be reachable by legal play (connected hive); the one-hive test
 *     relies on that exactly like the reference's BFS does.
 */
#ifndef HIVE_ABI_H
#define HIVE_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIVE_CELLS 144
#define HIVE_PIECES 22
#define HIVE_ACTIONS 1584          /* hive_engine/config.py:10 ACTION_SPACE */
#define HIVE_PLANES 56             /* hive_engine/config.py:20 STATE_FEATURES */
#define HIVE_SLOTS 11               /* piece slots per colour: Q B B S S G G G A A A (inventory_frame.py:47-99) */
#define HIVE_MASK_WORDS 66         /* legal set = 11 slots x one 6-word destination board (see hive_batch_legal) */
#define HIVE_IN_HAND 255
#define HIVE_LIST_CAP 256          /* capacity of one compacted legal-id row */

/* where action id a = (row*12 + col)*11 + slot sits inside one board's legal mask (see hive_batch_legal) */
#define HIVE_MASK_WORD(a) (((a) % HIVE_SLOTS) * 6 + ((((a) / HIVE_SLOTS) / 12) >> 1))
#define HIVE_MASK_BIT(a) (((((a) / HIVE_SLOTS) / 12 & 1) << 4) | (((a) / HIVE_SLOTS) % 12))
#define HIVE_MASK_TEST(mask_row, a) (((mask_row)[HIVE_MASK_WORD(a)] >> HIVE_MASK_BIT(a)) & 1u)

enum {
    HIVE_OK = 0,
    HIVE_E_ARG = -1,        /* bad argument (null pointer, n <= 0, unknown enum) */
    HIVE_E_DEVICE = -2,     /* HIP runtime error (no device, launch failure, OOM) */
    HIVE_E_ILLEGAL = -3,    /* at least one action was not in the legal set; those boards are unchanged */
    HIVE_E_STATE = -4       /* imported position is malformed */
};

/* One board in HBM: 64 bytes, 16-byte aligned.  Replaces the GamePlay object graph
 * (env_hive.py:26-57: white/black_pieces_set, state.turn, next_move_tiles). */
typedef struct HiveBoard {
    uint8_t pos[HIVE_PIECES];   /* cell of each piece, HIVE_IN_HAND if not placed */
    uint8_t lvl[11];            /* 4-bit stack index per piece: piece 2k = low nibble of lvl[k] */
    uint8_t turn;               /* game_state.py:38, starts at 1 */
    uint8_t flags;              /* bits 0-1 nmt_mode (0 rebuilt from occupancy, 1 start tile only,
                                   2 turn-2 tile only: env_hive.py:66-69,150-161);
                                   bit 2: the encode of this position inserts it into the
                                   history (env_hive.py:51,146,436-445) */
    uint8_t hist_len;           /* low nibble: entries visible to white's planes, high: black's */
    uint8_t rsv[28];
} HiveBoard;

/* History bit-planes of one board: [perspective][age, 0 = newest][0 = own, 1 = enemy][6 words].
 * Word r holds board rows 2r (bits 0-11) and 2r+1 (bits 16-27).  384 bytes.
 * Replaces GamePlay.history_white / history_black (env_hive.py:38-39,431-445). */
typedef struct HiveHistory {
    uint32_t m[2][4][2][6];
} HiveHistory;

typedef enum { HIVE_F32 = 0, HIVE_F16 = 1, HIVE_BF16 = 2 } HiveDType;
typedef enum {
    HIVE_HWC = 0,   /* [n][12][12][56], the reference's encode_board layout (channels-last) */
    HIVE_CHW = 1    /* [n][56][12][12], what api_hive.py:60 builds with transpose(2,0,1) */
} HiveLayout;

typedef struct HiveBatch HiveBatch;

const char *hive_last_error(void);
const char *hive_version(void);

/* Number of visible HIP devices (0 when none); never fails. */
int hive_device_count(void);

/* GamePlay.__init__/new_game for n boards (env_hive.py:26-97).  device = HIP ordinal. */
int hive_batch_create(int n, int device, HiveBatch **out);
int hive_batch_destroy(HiveBatch *h);
int hive_batch_size(const HiveBatch *h);
/* stream = hipStream_t (NULL = default stream). */
int hive_batch_set_stream(HiveBatch *h, void *stream);

/* GamePlay.new_game (env_hive.py:61-97) for the boards listed in idx (device int32[k]);
 * idx == NULL resets all n boards. */
int hive_batch_reset(HiveBatch *h, const int32_t *idx, int k);

/* GamePlay.move (env_hive.py:99-171) for every board: actions = device int32[n], -1 = pass
 * (also GamePlay.skip_turn, env_hive.py:493-496), -2 = leave this board untouched.
 * The reference applies illegal actions blindly (its assert is commented out,
 * env_hive.py:129-144); here a board whose action is not legal is left unchanged and
 * counted.  With sync != 0 the call waits and returns HIVE_E_ILLEGAL if any board
 * refused; with sync == 0 read the running total with hive_batch_illegal_count. */
int hive_batch_step(HiveBatch *h, const int32_t *actions, int sync);
int hive_batch_illegal_count(HiveBatch *h, int64_t *count /* host */);

/* GamePlay.actions (env_hive.py:182-183, 196-304): legal set of the side to move.
 * mask  = device uint32[n][HIVE_MASK_WORDS]: the legal set in the form GamePlay.pre_actions builds it
 *         (env_hive.py:196-285, {piece slot: [destination tiles]}): words 6s .. 6s+5 are the destination
 *         board of the mover's piece slot s -- word r holds rows 2r (bits 0-11, bit = column) and 2r+1
 *         (bits 16-27), the same bitboard words HiveHistory uses; action a = cell*11 + s is legal iff
 *         HIVE_MASK_TEST(row, a) (may be NULL);
 * count = device int32[n] (may be NULL);
 * list  = device int16[n][HIVE_LIST_CAP]: GamePlay.encode_action (env_hive.py:287-304) of that set, i.e. the
 *         ascending action ids GamePlay.actions() returns, -1 padded (may be NULL; in the stateless launch it
 *         needs mask != NULL); a board with more than HIVE_LIST_CAP legal ids keeps the first HIVE_LIST_CAP. */
int hive_batch_legal(HiveBatch *h, uint32_t *mask, int32_t *count, int16_t *list);

/* GamePlay.encode_board (env_hive.py:306-485): the 56 planes from the mover's side. */
int hive_batch_encode(HiveBatch *h, void *planes, HiveDType dtype, HiveLayout layout);

/* GamePlay.game_is_over (move_checker.py:140-165): over[n] int8 0/1,
 * winner[n] int8 0 none / 1 white / 2 black. */
int hive_batch_terminal(HiveBatch *h, int8_t *over, int8_t *winner);

/* Raw state exchange (deepcopy(env) in solo_play.py:158 / MCTS_chess.py:104):
 * boards = device HiveBoard[n], hist = device HiveHistory[n] (hist may be NULL). */
int hive_batch_export(HiveBatch *h, HiveBoard *boards, HiveHistory *hist);
int hive_batch_import(HiveBatch *h, const HiveBoard *boards, const HiveHistory *hist);

/* Stateless launches over caller-owned state arrays (used by the tree search, where
 * positions live in node pools).  Same semantics as the batch calls above. */
int hive_movegen_launch(const HiveBoard *boards, int n, uint32_t *mask, int32_t *count,
                        int16_t *list, void *stream);
/* (list, when given, must be 8-byte aligned -- its rows of HIVE_LIST_CAP ids leave the kernel as 8-byte stores; a
 * misaligned pointer is refused with HIVE_E_ARG.  With list != NULL the mask, the counts and the sorted id lists are
 * produced by ONE launch.) */
/* Launches without an id list of at least this many boards run the kernel's pair layout (one board = two lanes, 32
 * boards per workgroup: fewer instructions per board once the card is full either way; same bits).  boards > 0 sets the
 * threshold (INT_MAX: never), 0 restores the default (16384), < 0 only asks; returns the previous value.  Process-wide;
 * meant for tests and measurements, not for the hot path. */
int hive_movegen_pair_threshold(int boards);
/* workspace = device scratch of n * HIVE_CELLS * 8 bytes (the packed 56-bit-per-cell features). */
int hive_encode_launch(const HiveBoard *boards, const HiveHistory *hist, int n, void *planes,
                       HiveDType dtype, HiveLayout layout, void *workspace, void *stream);

/* The second half of hive_encode_launch on its own: packed features (uint64[n][144], bit p of cell c = plane p, as the
 * workspace of hive_encode_launch / the self-play records hold them) + history + the records' turn / history-length bytes
 * -> the plane tensor.  Turns stored self-play records back into GamePlay.encode_board planes (env_hive.py:306-447) on the
 * GPU; of `boards` only bytes 33 (turn) and 35 (history lengths) are read. */
int hive_expand_launch(const HiveBoard *boards, const HiveHistory *hist, const void *features, int n, void *planes,
                       HiveDType dtype, HiveLayout layout, void *stream);

/* GamePlay.game_is_over over caller-owned records. */
int hive_terminal_launch(const HiveBoard *boards, int n, int8_t *over, int8_t *winner, void *stream);
/* GamePlay.move over caller-owned records: actions as hive_batch_step; legal_mask (uint32[n][HIVE_MASK_WORDS]) may be
 * NULL when the actions are known to be legal (they come out of the search's own edge lists). */
int hive_step_launch(HiveBoard *boards, HiveHistory *hist, int n, const int32_t *actions, const uint32_t *legal_mask,
                     void *stream);
/* The same with the batch form's contract on refused moves (hive_batch_step / hive_batch_illegal_count): an action that
 * is not in its board's legal_mask row leaves that board unchanged AND adds one to *illegal_count (a device counter the
 * caller owns and zeroes; it accumulates over calls).  hive_step_launch refuses such actions silently; with
 * legal_mask == NULL nothing is tested and illegal_count must be NULL too.  (The reference applies illegal moves
 * blindly: its `assert check` is commented out, env_hive.py:129-144.) */
int hive_step_launch_counted(HiveBoard *boards, HiveHistory *hist, int n, const int32_t *actions, const uint32_t *legal_mask,
                             unsigned long long *illegal_count, void *stream);
/* Everything the tree search needs about n leaf positions in one call: planes (as hive_encode_launch),
 * legal mask / count (may be NULL) and game-over flags (may be NULL). */
int hive_leaf_launch(const HiveBoard *boards, const HiveHistory *hist, int n, void *planes, HiveDType dtype,
                     HiveLayout layout, void *workspace, uint32_t *mask, int32_t *count, int8_t *over, int8_t *winner,
                     void *stream);

/* Equal leaves of one batch (n <= 4096 rows): a row with need != 0 whose (HiveBoard, HiveHistory) bytes equal those of
 * an EARLIER needed row gets need = 0 and rep = that row's index; every other row gets rep = its own index.  The planes --
 * and so the network's answer -- are a function of those 448 bytes alone, so the evaluator runs the representative only
 * and the duplicates take its result (hive_nn_copy_rows).  The reference evaluates one position at a time
 * (api_hive.py:62-69 behind solo_play.py:188-197); a thousand lock-step games from the opening ask about the same few
 * positions.  keys = uint64[n] scratch; total (may be NULL): the number of rows switched off is SUBTRACTED from it
 * (the counter hive_search_leaf_need added to). */
int hive_leaf_dedup_launch(const HiveBoard *boards, const HiveHistory *hist, int n, int8_t *need, int32_t *rep,
                           uint64_t *keys, uint64_t *total, void *stream);

/* ---- A store of leaf evaluations that outlives a search.  The reference empties its tree on every move
 * (woker/solo_play.py:103-112), so the positions under the move just played -- and the openings of every later game -- are
 * evaluated again; a leaf's (p, v) is a pure function of its HiveBoard + HiveHistory, and this engine produces the same
 * bits for a position in any batch at any row, so a stored evaluation can stand in for a fresh one without changing the
 * search.  `capacity` entries (448 bytes of key material + 1585 floats each) in a ring: the oldest are overwritten.
 *   lookup  (after hive_leaf_dedup_launch, whose `keys` it takes): rows still flagged in need[] whose position is stored --
 *           64-bit key, then all 448 bytes compared -- are switched off; hit[row] = the slot, else -1; the number of rows
 *           switched off is subtracted from *total (may be NULL)
 *   update  (after the forward): row i takes (p, v) of slot hit[rep[i]] if that is >= 0 (rep may be NULL: identity); every
 *           row that was evaluated (need[i] still 1) is inserted.  p f32 [n][1584], v f32 [n] are the evaluator's outputs.
 * Clear the store whenever the network's weights change.  Not thread-safe; one stream at a time. */
typedef struct HiveLeafStore HiveLeafStore;
int hive_leaf_store_create(int device, int capacity, HiveLeafStore **out);
int hive_leaf_store_destroy(HiveLeafStore *s);
int hive_leaf_store_clear(HiveLeafStore *s, void *stream);
int hive_leaf_store_lookup(HiveLeafStore *s, const HiveBoard *boards, const HiveHistory *hist, int n, const uint64_t *keys,
                           int8_t *need, int32_t *hit, uint64_t *total, void *stream);
int hive_leaf_store_update(HiveLeafStore *s, const HiveBoard *boards, const HiveHistory *hist, int n, const uint64_t *keys,
                           const int8_t *need, const int32_t *rep, const int32_t *hit, float *p, float *v, void *stream);
int hive_leaf_store_stats(HiveLeafStore *s, uint64_t *served, uint64_t *inserted);   /* synchronises */

/* ---- One position, HOST buffers: the single-game surface of GamePlay without a round trip per question.
 * A HiveSingle owns a stream, a device block and a pinned host mirror; a call copies the position in, runs its kernels
 * as one launch chain, copies the answers out and synchronises once.  Not thread-safe; distinct handles are independent
 * (hive-alphazero_amd/env_hive.py lends one handle to each concurrent GamePlay call).  All pointers are HOST pointers. */
typedef struct HiveSingle HiveSingle;
int hive_single_create(int device, HiveSingle **out);
int hive_single_destroy(HiveSingle *h);

/* GamePlay.move / skip_turn / new_game + what they recompute (env_hive.py:61-97,99-171,493-496,196-304; game_is_over,
 * move_checker.py:140-165).  action >= 0: move; -1: pass / skip_turn; -2: no move (only the outputs of `rec`); -3: new
 * game (rec / hist ignored).  legal_before = the legal set of `rec` as an earlier call returned it (uint32[HIVE_MASK_WORDS],
 * may be NULL: it is then derived first); an action outside it is refused: HIVE_E_ILLEGAL, outputs untouched.
 * Outputs: the new record and history, its legal set (destination boards as hive_batch_legal), count, over, winner
 * (count / over / winner may be NULL). */
int hive_single_advance(HiveSingle *h, const HiveBoard *rec, const HiveHistory *hist, int action, const uint32_t *legal_before,
                        HiveBoard *rec_out, HiveHistory *hist_out, uint32_t *legal_after, int32_t *count, int8_t *over,
                        int8_t *winner);

/* GamePlay.encode_board (env_hive.py:306-485): planes = float[12][12][56] (HWC, the reference's layout) of the mover's side. */
int hive_single_encode(HiveSingle *h, const HiveBoard *rec, const HiveHistory *hist, float *planes);

#ifdef __cplusplus
}
#endif
#endif /* HIVE_ABI_H */
