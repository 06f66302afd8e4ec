"""Drivers for the caller goldens -- TEST INFRASTRUCTURE (never imported by the product).

`tests/golden/selfplay.json.gz` and `tests/golden/sl.json.gz` were written by the TRUE reference's callers
(woker/self_play_with_train.py::self_play_buffer, woker/sl.py::get_buffer; see oracle/gen_golden.py).  The product
no longer carries copies of those callers (the reference's own files run over this package by an
import swap; the GPU-native equivalents are mcts.SelfPlay and records.replay_recorded_games).  These two small
drivers reproduce what the goldens record -- rows of [planes, policy, value, [game_len, counter]] -- by stepping an
env / a HivePlayer the way the recorded runs did, so the goldens keep pinning the façade and the sequential search.
"""
import numpy as np

WHITE_RGB, BLACK_RGB = (250, 250, 250), (71, 71, 71)       # settings.py:3-4
LENGTH_CAP = 55                                             # hive_engine/config.py:22
PIECE_CODES = ("Q", "B1", "B2", "S1", "S2", "G1", "G2", "G3", "A1", "A2", "A3")     # woker/sl.py:33-46, slot order
COLS = "HIJKLMNOPQRS"                                       # hive_engine/config.py index_char
ROWS = [str(i) for i in range(7, 19)]                       # hive_engine/config.py index_number


def _white_result(env):
    if not env.game_is_over():
        return 0
    return {WHITE_RGB: 1, BLACK_RGB: -1}.get(env.state.winner, 0)


def _rows(log, white_result, draw_value):
    """log entries (planes, policy, side, nth move of that side) -> wire rows (self_play.py:178-191 / sl.py:213-229)."""
    per_side = {s: sum(1 for e in log if e[2] == s) for s in "WB"}
    out = []
    for planes, policy, side, nth in log:
        value = draw_value if white_result == 0 else (white_result if side == "W" else -white_result)
        out.append([planes, policy, value, [per_side[side], nth]])
    return out


def selfplay_game(env, searcher_for, sims=None):
    """One game of the recorded self-play run: `searcher_for(side)` returns the HivePlayer of side 0 / 1.

    Consumes numpy's global stream in the recorded order -- the search, then (turns 1-2) a uniform pick, then
    (while 0.7 - 0.15 * (turn + 1)/2 >= 0.1) a Dirichlet(0.5) mix and a weighted pick."""
    log, moves_made = [], [0, 0]
    while not env.game_is_over():
        side = env.state.player()
        player = searcher_for(side)
        if sims is not None:
            player.simulation_num_per_move = sims
        chosen, (policy, _) = player.action(env)
        moves_made[side] += 1
        legal = env.actions()
        turn = env.state.turn
        if turn <= 2:
            chosen = np.random.choice(legal)
        eps = 0.7 - int(turn + 1) / 2 * 0.15
        if eps >= 0.1 and len(legal):
            mix = (1 - eps) * np.array(policy)[legal] + eps * np.random.dirichlet([0.5] * len(legal))
            chosen = np.random.choice(legal, p=mix / mix.sum())
        tag = "WB"[side]
        log.append((env.encode_board(tag).tolist(), policy, tag, moves_made[side]))
        env.move(int(chosen))
        if env.state.turn >= LENGTH_CAP:
            break
    result = _white_result(env)
    for side in (0, 1):
        searcher_for(side).finish_game(result if side == 0 else -result)
    return _rows(log, result, draw_value=-1), result


def recorded_step_action(step):
    """[piece code, column letter, row number, 'W'/'B', bot flag] -> action id cell * 11 + slot."""
    code, col, row = step[0], step[1], step[2]
    return (COLS.index(col) * 12 + ROWS.index(row)) * 11 + PIECE_CODES.index(code)


def replay_recorded_game(env, steps, bot_weight=0.24):
    """The supervised-learning ingest of one recorded game through a GamePlay-like env, one step at a time."""
    log, moves_made = [], {"W": 0, "B": 0}
    for step in steps:
        side, bot = step[3], step[4]
        if "WB"[env.player()] != side:
            env.skip_turn()
        moves_made[side] += 1
        a = recorded_step_action(step)
        if a not in env.actions():
            log = []
            break
        target = np.zeros(1584)
        target[a] = bot_weight if bot == 1 else 1
        log.append((env.encode_board(side).tolist(), target.tolist(), side, moves_made[side]))
        env.move(a, with_skip=False)
    result = _white_result(env)
    return _rows(log, result, draw_value=0)
