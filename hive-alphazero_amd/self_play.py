"""self_play_buffer: mirror of the reference's woker/self_play.py::self_play_buffer (:116-193), the
drop-in caller of this package's GamePlay / HivePlayer.  Line 149 of the reference indexes a list
with a list (TypeError on move 1); the working form is woker/self_play_with_train.py:179 and is
what is used here.  One game per call, sequential search -- the reference's own process model; the
throughput path is hive_alphazero_amd.mcts.SelfPlay.
"""
from copy import deepcopy

import numpy as np

from .config import MAX_GAME_LENGTH, PIECE_BLACK, PIECE_WHITE
from .solo_play import HivePlayer


def self_play_buffer(cur, make_env=None, simulations=None):
    """cur: list of pipe lists (one popped per game, self_play.py:118).  Returns (data, [value_white])."""
    if make_env is None:
        from .env_hive import GamePlay
        board = GamePlay(HEIGHT_MAP=1050, WIDTH_MAP=900)
    else:
        board = make_env()
    pipes = cur.pop()
    white, black = HivePlayer(pipes=pipes), HivePlayer(pipes=pipes)
    if simulations is not None:
        white.simulation_num_per_move = black.simulation_num_per_move = simulations
    state_policy_player = []
    black_count = white_count = 0
    e = 0.7
    while not board.game_is_over():
        if board.state.player() == 0:
            action, policy = white.action(board)
            player = "W"
            white_count += 1
            counter = white_count
        else:
            action, policy = black.action(board)
            player = "B"
            black_count += 1
            counter = black_count
        if board.state.turn <= 2:
            action = np.random.choice(board.actions())
        policy = policy[0]
        error = e - int(board.state.turn + 1) / 2 * 0.15
        actions = board.actions()
        if error >= 0.1 and len(actions) != 0:
            p = np.array(policy)[actions]
            noise = np.random.dirichlet([0.5] * len(actions))
            p = (1 - error) * np.array(p) + error * noise
            p /= p.sum()
            action = np.random.choice(board.actions(), p=p)
        state = board.encode_board(player)
        state_policy_player.append([state.tolist(), policy, player, counter])
        board.move(int(action))
        if board.state.turn >= MAX_GAME_LENGTH:
            break

    value_white = 0
    if board.game_is_over():
        if board.state.winner == PIECE_WHITE:
            value_white = 1
        elif board.state.winner == PIECE_BLACK:
            value_white = -1
    white.finish_game(value_white)
    black.finish_game(-value_white)

    data = []
    for state, policy, player, counter in state_policy_player:
        value = value_white if player == "W" else -value_white
        game_lens = white_count if player == "W" else black_count
        if value_white == 0:
            value = -1
        data.append([state, policy, value, [game_lens, counter]])
    cur.append(pipes)
    return data, [value_white]
