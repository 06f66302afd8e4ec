"""get_buffer: mirror of the reference's woker/sl.py::get_buffer (:146-231), the supervised-learning
ingest that replays recorded human/bot games through the env (fourth "next" row).

A recorded game is a list of steps [piece code, column letter, row number, player 'W'/'B', bot flag]
(codes "Q", "B1".."B2", "S1".."S2", "G1".."G3", "A1".."A3"; woker/sl.py:33-46).  Out-of-turn steps
call skip_turn(), the policy target is one-hot (BOT_WEIGHT for bot moves) and the value is the final
result from the mover's side, 0 for unfinished games (unlike self-play's -1)."""
import numpy as np

from .config import BOT_WEIGHT, MAX_MAP_FULL, PIECE_BLACK, PIECE_WHITE, index_char, index_number

_TYPE_KEY = {"G": "<class 'pieces.Grasshopper'>", "B": "<class 'pieces.Beetle'>", "S": "<class 'pieces.Spider'>",
             "A": "<class 'pieces.Ant'>"}


def decode_piece(piece):                     # woker/sl.py:33-43
    if piece == "Q":
        return "<class 'pieces.Queen'>0"
    return _TYPE_KEY[piece[0]] + str(int(piece[1]) - 1)


def get_buffer(game, make_env=None):
    if make_env is None:
        from .env_hive import GamePlay
        board = GamePlay(HEIGHT_MAP=1050, WIDTH_MAP=900)
    else:
        board = make_env()
    state_policy_player = []
    black_count = white_count = 0
    for current_step in game:
        piece, x, y, player, bot = current_step[0], current_step[1], current_step[2], current_step[3], current_step[4]
        if (board.player() == 1 and player == "W") or (board.player() == 0 and player == "B"):
            board.skip_turn()
        if player == "W":
            white_count += 1
            counter = white_count
        else:
            black_count += 1
            counter = black_count
        yi = index_number.index(y)
        xi = index_char.index(x)
        end_tile = board.board_matrix[xi, yi]
        action = board.encode_action({decode_piece(piece): [end_tile]})
        if action[0] not in board.actions():
            state_policy_player = []
            break
        policy = np.zeros(MAX_MAP_FULL * MAX_MAP_FULL * 11)
        policy[action] = BOT_WEIGHT if bot == 1 else 1
        state = board.encode_board(player)
        state_policy_player.append([state.tolist(), policy, player, counter])
        board.move(action[0], with_skip=False)

    value_white = 0
    if board.game_is_over():
        if board.state.winner == PIECE_WHITE:
            value_white = 1
        elif board.state.winner == PIECE_BLACK:
            value_white = -1
    data = []
    for state, policy, player, counter in state_policy_player:
        value = value_white if player == "W" else value_white * -1
        game_lens = white_count if player == "W" else black_count
        if value_white == 0:
            value = 0
        data.append([state, policy.tolist(), value, [game_lens, counter]])
    return data, game
