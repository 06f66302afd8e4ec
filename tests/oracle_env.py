"""GamePlay-API adapter over the CPU oracle -- TEST INFRASTRUCTURE (BASELINE config C1: the
"CPU env_hive path" used to exercise host-side search logic without a GPU)."""
import copy

import numpy as np

from oracle import oracle_py as O

PIECE_WHITE = (250, 250, 250)
PIECE_BLACK = (71, 71, 71)


class _State:
    def __init__(self, g):
        self._g = g
        self.winner = None

    @property
    def turn(self):
        return self._g.turn

    def player(self):
        return 0 if self.turn % 2 == 1 else 1


class OracleGamePlay:
    def __init__(self, HEIGHT_MAP=None, WIDTH_MAP=None):
        self._g = O.OracleGame()
        self.state = _State(self._g)
        self.state_key = self._g.state_key()

    def game_is_over(self):
        over, w = self._g.game_is_over()
        if w == 1:
            self.state.winner = PIECE_WHITE
        elif w == 2:
            self.state.winner = PIECE_BLACK
        return over

    def move(self, a, with_skip=False):
        self._g.move(int(a))
        if int(a) == -1:
            self.state_key = self.state_key[:-1] + str(self.state.player())
        else:
            self.state_key = self._g.state_key()

    def actions(self):
        return self._g.actions()

    def encode_board(self, player="N"):
        return self._g.encode_board().astype(np.float64)

    def turn(self):
        return self._g.turn

    def player(self):
        return self.state.player()

    def decode_action(self, a):
        return divmod(int(a), 11)

    def skip_turn(self):                 # env_hive.py:493-496 (state_key is not updated there)
        self._g.skip_turn()

    def __deepcopy__(self, memo):        # solo_play.py:158 / MCTS_chess.py:104 copy the env once per simulation / child
        c = OracleGamePlay.__new__(OracleGamePlay)
        c._g = self._g.copy()
        c.state = _State(c._g)
        c.state.winner = self.state.winner
        c.state_key = self.state_key
        return c

    class _Tile:
        def __init__(self, x, y):
            self.index_xy = [x, y]

    board_matrix = np.array([[None] * 12 for _ in range(12)], dtype=object)

    def encode_action(self, action_list):
        from hive_alphazero_amd.config import SLOT_KEYS
        ids = set()
        for piece, tiles in action_list.items():
            for t in tiles:
                ids.add((t.index_xy[0] * 12 + t.index_xy[1]) * 11 + SLOT_KEYS.index(piece))
        return sorted(ids)


for _x in range(12):
    for _y in range(12):
        OracleGamePlay.board_matrix[_x, _y] = OracleGamePlay._Tile(_x, _y)
