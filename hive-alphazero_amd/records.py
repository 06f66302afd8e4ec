"""Self-play records in the reference's wire format (first "next" row).

woker/self_play.py:100-112,178-193 writes `play_<ts>.json` = list of
`[state[12][12][56], policy[1584], value, [game_len, counter]]`; woker/optimize.py:42-65 loads it
and applies `value * 0.99 ** (game_len - step)`.  The GPU engine keeps, per ply, the packed 56-bit
features of every game (1,152 B instead of the 64 KB float64 planes), the visit policy and the mover;
this module expands finished games into those entries.
"""
import json
import os
from datetime import datetime

import numpy as np

from .config import DISCOUNTED_REWARD


def unpack_features(words, turn, history_planes=None):
    """uint64[144] packed features (+ the raw turn number) -> float64 [12,12,56] planes as
    GamePlay.encode_board returns them.  History planes 36..43 are not part of the packed word; pass
    them (bool [12,12,8]) when the trainer needs them."""
    w = np.asarray(words).astype(np.uint64).reshape(144)
    bits = ((w[:, None] >> np.arange(56, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.float64)
    planes = bits.reshape(12, 12, 56)
    planes[:, :, 31] = float(turn)
    if history_planes is not None:
        planes[:, :, 36:44] = history_planes
    return planes


def history_planes(hist_words, hist_len):
    """uint32[4,2,6] history bitboards of the mover's perspective (+ how many entries are valid) ->
    bool [12,12,8] planes 36..43 (env_hive.py:431-434)."""
    out = np.zeros((12, 12, 8))
    hw = np.asarray(hist_words).astype(np.uint32).reshape(4, 2, 6)
    for age in range(min(int(hist_len), 4)):
        for k in range(2):
            for r in range(6):
                v = int(hw[age, k, r])
                for bit in range(32):
                    if (v >> bit) & 1:
                        out[2 * r + (bit >> 4), bit & 15, 2 * age + k] = 1.0
    return out


def game_entries(plies, value_white):
    """plies: list of (planes [12,12,56], policy [1584], player 'W'/'B') in game order ->
    the reference's data rows (self_play.py:178-191)."""
    counts = {"W": sum(1 for p in plies if p[2] == "W"), "B": sum(1 for p in plies if p[2] == "B")}
    seen = {"W": 0, "B": 0}
    out = []
    for planes, policy, player in plies:
        seen[player] += 1
        value = value_white if player == "W" else -value_white
        if value_white == 0:
            value = -1                      # draw or length cap counts as a loss for both sides
        out.append([np.asarray(planes).tolist(), [float(x) for x in policy], value, [counts[player], seen[player]]])
    return out


def write_game_data_to_file(path, data):    # woker/sl.py:49-60
    with open(path, "wt") as f:
        json.dump(data, f)


def flush_buffer(buffer, datapath="../dataSelf"):       # self_play.py:100-112
    os.makedirs(datapath, exist_ok=True)
    game_id = datetime.now().strftime("%Y%m%d-%H%M%S.%f")
    path = os.path.join(datapath, "play_%s.json" % game_id)
    write_game_data_to_file(path, buffer)
    return path


def load_data(filename):                     # woker/optimize.py:42-65
    with open(filename, "rt") as f:
        data = json.load(f)
    rows = []
    for state, policy, value, game_lens in data:
        game_len, step = game_lens[0], game_lens[1]
        if step != game_len:
            value = value * DISCOUNTED_REWARD ** (game_len - step)
        rows.append([np.array(state), np.array(policy, dtype=np.float32), value])
    return rows
