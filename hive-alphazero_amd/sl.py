"""Supervised-learning ingest: recorded human/bot games -> training rows, many games in lock step.

The reference replays ONE recorded game per call through a Python `GamePlay` (woker/sl.py::get_buffer, :146-231):
skip the turn when the recorded mover is not the side to move, look the step up in the legal list, encode the
planes, move.  Here a whole set of games advances together on the GPU: the replay schedule (which game skips,
which action it plays) is known on the host up front -- the turn parity only depends on how many steps and skips
came before -- so one ingest round is three batched kernel launches for all games (skip step, legal mask + planes,
move), and the only device -> host traffic per round is the planes and one legality bit per game.

A recorded step is [piece code, column letter, row number, 'W' / 'B', bot flag] with codes Q, B1-B2, S1-S2, G1-G3,
A1-A3 (woker/sl.py:33-46).  Rows are the reference's: [planes[12][12][56], one-hot policy (BOT_WEIGHT for a bot's
move, :193-198), value from the mover's side (0 for an unfinished or drawn game, :226-227), [game_len, counter]];
a game with a step that is not legal yields no rows (:187-190).
"""
import numpy as np
import torch

from ._lib import HIVE_MASK_WORDS
from .batch import BoardBatch
from .config import ACTION_SPACE, BOT_WEIGHT, index_char, index_number

_CODES = ("Q", "B1", "B2", "S1", "S2", "G1", "G2", "G3", "A1", "A2", "A3")       # slot order of inventory_frame.py:47-99
_SLOT_KEY_OF_CODE = {"Q": "<class 'pieces.Queen'>", "B": "<class 'pieces.Beetle'>", "S": "<class 'pieces.Spider'>",
                     "G": "<class 'pieces.Grasshopper'>", "A": "<class 'pieces.Ant'>"}


def decode_piece(code):
    """woker/sl.py:33-43: 'G3' -> "<class 'pieces.Grasshopper'>2" (the key GamePlay.white_pieces_set uses)."""
    return _SLOT_KEY_OF_CODE[code[0]] + str(int(code[1:]) - 1 if len(code) > 1 else 0)


def step_action(step):
    """Recorded step -> action id cell * 11 + slot (env_hive.py:287-304 for a single {piece: [tile]})."""
    cell = index_char.index(step[1]) * 12 + index_number.index(step[2])
    return cell * 11 + _CODES.index(step[0])


def _schedule(game):
    """Host-side plan of one recorded game: per step (skip first?, action id, side, bot flag, nth move of the side)."""
    plan, to_move, made = [], 0, [0, 0]
    for step in game:
        side = 0 if step[3] == "W" else 1
        skip = side != to_move
        made[side] += 1
        plan.append((skip, step_action(step), side, step[4] == 1, made[side]))
        to_move = 1 - side
    return plan


def get_buffers(games, device=None, plane_dtype=torch.float16):
    """Rows of every recorded game in `games` (list of step lists) -> list of row lists, same order."""
    G = len(games)
    if G == 0:
        return []
    plans = [_schedule(g) for g in games]
    env = BoardBatch(G, device)
    dev = env.device
    live = [True] * G                       # still replaying and every step so far was legal
    logs = [[] for _ in range(G)]           # (planes, action, side, bot, nth)
    try:
        for r in range(max(len(p) for p in plans)):
            rows = [g for g in range(G) if live[g] and r < len(plans[g])]
            if not rows:
                break
            skip = np.full(G, -2, dtype=np.int32)
            act = np.full(G, -2, dtype=np.int32)
            for g in rows:
                if plans[g][r][0]:
                    skip[g] = -1
                act[g] = plans[g][r][1]
            if (skip == -1).any():
                env.step(skip, sync=False)                              # GamePlay.skip_turn for those games
            mask, _, _ = env.legal()
            a = torch.from_numpy(np.where(act >= 0, act, 0)).to(dev).long()
            cell, slot = a // 11, a % 11
            word = slot * 6 + (cell // 12) // 2                         # HIVE_MASK_WORD / HIVE_MASK_BIT (hive_abi.h)
            bit = (((cell // 12) & 1) << 4) | (cell % 12)
            hit = (mask.view(G, HIVE_MASK_WORDS).gather(1, word.view(-1, 1)).view(-1) >> bit.to(torch.int32)) & 1
            planes = env.encode(plane_dtype, "hwc")
            idx = torch.tensor(rows, device=dev)
            legal = hit[idx].cpu().numpy().astype(bool)
            host_planes = planes[idx].cpu().numpy()
            for k, g in enumerate(rows):
                if not legal[k]:
                    live[g], logs[g], act[g] = False, [], -2            # sl.py:187-190
                else:
                    _, action, side, bot, nth = plans[g][r]
                    logs[g].append((host_planes[k], action, side, bot, nth))
            env.step(act, sync=False)
        over, winner = env.terminal()
        over, winner = over.cpu().numpy(), winner.cpu().numpy()
        bad = env.illegal_count()
        assert bad == 0, f"{bad} recorded steps passed the legal mask but were refused by the env"
    finally:
        env.close()
    out = []
    for g in range(G):
        white = 0 if not over[g] else (1 if winner[g] == 1 else (-1 if winner[g] == 2 else 0))
        total = [sum(1 for e in logs[g] if e[2] == s) for s in (0, 1)]
        rows = []
        for planes, action, side, bot, nth in logs[g]:
            policy = np.zeros(ACTION_SPACE)
            policy[action] = BOT_WEIGHT if bot else 1
            value = 0 if white == 0 else (white if side == 0 else -white)
            rows.append([planes.astype(np.float64).tolist(), policy.tolist(), value, [total[side], nth]])
        out.append(rows)
    return out


def get_buffer(game, device=None):
    """woker/sl.py::get_buffer's signature for one recorded game: -> (rows, game)."""
    return get_buffers([game], device)[0], game
