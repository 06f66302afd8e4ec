"""Random playouts on the GPU (product path): used to build synthetic position corpora
 and as a simple end-to-end exercise of step + movegen + terminal."""
import torch

from .batch import BoardBatch

MAX_GAME_LENGTH = 55    # hive_engine/config.py:22 (reference)


def pick_uniform(count, lst, gen):
    """One uniformly random legal action per board from the compacted id lists; -1 (pass) when empty."""
    n = count.shape[0]
    r = torch.rand((n,), device=count.device, generator=gen)
    idx = torch.clamp((r * count.float()).long(), max=255)
    idx = torch.minimum(idx, torch.clamp(count.long() - 1, min=0))
    a = lst.long().gather(1, idx.view(-1, 1)).view(-1)
    return torch.where(count > 0, a, torch.full_like(a, -1)).to(torch.int32)


def random_positions(n, seed=0, device=None, games=None):
    """n HiveBoard records (uint8 [n,64], on the GPU) sampled at every ply of uniformly random
    playouts; plies are ~uniform over turns 1..54."""
    games = games or max(8, (n + 49) // 50)
    B = BoardBatch(games, device)
    gen = torch.Generator(device=B.device)
    gen.manual_seed(seed)
    out = []
    total = 0
    alive = torch.ones((games,), dtype=torch.bool, device=B.device)
    while total < n:
        boards, _ = B.export_state()
        over, _ = B.terminal()
        turn = boards[:, 33]
        out.append(boards[alive])
        total += int(alive.sum().item())
        alive = alive & (over == 0) & (turn < MAX_GAME_LENGTH - 1)
        if not bool(alive.any().item()):
            B.reset()
            alive[:] = True
            continue
        _, count, lst = B.legal(want_list=True)
        a = pick_uniform(count, lst, gen)
        a = torch.where(alive, a, torch.full_like(a, -2))
        B.step(a, sync=False)
    B.close()
    return torch.cat(out, 0)[:n].contiguous()
