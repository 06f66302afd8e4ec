"""Whole-game self-play (1024 games x 50 sims from the opening, engine only) with the leaf-batch economies switched off
and on, in ONE process on one box (boxes differ by 2-5 %): every row evaluated / unread rows skipped / + equal leaves shared."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_alphazero_amd import mcts
from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
torch.manual_seed(0)
inf = InferenceNet(ChessNet().cuda().eval(), dtype=torch.bfloat16)
G, SIMS = 1024, 50
forms = (("every row", dict(skip_unread_rows=False)), ("skip unread", dict(share_equal_leaves=False)), ("skip + share equal", dict()))
warm = mcts.SelfPlay(G, SIMS, inf, seed=1234, keep_records=False); warm.play_ply(); torch.cuda.synchronize(); warm.close()
for rnd in range(2):
    for name, opt in forms:
        sp = mcts.SelfPlay(G, SIMS, inf, seed=1234, keep_records=False, game_ids=range(G), search_options=opt)
        torch.cuda.synchronize(); t0 = time.perf_counter(); plies = 0
        while True:
            sp.play_ply(); plies += 1
            if sp.running() == 0 or plies > 60: break
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        rows = int(sp.search.evals_run.item()) if sp.search.skip_unread_rows else sp.search.evals_launched
        print(f"{name:20s} {sp.finished / el * 60:8.1f} games/min  {el:6.2f} s  rows evaluated {rows} of {sp.search.evals_launched}  "
              f"results W{sp.white_wins} B{sp.black_wins} D{sp.draws}", flush=True)
        sp.close()
