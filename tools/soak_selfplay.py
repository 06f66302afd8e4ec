"""Soak: thousands of complete self-play games through the GPU engine; every move must be legal."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_alphazero_amd import mcts
from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
plies = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(0)
net = InferenceNet(ChessNet().cuda().eval())
sp = mcts.SelfPlay(1024, 8, net, seed=11)
t0 = time.time()
for i in range(plies):
    sp.play_ply()
    if i % 50 == 49:
        torch.cuda.synchronize()
        print(f"ply {i+1}: finished {sp.finished} W {sp.white_wins} B {sp.black_wins} D {sp.draws} illegal {sp.env.illegal_count()} "
              f"mem {torch.cuda.memory_allocated() >> 20} MiB  {time.time() - t0:.1f}s", flush=True)
sp._retire_finished()
assert sp.env.illegal_count() == 0
rows = sp.finished_game_rows(0)
print("ok: games", sp.finished, "rows of first game", len(rows))
