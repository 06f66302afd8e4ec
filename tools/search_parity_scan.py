"""Differential scan: GPU tree search (noise off, transposition merging on) against the sequential reference mirror
(solo_play.HivePlayer over the HIP-backed GamePlay) on many random mid-game positions -- policy vector, move, visit total."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_search import _host_stub_evaluator
from mcts_stub import StubPipe
import hive_alphazero_amd.solo_play as sp
from hive_alphazero_amd import batch, mcts
from hive_alphazero_amd.env_hive import GamePlay
sp.SEARCH_THREADS = 1; sp.noise_eps = 0.0
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = hits_total = done = 0
for seed in range(1000, 1000 + cases):
    rng = np.random.default_rng(seed)
    plies = int(rng.integers(4, 53))
    g = GamePlay(1050, 900)
    for _ in range(plies):
        acts = g.actions()
        if g.game_is_over():
            break
        g.move(int(acts[rng.integers(len(acts))]) if acts else -1)
    if g.game_is_over() or g.state.turn >= 55:
        continue
    B = batch.BoardBatch(1); B.import_state(g._rec.reshape(1, 64), g._hist.reshape(1, 384)); rb, rh = B.export_state()
    player = sp.HivePlayer(pipes=[StubPipe()]); player.simulation_num_per_move = sims
    np.random.seed(0); move, (rpol, rvis) = player.action(g)
    ts = mcts.TreeSearch(1, sims, _host_stub_evaluator, plane_dtype=torch.float32, noise_eps=0.0)
    a, pol, n = ts.search(rb, rh)
    diff = float(np.abs(pol[0].cpu().numpy().astype(np.float64) - np.asarray(rpol, dtype=np.float64)).max())
    hits = int(ts.transposition_hits()[0]); hits_total += hits; done += 1
    ok = diff < 1e-6 and int(a[0]) == move and int(n[0]) == int(rvis)
    if not ok:
        bad += 1
        print(f"MISMATCH seed {seed} turn {g.state.turn} hits {hits} maxdiff {diff:.2e} action {int(a[0])} vs {move} n {int(n[0])} vs {rvis}", flush=True)
    ts.close(); B.close()
print(f"{done} positions x {sims} sims: {bad} mismatches, {hits_total} descents through shared entries")
