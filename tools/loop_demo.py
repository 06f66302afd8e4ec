"""The whole AlphaZero loop of the reference on one MI355X, end to end (self_play.py -> optimize.py/train.py -> workers
reload the weights): GPU self-play -> the reference's [state, policy, value, [len, counter]] rows -> Trainer steps on the
same ChessNet -> InferenceNet.refresh -> next round of self-play.  A demonstration, sized to run in about a minute."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_alphazero_amd import mcts
from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet, Trainer

games, sims, rounds = 256, 16, 2
torch.manual_seed(0)
net = ChessNet().cuda()
evaluator = InferenceNet(net)
trainer = Trainer(net)
for rnd in range(rounds):
    net.eval()
    sp = mcts.SelfPlay(games, sims, evaluator, seed=100 + rnd)
    sp.max_finished_kept = 48
    t0 = time.perf_counter()
    while len(sp.finished_games) < sp.max_finished_kept and sp.plies < 80:
        sp.play_ply()
    sp._retire_finished()
    torch.cuda.synchronize()
    t_sp = time.perf_counter() - t0
    rows = [r for k in range(len(sp.finished_games)) for r in sp.finished_game_rows(k)]
    illegal = sp.env.illegal_count()
    results = (sp.white_wins, sp.black_wins, sp.draws)
    sp.close()
    states = torch.from_numpy(np.stack([np.asarray(r[0], dtype=np.float32).transpose(2, 0, 1) for r in rows]))
    policies = torch.from_numpy(np.stack([np.asarray(r[1], dtype=np.float32) for r in rows]))
    values = torch.tensor([float(r[2]) for r in rows])
    t0 = time.perf_counter()
    losses = []
    g = torch.Generator().manual_seed(rnd)
    for _ in range(20):
        idx = torch.randint(0, len(rows), (min(256, len(rows)),), generator=g)
        losses.append(trainer.step(states[idx], policies[idx], values[idx]))
    trainer.end_epoch()
    torch.cuda.synchronize()
    t_tr = time.perf_counter() - t0
    evaluator.refresh(net)
    print(f"round {rnd}: {sp.plies} plies of {games} games x {sims} sims in {t_sp:.1f} s, W/B/draw {results}, illegal {illegal}; "
          f"{len(rows)} rows from {len(sp.finished_games)} kept games; 20 training steps in {t_tr:.1f} s, loss {losses[0]:.3f} -> {losses[-1]:.3f}",
          flush=True)
    assert illegal == 0 and all(np.isfinite(l) for l in losses)
print("ok")
