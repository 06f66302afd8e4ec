"""The whole AlphaZero loop of the reference on one MI355X, end to end (self_play.py -> optimize.py/train.py -> workers
reload the weights): GPU self-play with packed per-ply records -> the trainer's tensors widened ON THE GPU from the packed
features (records.dataset_tensors_gpu_packed: the same planes, policies and 0.99 ** k discounted values optimize.py:42-65
builds from the reference's JSON rows; that equivalence is what tests/test_gpu_scale.py and tests/test_host_cpu.py check)
-> Trainer steps on the same ChessNet -> InferenceNet.refresh -> next round of self-play.  Sized to run in well under a minute."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_alphazero_amd import mcts, records
from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet, Trainer

games, sims, rounds = 256, 16, 2
torch.manual_seed(0)
net = ChessNet().cuda()
evaluator = InferenceNet(net)
trainer = Trainer(net)
for rnd in range(rounds):
    net.eval()
    sp = mcts.SelfPlay(games, sims, evaluator, seed=100 + rnd, game_ids=range(rnd * games, (rnd + 1) * games),
                       packed_records=True, max_finished_kept=2 * games)
    t0 = time.perf_counter()
    batches = []
    while sp.running() and sp.plies < 80:
        sp.play_ply()
        b = sp.drain_finished_packed()
        if b is not None:
            batches.append(b)
    torch.cuda.synchronize()
    t_sp = time.perf_counter() - t0
    illegal, results = sp.env.illegal_count(), (sp.white_wins, sp.black_wins, sp.draws)
    assert sp.dropped_games == 0 and sp.unrecorded_games == 0
    sp.close()
    packed = records.concat_packed(batches)
    t0 = time.perf_counter()
    states, policies, values = records.dataset_tensors_gpu_packed(packed, dtype=torch.float32, layout="chw")
    torch.cuda.synchronize()
    t_ds = time.perf_counter() - t0
    n = states.shape[0]
    t0 = time.perf_counter()
    losses = []
    g = torch.Generator(device="cuda").manual_seed(rnd)
    for _ in range(20):
        idx = torch.randint(0, n, (min(512, n),), generator=g, device="cuda")
        losses.append(trainer.step(states[idx], policies[idx], values[idx]))
    trainer.end_epoch()
    torch.cuda.synchronize()
    t_tr = time.perf_counter() - t0
    evaluator.refresh(net)
    print(f"round {rnd}: {sp.plies} plies of {games} games x {sims} sims in {t_sp:.1f} s, W/B/draw {results}, illegal {illegal}; "
          f"{n} rows of {records.packed_games(packed)} games widened on the GPU in {t_ds * 1e3:.0f} ms; "
          f"20 training steps of 512 in {t_tr:.1f} s, loss {losses[0]:.3f} -> {losses[-1]:.3f}", flush=True)
    assert illegal == 0 and all(np.isfinite(l) for l in losses) and records.packed_games(packed) == games
print("ok")
