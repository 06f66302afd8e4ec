"""Stress run of the piece kernel (its pin waves synchronise through LDS counters): random batch sizes including
ragged tails, movegen and leaf (both colours + features) variants, several streams, results compared across repeats."""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hive_alphazero_amd as h
from hive_alphazero_amd import batch, playout
from hive_alphazero_amd._lib import HIVE_MASK_WORDS
L = h.load()
pool = playout.random_positions(65536, seed=9)
g = torch.Generator().manual_seed(1)
streams = [torch.cuda.Stream() for _ in range(4)]
t0, launches, boards = time.time(), 0, 0
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
ref = {}
while time.time() - t0 < budget:
    n = int(torch.randint(1, 65537, (1,), generator=g).item()) if launches % 3 else int(torch.randint(1, 200, (1,), generator=g).item())
    off = int(torch.randint(0, 65536 - n + 1, (1,), generator=g).item())
    b = pool[off:off + n].contiguous()
    outs = []
    for s in streams:
        with torch.cuda.stream(s):
            mask, count, lst = batch.movegen(b, want_list=(launches % 2 == 0))
            outs.append((mask, count))
    torch.cuda.synchronize()
    for m, c in outs[1:]:
        assert torch.equal(m, outs[0][0]) and torch.equal(c, outs[0][1]), ("streams disagree", n, off)
    # the same boards inside a different batch must give the same rows
    k = min(n, 64)
    m2, c2, _ = batch.movegen(pool[off:off + k].contiguous())
    assert torch.equal(m2, outs[0][0][:k]) and torch.equal(c2, outs[0][1][:k]), ("batch-size dependence", n, off)
    launches += 5; boards += 4 * n + k
print(f"ok: {launches} launches, {boards / 1e6:.1f} M boards in {time.time() - t0:.1f} s, no disagreement")
