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
pairs = 0
while time.time() - t0 < budget:
    n = int(torch.randint(1, 65537, (1,), generator=g).item()) if launches % 3 else int(torch.randint(1, 200, (1,), generator=g).item())
    off = int(torch.randint(0, 65536 - n + 1, (1,), generator=g).item())
    b = pool[off:off + n].contiguous()
    outs = []
    for s in streams:
        with torch.cuda.stream(s):
            mask, count, lst = batch.movegen(b, want_list=(launches % 2 == 0))
            outs.append((mask, count, lst))
    torch.cuda.synchronize()
    for m, c, l in outs[1:]:
        assert torch.equal(m, outs[0][0]) and torch.equal(c, outs[0][1]), ("streams disagree", n, off)
        assert l is None or torch.equal(l, outs[0][2]), ("streams disagree on the id lists", n, off)
    # the same boards inside a different batch must give the same rows (launches of >= 16,384 boards run the pair layout,
    # the 64-board launch the quad layout: this is also layout against layout)
    k = min(n, 64)
    m2, c2, l2 = batch.movegen(pool[off:off + k].contiguous(), want_list=True)
    assert torch.equal(m2, outs[0][0][:k]) and torch.equal(c2, outs[0][1][:k]), ("batch-size dependence", n, off)
    assert outs[0][2] is None or torch.equal(l2, outs[0][2][:k]), ("batch-size dependence of the id lists", n, off)
    pairs = pairs + 1 if n >= 16384 else pairs
    launches += 5; boards += 4 * n + k
print(f"ok: {launches} launches ({pairs} rounds in the pair layout), {boards / 1e6:.1f} M boards in {time.time() - t0:.1f} s, no disagreement")
