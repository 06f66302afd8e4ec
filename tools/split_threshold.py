"""Leaf-batch sizes at which the two-chain tower pays: InferenceNet forward (HIP graph) with split_streams off / on."""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
torch.manual_seed(0)
net = ChessNet().cuda().eval()
one = InferenceNet(net, dtype=torch.bfloat16); one.split_streams = False
two = InferenceNet(net, dtype=torch.bfloat16)
for B in (256, 384, 512, 768, 1024, 2048, 4096):
    x = (torch.rand((B, 12, 12, 56), device="cuda") < 0.08).to(torch.bfloat16)
    t = {}
    for name, inf in (("one chain", one), ("two chains", two)):
        inf(x); inf(x)
    for _ in range(6):
        for name, inf in (("one chain", one), ("two chains", two)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): inf(x)
            e1.record(); torch.cuda.synchronize()
            t.setdefault(name, []).append(e0.elapsed_time(e1) / 5)
    a, b = statistics.median(t["one chain"]), statistics.median(t["two chains"])
    print(f"B = {B:5d}: one chain {a:7.3f} ms   two chains {b:7.3f} ms   ({(b / a - 1) * 100:+.1f} %)", flush=True)
