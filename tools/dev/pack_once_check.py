import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hive_alphazero_amd import alpha_net as A
out = []
for once in (False, False, True, True, False):
    torch.manual_seed(3)
    net = A.ChessNet().cuda()
    tr = A.Trainer(net)
    tr.model.pack_once = once
    gg = torch.Generator(device="cuda").manual_seed(9)
    x = (torch.rand((16, 56, 12, 12), device="cuda", generator=gg) < 0.1).float()
    pol = torch.softmax(torch.randn((16, 1584), device="cuda", generator=gg), dim=1)
    val = torch.rand((16,), device="cuda", generator=gg) * 2 - 1
    tr.model.train()
    loss = tr.loss(x, pol, val)
    loss.backward()
    g = {k: p.grad.clone() for k, p in net.named_parameters()}
    out.append((once, loss.item(), g))
    print(once, repr(loss.item()))
for i in range(1, len(out)):
    a, b = out[i - 1], out[i]
    worst = max(((a[2][k] - b[2][k]).abs().max().item() / (b[2][k].abs().max().item() + 1e-12), k) for k in a[2])
    ndiff = sum(1 for k in a[2] if not torch.equal(a[2][k], b[2][k]))
    print(a[0], "->", b[0], "loss diff", a[1] - b[1], "grads differing", ndiff, "of", len(a[2]), "worst rel", worst)
