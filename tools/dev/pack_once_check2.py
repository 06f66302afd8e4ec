import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hive_alphazero_amd import alpha_net as A
import hive_alphazero_amd as h
L = h.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
torch.manual_seed(3)
net = A.ChessNet().cuda()
tr = A.Trainer(net)
m = tr.model
fwd, bwd = m._packed_tower(torch.device("cuda:0"))
torch.cuda.synchronize()
one = torch.zeros(9 * 256 * 256, dtype=torch.bfloat16, device="cuda")
j = 0
for i in range(19):
    blk = getattr(net, "res_%i" % i)
    for w in (blk.conv1.weight, blk.conv2.weight):
        wsrc, cl = A._weight_layout(w)
        for trn, got in ((0, fwd), (1, bwd)):
            L.hive_nn_pack_conv3x3_weights(P(wsrc), 256, trn, cl, P(one), None)
            torch.cuda.synchronize()
            if not torch.equal(one.view(torch.int16), got[j].view(torch.int16)):
                print("layer", j, "form", trn, "differs", (one.float() - got[j].float()).abs().max().item(), cl, w.stride())
        j += 1
print("packs compared")
outs = {}
for once in (False, True):
    m.pack_once = once
    acts = []
    hk = net.outblock.register_forward_pre_hook(lambda mod, inp: acts.append(inp[0].detach().clone()))
    gg = torch.Generator(device="cuda").manual_seed(9)
    x = (torch.rand((16, 56, 12, 12), device="cuda", generator=gg) < 0.1).float()
    m.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        p, v = m(x.contiguous(memory_format=torch.channels_last))
    hk.remove()
    outs[once] = (acts[0], p.detach().clone(), v.detach().clone())
for k in range(3):
    a, b = outs[False][k], outs[True][k]
    print(k, torch.equal(a, b), (a.float() - b.float()).abs().max().item())
