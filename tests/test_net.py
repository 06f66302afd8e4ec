"""ChessNet mirror + InferenceNet against outputs of the reference's alpha_zero/alpha_net.py
(tests/golden/net.json, produced by oracle/gen_golden.py net, which also asserted that a same-seed
init gives identical tensors to the reference and that the state_dict keys are the same 255)."""
import gzip
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _inputs(gold):
    with gzip.open(os.path.join(GOLD, "games_full.json.gz"), "rt") as f:
        games = json.load(f)["games"]
    xs = []
    for gi, ply in gold["picks"]:
        rec = games[gi]["plies"][ply]
        pl = np.zeros((12, 12, 56), dtype=np.float32)
        pl.reshape(-1)[rec["planes"]] = 1.0
        pl[:, :, 31] = rec["t"]
        xs.append(pl)
    return torch.from_numpy(np.stack(xs))          # [B,12,12,56] (HWC, the env's layout)


def _gold():
    with open(os.path.join(GOLD, "net.json")) as f:
        return json.load(f)


def _check(p, v, gold, atol_p, atol_v):
    p, v = p.float().cpu(), v.float().cpu().view(-1)
    assert np.allclose(v.numpy(), gold["v"], atol=atol_v), (v.numpy().tolist(), gold["v"], atol_v)
    for b, top in enumerate(gold["p_top"]):
        for i, val in top:
            assert abs(float(p[b, i]) - val) <= atol_p
        assert np.allclose(p[b, :16].numpy(), gold["p_first16"][b], atol=atol_p)
        assert abs(float(p[b].sum()) - 1.0) < 1e-3


def test_chessnet_matches_reference_cpu_fp32():
    from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
    gold = _gold()
    torch.manual_seed(gold["seed"])
    net = ChessNet().eval()
    sd = net.state_dict()
    assert len(sd) == gold["n_keys"] == 255
    assert sum(p.numel() for p in net.parameters()) == gold["n_params"] == 51803188
    for k in ("conv.conv1.weight", "conv.bn1.running_mean", "res_0.conv1.weight", "res_18.bn2.num_batches_tracked",
              "outblock.conv.weight", "outblock.bn.weight", "outblock.fc1.bias", "outblock.fc2.weight",
              "outblock.conv1.bias", "outblock.bn1.running_var", "outblock.fc.weight"):
        assert k in sd
    x = _inputs(gold)
    with torch.no_grad():
        p, v = net(x.permute(0, 3, 1, 2))
    _check(p, v, gold, 1e-6, 1e-5)          # tolerance: fp32 same-op replay
    inf = InferenceNet(net, dtype=torch.float32, device="cpu", use_graph=False)
    p2, v2 = inf(x)
    _check(p2, v2, gold, 1e-6, 1e-5)        # BN folding + NHWC head permutation are exact up to fp32 rounding
    # checkpoint round trip with the reference's container format (train.py:35-38,50-51)
    net2 = ChessNet()
    net2.load_state_dict({"state_dict": sd}["state_dict"])


@pytest.mark.gpu
def test_inference_net_gpu_tolerances():
    assert torch.cuda.is_available()
    from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
    gold = _gold()
    torch.manual_seed(gold["seed"])
    net = ChessNet().eval()
    x = _inputs(gold).cuda()
    with torch.no_grad():
        p, v = net.cuda()(x.permute(0, 3, 1, 2))
    _check(p, v, gold, 2e-6, 2e-5)                              # fp32 on the GPU
    for dt, atol_p, atol_v in ((torch.float32, 2e-6, 2e-5), (torch.bfloat16, 4e-4, 5e-2), (torch.float16, 1e-4, 1e-2)):
        inf = InferenceNet(net, dtype=dt)
        p2, v2 = inf(x)
        _check(p2, v2, gold, atol_p, atol_v)                    # stated tolerance of the reduced-precision engine
        p3, v3 = inf(x)                                         # graph replay: same result up to the library
        # MIOpen / hipBLASLt use split-K float atomics: repeated runs differ in the last bits (measured <= 2e-6)
        dp, dv = float((p2 - p3).abs().max()), float((v2 - v3).abs().max())
        # the policy logits leave the bf16/fp16 GEMM rounded to 8/11 bits; a split-K reordering can flip the last one
        assert dp <= (1e-6 if dt == torch.float32 else 2e-4) and dv <= 2e-2, (str(dt), dp, dv)


@pytest.mark.gpu
def test_hip_conv3x3_matches_torch():
    """hive_nn_conv3x3 (MFMA implicit GEMM, fused bias/skip/ReLU) against F.conv2d in fp32 on the same
    bf16-rounded operands.  Tolerance: one bf16 rounding of the output (2^-8 relative) + fp32
    accumulation-order noise."""
    assert torch.cuda.is_available()
    import ctypes
    import torch.nn.functional as F
    import hive_alphazero_amd as h
    from hive_alphazero_amd.alpha_net import _frag_major
    L = h.load()
    g = torch.Generator(device="cuda").manual_seed(1)
    for cin, B in ((256, 5), (56, 3)):
        x = torch.randn((B, 12, 12, cin), device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn((256, cin, 3, 3), device="cuda", generator=g) * (2.0 / (9 * cin)) ** 0.5)
        bias = torch.randn((256,), device="cuda", generator=g)
        res = torch.randn((B, 12, 12, 256), device="cuda", generator=g).to(torch.bfloat16)
        wt = _frag_major(w, x.device)
        wq = w.to(torch.bfloat16).float()
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), wq, bias, padding=1)
        for use_res in (False, True):
            for relu in (0, 1):
                want = ref + (res.float().permute(0, 3, 1, 2) if use_res else 0)
                if relu:
                    want = torch.relu(want)
                want = want.permute(0, 2, 3, 1)
                y = torch.full((B, 12, 12, 256), float("nan"), dtype=torch.bfloat16, device="cuda")
                rc = L.hive_nn_conv3x3(ctypes.c_void_p(x.data_ptr()), cin, ctypes.c_void_p(wt.data_ptr()),
                                       ctypes.c_void_p(bias.data_ptr()),
                                       ctypes.c_void_p(res.data_ptr()) if use_res else None,
                                       ctypes.c_void_p(y.data_ptr()), B, relu, None)
                assert rc == 0
                torch.cuda.synchronize()
                err = (y.float() - want).abs()
                tol = 1e-2 * want.abs() + 2e-2
                assert bool((err <= tol).all()), (cin, use_res, relu, float(err.max()))
                assert float(err.mean()) < 5e-3


@pytest.mark.gpu
def test_inference_net_hip_vs_torch_backend():
    assert torch.cuda.is_available()
    from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
    gold = _gold()
    torch.manual_seed(gold["seed"])
    net = ChessNet().eval().cuda()
    x = _inputs(gold).cuda()
    a = InferenceNet(net, dtype=torch.bfloat16, conv="hip")
    b = InferenceNet(net, dtype=torch.bfloat16, conv="torch")
    pa, va = a(x)
    pb, vb = b(x)
    _check(pa, va, gold, 2e-4, 3e-2)
    assert float((pa - pb).abs().max()) < 2e-4 and float((va - vb).abs().max()) < 3e-2


def _toy_batch(n=8, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand((n, 56, 12, 12), generator=g) < 0.05).float()
    x[:, 31] = 9.0
    pol = torch.zeros((n, 1584))
    pol[torch.arange(n), torch.randint(0, 1584, (n,), generator=g)] = 1.0
    val = torch.where(torch.rand((n,), generator=g) < 0.5, -1.0, 1.0)
    return x, pol, val


def test_trainer_loss_matches_reference_formula_cpu():
    """AlphaLoss = (v - z)^2 + sum(-pi * log(1e-6 + p)) (alpha_net.py:98-115); one Adam step lowers it."""
    from hive_alphazero_amd.alpha_net import ChessNet, Trainer
    torch.manual_seed(1)
    net = ChessNet()
    tr = Trainer(net)
    x, pol, val = _toy_batch(4)
    net.eval()
    with torch.no_grad():
        p, v = net(x)
        want = ((v[:, 0] - val) ** 2 + torch.sum(-pol * torch.log(1e-6 + p), 1)).mean()
        got = tr.loss(x, pol, val)
    assert abs(float(want) - float(got)) < 1e-5
    l0 = tr.step(x, pol, val)
    l1 = tr.step(x, pol, val)
    assert l1 < l0


@pytest.mark.gpu
def test_trainer_bf16_autocast_close_to_fp32():
    assert torch.cuda.is_available()
    from hive_alphazero_amd.alpha_net import ChessNet, Trainer
    torch.manual_seed(1)
    net = ChessNet().cuda()
    x, pol, val = _toy_batch(16)
    net.eval()
    fp32 = Trainer(net, autocast_dtype=None)
    bf16 = Trainer(net, autocast_dtype=torch.bfloat16)
    with torch.no_grad():
        a, b = float(fp32.loss(x, pol, val)), float(bf16.loss(x, pol, val))
    assert abs(a - b) < 0.05 * abs(a) + 0.05        # stated tolerance of the bf16 forward on the loss
    l0 = bf16.step(x, pol, val)
    for _ in range(3):
        l1 = bf16.step(x, pol, val)
    assert l1 < l0


@pytest.mark.gpu
def test_hip_resblock_matches_two_convs():
    """hive_nn_resblock (both convolutions of a residual block in one launch, intermediate in LDS) against the two
    hive_nn_conv3x3 launches it replaces: identical arithmetic, so bit-identical bf16 outputs."""
    assert torch.cuda.is_available()
    import ctypes
    import hive_alphazero_amd as h
    from hive_alphazero_amd.alpha_net import _frag_major
    L = h.load()
    g = torch.Generator(device="cuda").manual_seed(2)
    B = 7
    x = torch.randn((B, 12, 12, 256), device="cuda", generator=g).to(torch.bfloat16)
    w1 = _frag_major(torch.randn((256, 256, 3, 3), device="cuda", generator=g) * 0.03, x.device)
    w2 = _frag_major(torch.randn((256, 256, 3, 3), device="cuda", generator=g) * 0.03, x.device)
    b1 = torch.randn((256,), device="cuda", generator=g)
    b2 = torch.randn((256,), device="cuda", generator=g)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    o = torch.empty_like(x); y_ref = torch.empty_like(x); y = torch.full_like(x, float("nan"))
    assert L.hive_nn_conv3x3(P(x), 256, P(w1), P(b1), None, P(o), B, 1, None) == 0
    assert L.hive_nn_conv3x3(P(o), 256, P(w2), P(b2), P(x), P(y_ref), B, 1, None) == 0
    assert L.hive_nn_resblock(P(x), P(w1), P(b1), P(w2), P(b2), P(y), B, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref)
    assert L.hive_nn_resblock(P(x), P(w1), P(b1), P(w2), P(b2), P(x), B, None) == -1     # in-place is refused


_DDP_WORKER = r"""
import sys, json
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from hive_alphazero_amd import dist as hd
from hive_alphazero_amd.alpha_net import ChessNet, Trainer
rank, local_rank, world = hd.init("gloo")
torch.set_num_threads(3)
torch.manual_seed(0)                      # same initial weights on every rank
net = ChessNet()
before = float(sum(p.double().abs().sum() for p in net.parameters()))
tr = Trainer(net, lr=1e-3)                # picks DistributedDataParallel up from the process group
assert tr.model is not tr.net
g = torch.Generator().manual_seed(100 + rank)   # a different shard of rows per rank
losses = []
for _ in range(2):
    x = (torch.rand(3, 56, 12, 12, generator=g) < 0.1).float()
    pi = torch.softmax(torch.randn(3, 1584, generator=g), 1)
    z = torch.sign(torch.randn(3, generator=g))
    losses.append(tr.step(x, pi, z))
after = float(sum(p.double().abs().sum() for p in net.parameters()))
sums = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
dist.all_gather(sums, torch.tensor([after], dtype=torch.float64))
if rank == 0:
    print(json.dumps({"before": before, "after": [float(s) for s in sums], "losses": losses}))
dist.destroy_process_group()
"""


def test_trainer_ddp_world_size_2_gloo(tmp_path):
    """the one place a collective belongs.  Two ranks train on different rows; the gradient
    all-reduce keeps their weights identical."""
    import subprocess
    import sys
    script = tmp_path / "ddp_worker.py"
    script.write_text(_DDP_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29583")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29583", str(script), root],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["after"][0] == res["after"][1] != res["before"]
    assert all(np.isfinite(l) for l in res["losses"])
