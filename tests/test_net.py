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
@pytest.mark.parametrize("dtype,rel,abs_", [(torch.bfloat16, 1e-2, 2e-2), (torch.float16, 1.5e-3, 2.5e-3)])
def test_hip_conv3x3_matches_torch(dtype, rel, abs_):
    """hive_nn_conv3x3_dt (MFMA implicit GEMM, fused bias/skip/ReLU) against F.conv2d in fp32 on the same
    16-bit-rounded operands.  Tolerance: one rounding of the output to the format (2^-8 relative for bf16, 2^-11 for
    fp16) + fp32 accumulation-order noise."""
    assert torch.cuda.is_available()
    import ctypes
    import torch.nn.functional as F
    import hive_alphazero_amd as h
    from hive_alphazero_amd import _lib
    from hive_alphazero_amd.alpha_net import _frag_major
    L = h.load()
    dt = _lib.BF16 if dtype == torch.bfloat16 else _lib.F16
    g = torch.Generator(device="cuda").manual_seed(1)
    for cin, B in ((256, 5), (56, 3)):
        x = torch.randn((B, 12, 12, cin), device="cuda", generator=g).to(dtype)
        w = (torch.randn((256, cin, 3, 3), device="cuda", generator=g) * (2.0 / (9 * cin)) ** 0.5)
        bias = torch.randn((256,), device="cuda", generator=g)
        res = torch.randn((B, 12, 12, 256), device="cuda", generator=g).to(dtype)
        wt = _frag_major(w, x.device, dtype)
        wq = w.to(dtype).float()
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), wq, bias, padding=1)
        for use_res in (False, True):
            for relu in (0, 1):
                want = ref + (res.float().permute(0, 3, 1, 2) if use_res else 0)
                if relu:
                    want = torch.relu(want)
                want = want.permute(0, 2, 3, 1)
                y = torch.full((B, 12, 12, 256), float("nan"), dtype=dtype, device="cuda")
                rc = L.hive_nn_conv3x3_dt(ctypes.c_void_p(x.data_ptr()), cin, ctypes.c_void_p(wt.data_ptr()),
                                          ctypes.c_void_p(bias.data_ptr()),
                                          ctypes.c_void_p(res.data_ptr()) if use_res else None,
                                          ctypes.c_void_p(y.data_ptr()), B, relu, dt, None)
                assert rc == 0
                torch.cuda.synchronize()
                err = (y.float() - want).abs()
                tol = rel * want.abs() + abs_
                assert bool((err <= tol).all()), (cin, use_res, relu, float(err.max()))
                assert float(err.mean()) < abs_ / 4
                if dtype == torch.bfloat16:               # the original entry point = the bf16 instantiation
                    y0 = torch.empty_like(y)
                    assert L.hive_nn_conv3x3(ctypes.c_void_p(x.data_ptr()), cin, ctypes.c_void_p(wt.data_ptr()),
                                             ctypes.c_void_p(bias.data_ptr()),
                                             ctypes.c_void_p(res.data_ptr()) if use_res else None,
                                             ctypes.c_void_p(y0.data_ptr()), B, relu, None) == 0
                    assert torch.equal(y0, y)
    assert L.hive_nn_conv3x3_dt(ctypes.c_void_p(x.data_ptr()), 56, ctypes.c_void_p(wt.data_ptr()), ctypes.c_void_p(bias.data_ptr()),
                                None, ctypes.c_void_p(y.data_ptr()), 1, 1, _lib.F32, None) == -1     # fp32 is not a kernel dtype


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


def test_trainer_fused_kernels_need_the_gpu():
    """The fused BatchNorm / convolution kernels are bf16 GPU code: on a CPU-resident net the default trainer runs the
    plain modules, asking for the kernels explicitly fails loudly, and FusedTrainNet in eval mode is the net itself."""
    from hive_alphazero_amd.alpha_net import ChessNet, FusedTrainNet, Trainer
    torch.manual_seed(0)
    net = ChessNet()
    assert Trainer(net).fused is False
    with pytest.raises(ValueError):
        Trainer(net, fused=True)
    net.eval()
    x, _, _ = _toy_batch(2)
    with torch.no_grad():
        p0, v0 = net(x)
        p1, v1 = FusedTrainNet(net)(x)
    assert torch.equal(p0, p1) and torch.equal(v0, v1)


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


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_hip_tower_matches_resblock_chain(dtype):
    """hive_nn_tower (the whole residual tower in one launch, boards resident in LDS across blocks) in its three
    workgroup forms against a chain of hive_nn_resblock_dt launches: identical arithmetic and rounding points, so
    bit-identical outputs -- for both 16-bit formats, odd batch sizes (a half-empty two-board group) and one block."""
    assert torch.cuda.is_available()
    import ctypes
    import hive_alphazero_amd as h
    from hive_alphazero_amd import _lib
    L = h.load()
    dt = _lib.BF16 if dtype == torch.bfloat16 else _lib.F16
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    g = torch.Generator(device="cuda").manual_seed(3)
    for B, nblk in ((1, 2), (5, 3), (64, 1), (257, 2)):
        x = torch.randn((B, 144, 256), device="cuda", generator=g).to(dtype)
        w = (torch.randn((2 * nblk, 9 * 8 * 16 * 64 * 8), device="cuda", generator=g) * 0.02).to(dtype)
        bias = torch.randn((2 * nblk, 256), device="cuda", generator=g) * 0.1
        bufs = [x, torch.empty_like(x), torch.empty_like(x)]
        cur = 0
        for i in range(nblk):
            nxt = 1 if cur != 1 else 2
            assert L.hive_nn_resblock_dt(P(bufs[cur]), P(w[2 * i]), P(bias[2 * i]), P(w[2 * i + 1]), P(bias[2 * i + 1]),
                                         P(bufs[nxt]), B, dt, None) == 0
            cur = nxt
        want = bufs[cur]
        for mode in (0, 1, 2, 3):
            y = torch.full_like(x, float("nan"))
            assert L.hive_nn_tower(P(x), P(w), P(bias), P(y), B, nblk, dt, mode, None) == 0
            torch.cuda.synchronize()
            assert torch.equal(y, want), (B, nblk, mode)
    assert L.hive_nn_tower(P(x), P(w), P(bias), P(x), B, nblk, dt, 0, None) == -1          # in place is refused
    assert L.hive_nn_tower(P(x), P(w), P(bias), P(y), B, nblk, _lib.F32, 0, None) == -1


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


@pytest.mark.gpu
@pytest.mark.parametrize("with_res,relu", [(False, True), (True, True), (False, False)])
@pytest.mark.parametrize("C", [256, 128, 1])
def test_hip_bn_act_matches_torch(with_res, relu, C):
    """csrc/hive_train.hip against torch.nn.BatchNorm2d (training mode, fp32 math on the same bf16 inputs):
    output, input / skip / parameter gradients, running statistics.  Tolerances: one bf16 rounding of the output
    (2^-8 relative) on values of order 1; the gradient sums run over 144 * batch terms in fp32."""
    assert torch.cuda.is_available()
    from hive_alphazero_amd.alpha_net import bn_act, bn_act_ok
    torch.manual_seed(3)
    B = 24 if C == 256 else 32           # C = 128 / 1: the heads' BatchNorms (columns of a channel folded; 32 * 144 * 1 = 18 rows of 256)
    bn = torch.nn.BatchNorm2d(C).cuda()
    ref = torch.nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
        ref.weight.copy_(bn.weight); ref.bias.copy_(bn.bias)
    x = (torch.randn((B, C, 12, 12), device="cuda") * 1.7 + 0.3).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    r = torch.randn((B, C, 12, 12), device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn((B, C, 12, 12), device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert bn_act_ok(x)
    if C == 1:
        assert not bn_act_ok(x[:3])      # 3 * 144 values are not a whole number of 256-wide rows: the caller keeps the library path
    x1, r1 = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    y = bn_act(x1, bn, residual=r1 if with_res else None, relu=relu)
    y.backward(dy)
    x2, r2 = x.float().requires_grad_(True), r.float().requires_grad_(True)
    z = ref(x2)
    if with_res:
        z = z + r2
    if relu:
        z = torch.relu(z)
    # the HIP backward masks with the bf16 output it stored; give the reference the same mask where they disagree at 0
    z.backward(dy.float())
    assert (y.float() - z).abs().max().item() <= 2 ** -7 * max(1.0, z.abs().max().item())
    assert torch.allclose(bn.running_mean, ref.running_mean, atol=1e-4) and torch.allclose(bn.running_var, ref.running_var, rtol=1e-3)
    assert int(bn.num_batches_tracked) == 1
    scale = x2.grad.abs().max().item()
    assert (x1.grad.float() - x2.grad).abs().max().item() <= 0.02 * scale + 1e-3
    if with_res:
        assert (r1.grad.float() - r2.grad).abs().max().item() <= 2 ** -7 * r2.grad.abs().max().item() + 1e-6
    gs = ref.weight.grad.abs().max().item()
    assert (bn.weight.grad - ref.weight.grad).abs().max().item() <= 0.01 * gs
    assert (bn.bias.grad - ref.bias.grad).abs().max().item() <= 0.01 * ref.bias.grad.abs().max().item() + 1e-2


@pytest.mark.gpu
def test_trainer_fused_bn_close_to_library_path():
    """Trainer(fused=True) (HIP BatchNorm + skip + ReLU) against Trainer(fused=False) (torch/MIOpen) on the same
    weights and batch: same loss within bf16 noise, same gradient direction, same running statistics, and it trains."""
    assert torch.cuda.is_available()
    import copy
    from hive_alphazero_amd.alpha_net import ChessNet, Trainer
    torch.manual_seed(5)
    net_a = ChessNet().cuda()
    net_b = copy.deepcopy(net_a)
    net_c = copy.deepcopy(net_a)
    x, pol, val = _toy_batch(32)
    ta, tb, tc = Trainer(net_a, fused=True), Trainer(net_b, fused=False), Trainer(net_c, autocast_dtype=None)
    assert ta.fused and not tb.fused and not tc.fused
    grads = []
    losses = []
    for t, n in ((ta, net_a), (tb, net_b), (tc, net_c)):
        t.model.train()
        l = t.loss(x, pol, val)
        l.backward()
        losses.append(float(l.detach()))
        grads.append(torch.cat([p.grad.flatten().float() for p in n.parameters()]))
    la, lb, lc = losses
    assert abs(la - lc) < 0.02 * abs(lc) + 0.02 and abs(lb - lc) < 0.02 * abs(lc) + 0.02
    cos = lambda u, v: torch.nn.functional.cosine_similarity(u, v, dim=0).item()
    fused_vs_fp32, lib_vs_fp32 = cos(grads[0], grads[2]), cos(grads[1], grads[2])
    # bf16 gradients of a 40-layer random-init tower are noisy either way; the fused path must be at least as close to
    # the fp32 gradient as the library bf16 path is (it rounds once per layer instead of three times)
    assert fused_vs_fp32 > 0.85 and fused_vs_fp32 >= lib_vs_fp32 - 0.02, (fused_vs_fp32, lib_vs_fp32)   # measured 0.896 vs 0.897
    assert torch.allclose(net_a.res_7.bn2.running_mean, net_b.res_7.bn2.running_mean, atol=2e-2)
    assert torch.allclose(net_a.res_7.bn2.running_var, net_b.res_7.bn2.running_var, rtol=5e-2, atol=1e-3)
    assert sorted(net_a.state_dict().keys()) == sorted(net_b.state_dict().keys())
    l0 = ta.step(x, pol, val)
    for _ in range(3):
        l1 = ta.step(x, pol, val)
    assert l1 < l0


@pytest.mark.gpu
@pytest.mark.parametrize("cin,weights_cl", [(256, False), (256, True), (56, True)])
def test_hip_training_conv_matches_torch(cin, weights_cl):
    """The training-step convolution (alpha_net._Conv3x3): forward and data gradient on hive_nn_conv3x3 (the data
    gradient through the transposed / rotated weight packing), weight gradient on the library path -- against
    F.conv2d autograd in fp32 on the same bf16-rounded operands.  Tolerance: bf16 output rounding (2^-8) on sums of
    up to 2304 products accumulated in fp32."""
    assert torch.cuda.is_available()
    import torch.nn.functional as F
    from hive_alphazero_amd.alpha_net import conv3x3
    torch.manual_seed(11)
    B = 8
    conv = torch.nn.Conv2d(cin, 256, 3, padding=1, bias=(cin == 56)).cuda()
    if weights_cl:
        conv = conv.to(memory_format=torch.channels_last)                   # as Trainer keeps the network
    with torch.no_grad():
        conv.weight.copy_(conv.weight.to(torch.bfloat16).float())          # operands exactly representable in bf16
    x = torch.randn((B, cin, 12, 12), device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn((B, 256, 12, 12), device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x1 = x.clone().requires_grad_(cin == 256)
    y = conv3x3(x1, conv)
    y.backward(dy)
    gw, gb = conv.weight.grad.clone(), (conv.bias.grad.clone() if conv.bias is not None else None)
    conv.weight.grad = None
    if conv.bias is not None:
        conv.bias.grad = None
    x2 = x.float().requires_grad_(True)
    z = F.conv2d(x2, conv.weight, conv.bias, padding=1)
    z.backward(dy.float())
    tol = lambda ref: 2 ** -7 * ref.abs().max().item()
    assert (y.float() - z).abs().max().item() <= tol(z)
    if cin == 256:
        assert (x1.grad.float() - x2.grad).abs().max().item() <= tol(x2.grad)
    assert (gw - conv.weight.grad).abs().max().item() <= 2 * tol(conv.weight.grad)
    if gb is not None:
        assert (gb - conv.bias.grad).abs().max().item() <= 1e-3 * conv.bias.grad.abs().max().item() + 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("weights_cl", [False, True])
def test_hip_weights_packed_in_one_launch_equal_the_single_packs(weights_cl):
    """hive_nn_pack_conv3x3_weights_multi (all tower convolutions of a training step, forward and data-gradient form,
    one launch) against hive_nn_pack_conv3x3_weights per weight: the same bytes; and a FusedTrainNet step with and
    without it: the same loss and gradients, bit for bit."""
    assert torch.cuda.is_available()
    import ctypes
    import hive_alphazero_amd as h
    from hive_alphazero_amd import alpha_net as A
    L = h.load()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    g = torch.Generator(device="cuda").manual_seed(5)
    ws = [torch.randn((256, 256, 3, 3), device="cuda", generator=g) for _ in range(5)]
    if weights_cl:
        ws = [w.contiguous(memory_format=torch.channels_last) for w in ws]
    table = torch.tensor([w.data_ptr() for w in ws], dtype=torch.int64).cuda()
    fwd = torch.zeros((5, 9 * 256 * 256), dtype=torch.bfloat16, device="cuda")
    bwd = torch.zeros_like(fwd)
    assert L.hive_nn_pack_conv3x3_weights_multi(P(table), 5, int(weights_cl), P(fwd), P(bwd), None) == 0, L.hive_last_error()
    one = torch.zeros(9 * 256 * 256, dtype=torch.bfloat16, device="cuda")
    for i, w in enumerate(ws):
        for tr, got in ((0, fwd), (1, bwd)):
            assert L.hive_nn_pack_conv3x3_weights(P(w), 256, tr, int(weights_cl), P(one), None) == 0
            torch.cuda.synchronize()
            assert torch.equal(one.view(torch.int16), got[i].view(torch.int16)), (i, tr)
    assert L.hive_nn_pack_conv3x3_weights_multi(None, 5, 0, P(fwd), P(bwd), None) == -1
    assert L.hive_nn_pack_conv3x3_weights_multi(P(table), 0, 0, P(fwd), P(bwd), None) == -1
    if not weights_cl:
        return
    # a whole forward + backward both ways: the tower's output bit for bit (this repo's kernels, deterministic); the heads
    # and the loss run through the libraries, whose kernel choice depends on the allocator's state (measured: the policy
    # differs by 6e-6 between two passes that differ only in when buffers were allocated), so loss and gradients are
    # compared within a tolerance
    res = {}
    for once in (False, True):
        torch.manual_seed(3)
        net = A.ChessNet().cuda()
        tr = A.Trainer(net)
        assert isinstance(tr.model, A.FusedTrainNet)
        tr.model.pack_once = once
        tr.model.keep_tower_output = True
        gg = torch.Generator(device="cuda").manual_seed(9)
        x = (torch.rand((16, 56, 12, 12), device="cuda", generator=gg) < 0.1).float()
        pol = torch.softmax(torch.randn((16, 1584), device="cuda", generator=gg), dim=1)
        val = torch.rand((16,), device="cuda", generator=gg) * 2 - 1
        tr.model.train()
        loss = tr.loss(x, pol, val)
        loss.backward()
        res[once] = (loss.detach().clone(), {k: p_.grad.clone() for k, p_ in net.named_parameters()}, tr.model.tower_output.clone())
    assert torch.equal(res[True][2], res[False][2])
    assert abs(res[True][0].item() - res[False][0].item()) <= 1e-5
    for k, a in res[True][1].items():
        if k.endswith("weight") and a.dim() == 4:      # (biases in front of a BatchNorm have a zero gradient up to rounding)
            b = res[False][1][k]
            assert (a - b).abs().max().item() <= 2e-2 * b.abs().max().item() + 1e-7, k


@pytest.mark.gpu
def test_batchnorm_statistics_added_up_by_the_convolution_launch():
    """conv3x3(..., stats=d) + bn_act(..., stats=d): the convolution's epilogue adds up the BatchNorm's per-channel sums
    (hive_nn_conv72_stats), the BatchNorm skips its own pass over the tensor (hive_nn_bn_act_fwd_partial).  Against the
    unfused pair on the same operands: the same convolution output, statistics equal to fp32 summation order (1e-5), the
    normalised output within one bf16 rounding, gradients likewise."""
    assert torch.cuda.is_available()
    from hive_alphazero_amd.alpha_net import ResBlock, bn_act, conv3x3
    torch.manual_seed(22)
    B = 12
    blk = ResBlock().cuda().to(memory_format=torch.channels_last)
    x = torch.randn((B, 256, 12, 12), device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn((B, 256, 12, 12), device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = {}
    for fused in (False, True):
        blk.zero_grad(set_to_none=True)
        blk.bn1.reset_running_stats()
        xi = x.clone().requires_grad_(True)
        d = {} if fused else None
        y = bn_act(conv3x3(xi, blk.conv1, stats=d), blk.bn1, stats=d)
        assert not d                                  # the partial sums were consumed
        y.backward(dy)
        res[fused] = (y.detach().float(), xi.grad.float(), blk.conv1.weight.grad.clone(), blk.bn1.weight.grad.clone(),
                      blk.bn1.running_mean.clone(), blk.bn1.running_var.clone())
    a, b = res[True], res[False]
    assert (a[4] - b[4]).abs().max().item() <= 1e-5 * b[4].abs().max().item() + 1e-7
    assert (a[5] - b[5]).abs().max().item() <= 1e-5 * b[5].abs().max().item()
    assert (a[0] - b[0]).abs().max().item() <= 2 ** -7 * b[0].abs().max().item()
    assert (a[0] != b[0]).float().mean().item() < 0.01          # (only values that sit on a rounding boundary move)
    assert (a[1] - b[1]).abs().max().item() <= 2 ** -6 * b[1].abs().max().item()
    assert (a[2] - b[2]).abs().max().item() <= 2 ** -6 * b[2].abs().max().item()
    assert (a[3] - b[3]).abs().max().item() <= 1e-2 * b[3].abs().max().item()


@pytest.mark.gpu
def test_skip_gradient_summed_inside_the_data_gradient_convolution():
    """A residual block's input reaches the output on two ways (conv1 and the skip connection).  Linked
    (FusedTrainNet.fuse_skip_grad), bn2's backward hands the skip gradient to conv1's backward, whose data-gradient launch
    adds it in fp32 before its one rounding (hive_nn_conv72_add) -- instead of autograd adding two bf16 tensors.  Against the
    unlinked operators on the same block: parameter gradients bit for bit (they do not depend on the sum), the input
    gradient within one bf16 rounding of the sum."""
    assert torch.cuda.is_available()
    from hive_alphazero_amd.alpha_net import ResBlock, bn_act, conv3x3
    torch.manual_seed(21)
    B = 10
    blk = ResBlock().cuda().to(memory_format=torch.channels_last)
    x = torch.randn((B, 256, 12, 12), device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn((B, 256, 12, 12), device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = {}
    for linked in (False, True):
        blk.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        s = xi * 1                                    # (a non-leaf input, as inside the tower)
        link = {} if linked else None
        out = bn_act(conv3x3(s, blk.conv1, link=link), blk.bn1)
        y = bn_act(conv3x3(out, blk.conv2), blk.bn2, residual=s, link=link)
        y.backward(dy)
        assert not link                               # the handed-over gradient was consumed
        res[linked] = (y.detach().clone(), xi.grad.clone(), [p.grad.clone() for p in blk.parameters()])
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][2], res[False][2]):
        assert torch.equal(a, b)
    gx_l, gx_u = res[True][1].float(), res[False][1].float()
    assert (gx_l - gx_u).abs().max().item() <= 2 ** -7 * gx_u.abs().max().item()
    assert not torch.equal(gx_l, gx_u)               # (one rounding instead of two: not the same bits)


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 8, 37, 512])
def test_hip_weight_gradient_matches_fp32_convolution_backward(batch):
    """hive_nn_conv3x3_wgrad (pixels as the MFMA contraction, both operands through transposing LDS reads, split over
    board ranges whose partial sums a second kernel adds) against aten.convolution_backward in fp32 on the same bf16 operands: every one of the
    9 x 256 x 256 entries.  Products of bf16 values are exact in fp32, so only the summation order differs:
    tolerance 2e-5 of the largest entry (sums of up to batch * 144 terms).  Batches 1 / 37 exercise single-board
    and ragged board ranges; the border taps (zero halo) carry full weight because every pixel is non-zero."""
    assert torch.cuda.is_available()
    import ctypes
    import hive_alphazero_amd as h
    L = h.load()
    g = torch.Generator(device="cuda").manual_seed(100 + batch)
    x = torch.randn((batch, 256, 12, 12), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn((batch, 256, 12, 12), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    taps = torch.full((3, 3, 256, 256), float("nan"), dtype=torch.float32, device="cuda")      # must be overwritten
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    ws = torch.full((L.hive_nn_wgrad_workspace_floats(),), float("nan"), dtype=torch.float32, device="cuda")    # scratch: any contents
    rc = L.hive_nn_conv3x3_wgrad(p(x), p(dy), p(taps), batch, p(ws), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, L.hive_last_error()
    torch.cuda.synchronize()
    got = taps.permute(2, 3, 0, 1)
    want = torch.ops.aten.convolution_backward(dy.float(), x.float(), torch.zeros((256, 256, 3, 3), device="cuda"), None, (1, 1), (1, 1),
                                               (1, 1), False, (0, 0), 1, (False, True, False))[1]
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item()
    assert err <= 2e-5 * want.abs().max().item() + 1e-6, (err, want.abs().max().item())
    # a second call overwrites dw, it does not accumulate, and the fixed summation order makes it bit-identical
    first = taps.clone()
    L.hive_nn_conv3x3_wgrad(p(x), p(dy), p(taps), batch, p(ws), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(taps, first)
    # the same sums in a torch.channels_last weight's memory order ([k][ty][tx][c], hive_nn_conv3x3_wgrad_layout 1)
    cl = torch.full((256, 3, 3, 256), float("nan"), dtype=torch.float32, device="cuda")
    assert L.hive_nn_conv3x3_wgrad_layout(p(x), p(dy), p(cl), batch, p(ws), 1, None) == 0, L.hive_last_error()
    torch.cuda.synchronize()
    assert torch.equal(cl.permute(1, 2, 0, 3), first)
    assert L.hive_nn_conv3x3_wgrad_layout(p(x), p(dy), p(cl), batch, p(ws), 2, None) == -1


@pytest.mark.gpu
def test_inference_net_refresh_keeps_graphs_and_takes_new_weights():
    """After a training iteration the evaluator takes the new weights in place (the reference's workers re-read the
    checkpoint, self_play.py:37-75): the captured graph of a batch size keeps replaying, now with the new network."""
    assert torch.cuda.is_available()
    from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
    torch.manual_seed(1)
    net_a = ChessNet().cuda().eval()
    torch.manual_seed(2)
    net_b = ChessNet().cuda().eval()
    with torch.no_grad():                      # make the BatchNorm statistics differ too
        for m in net_b.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.1, 0.1); m.running_var.uniform_(0.8, 1.2)
    x = (torch.rand((32, 12, 12, 56), device="cuda") < 0.1).to(torch.bfloat16)
    inf = InferenceNet(net_a)
    pa, va = inf(x)
    graphs = dict(inf._graphs)
    inf.refresh(net_b)
    pb, vb = inf(x)
    assert inf._graphs == graphs and 32 in inf.graph_batches      # no re-capture
    fresh_p, fresh_v = InferenceNet(net_b)(x)
    assert torch.equal(pb, fresh_p) and torch.equal(vb, fresh_v)
    assert (pa - pb).abs().max().item() > 1e-5                  # and it is not the old network any more
    # a large batch lets TunableOp pick the head GEMMs before the capture -- and hands the process-wide switch back
    import torch.cuda.tunable as tn
    was = tn.is_enabled()
    big = (torch.rand((256, 12, 12, 56), device="cuda") < 0.1).to(torch.bfloat16)
    p256, _ = inf(big)
    assert tn.is_enabled() == was and 256 in inf.graph_batches
    assert float((p256.sum(1) - 1).abs().max().item()) < 1e-3


def test_train_matches_reference_training_run_cpu():
    """alpha_net.train against the TRUE reference's train() (tests/golden/train.json, oracle/gen_golden.py train): same
    seeded init, the same 20 synthetic rows, one epoch of ten batches of two on the CPU in fp32 -- the reported loss
    (the reference prints it with three decimals) and sums / sums of squares of tensors of the trained network, incl.
    BatchNorm running statistics and the step counter."""
    from hive_alphazero_amd.alpha_net import ChessNet, train
    with open(os.path.join(GOLD, "train.json")) as f:
        gold = json.load(f)
    rng = np.random.default_rng(7)
    ds = np.empty((20, 3), dtype=object)
    for i in range(20):
        ds[i, 0] = (rng.random((12, 12, 56)) < 0.1).astype(np.float32)
        p = rng.random(1584).astype(np.float32)
        ds[i, 1] = p / p.sum()
        ds[i, 2] = float(rng.choice([-1.0, 1.0]))
    torch.manual_seed(gold["init_seed"])
    net = ChessNet()
    losses = train(net, ds, 0, gold["epochs"], cpu=gold["cpu"], batch_size=gold["batch_size"], log=lambda *_: None)
    assert abs(losses[0] - gold["printed_loss_3dp"]) < 1e-3
    sd = net.state_dict()
    for k, (s1, s2) in gold["digest"].items():
        t = sd[k].double()
        assert abs(float(t.sum()) - s1) <= 1e-4 * max(1.0, abs(s1)), k
        assert abs(float((t ** 2).sum()) - s2) <= 1e-4 * max(1.0, abs(s2)), k


# ------------------------------------------------------------------------------------------------------------------
# a22 on a signal larger than the tolerance: 64 golden positions x two weight sets of the TRUE reference's ChessNet
# (tests/golden/net_wide.npz, oracle/gen_golden.py net_wide): the seeded init, and the same init with peaked heads
# (policy logits x 30: max p 0.10 .. 0.88; value stretched over (-0.98, 0.98)).

def _wide():
    return np.load(os.path.join(GOLD, "net_wide.npz"))


def _wide_inputs(wide, rows=None):
    with gzip.open(os.path.join(GOLD, "games_full.json.gz"), "rt") as f:
        games = json.load(f)["games"]
    xs, legal = [], []
    for gi, ply in (wide["picks"] if rows is None else wide["picks"][rows]):
        rec = games[int(gi)]["plies"][int(ply)]
        pl = np.zeros((12, 12, 56), dtype=np.float32)
        pl.reshape(-1)[rec["planes"]] = 1.0
        pl[:, :, 31] = rec["t"]
        xs.append(pl)
        legal.append(np.asarray(rec["legal"], dtype=np.int64))
    return torch.from_numpy(np.stack(xs)), legal


def _wide_net(wide, tag):
    """The build's ChessNet with the fixture's weights: same-seed init (asserted identical to the reference's by
    oracle/gen_golden.py net), plus -- for "peak" -- the documented edit of the two heads (gen_golden.peak_state_dict)."""
    from hive_alphazero_amd.alpha_net import ChessNet
    torch.manual_seed(int(wide["seed"]))
    net = ChessNet().eval()
    if tag == "peak":
        ps, vs, vshift = (float(t) for t in wide["peak"])
        sd = net.state_dict()
        sd["outblock.fc.weight"] *= ps
        sd["outblock.fc.bias"] *= ps
        sd["outblock.fc2.weight"] *= vs
        sd["outblock.fc2.bias"].mul_(vs).add_(vshift)
        net.load_state_dict(sd)
    return net


def _wide_metrics(p, v, p_ref, v_ref, legal):
    """max |d log p| over the legal moves, max KL(p_ref || p), top-1 / top-5 agreement (ranks among the legal moves),
    max |dv|."""
    p, v = p.double().cpu().numpy(), v.double().cpu().numpy().reshape(-1)
    dlog, kl, top1, top5 = 0.0, 0.0, 0, 0
    for b, lg in enumerate(legal):
        if len(lg) == 0:
            continue
        pr, pm = p_ref[b].astype(np.float64), p[b]
        seen = lg[pr[lg] > 1e-30]                 # (peaked logits push a few legal moves below fp32's range: log 0 on both sides)
        with np.errstate(divide="ignore"):
            dlog = max(dlog, float(np.abs(np.log(pm[seen]) - np.log(pr[seen])).max()))
        kl = max(kl, float((pr * (np.log(pr + 1e-300) - np.log(pm + 1e-300))).sum()))
        order_r, order_m = lg[np.argsort(-pr[lg], kind="stable")], lg[np.argsort(-pm[lg], kind="stable")]
        top1 += int(order_r[0] == order_m[0])
        top5 += int(set(order_r[:5]) == set(order_m[:5]))
    n = sum(1 for lg in legal if len(lg))
    return {"dlogp": dlog, "kl": kl, "top1": top1 / n, "top5": top5 / n, "dv": float(np.abs(v - v_ref).max())}


@pytest.mark.parametrize("tag", ["init", "peak"])
def test_chessnet_matches_reference_wide_cpu_fp32(tag):
    """fp32 on the CPU reproduces the reference's outputs on 12 of the 64 positions (the full set runs on the GPU)."""
    wide = _wide()
    rows = np.arange(0, 64, 6)
    x, legal = _wide_inputs(wide, rows)
    net = _wide_net(wide, tag)
    with torch.no_grad():
        p, v = net(x.permute(0, 3, 1, 2))
    m = _wide_metrics(p, v, wide["p_" + tag][rows], wide["v_" + tag][rows], legal)
    assert m["dlogp"] < 2e-4 and m["kl"] < 1e-6 and m["top1"] == 1.0 and m["dv"] < 2e-5, m


# bounds of the leaf evaluator per weight set and engine: (max |d log p| on the legal moves, max KL(p_ref || p),
# min top-1 agreement, min top-5 agreement, max |dv|).  The peaked set multiplies the logits -- and with them every
# absolute logit error -- by 30, so its log-probability bounds are ~30 x those of the seeded init.
# Measured on MI355X (round 3): init  fp16 0.0034 / 6.5e-7 / 1 / 1 / 7e-4      bf16 0.025 / 2.9e-5 / 0.97 / 0.97 / 5.5e-3
#                              peak  fp16 0.052  / 9.7e-4 / 1 / 1 / 1.4e-3    bf16 0.47  / 0.046  / 0.98 / 0.95 / 0.031
# (max |d log p| over the legal moves, max KL, top-1 agreement, top-5 agreement, max |dv|) per engine: <= 1.3x the values
# measured in round 4 (hand-written heads: fp32 logits) -- init: fp16 0.00341 / 6.7e-7 / 1 / 1 / 7.5e-4, bf16 0.0248 / 2.6e-5 /
# 0.984 / 0.969 / 5.6e-3; peak: fp16 0.127 / 4.9e-4 / 1 / 1 / 1.4e-3, bf16 0.817 / 0.0222 / 0.984 / 0.969 / 0.0309.  The kernels are
# deterministic, so these are the numbers every box produces; the agreement floors leave room for one more of the 64 positions.
WIDE_BOUNDS = {
    "init": {("float32", "torch"): (5e-4, 1e-6, 1.0, 1.0, 1e-4),
             ("float16", "hip"): (4.5e-3, 9e-7, 0.98, 0.98, 1e-3),
             ("bfloat16", "hip"): (3.3e-2, 3.5e-5, 0.95, 0.93, 7.3e-3)},
    "peak": {("float32", "torch"): (2e-3, 1e-6, 1.0, 1.0, 1e-4),
             ("float16", "hip"): (0.17, 6.5e-4, 0.98, 0.98, 1.8e-3),
             ("bfloat16", "hip"): (1.06, 2.9e-2, 0.95, 0.93, 4.0e-2)},
}


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["init", "peak"])
def test_inference_net_wide_parity(tag):
    """The engines that are timed (bf16 / fp16 hand-written convolutions) and the fp32 path against the TRUE reference's
    outputs on all 64 positions, on metrics that mean something for a policy: log-probability error on the legal moves,
    KL divergence, agreement of the best and the five best legal moves, value error."""
    assert torch.cuda.is_available()
    from hive_alphazero_amd.alpha_net import InferenceNet
    wide = _wide()
    x, legal = _wide_inputs(wide)
    net = _wide_net(wide, tag).cuda()
    report = {}
    for (dt, conv) in WIDE_BOUNDS[tag]:
        inf = InferenceNet(net, dtype=getattr(torch, dt), conv=conv)
        p, v = inf(x.cuda())
        report[(dt, conv)] = m = _wide_metrics(p, v, wide["p_" + tag], wide["v_" + tag], legal)
        print(f"net_wide[{tag}] {dt}/{conv}: " + ", ".join(f"{k} {val:.3g}" for k, val in m.items()))
    for key, bound in WIDE_BOUNDS[tag].items():
        m = report[key]
        assert m["dlogp"] <= bound[0] and m["kl"] <= bound[1] and m["top1"] >= bound[2] and m["top5"] >= bound[3] \
            and m["dv"] <= bound[4], (tag, key, m, bound)
    # fp16 carries three more mantissa bits than bf16: it must be the closer engine
    assert report[("float16", "hip")]["dlogp"] < report[("bfloat16", "hip")]["dlogp"]


@pytest.mark.gpu
def test_reduced_precision_search_agrees_with_fp32_search():
    """What bf16 / fp16 leaf evaluation does to the SEARCH: 256 positions x 50 simulations, root noise off, the same
    trees searched with the fp32 evaluator (the reference's precision, api_hive.py:62-69) and with the bf16 / fp16
    engines, peaked weight set (a random-init network gives near-uniform priors: every move is then a near tie and
    the comparison says nothing).  Reported: fraction of positions with the same move, max and mean |d pi|."""
    assert torch.cuda.is_available()
    from hive_alphazero_amd import batch, mcts, playout
    from hive_alphazero_amd.alpha_net import InferenceNet
    wide = _wide()
    net = _wide_net(wide, "peak").cuda()
    G, sims = 256, 50
    boards = playout.random_positions(G, seed=11)
    B = batch.BoardBatch(G)
    B.import_state(boards)
    rb, rh = B.export_state()
    over, _ = B.terminal()
    live = (over == 0).cpu().numpy() & (rb[:, 33].cpu().numpy() < 55)
    res = {}
    for name, (dt, conv) in {"fp32": (torch.float32, "torch"), "fp16": (torch.float16, "hip"), "bf16": (torch.bfloat16, "hip")}.items():
        ts = mcts.TreeSearch(G, sims, InferenceNet(net, dtype=dt, conv=conv), plane_dtype=dt, noise_eps=0.0, seed=1)
        action, policy, _ = ts.search(rb, rh)
        res[name] = (action.cpu().numpy().copy(), policy.cpu().numpy().copy())
        ts.close()
    B.close()
    # measured (round 3): fp16 same move 1.000, mean per-position max |d pi| 0.0028; bf16 0.977 and 0.024
    floors = {"fp16": (0.97, 0.02), "bf16": (0.93, 0.08)}
    for name, (min_agree, max_dpi) in floors.items():
        agree = float((res[name][0][live] == res["fp32"][0][live]).mean())
        dpi = np.abs(res[name][1][live] - res["fp32"][1][live])
        print(f"search {name} vs fp32: same move {agree:.3f}, max |d pi| {dpi.max():.3f}, mean of per-position max {dpi.max(1).mean():.4f}")
        assert agree >= min_agree and dpi.max(1).mean() <= max_dpi, (name, agree, float(dpi.max()))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_hip_heads_match_fp32_math_on_the_same_operands(dtype):
    """hive_nn_heads (both 1x1 convolutions, the 18432 -> 1584 policy FC with softmax, the value MLP with tanh:
    alpha_net.py:56-80) through the C ABI against the same arithmetic in fp64 on the same 16-bit operands with the same
    rounding points (the convolutions' outputs rounded to the 16-bit type, everything else fp32 or better); a board's
    outputs must not depend on the batch it sits in nor on its position (bit for bit), and repeat exactly."""
    assert torch.cuda.is_available()
    import ctypes
    from hive_alphazero_amd import _lib
    from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
    L = _lib.load()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    DT = _lib.BF16 if dtype == torch.bfloat16 else _lib.F16
    torch.manual_seed(5)
    net = ChessNet().cuda().eval()
    with torch.no_grad():                      # heads with some contrast
        net.outblock.fc.weight.mul_(20.0)
        net.outblock.fc1.weight.mul_(5.0)
        for bn in (net.outblock.bn, net.outblock.bn1):
            bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.5, 1.5); bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
    inf = InferenceNet(net, dtype=dtype, use_graph=False)
    wc, bc, wf, bf_, w1, b1, w2, b2 = inf.h_heads
    gen = torch.Generator(device="cuda").manual_seed(6)

    def run(x):
        B = x.shape[0]
        ws = torch.empty((int(L.hive_nn_heads_workspace_bytes(B)),), dtype=torch.uint8, device="cuda")
        p = torch.empty((B, 1584), device="cuda")
        v = torch.empty((B,), device="cuda")
        _lib.check(L.hive_nn_heads(P(x), B, DT, P(wc), P(bc), P(wf), P(bf_), P(w1), P(b1), P(w2), P(b2), P(ws), P(p), P(v),
                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        return p, v

    x = torch.relu(torch.randn((37, 12, 12, 256), device="cuda", generator=gen)).to(dtype)
    p, v = run(x)
    # the same arithmetic in fp64 on the engine's own 16-bit weights
    flat = x.reshape(37 * 144, 256).double()
    pc = torch.relu((flat @ inf.pconv[0].double().t() + inf.pconv[1].double()).to(dtype)).double()     # bias as the engine holds it
    bp32 = bc[:128].double()
    pc = torch.relu((flat @ inf.pconv[0].double().t() + bp32).to(dtype)).double().reshape(37, 144 * 128)
    logits = pc @ inf.fc[0].double().t() + bf_.double()
    want_p = torch.softmax(logits, 1)
    vc = torch.relu((flat @ inf.vconv[0].double().t() + bc[128].double()).to(dtype)).double().reshape(37, 144)
    want_v = torch.tanh(torch.relu(vc @ w1.double() + b1.double()) @ w2.double() + b2.double())      # (w1 is stored transposed)
    dp = float((p.double() - want_p).abs().max())
    dl = float((torch.log(p.double() + 1e-30) - torch.log(want_p + 1e-30)).abs().max())
    dv = float((v.double() - want_v).abs().max())
    print(f"heads {dtype}: max |dp| {dp:.3g} (max p {float(want_p.max()):.3g}), max |d log p| {dl:.3g}, max |dv| {dv:.3g}")
    # (a convolution output that falls on a rounding boundary may round the other way in fp32: a few 16-bit ulps of one input)
    assert dl < 2e-2 and dv < 2e-3 and abs(float(p.sum(1).min()) - 1) < 1e-5
    p2, v2 = run(x)
    assert torch.equal(p, p2) and torch.equal(v, v2)
    # position and batch independence, bit for bit: the 37 boards inside a batch of 1031, shifted by 500 rows
    big = torch.relu(torch.randn((1031, 12, 12, 256), device="cuda", generator=gen)).to(dtype)
    big[500:537] = x
    pb, vb = run(big)
    assert torch.equal(pb[500:537], p) and torch.equal(vb[500:537], v)
    assert torch.isfinite(pb).all() and torch.isfinite(vb).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_hip_tower72_matches_resblock_chain(dtype):
    """hive_nn_tower72 (the 72-tile assembly tower: two boards per workgroup, 288 accumulator registers per lane,
    csrc/gen_tower_asm.py) against the launch-per-block chain hive_nn_resblock_dt, through the C ABI, bit for bit:
    one and several blocks, odd batches (the tail workgroup repeats its board and stores it once), more boards than one
    round of the chip, and with a row list (hive_nn_compact_rows of random need flags: unlisted boards keep their bytes)."""
    assert torch.cuda.is_available()
    import ctypes
    from hive_alphazero_amd import _lib
    L = _lib.load()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    DT = _lib.BF16 if dtype == torch.bfloat16 else _lib.F16
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    gen = torch.Generator(device="cuda").manual_seed(9)
    for B, nblk, use_rows in ((2, 1, False), (5, 1, False), (7, 2, False), (64, 3, True), (1, 2, False), (600, 19, True), (1031, 4, False)):
        x = torch.relu(torch.randn((B, 144, 256), device="cuda", generator=gen)).to(dtype)
        w = (torch.randn((2 * nblk, 9 * 8 * 16 * 64 * 8), device="cuda", generator=gen) * 0.015).to(dtype)
        bias = torch.randn((2 * nblk, 256), device="cuda", generator=gen) * 0.1
        bufs = [x, torch.zeros_like(x), torch.zeros_like(x)]
        cur = 0
        for i in range(nblk):
            nxt = 1 if cur != 1 else 2
            _lib.check(L.hive_nn_resblock_dt(P(bufs[cur]), P(w[2 * i]), P(bias[2 * i]), P(w[2 * i + 1]), P(bias[2 * i + 1]),
                                             P(bufs[nxt]), B, DT, st()))
            cur = nxt
        want = bufs[cur]
        y = torch.full_like(x, 7.0)
        rows = nrows = need = None
        if use_rows:
            need = (torch.rand((B,), device="cuda", generator=gen) < 0.9).to(torch.int8)
            rows = torch.full((B,), -1, dtype=torch.int32, device="cuda")
            nrows = torch.zeros((1,), dtype=torch.int32, device="cuda")
            _lib.check(L.hive_nn_compact_rows(P(need), B, P(rows), P(nrows), st()))
            k = int(nrows.item())
            assert k == int(need.sum().item()) and torch.equal(rows[:k].long(), torch.nonzero(need).flatten())
        _lib.check(L.hive_nn_tower72(P(x), P(w), P(bias), P(y), B, nblk, DT, P(rows), P(nrows), st()))
        torch.cuda.synchronize()
        if use_rows:
            sel = need.bool()
            assert torch.equal(y[sel], want[sel]) and bool((y[~sel] == 7.0).all()), (B, nblk)
        else:
            assert torch.equal(y, want), (B, nblk, int((y != want).sum()))
    # the balanced launch (one workgroup per CU, the pairs' blocks dealt evenly: pairs cut at a share boundary are handed
    # from one workgroup to the next through y and a release / acquire flag): same bits, with and without a row list,
    # fewer pairs than CUs, more pairs than CUs with every kind of remainder
    for B, nblk, use_rows in ((7, 2, False), (300, 19, True), (700, 19, False), (1024, 19, True), (1031, 3, False), (2500, 5, True)):
        x = torch.relu(torch.randn((B, 144, 256), device="cuda", generator=gen)).to(dtype)
        w = (torch.randn((2 * nblk, 9 * 8 * 16 * 64 * 8), device="cuda", generator=gen) * 0.015).to(dtype)
        bias = torch.randn((2 * nblk, 256), device="cuda", generator=gen) * 0.1
        rows = nrows = need = None
        if use_rows:
            need = (torch.rand((B,), device="cuda", generator=gen) < 0.9).to(torch.int8)
            rows = torch.full((B,), -1, dtype=torch.int32, device="cuda")
            nrows = torch.zeros((1,), dtype=torch.int32, device="cuda")
            _lib.check(L.hive_nn_compact_rows(P(need), B, P(rows), P(nrows), st()))
        want = torch.full_like(x, 7.0)
        _lib.check(L.hive_nn_tower72(P(x), P(w), P(bias), P(want), B, nblk, DT, P(rows), P(nrows), st()))      # (checked against the chain above)
        y = torch.full_like(x, 7.0)
        ws = torch.empty((int(L.hive_nn_tower72_plan_bytes(B)),), dtype=torch.uint8, device="cuda")
        for _ in range(2):                                            # (the plan and its flags are rebuilt by every launch)
            _lib.check(L.hive_nn_tower72_balanced(P(x), P(w), P(bias), P(y), B, nblk, DT, P(rows), P(nrows), P(ws), st()))
            torch.cuda.synchronize()
            assert torch.equal(y, want), (B, nblk, use_rows, int((y != want).sum()))
    # ONE convolution on the same kernel (hive_nn_conv72: what the training step's forward / data-gradient convolutions use)
    # == hive_nn_conv3x3_dt, with and without ReLU, odd and even batches, negative outputs kept when there is no ReLU
    for B in (1, 6, 513):
        x = torch.randn((B, 144, 256), device="cuda", generator=gen).to(dtype)
        w1 = (torch.randn((9 * 8 * 16 * 64 * 8,), device="cuda", generator=gen) * 0.015).to(dtype)
        b1 = torch.randn((256,), device="cuda", generator=gen) * 0.1
        for relu in (0, 1):
            want, got = torch.full_like(x, 7.0), torch.full_like(x, 7.0)
            _lib.check(L.hive_nn_conv3x3_dt(P(x), 256, P(w1), P(b1), None, P(want), B, relu, DT, st()))
            _lib.check(L.hive_nn_conv72(P(x), P(w1), P(b1), P(got), B, relu, DT, st()))
            torch.cuda.synchronize()
            assert torch.equal(got, want), (B, relu, int((got != want).sum()))
            assert relu or bool((got.float() < 0).any())
            # ... + residual (hive_nn_conv72_add), into a third buffer and in place over the residual
            r = torch.randn((B, 144, 256), device="cuda", generator=gen).to(dtype)
            want = torch.full_like(x, 7.0)
            _lib.check(L.hive_nn_conv3x3_dt(P(x), 256, P(w1), P(b1), P(r), P(want), B, relu, DT, st()))
            got = torch.full_like(x, 7.0)
            _lib.check(L.hive_nn_conv72_add(P(x), P(w1), P(b1), P(r), P(got), B, relu, DT, st()))
            torch.cuda.synchronize()
            assert torch.equal(got, want), ("add", B, relu, int((got != want).sum()))
            inplace = r.clone()
            _lib.check(L.hive_nn_conv72_add(P(x), P(w1), P(b1), P(inplace), P(inplace), B, relu, DT, st()))
            torch.cuda.synchronize()
            assert torch.equal(inplace, want), ("add in place", B, relu, int((inplace != want).sum()))
    assert L.hive_nn_conv72_add(P(x), P(w1), P(b1), None, P(got), B, 0, DT, st()) != 0
    # ... with the BatchNorm statistics of the output added up by the launch (hive_nn_conv72_stats): same output bits; the
    # partial sums of all workgroups together = the per-channel sum / sum of squares of the STORED values (fp32 sums of up
    # to 144 * B terms in another order: 1e-5 relative to the sum of magnitudes)
    for B in (2, 6, 512):
        x = torch.randn((B, 144, 256), device="cuda", generator=gen).to(dtype)
        want, got = torch.full_like(x, 7.0), torch.full_like(x, 7.0)
        part = torch.full((B // 2, 2, 256), float("nan"), dtype=torch.float32, device="cuda")
        _lib.check(L.hive_nn_conv72(P(x), P(w1), P(b1), P(want), B, 0, DT, st()))
        _lib.check(L.hive_nn_conv72_stats(P(x), P(w1), P(b1), P(got), B, 0, DT, P(part), st()))
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        yf = want.double().reshape(-1, 256)
        s1, s2 = part[:, 0].double().sum(0), part[:, 1].double().sum(0)
        assert torch.isfinite(part).all()
        assert (s1 - yf.sum(0)).abs().max().item() <= 1e-5 * yf.abs().sum(0).max().item()
        assert (s2 - (yf * yf).sum(0)).abs().max().item() <= 1e-5 * (yf * yf).sum(0).max().item()
        # every workgroup's row = its own two boards
        assert (part[0, 0].double() - want[:2].double().reshape(-1, 256).sum(0)).abs().max().item() <= 1e-4 * want[:2].double().abs().reshape(-1, 256).sum(0).max().item()
    assert L.hive_nn_conv72_stats(P(x), P(w1), P(b1), P(got), 5, 0, DT, P(part), st()) != 0      # an odd batch is refused
    # arguments: rows without a count, aliased output
    assert L.hive_nn_tower72(P(x), P(w), P(bias), P(y), B, nblk, DT, P(y), None, st()) != 0
    assert L.hive_nn_tower72(P(x), P(w), P(bias), P(x), B, nblk, DT, None, None, st()) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_inference_net_tower_forms_give_the_same_bits(dtype):
    """InferenceNet(tower=1|2|3) (the whole residual tower in one hive_nn_tower launch, three workgroup forms) against the
    default launch-per-block chain: identical policy and value outputs, eagerly and through the captured HIP graph, and
    after refresh() with new weights (the one weight buffer the tower reads and the per-block views share storage)."""
    assert torch.cuda.is_available()
    from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
    torch.manual_seed(1)
    net = ChessNet().cuda().eval()
    x = (torch.rand((37, 12, 12, 56), device="cuda") < 0.08).to(dtype)
    x[..., 31] = 17.0
    ref = InferenceNet(net, dtype=dtype, tower=0)
    p0, v0 = ref(x)
    # leaf batches >= 512 run as two half-batch chains on two streams: same bits as one chain, eagerly and replayed
    xl = (torch.rand((640, 12, 12, 56), device="cuda") < 0.08).to(dtype)
    one = InferenceNet(net, dtype=dtype, tune_gemms=False, tower=0)
    one.split_streams = False
    two = InferenceNet(net, dtype=dtype, tune_gemms=False, tower=0)
    assert two.split_streams
    pa, va = one(xl)
    for _ in range(3):
        pb, vb = two(xl)
        assert torch.equal(pa, pb) and torch.equal(va, vb)
    # more than 1024 rows: chains of 512 boards (the last one shorter) dealt onto the two streams; an odd row count too
    for rows in (2304, 1025):
        xb = (torch.rand((rows, 12, 12, 56), device="cuda") < 0.08).to(dtype)
        pa, va = one(xb)
        for _ in range(2):
            pb, vb = two(xb)
            assert torch.equal(pa, pb) and torch.equal(va, vb), rows
    # the default ("auto") takes the 72-tile assembly tower for batches that fill its rounds: same bits as the chain
    auto = InferenceNet(net, dtype=dtype, tune_gemms=False)
    assert auto._tower_form(1024) == 72 and auto._tower_form(640) == 72 and auto._tower_form(400) == 0 and auto._tower_form(37) == 0
    xb = (torch.rand((1000, 12, 12, 56), device="cuda") < 0.08).to(dtype)
    pa, va = one(xb)
    for _ in range(2):
        pb, vb = auto(xb)
        assert torch.equal(pa, pb) and torch.equal(va, vb)
    engines = {t: InferenceNet(net, dtype=dtype, tower=t) for t in (1, 2, 3, 72)}
    for t, inf in engines.items():
        p, v = inf(x)
        assert torch.equal(p, p0) and torch.equal(v, v0), t
        p, v = inf(x)                                  # graph replay
        assert torch.equal(p, p0) and torch.equal(v, v0), t
    torch.manual_seed(2)
    net2 = ChessNet().cuda().eval()
    ref.refresh(net2)
    p1, v1 = ref(x)
    assert not torch.equal(p1, p0)
    for t, inf in engines.items():
        inf.refresh(net2)
        p, v = inf(x)
        assert torch.equal(p, p1) and torch.equal(v, v1), t


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_inference_net_row_selection_keeps_the_selected_rows_bit_identical(dtype):
    """InferenceNet(planes, need=flags): the tower's kernels skip the boards flagged 0 (hive_nn_conv3x3_sel /
    hive_nn_resblock_sel; what the reference does by never calling its model for a finished or capped leaf,
    solo_play.py:169-197).  The rows flagged 1 must carry exactly the bits of the unselected forward -- eagerly, through
    the captured graph, on the two-stream form (>= 512 rows) and when the selection changes between replays -- and the
    skipped rows must stay finite."""
    assert torch.cuda.is_available()
    from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
    torch.manual_seed(3)
    net = ChessNet().cuda().eval()
    gen = torch.Generator(device="cuda").manual_seed(5)
    for B, form in ((37, 0), (640, 0), (45, 72), (1000, 72)):      # (72: the assembly tower takes the rows as a compacted list)
        x = (torch.rand((B, 12, 12, 56), device="cuda", generator=gen) < 0.08).to(dtype)
        full = InferenceNet(net, dtype=dtype, tune_gemms=False, tower=0)
        p0, v0 = full(x)
        for use_graph in (False, True):
            inf = InferenceNet(net, dtype=dtype, tune_gemms=False, use_graph=use_graph, tower=form)
            for frac in (0.9, 0.5, 1.0, 0.0, 0.97):
                need = (torch.rand((B,), device="cuda", generator=gen) < frac).to(torch.int8)
                p, v = inf(x, need=need)
                sel = need.bool()
                assert torch.equal(p[sel], p0[sel]) and torch.equal(v[sel], v0[sel]), (B, use_graph, frac)
                assert torch.isfinite(p).all() and torch.isfinite(v).all()
            p, v = inf(x)                                     # and back to every row
            assert torch.equal(p, p0) and torch.equal(v, v0)
    # the selection really skips work: all rows off must be much cheaper than all rows on
    inf = InferenceNet(net, dtype=dtype, tune_gemms=False)
    x = (torch.rand((1024, 12, 12, 56), device="cuda", generator=gen) < 0.08).to(dtype)
    on, off = torch.ones(1024, dtype=torch.int8, device="cuda"), torch.zeros(1024, dtype=torch.int8, device="cuda")
    def ms(need):
        inf(x, need=need)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            inf(x, need=need)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 3
    t_on, t_off = ms(on), ms(off)
    print(f"1024-row forward: every row {t_on:.3f} ms, no row {t_off:.3f} ms")
    assert t_off < 0.25 * t_on


@pytest.mark.gpu
def test_inference_net_equal_rows_take_their_representatives_tower_output():
    """InferenceNet(planes, need, rep): a duplicate row (switched off in `need`) gets the tower output of its
    representative copied in before the heads (hive_nn_copy_rows), so EVERY row's p / v -- also a duplicate's -- carry the
    bits of the plain forward: the head GEMMs still see each row at its own position (their results depend on the row
    position at the last ulp: tools/row_position.py)."""
    assert torch.cuda.is_available()
    from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
    torch.manual_seed(4)
    net = ChessNet().cuda().eval()
    gen = torch.Generator(device="cuda").manual_seed(6)
    for B, use_graph, form in ((96, False, 0), (1024, True, "auto"), (88, False, 72)):
        base = (torch.rand((B // 8, 12, 12, 56), device="cuda", generator=gen) < 0.08).to(torch.bfloat16)
        which = torch.randint(0, B // 8, (B,), device="cuda", generator=gen)
        x = base[which]
        w = which.cpu().numpy()
        first = {}
        rep = np.arange(B, dtype=np.int32)
        for i in range(B):
            rep[i] = first.setdefault(int(w[i]), i)
        need = torch.from_numpy((rep == np.arange(B)).astype(np.int8)).cuda()
        trep = torch.from_numpy(rep).cuda()
        plain = InferenceNet(net, dtype=torch.bfloat16, use_graph=use_graph, tower=0)
        p0, v0 = plain(x)
        inf = InferenceNet(net, dtype=torch.bfloat16, use_graph=use_graph, tower=form)
        for _ in range(2):
            p, v = inf(x, need=need, rep=trep)
            assert torch.equal(p, p0) and torch.equal(v, v0), (B, use_graph)
        assert int(need.sum()) <= B // 8
        p, v = inf(x)                                          # back to the plain call on the same graph
        assert torch.equal(p, p0) and torch.equal(v, v0)
