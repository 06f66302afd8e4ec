"""ChessNet: mirror of the reference's alpha_zero/alpha_net.py (:25-95) + an MI355X inference engine.

`ChessNet` keeps the reference's module names, construction order (so torch.manual_seed(k)
yields the same initial tensors) and forward arithmetic, hence the same 255 state_dict keys:
checkpoints ({'state_dict': ...}, alpha_zero/train.py:35-51) load both ways.

`InferenceNet` is how leaf batches are evaluated on the GPU: eval-mode BatchNorm folded into the
convolutions, bf16 (or fp16/fp32) weights, channels-last activations so the 3x3 convolutions
run as NHWC implicit GEMMs on the MFMA units (MIOpen / hipBLASLt), softmax in fp32, and the
whole forward captured in a HIP graph per batch size.  It consumes the planes exactly as the env
kernels emit them ([B,12,12,56] channels-last), so there is no transpose between encode and
conv1.
"""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from .config import LOSS_WEIGHT, MAX_MAP_FULL, STATE_FEATURES

ACTIONS = MAX_MAP_FULL * MAX_MAP_FULL * 11


class board_data(torch.utils.data.Dataset):            # alpha_net.py:14-23
    def __init__(self, dataset):
        self.X = dataset[:, 0]
        self.y_p, self.y_v = dataset[:, 1], dataset[:, 2]

    def __len__(self):
        return len(self.X)

    def __getitem__(self, idx):
        return self.X[idx].transpose(2, 0, 1), self.y_p[idx], self.y_v[idx]


class ConvBlock(nn.Module):                             # alpha_net.py:25-34
    def __init__(self):
        super().__init__()
        self.action_size = ACTIONS
        self.conv1 = nn.Conv2d(STATE_FEATURES, 256, 3, stride=1, padding=1)
        self.bn1 = nn.BatchNorm2d(256)

    def forward(self, s):
        return F.relu(self.bn1(self.conv1(s)))


class ResBlock(nn.Module):                              # alpha_net.py:36-54
    def __init__(self, inplanes=256, planes=256, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        out = out + x
        return F.relu(out)


class OutBlock(nn.Module):                              # alpha_net.py:56-80
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(256, 1, kernel_size=1)                    # value head
        self.bn = nn.BatchNorm2d(1)
        self.fc1 = nn.Linear(MAX_MAP_FULL * MAX_MAP_FULL, 64)
        self.fc2 = nn.Linear(64, 1)
        self.conv1 = nn.Conv2d(256, 128, kernel_size=1)                 # policy head
        self.bn1 = nn.BatchNorm2d(128)
        self.logsoftmax = nn.LogSoftmax(dim=1)
        self.fc = nn.Linear(MAX_MAP_FULL * MAX_MAP_FULL * 128, ACTIONS)

    def forward(self, s):
        v = F.relu(self.bn(self.conv(s)))
        v = v.reshape(-1, MAX_MAP_FULL * MAX_MAP_FULL)
        v = F.relu(self.fc1(v))
        v = torch.tanh(self.fc2(v))
        p = F.relu(self.bn1(self.conv1(s)))
        p = p.reshape(-1, MAX_MAP_FULL * MAX_MAP_FULL * 128)
        p = self.fc(p)
        p = self.logsoftmax(p).exp()
        return p, v


class ChessNet(nn.Module):                              # alpha_net.py:82-95
    def __init__(self):
        super().__init__()
        self.conv = ConvBlock()
        for block in range(19):
            setattr(self, "res_%i" % block, ResBlock())
        self.outblock = OutBlock()

    def forward(self, s):
        s = self.conv(s)
        for block in range(19):
            s = getattr(self, "res_%i" % block)(s)
        return self.outblock(s)


class AlphaLoss(nn.Module):                             # alpha_net.py:98-115
    def forward(self, y_value, value, y_policy, policy):
        value_error = (value - y_value) ** 2
        policy_error = torch.sum((-policy * (1e-6 + y_policy.float()).float().log()), 1)
        return (value_error.view(-1).float() * LOSS_WEIGHT["value"] + policy_error * LOSS_WEIGHT["policy"]).mean()


def train(net, dataset, epoch_start=0, epoch_stop=10, cpu=0, batch_size=512, lr=0.001, log=print):
    """alpha_net.py:117-162 (Adam 1e-3, MultiStepLR [100,200,300,400] x0.2, AlphaLoss); the loss plot
    is left to the caller.  Returns the per-epoch mean losses."""
    torch.manual_seed(cpu)
    dev = next(net.parameters()).device
    net.train()
    criterion = AlphaLoss()
    optimizer = torch.optim.Adam(net.parameters(), lr=lr)
    scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[100, 200, 300, 400], gamma=0.2)
    loader = torch.utils.data.DataLoader(board_data(dataset), batch_size=batch_size, shuffle=True, num_workers=0)
    losses_per_epoch = []
    for epoch in range(epoch_start, epoch_stop):
        total, nb = 0.0, 0
        for state, policy, value in loader:
            state, policy, value = state.to(dev).float(), policy.float().to(dev), value.to(dev).float()
            optimizer.zero_grad()
            policy_pred, value_pred = net(state)
            loss = criterion(value_pred[:, 0], value, policy_pred, policy)
            loss.backward()
            optimizer.step()
            total += loss.item()
            nb += 1
        scheduler.step()
        losses_per_epoch.append(total / max(nb, 1))
        log("epoch %d loss %.4f" % (epoch + 1, losses_per_epoch[-1]))
        if len(losses_per_epoch) > 100 and \
                abs(sum(losses_per_epoch[-4:-1]) / 3 - sum(losses_per_epoch[-16:-13]) / 3) <= 0.01:
            break
    return losses_per_epoch


def _cl_rows(t):
    """[B,C,12,12] channels-last bf16 tensor -> (contiguous-as-[B*144][C] tensor, rows)."""
    if t.dtype != torch.bfloat16 or not t.is_contiguous(memory_format=torch.channels_last):
        t = t.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    return t, t.shape[0] * t.shape[2] * t.shape[3]


class _BNAct(torch.autograd.Function):
    """Training-mode BatchNorm2d(C) [+ skip] [+ ReLU] on the hand-written HIP kernels (csrc/hive_train.hip,
    include/hive_nn.h): bf16 channels-last activations, fp32 statistics and parameter gradients.  C = 256 (the tower) or a
    smaller power of two (the heads' 128 and 1; bn_act_ok tells whether a shape qualifies)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, bn, relu, track=True, link=None, stats=None):
        """link: a dict shared with the _Conv3x3 that consumes the same `residual` tensor (a residual block's first
        convolution): the backward then leaves the skip gradient there instead of returning it, and that convolution's
        data gradient adds it in its epilogue (one pass instead of a convolution and autograd's elementwise add)."""
        from . import _lib
        L = _lib.load()
        x, rows = _cl_rows(x)
        res = None
        if residual is not None:
            res, _ = _cl_rows(residual)
        dev = x.device
        C = x.shape[1]
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty(C, dtype=torch.float32, device=dev)
        ws = torch.empty(L.hive_nn_bn_workspace_floats(), dtype=torch.float32, device=dev)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        part = stats.pop("partial", None) if stats is not None else None
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if part is not None and C == 256:      # the convolution in front has added up its output already (hive_nn_conv72_stats)
            _lib.check(L.hive_nn_bn_act_fwd_partial(p(x), p(res), p(gamma), p(beta), p(bn.running_mean), p(bn.running_var),
                                                    float(bn.momentum), float(bn.eps), p(y), p(mean), p(invstd), p(ws), p(part),
                                                    part.shape[0], rows, int(relu), st))
        else:
            _lib.check(L.hive_nn_bn_act_fwd(p(x), p(res), p(gamma), p(beta), p(bn.running_mean), p(bn.running_var),
                                            float(bn.momentum), float(bn.eps), p(y), p(mean), p(invstd), p(ws), rows, C,
                                            int(relu), st))
        if track:
            bn.num_batches_tracked += 1
        ctx.save_for_backward(x, y, gamma, mean, invstd)
        ctx.relu, ctx.has_res, ctx.rows = bool(relu), residual is not None, rows
        ctx.link = link if residual is not None else None
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        L = _lib.load()
        x, y, gamma, mean, invstd = ctx.saved_tensors
        dy, _ = _cl_rows(dy)
        dev = x.device
        C = x.shape[1]
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        ws = torch.empty(L.hive_nn_bn_workspace_floats(), dtype=torch.float32, device=dev)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(L.hive_nn_bn_act_bwd(p(dy), p(x), p(y), p(gamma), p(mean), p(invstd), p(dx), p(dres), p(dgamma),
                                        p(dbeta), p(ws), ctx.rows, C, int(ctx.relu),
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        if ctx.link is not None:
            ctx.link["dres"] = dres                 # picked up (and added to its own dx) by the linked convolution's backward
            dres = None
        return dx, dgamma, dbeta, dres, None, None, None, None, None


_CONST_CACHE = {}


def _const(kind, cin, dev):
    """Per-device constants of the training convolutions: a zero bias, and a bf16 weight-shaped tensor for the
    library's weight-gradient call (it reads only the shape and memory format of that argument)."""
    key = (kind, cin, dev)
    if key not in _CONST_CACHE:
        if kind == "zero_bias":
            _CONST_CACHE[key] = torch.zeros(256, dtype=torch.float32, device=dev)
        else:
            _CONST_CACHE[key] = torch.zeros((256, cin, 3, 3), dtype=torch.bfloat16, device=dev).contiguous(
                memory_format=torch.channels_last)
    return _CONST_CACHE[key]


class _Conv3x3(torch.autograd.Function):
    """3x3 convolution of the training step on the hand-written MFMA kernels: forward and data gradient are
    hive_nn_conv3x3 (the data gradient = the same kernel on dy with transposed, 180-degree-rotated weights, packed by
    hive_nn_pack_conv3x3_weights); the weight gradient of the 256 -> 256 convolutions is hive_nn_conv3x3_wgrad (pixels as
    the contraction dimension, transposing LDS reads); only the 56-channel stem's stays on the library (MIOpen) path."""
    hip_wgrad = True
    asm_conv = True          # forward / data gradient of the 256 -> 256 convolutions on hive_nn_conv72

    @staticmethod
    def forward(ctx, x, weight, bias, packed=None, packed_t=None, link=None, stats=None):
        """packed / packed_t: this weight already in the kernels' layout (forward / data-gradient form), as
        FusedTrainNet packs all tower convolutions of a step in one launch; None: packed here.  link: see _BNAct."""
        from . import _lib
        L = _lib.load()
        if x.dtype != torch.bfloat16 or not x.is_contiguous(memory_format=torch.channels_last):
            x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        B, cin = x.shape[0], x.shape[1]
        dev = x.device
        cinp = (cin + 63) // 64 * 64
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        if packed is not None:
            wp = packed
        else:
            wp = torch.empty(9 * cinp * 256, dtype=torch.bfloat16, device=dev)
            wsrc, wcl = _weight_layout(weight)
            _lib.check(L.hive_nn_pack_conv3x3_weights(p(wsrc), cin, 0, wcl, p(wp), st))
        ctx.packed_t = packed_t
        ctx.link = link
        b = bias if bias is not None else _const("zero_bias", 0, dev)
        y = torch.empty((B, 256, 12, 12), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last)
        if cin == 256 and _Conv3x3.asm_conv and stats is not None and B % 2 == 0:
            # ... and the BatchNorm behind it gets its per-channel sums from this launch (stats: a dict shared with that bn_act)
            part = torch.empty((B // 2, 2, 256), dtype=torch.float32, device=dev)
            _lib.check(L.hive_nn_conv72_stats(p(x), p(wp), p(b), p(y), B, 0, _lib.BF16, p(part), st))
            stats["partial"] = part
        elif cin == 256 and _Conv3x3.asm_conv:       # the 72-tile assembly kernel (two boards per workgroup): same bits
            _lib.check(L.hive_nn_conv72(p(x), p(wp), p(b), p(y), B, 0, _lib.BF16, st))
        else:
            _lib.check(L.hive_nn_conv3x3(p(x), cin, p(wp), p(b), None, p(y), B, 0, st))
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        L = _lib.load()
        x, weight = ctx.saved_tensors
        if dy.dtype != torch.bfloat16 or not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        B, cin = x.shape[0], x.shape[1]
        dev = x.device
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        dx = None
        if ctx.needs_input_grad[0]:
            wt = ctx.packed_t
            if wt is None:
                wt = torch.empty(9 * 256 * 256, dtype=torch.bfloat16, device=dev)
                wsrc, wcl = _weight_layout(weight)
                _lib.check(L.hive_nn_pack_conv3x3_weights(p(wsrc), cin, 1, wcl, p(wt), st))
            extra = ctx.link.pop("dres", None) if ctx.link is not None else None      # the skip gradient of the same input
            if extra is not None and cin == 256 and _Conv3x3.asm_conv:
                dx = extra                                  # dx = dgrad(dy) + dskip, summed in fp32, written over dskip
                _lib.check(L.hive_nn_conv72_add(p(dy), p(wt), p(_const("zero_bias", 0, dev)), p(extra), p(dx), B, 0, _lib.BF16, st))
                extra = None
            else:
                dx = torch.empty_like(x)
                if cin == 256 and _Conv3x3.asm_conv:
                    _lib.check(L.hive_nn_conv72(p(dy), p(wt), p(_const("zero_bias", 0, dev)), p(dx), B, 0, _lib.BF16, st))
                else:
                    _lib.check(L.hive_nn_conv3x3(p(dy), 256, p(wt), p(_const("zero_bias", 0, dev)), None, p(dx), B, 0, st))
            if extra is not None:
                dx = dx + extra
        if cin == 256 and _Conv3x3.hip_wgrad:
            if weight.is_contiguous(memory_format=torch.channels_last) and not weight.is_contiguous():
                # the gradient in the parameter's own memory order [k][ty][tx][c]: autograd keeps it as it is (a gradient with
                # other strides is copied into the parameter's layout when it is accumulated: 38 strided copies per step)
                dw = torch.empty_strided((256, 256, 3, 3), weight.stride(), dtype=torch.float32, device=dev)
                _lib.check(L.hive_nn_conv3x3_wgrad_layout(p(x), p(dy), p(dw), B, p(_wgrad_workspace(L, dev)), 1, st))
            else:
                taps = torch.empty((3, 3, 256, 256), dtype=torch.float32, device=dev)
                _lib.check(L.hive_nn_conv3x3_wgrad(p(x), p(dy), p(taps), B, p(_wgrad_workspace(L, dev)), st))
                dw = taps.permute(2, 3, 0, 1)                  # [k][c][ty][tx] as a view of [ty][tx][k][c]
        else:
            dw = torch.ops.aten.convolution_backward(dy, x, _const("weight_like", cin, dev), None, (1, 1), (1, 1), (1, 1), False,
                                                     (0, 0), 1, (False, True, False))[1].float()
        db = dy.float().sum(dim=(0, 2, 3)) if ctx.has_bias else None
        return dx, dw, db, None, None, None, None


def _wgrad_workspace(L, dev):
    """Per-device partial-sum scratch of hive_nn_conv3x3_wgrad (include/hive_nn.h)."""
    key = ("wgrad_ws", dev.index)
    if key not in _CONST_CACHE:
        _CONST_CACHE[key] = torch.empty((L.hive_nn_wgrad_workspace_floats(),), dtype=torch.float32, device=dev)
    return _CONST_CACHE[key]


def _weight_layout(w):
    """(tensor whose storage the pack kernel can index, channels_last flag) for an nn.Conv2d weight."""
    if w.is_contiguous():
        return w, 0
    if w.is_contiguous(memory_format=torch.channels_last):
        return w, 1
    return w.contiguous(), 0


def conv3x3(x, conv, packed=None, packed_t=None, link=None, stats=None):
    """conv(x) for a 3x3 / stride 1 / padding 1 nn.Conv2d with 256 output channels, through the HIP kernel.  stats: a dict
    shared with the bn_act that normalises the result (the convolution launch then adds up the BatchNorm statistics)."""
    return _Conv3x3.apply(x, conv.weight, conv.bias, packed, packed_t, link, stats)


def bn_act_ok(x):
    """Whether the HIP BatchNorm kernels take this [B, C, H, W] activation: C a power of two <= 256 and the
    [B*H*W][C] matrix a whole number of 256-wide rows (include/hive_nn.h)."""
    C = x.shape[1]
    return x.is_cuda and 1 <= C <= 256 and (C & (C - 1)) == 0 and (x.shape[0] * x.shape[2] * x.shape[3] * C) % 256 == 0


def bn_act(x, bn, residual=None, relu=True, track=True, link=None, stats=None):
    """relu(bn(x) + residual) in training mode through the HIP kernels (updates bn's running statistics; track=False
    leaves num_batches_tracked to the caller, who can bump all layers' counters in one launch; link: see _BNAct)."""
    return _BNAct.apply(x, bn.weight, bn.bias, residual, bn, relu, track, link, stats)


class FusedTrainNet(nn.Module):
    """ChessNet.forward for the training step on MI355X: same modules and parameters (so the optimizer, DDP and the
    state_dict see the reference's network), but every BatchNorm + skip + ReLU of the 256-channel tower is one fused
    HIP forward / backward (alpha_net.py:25-54); the two heads stay on the library path."""

    pack_once = True         # the 38 tower convolutions' weights packed by ONE launch per step (both forms)
    hip_head_bn = True       # the heads' BatchNorm2d(1) / BatchNorm2d(128) + ReLU on the HIP kernels too
    fuse_skip_grad = True    # a block input's two gradients (conv1's data gradient + the skip's) summed in the convolution launch
    fuse_bn_stats = True     # a tower convolution adds up the statistics of the BatchNorm behind it (no separate pass over its output)
    keep_tower_output = False  # tests: keep the last forward's tower output (res_18's result) in self.tower_output

    def __init__(self, net, hip_conv=True):
        super().__init__()
        self.net = net
        self.hip_conv = hip_conv
        self._pack = None        # (data_ptr tuple, device pointer table, forward-form buffer, data-gradient-form buffer)

    def _packed_tower(self, dev):
        """[38][589824] bf16 twice: every ResBlock convolution's weight in the forward and the data-gradient form of
        hive_nn_conv3x3 / hive_nn_conv72 (hive_nn_pack_conv3x3_weights_multi).  The buffers are reused step after step: a
        step's backward has run on this stream before the next forward packs again."""
        from . import _lib
        L = _lib.load()
        ws = []
        for i in range(19):
            blk = getattr(self.net, "res_%i" % i)
            ws += [blk.conv1.weight, blk.conv2.weight]
        lay = [_weight_layout(w) for w in ws]
        if any(t is not w for (t, _), w in zip(lay, ws)) or len({cl for _, cl in lay}) != 1:
            return None                                    # mixed or strided storage: each convolution packs its own
        ptrs = tuple(w.data_ptr() for w in ws)
        if self._pack is None or self._pack[0] != ptrs or self._pack[1].device != dev:
            table = torch.tensor(ptrs, dtype=torch.int64).to(dev)
            fwd = torch.empty((len(ws), 9 * 256 * 256), dtype=torch.bfloat16, device=dev)
            self._pack = (ptrs, table, fwd, torch.empty_like(fwd))
        _, table, fwd, bwd = self._pack
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(L.hive_nn_pack_conv3x3_weights_multi(ctypes.c_void_p(table.data_ptr()), len(ws), lay[0][1],
                                                        ctypes.c_void_p(fwd.data_ptr()), ctypes.c_void_p(bwd.data_ptr()), st))
        return fwd, bwd

    def forward(self, s):
        net = self.net
        if not (net.training and s.is_cuda):      # eval mode (running statistics) and CPU runs use the modules as they are
            return net(s)
        if not self.hip_conv:
            conv = lambda t, m, *a, **k: m(t)
            packed = None
        else:
            conv = conv3x3
            packed = self._packed_tower(s.device) if self.pack_once else None
        pk = (lambda j: (packed[0][j], packed[1][j])) if packed is not None else (lambda j: (None, None))
        s = bn_act(conv(s, net.conv.conv1), net.conv.bn1, track=False)
        counters = [net.conv.bn1.num_batches_tracked]
        fs = self.fuse_bn_stats and self.hip_conv
        for i in range(19):
            blk = getattr(net, "res_%i" % i)
            # the block's input feeds conv1 and the skip connection: its two gradients are summed inside conv1's data-gradient
            # launch (hive_nn_conv72_add) when the two operators are linked
            link = {} if (self.fuse_skip_grad and self.hip_conv and s.requires_grad) else None
            st1 = {} if fs else None
            st2 = {} if fs else None
            out = bn_act(conv(s, blk.conv1, *pk(2 * i), link=link, stats=st1), blk.bn1, track=False, stats=st1)
            s = bn_act(conv(out, blk.conv2, *pk(2 * i + 1), stats=st2), blk.bn2, residual=s, track=False, link=link, stats=st2)
            counters += [blk.bn1.num_batches_tracked, blk.bn2.num_batches_tracked]
        if self.keep_tower_output:
            self.tower_output = s.detach()
        out, hb = net.outblock, self.hip_head_bn
        # OutBlock.forward (alpha_net.py:67-80) with the two BatchNorms on the HIP kernels
        v = out.conv(s)
        pl = out.conv1(s)
        if hb and bn_act_ok(v) and bn_act_ok(pl):
            v = bn_act(v, out.bn, track=False)
            pl = bn_act(pl, out.bn1, track=False)
            counters += [out.bn.num_batches_tracked, out.bn1.num_batches_tracked]
        else:
            v = F.relu(out.bn(v))
            pl = F.relu(out.bn1(pl))
        torch._foreach_add_(counters, 1)          # nn.BatchNorm2d's num_batches_tracked += 1, all layers in one launch
        v = v.reshape(-1, MAX_MAP_FULL * MAX_MAP_FULL)
        v = F.relu(out.fc1(v))
        v = torch.tanh(out.fc2(v))
        pl = pl.reshape(-1, MAX_MAP_FULL * MAX_MAP_FULL * 128)
        pl = out.fc(pl)
        return out.logsoftmax(pl).exp(), v


class Trainer:
    """Training step of alpha_net.py:117-162 laid out for MI355X (second "next" row):
    channels-last activations, bf16 autocast on the MFMA units with fp32 master weights and fp32
    loss, Adam lr 1e-3 + MultiStepLR like the reference.  With torch.distributed initialised the
    model is wrapped in DistributedDataParallel (gradient all-reduce over RCCL/xGMI, bucketed and
    overlapped with backward by DDP); self-play itself never needs a collective.  `fused` (default: on for bf16 on a
    GPU) routes the tower's BatchNorm + skip + ReLU through the HIP kernels of csrc/hive_train.hip (FusedTrainNet)."""

    def __init__(self, net, lr=0.001, autocast_dtype=torch.bfloat16, ddp=None, fused=None, freeze_gc=False):
        """freeze_gc: move everything allocated so far (torch, numpy, the model) into the garbage collector's permanent
        generation (gc.freeze()).  A 16 ms step is short enough to notice CPython's full collections: in a process with
        torch loaded one of them walks ~10^6 live objects and takes 80-110 ms, about once every few dozen steps
        (profiles/r03_training_step.md: the step it lands in reads 95-126 ms).  Frozen, those passes only walk what was
        allocated since."""
        import torch.distributed as dist
        self.device = next(net.parameters()).device
        self.net = net.to(memory_format=torch.channels_last)
        self.autocast_dtype = autocast_dtype if self.device.type == "cuda" else None
        use_ddp = ddp if ddp is not None else (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        self.fused = (self.autocast_dtype == torch.bfloat16) if fused is None else bool(fused)
        if self.fused and self.autocast_dtype != torch.bfloat16:
            raise ValueError("the fused BatchNorm kernels are bf16: use autocast_dtype=torch.bfloat16 on a GPU")
        self.model = FusedTrainNet(self.net) if self.fused else self.net
        if use_ddp:
            ids = [self.device.index] if self.device.type == "cuda" else None
            self.model = torch.nn.parallel.DistributedDataParallel(self.model, device_ids=ids, bucket_cap_mb=64)
        self.criterion = AlphaLoss()
        # the reference's optimizer (alpha_net.py:117-121: optim.Adam, lr as given); on a GPU its single fused multi-tensor
        # kernel (0.3 instead of 0.9 ms per step for the 255 parameter tensors: profiles/r03_training_step.md)
        self.optimizer = torch.optim.Adam(self.net.parameters(), lr=lr, fused=True) if self.device.type == "cuda" \
            else torch.optim.Adam(self.net.parameters(), lr=lr)
        self.scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=[100, 200, 300, 400], gamma=0.2)
        if freeze_gc:
            import gc
            gc.collect()
            gc.freeze()

    def loss(self, state_nchw, policy, value):
        x = state_nchw.to(self.device).float().contiguous(memory_format=torch.channels_last)
        policy, value = policy.to(self.device).float(), value.to(self.device).float()
        if self.autocast_dtype is not None:
            with torch.autocast("cuda", dtype=self.autocast_dtype):
                p, v = self.model(x)
        else:
            p, v = self.model(x)
        return self.criterion(v.float()[:, 0], value, p.float(), policy)

    def step(self, state_nchw, policy, value):
        self.model.train()
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss(state_nchw, policy, value)
        loss.backward()
        self.optimizer.step()
        return float(loss.detach())

    def end_epoch(self):
        self.scheduler.step()


def _fold(conv, bn):
    """eval-mode BatchNorm folded into the preceding convolution (fp32 math)."""
    w = conv.weight.detach().float()
    b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    return w * scale.view(-1, 1, 1, 1), bn.bias.detach().float() + (b - bn.running_mean.detach().float()) * scale


def _frag_major(w, dev, dtype=torch.bfloat16):
    """[256, cin, 3, 3] fp32 -> bf16 / fp16 [9][cinp/32][16][64][8]: the MFMA-fragment-major weight layout of
    hive_nn_conv3x3 (include/hive_nn.h).  tap = dy*3+dx; cin zero-padded to a multiple of 64; inside a
    (tap, 32-channel k-step, 16-output-channel tile) block, lane = (c%32)//8*16 + k%16 holds 8 consecutive c."""
    k, cin = w.shape[0], w.shape[1]
    cinp = (cin + 63) // 64 * 64
    t = w.permute(2, 3, 0, 1).reshape(9, k, cin)
    if cinp != cin:
        t = torch.cat([t, torch.zeros(9, k, cinp - cin, dtype=t.dtype, device=t.device)], dim=2)
    t = t.reshape(9, 16, 16, cinp // 32, 4, 8)          # tap, m-tile, row, k-step, k-group, 8
    t = t.permute(0, 3, 1, 4, 2, 5)                     # tap, k-step, m-tile, k-group, row, 8
    return t.to(dev, dtype).contiguous()


class InferenceNet:
    """Batched leaf evaluator for a ChessNet (eval mode).  __call__(planes_hwc) -> (p fp32 [B,1584], v fp32 [B]).

    conv = "hip" (default for bf16 / fp16 on a GPU): the 39 3x3 convolutions run in the hand-written MFMA kernels
    (hive_nn_conv3x3_dt for the stem, hive_nn_resblock_dt per residual block) with bias / skip / ReLU fused;
    conv = "torch": MIOpen through F.conv2d (the fp32 path, the reference's own precision, api_hive.py:62-69).
    tower: "auto" (default) = the 72-tile assembly tower hive_nn_tower72 for batches that fill its rounds, else 0;
    0 = one hive_nn_resblock_dt launch per residual block; 72 = hive_nn_tower72 always; 1 / 2 / 3 = the whole tower in one
    hive_nn_tower launch (its boards_per_group modes).  All forms produce the same bits."""

    FP16_HEADROOM = 8.0                  # fp16 is chosen only while every probed magnitude stays below 65504 / 8

    def __init__(self, net, dtype=None, device=None, use_graph=True, conv=None, tune_gemms=True, tower="auto"):
        """dtype None = choose on measurement (api_hive.py:56-69 evaluates in fp32; fp16 reproduces its search in 100 % of
        the test positions, bf16 in 97.7 %, tests/test_net.py): fp16 on a GPU when `range_probe` shows the BatchNorm-folded
        weights and every layer's activations on a batch of playout positions at least FP16_HEADROOM below the fp16
        maximum, else bf16 (whose exponent range is fp32's); fp32 on the CPU.  `precision_report` holds the numbers."""
        dev = torch.device(device) if device is not None else next(net.parameters()).device
        auto = dtype is None
        if auto:
            dtype = torch.float16 if dev.type == "cuda" and conv in (None, "hip") else torch.float32
        self.device, self.dtype, self.use_graph = dev, dtype, use_graph and dev.type == "cuda"
        self.precision_report = {"requested": "auto" if auto else str(dtype).replace("torch.", "")}
        hip_ok = dev.type == "cuda" and dtype in (torch.bfloat16, torch.float16)
        if conv is None:
            conv = "hip" if hip_ok else "torch"
        if conv == "hip" and not hip_ok:
            raise ValueError("the HIP convolution path is bf16 or fp16 on a GPU")
        self.conv, self.tower = conv, (tower if tower == "auto" else int(tower))
        # large leaf batches: let PyTorch's TunableOp pick the hipBLASLt solutions of the head GEMMs once, before the graph
        # is captured (the default heuristic runs the 1024 x 18432 x 1584 policy FC at 178 us, the tuned pick at 92 us)
        self.tune_gemms = tune_gemms and dev.type == "cuda"
        self.hip_heads = conv == "hip"   # the heads on hive_nn_heads (False: the library GEMMs, as the fp32 path)
        self.fuse_blocks = True          # hive_nn_resblock: both convolutions of a residual block in one launch
        self.split_streams = True        # leaf batches >= 512: two independent half-batch chains on two streams
        self._side_stream = None
        if conv == "hip":
            from . import _lib
            self._L = _lib.load()
        for name, value in self._folded(net.eval()).items():
            setattr(self, name, value)
        self._graphs = {}
        self.weights_version = 0
        self._tunable_before = None        # TunableOp's process-wide switch as it was before this object turned it on
        import threading
        self._lock = threading.Lock()      # capture/replay share static buffers: one caller at a time
        if auto and self.dtype == torch.float16:
            rep = self.range_probe()
            self.precision_report.update(rep)
            if not rep["fp16_safe"]:
                # this checkpoint's folded weights / activations come within FP16_HEADROOM of the fp16 maximum (or beyond):
                # re-fold in bf16, whose exponent range is fp32's
                self.dtype = torch.bfloat16
                for name, value in self._folded(net.eval()).items():
                    setattr(self, name, value)
        self.precision_report["dtype"] = str(self.dtype).replace("torch.", "")

    @torch.no_grad()
    def range_probe(self, planes_hwc=None, positions=64):
        """Largest magnitudes the 16-bit engine meets: max |BatchNorm-folded weight| and, on one batch of positions (given,
        or `positions` random-playout positions encoded by the env kernels), max |activation| after every convolution of
        the tower (launch-per-convolution form, so the intermediate of a residual block is seen too) and in the heads.
        An overflow shows as inf (the epilogues round to the 16-bit type without saturating).  fp16_safe = everything is
        finite and below 65504 / FP16_HEADROOM."""
        if self.conv != "hip":
            raise ValueError("range_probe measures the 16-bit HIP engine")
        import ctypes
        from . import _lib, playout
        dev = self.device
        if planes_hwc is None:
            boards = playout.random_positions(positions, seed=20260, device=dev.index)
            n = boards.shape[0]
            hist = torch.zeros((n, 384), dtype=torch.uint8, device=dev)
            ws = torch.empty((n * 144,), dtype=torch.int64, device=dev)
            planes_hwc = torch.empty((n, 12, 12, 56), dtype=self.dtype, device=dev)
            P = lambda t: ctypes.c_void_p(t.data_ptr())
            _lib.check(self._L.hive_encode_launch(P(boards), P(hist), n, P(planes_hwc), _lib.F16 if self.dtype == torch.float16 else _lib.BF16,
                                                  _lib.HWC, P(ws), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        x = planes_hwc.to(self.dtype).contiguous()
        B = x.shape[0]
        wmax = max(float(t.float().abs().max()) for t in (self.h_stem[0], self.h_tower[0], self.pconv[0], self.vconv[0], self.fc[0]))
        amax = []
        bufs = [torch.zeros((B, 12, 12, 256), dtype=self.dtype, device=dev) for _ in range(3)]
        s = self._conv_hip(x, 56, self.h_stem[0], self.h_stem[1], None, bufs[0])
        amax.append(float(s.float().abs().max()))
        cur = 0
        for w1, b1, w2, b2 in self.h_blocks:
            o = self._conv_hip(s, 256, w1, b1, None, bufs[(cur + 1) % 3])
            amax.append(float(o.float().abs().max()))
            s = self._conv_hip(o, 256, w2, b2, s, bufs[(cur + 2) % 3])
            amax.append(float(s.float().abs().max()))
            cur = (cur + 2) % 3
        flat = s.reshape(B * 144, 256)
        hp = F.relu(F.linear(flat, *self.pconv))
        logits = F.linear(hp.reshape(B, 144 * 128), *self.fc)
        hv = F.relu(F.linear(flat, *self.vconv))
        heads = [float(t.float().abs().max()) for t in (hp, logits, hv)]
        top = max([wmax] + amax + heads)
        limit = 65504.0 / self.FP16_HEADROOM
        import math
        return {"probe_positions": int(B), "max_abs_folded_weight": wmax, "max_abs_activation_tower": max(amax),
                "max_abs_activation_per_conv": [round(a, 3) for a in amax], "max_abs_heads": max(heads),
                "fp16_limit_with_headroom": limit, "fp16_safe": bool(math.isfinite(top) and top < limit)}

    def _folded(self, net):
        """Every weight tensor of the forward, BatchNorm folded, in the layouts the kernels read."""
        dev, dtype, conv = self.device, self.dtype, self.conv
        cl = torch.channels_last
        out = {}
        if conv == "hip":
            w, b = _fold(net.conv.conv1, net.conv.bn1)
            out["h_stem"] = (_frag_major(w, dev, dtype), b.to(dev, torch.float32).contiguous())
            # the 38 tower convolutions back to back in ONE buffer ([38][9][8][16][64][8] + [38][256] biases: what
            # hive_nn_tower reads); the per-block tuples below are views into it
            tw = torch.empty((38, 9 * 8 * 16 * 64 * 8), dtype=dtype, device=dev)
            tb = torch.empty((38, 256), dtype=torch.float32, device=dev)
            h_blocks = []
            for i in range(19):
                rb = getattr(net, "res_%i" % i)
                for j, (cv, bn) in enumerate(((rb.conv1, rb.bn1), (rb.conv2, rb.bn2))):
                    wj, bj = _fold(cv, bn)
                    tw[2 * i + j].copy_(_frag_major(wj, dev, dtype).reshape(-1))
                    tb[2 * i + j].copy_(bj)
                h_blocks.append((tw[2 * i], tb[2 * i], tw[2 * i + 1], tb[2 * i + 1]))
            out["h_tower"] = (tw, tb)
            out["h_blocks"] = h_blocks

        def prep(w, b):
            return (w.to(dev, dtype).contiguous(memory_format=cl), b.to(dev, dtype))

        out["stem"] = prep(*_fold(net.conv.conv1, net.conv.bn1))
        blocks = []
        for i in range(19):
            rb = getattr(net, "res_%i" % i)
            blocks.append((prep(*_fold(rb.conv1, rb.bn1)), prep(*_fold(rb.conv2, rb.bn2))))
        out["blocks"] = blocks
        ob = net.outblock
        # the two 1x1 head convolutions are plain GEMMs over the NHWC activations ([B*144, 256] x [256, k])
        wv, bv = _fold(ob.conv, ob.bn)
        wp, bp = _fold(ob.conv1, ob.bn1)
        out["vconv"] = (wv.reshape(1, 256).to(dev, dtype).contiguous(), bv.to(dev, dtype))
        out["pconv"] = (wp.reshape(128, 256).to(dev, dtype).contiguous(), bp.to(dev, dtype))
        # the policy FC consumes the head in NHWC order: permute its columns once (c*144+hw -> hw*128+c)
        wfc = ob.fc.weight.detach().float().view(ACTIONS, 128, 144).permute(0, 2, 1).reshape(ACTIONS, 144 * 128)
        out["fc"] = (wfc.to(dev, dtype).contiguous(), ob.fc.bias.detach().to(dev, dtype))
        out["fc1"] = (ob.fc1.weight.detach().to(dev, torch.float32), ob.fc1.bias.detach().to(dev, torch.float32))
        out["fc2"] = (ob.fc2.weight.detach().to(dev, torch.float32), ob.fc2.bias.detach().to(dev, torch.float32))
        if conv == "hip":
            # hive_nn_heads (include/hive_nn.h): both 1x1 convolutions as one [144 x 256] matrix of MFMA A fragments (rows
            # 0..127 policy, 128 value, the rest zero), the policy FC as B fragments over k = pixel * 128 + channel
            wc = torch.zeros((144, 256), dtype=torch.float32, device=wp.device)
            wc[:128] = wp.reshape(128, 256)
            wc[128] = wv.reshape(256)
            bc = torch.zeros((144,), dtype=torch.float32, device=wp.device)
            bc[:128] = bp
            bc[128] = bv.reshape(())
            wc = wc.reshape(9, 16, 8, 4, 8).permute(0, 2, 3, 1, 4)          # M tile, k-step, k-group, row, 8
            wf = wfc.reshape(99, 16, 576, 4, 8).permute(0, 2, 3, 1, 4)      # action tile, k-step, k-group, action % 16, 8
            out["h_heads"] = (wc.to(dev, dtype).contiguous(), bc.to(dev).contiguous(), wf.to(dev, dtype).contiguous(),
                              ob.fc.bias.detach().to(dev, torch.float32).contiguous(),
                              ob.fc1.weight.detach().to(dev, torch.float32).t().contiguous(),      # [144][64]
                              ob.fc1.bias.detach().to(dev, torch.float32).contiguous(),
                              ob.fc2.weight.detach().to(dev, torch.float32).reshape(64).contiguous(),
                              ob.fc2.bias.detach().to(dev, torch.float32).reshape(1).contiguous())
        return out

    def refresh(self, net):
        """New weights (same architecture) after a training iteration -- what the reference's workers do by re-reading
        the checkpoint (self_play.py:37-75): the folded tensors are copied INTO the existing buffers, so the captured
        HIP graphs (which hold their addresses) and the tuned GEMM choices stay valid."""
        def copy_into(dst, src):
            if isinstance(dst, torch.Tensor):
                dst.copy_(src)
            else:
                for d, s_ in zip(dst, src):
                    copy_into(d, s_)
        with self._lock, torch.no_grad():
            self.weights_version += 1          # (consumers that keep evaluations -- mcts.TreeSearch(reuse_store=...) -- drop them)
            for name, value in self._folded(net).items():
                copy_into(getattr(self, name), value)
            torch.cuda.synchronize(self.device) if self.device.type == "cuda" else None

    def _conv_hip(self, x, cin, w, b, res, out, need=None):
        import ctypes
        from ._lib import BF16, F16, check
        st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(self._L.hive_nn_conv3x3_sel(ctypes.c_void_p(x.data_ptr()), cin, ctypes.c_void_p(w.data_ptr()),
                                          ctypes.c_void_p(b.data_ptr()),
                                          ctypes.c_void_p(res.data_ptr()) if res is not None else None,
                                          ctypes.c_void_p(out.data_ptr()), x.shape[0], 1,
                                          BF16 if self.dtype == torch.bfloat16 else F16,
                                          ctypes.c_void_p(need.data_ptr()) if need is not None else None, st))
        return out

    def _tower_form(self, B):
        """The launch form of the 19 residual blocks for a batch of B boards: self.tower, with "auto" resolved.
        The 72-tile tower (72: hive_nn_tower72_balanced, one 2-board workgroup per CU, the pairs' blocks dealt evenly over
        the chip) costs ~3.9 us per evaluated board once the chip is full (220 pairs and more); the launch-per-block chain
        costs ~4.6 us per board and wins below that (profiles/r04_net_tower72.md).  auto = the tower from 440 boards on."""
        if self.tower != "auto":
            return self.tower
        return 72 if self.fuse_blocks and B >= 440 else 0

    def _tower_hip(self, x_hwc, need=None, rep=None):
        """need: int8[B] on the device or None -- boards flagged 0 are skipped by every kernel of the tower (their rows of
        the activation buffers keep stale, finite values; the heads compute on them and nobody reads the result).
        rep: int32[B] or None -- row i takes the tower output of row rep[i] (equal leaves, hive_leaf_dedup_launch)
        before the heads run, i.e. exactly the activations it would have computed itself."""
        out = self._tower_hip_rows(x_hwc, need)
        if rep is not None:
            import ctypes
            from ._lib import check
            nhwc = out.permute(0, 2, 3, 1)                   # the contiguous [B,12,12,256] buffer behind the NCHW view
            assert nhwc.is_contiguous()
            check(self._L.hive_nn_copy_rows(ctypes.c_void_p(nhwc.data_ptr()), ctypes.c_void_p(rep.data_ptr()), nhwc.shape[0],
                                            144 * 256 * nhwc.element_size(),
                                            ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return out

    def _tower_hip_rows(self, x_hwc, need=None):
        B = x_hwc.shape[0]
        x_hwc = x_hwc.contiguous()
        tower = self._tower_form(B)
        if need is not None and ((tower and tower != 72) or not self.fuse_blocks):
            need = None                      # the launch-per-block form and the 72-tile tower take the selection
        # a skipped board's rows must hold finite numbers: eager calls start from zeros; a captured graph owns its buffers
        # for good and _call_locked replays it once with every board selected before the first real call
        alloc = torch.zeros if need is not None and not torch.cuda.is_current_stream_capturing() else torch.empty
        bufs = [alloc((B, 12, 12, 256), dtype=self.dtype, device=self.device) for _ in range(3)]
        s = self._conv_hip(x_hwc, 56, self.h_stem[0], self.h_stem[1], None, bufs[0], need)
        cur = 0
        import ctypes
        from ._lib import BF16, F16, check
        dt = BF16 if self.dtype == torch.bfloat16 else F16
        st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

        def needp(lo):
            return ctypes.c_void_p(need.data_ptr() + lo) if need is not None else None

        if tower == 72:
            # the 72-tile assembly tower (hive_nn_tower72): two boards per workgroup, all 19 blocks in one launch; the
            # boards to evaluate travel as a compacted row list (hive_nn_compact_rows of the need flags)
            tw, tb = self.h_tower
            rows = nrows = None
            if need is not None:
                rows = torch.empty((B,), dtype=torch.int32, device=self.device)
                nrows = torch.empty((1,), dtype=torch.int32, device=self.device)
                check(self._L.hive_nn_compact_rows(ctypes.c_void_p(need.data_ptr()), B, ctypes.c_void_p(rows.data_ptr()),
                                                   ctypes.c_void_p(nrows.data_ptr()), st))
            plan = torch.empty((int(self._L.hive_nn_tower72_plan_bytes(B)),), dtype=torch.uint8, device=self.device)
            check(self._L.hive_nn_tower72_balanced(ctypes.c_void_p(s.data_ptr()), ctypes.c_void_p(tw.data_ptr()),
                                                   ctypes.c_void_p(tb.data_ptr()), ctypes.c_void_p(bufs[1].data_ptr()), B, 19, dt,
                                                   ctypes.c_void_p(rows.data_ptr()) if rows is not None else None,
                                                   ctypes.c_void_p(nrows.data_ptr()) if nrows is not None else None,
                                                   ctypes.c_void_p(plan.data_ptr()), st))
            return bufs[1].permute(0, 3, 1, 2)
        if tower:
            tw, tb = self.h_tower
            check(self._L.hive_nn_tower(ctypes.c_void_p(s.data_ptr()), ctypes.c_void_p(tw.data_ptr()),
                                        ctypes.c_void_p(tb.data_ptr()), ctypes.c_void_p(bufs[1].data_ptr()), B, 19, dt,
                                        tower, st))
            return bufs[1].permute(0, 3, 1, 2)
        if self.fuse_blocks and self.split_streams and B >= 512:
            # Boards are independent: only the 19 blocks of ONE board are ordered.  A single stream makes every block a
            # chip-wide barrier (the launch boundary) and keeps every CU's two workgroups in the same phase; independent
            # chains on two streams run one chain's tails, staging and epilogues under the other's tap loops.  Measured
            # (tools/two_stream_chain.py, same bits): 1024 boards as 2 x 512: 246 instead of 260 us per block; 4096 boards
            # as chains of 512 dealt onto two streams: 955 instead of 1003 (two chains of 2048: 998); more than two chains
            # in flight, or chains of fewer than 256 boards, are slower again (they de-phase the weight streams).
            if B <= 1024:
                bounds = [(0, B // 2), (B // 2, B)]
            else:
                bounds = [(lo, min(lo + 512, B)) for lo in range(0, B, 512)]
            main = torch.cuda.current_stream(self.device)
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(self.device)
            side = self._side_stream
            side.wait_stream(main)
            for k, (lo, hi) in enumerate(bounds):
                stp = ctypes.c_void_p((main if k % 2 == 0 else side).cuda_stream)
                cur = 0
                for w1, b1, w2, b2 in self.h_blocks:
                    nxt = (cur + 1) % 3
                    check(self._L.hive_nn_resblock_sel(ctypes.c_void_p(bufs[cur][lo:].data_ptr()), ctypes.c_void_p(w1.data_ptr()),
                                                       ctypes.c_void_p(b1.data_ptr()), ctypes.c_void_p(w2.data_ptr()),
                                                       ctypes.c_void_p(b2.data_ptr()), ctypes.c_void_p(bufs[nxt][lo:].data_ptr()),
                                                       hi - lo, dt, needp(lo), stp))
                    cur = nxt
            if not torch.cuda.is_current_stream_capturing():      # (a captured graph owns its memory pool)
                for t in bufs:
                    t.record_stream(side)
            main.wait_stream(side)
            return bufs[cur].permute(0, 3, 1, 2)
        for w1, b1, w2, b2 in self.h_blocks:
            if self.fuse_blocks:
                s2 = bufs[(cur + 1) % 3]                                     # whole residual block in one launch
                check(self._L.hive_nn_resblock_sel(ctypes.c_void_p(s.data_ptr()), ctypes.c_void_p(w1.data_ptr()),
                                                   ctypes.c_void_p(b1.data_ptr()), ctypes.c_void_p(w2.data_ptr()),
                                                   ctypes.c_void_p(b2.data_ptr()), ctypes.c_void_p(s2.data_ptr()), B, dt,
                                                   needp(0), st))
                s, cur = s2, (cur + 1) % 3
            else:
                o = self._conv_hip(s, 256, w1, b1, None, bufs[(cur + 1) % 3])
                s2 = self._conv_hip(o, 256, w2, b2, s, bufs[(cur + 2) % 3])  # relu(conv + bias + skip)
                s, cur = s2, (cur + 2) % 3
        return s.permute(0, 3, 1, 2)             # NCHW view with channels-last strides

    def _forward(self, x_hwc, need=None, rep=None):
        # x_hwc: [B,12,12,56] in self.dtype; viewed as NCHW with channels-last strides (zero copy)
        if self.conv == "hip":
            s = self._tower_hip(x_hwc, need, rep)
        else:
            x = x_hwc.permute(0, 3, 1, 2)
            s = F.relu(F.conv2d(x, self.stem[0], self.stem[1], padding=1))
            for (w1, b1), (w2, b2) in self.blocks:
                o = F.relu(F.conv2d(s, w1, b1, padding=1))
                o = F.conv2d(o, w2, b2, padding=1)
                s = F.relu(o + s)
        B = s.shape[0]
        if self.conv == "hip" and self.hip_heads:
            return self._heads_hip(s.permute(0, 2, 3, 1))
        flat = s.permute(0, 2, 3, 1).reshape(B * 144, 256)             # NHWC rows (a view: s is channels-last)
        v = F.relu(F.linear(flat, *self.vconv)).float().reshape(B, 144)
        v = F.relu(F.linear(v, *self.fc1))
        v = torch.tanh(F.linear(v, *self.fc2)).reshape(B)
        p = F.relu(F.linear(flat, *self.pconv)).reshape(B, 144 * 128)  # NHWC flatten (matches self.fc columns)
        p = F.linear(p, *self.fc).float()
        return torch.softmax(p, dim=1), v

    def _heads_hip(self, nhwc):
        """Both heads on the hand-written kernels (hive_nn_heads): no library GEMM, nothing to tune at start-up."""
        import ctypes
        from ._lib import BF16, F16, check
        assert nhwc.is_contiguous()
        B = nhwc.shape[0]
        ws = torch.empty((int(self._L.hive_nn_heads_workspace_bytes(B)),), dtype=torch.uint8, device=self.device)
        p = torch.empty((B, ACTIONS), dtype=torch.float32, device=self.device)
        v = torch.empty((B,), dtype=torch.float32, device=self.device)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        wc, bc, wf, bf_, w1, b1, w2, b2 = self.h_heads
        check(self._L.hive_nn_heads(P(nhwc), B, BF16 if self.dtype == torch.bfloat16 else F16, P(wc), P(bc), P(wf), P(bf_), P(w1), P(b1),
                                    P(w2), P(b2), P(ws), P(p), P(v), ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return p, v

    @torch.no_grad()
    def _tune(self, static_in):
        try:
            import torch.cuda.tunable as tn
        except ImportError:
            return
        import os
        import tempfile
        if not tn.is_enabled():
            # the results file TunableOp writes at exit goes to the temp directory, not into the caller's cwd
            tn.set_filename(os.path.join(tempfile.gettempdir(), "hive_tunableop_%d.csv" % os.getpid()))
        if self._tunable_before is None:
            self._tunable_before = bool(tn.is_enabled())
        tn.enable(True)                    # on until the graph of this batch size is captured (it keeps the solutions picked)
        tn.set_max_tuning_duration(1000)
        tn.tuning_enable(True)
        try:
            self._forward(static_in)
            torch.cuda.synchronize(self.device)
        finally:
            tn.tuning_enable(False)

    @property
    def batch_independent_bits(self):
        """A position's (p, v) are the same bits in any batch at any row: every tower form is bit-identical and hive_nn_heads
        sums over a fixed K split (the library heads' results depended on the row position in the last ulp)."""
        return self.conv == "hip" and self.hip_heads

    @property
    def graph_batches(self):
        """Batch sizes a HIP graph has been captured for (in the current or an earlier launch form)."""
        return {k[0] for k in self._graphs}

    @property
    def accepts_need(self):
        """__call__ honours the row selection of hive_search_leaf_need (mcts.TreeSearch asks): the HIP tower only -- the
        library path evaluates every row, and says so, so that the search's rows-evaluated counter stays true."""
        return self.conv == "hip" and self.tower in (0, 72, "auto") and self.fuse_blocks

    accepts_rep = accepts_need   # ... and the representatives of equal rows (hive_leaf_dedup_launch)

    def __call__(self, planes_hwc, need=None, rep=None):
        """need: int8[B] on the device (1 = this row's p / v will be read) or None = every row.  Rows flagged 0 come back
        with unspecified (finite) numbers; the tower's kernels skip their boards (hive_nn_resblock_sel).
        rep: int32[B] or None -- equal rows (hive_leaf_dedup_launch): row i, switched off in `need`, gets the tower output
        of row rep[i] copied in before the heads, so its p / v are the bits it would have produced itself."""
        with self._lock:
            hip = self.accepts_need
            p, v = self._call_locked(planes_hwc, need if hip else None, rep if hip else None)
            return (p.clone(), v.clone()) if self.use_graph else (p, v)

    def _call_locked(self, planes_hwc, need=None, rep=None):
        B = planes_hwc.shape[0]
        if planes_hwc.dtype != self.dtype:
            planes_hwc = planes_hwc.to(self.dtype)
        if not self.use_graph:
            return self._forward(planes_hwc, need, rep)
        hip = self.conv == "hip"
        key = (B, self.tower, self.fuse_blocks, self.split_streams, self.hip_heads)      # a captured graph keeps the launch form it saw
        g = self._graphs.get(key)
        if g is None:
            static_in = torch.zeros_like(planes_hwc)
            static_in.copy_(planes_hwc)
            static_need = torch.ones((B,), dtype=torch.int8, device=self.device) if hip else None
            static_rep = torch.arange(B, dtype=torch.int32, device=self.device) if hip else None
            try:
                s = torch.cuda.Stream(self.device)
                s.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(s):
                    if self.tune_gemms and B >= 256 and not (hip and self.hip_heads):
                        self._tune(static_in)
                    for _ in range(2):
                        self._forward(static_in, static_need, static_rep)
                torch.cuda.current_stream(self.device).wait_stream(s)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    out = self._forward(static_in, static_need, static_rep)
                if hip:
                    graph.replay()             # every board selected: the graph's activation buffers now hold finite rows
                g = [graph, static_in, out, static_need, True, static_rep, True]
                self._graphs[key] = g
            finally:
                if self._tunable_before is not None:
                    # TunableOp's switch is process-global: hand it back as it was -- also when tuning or the capture
                    # failed --, so that whatever else runs in this process (a Trainer, another library) keeps its own
                    # GEMM selection; the captured graph holds the picked kernels
                    import torch.cuda.tunable as tn
                    tn.enable(self._tunable_before)
                    self._tunable_before = None
        graph, static_in, out, static_need, all_rows, static_rep, identity = g
        static_in.copy_(planes_hwc)
        if need is not None:
            static_need.copy_(need)
            g[4] = False
        elif not all_rows:
            static_need.fill_(1)
            g[4] = True
        if rep is not None:
            static_rep.copy_(rep)
            g[6] = False
        elif not identity:
            static_rep.copy_(torch.arange(B, dtype=torch.int32, device=self.device))
            g[6] = True
        graph.replay()
        return out
