"""Winograd F(2x2, 3x3) for ONE 256 -> 256 convolution of the tower at 1024 boards, measured before anyone argues about it
(VERDICT round 3, item 8).  Transforms and the 16 component GEMMs run as separate, individually efficient steps (torch
elementwise kernels / hipBLASLt batched GEMM) -- the structure a Winograd convolution whose components do not fit one
workgroup's registers must have (16 components x 288 tiles x 256 channels of fp32 accumulators per board pair
are 4.7 MB; a workgroup holds 0.29 MB) -- and are timed one by one against the direct MFMA kernel hive_nn_conv3x3_dt.
Accuracy: max |d| against fp32 F.conv2d, next to the direct 16-bit kernel's.   usage: winograd_probe.py [boards]"""
import ctypes, os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hive_alphazero_amd import _lib
from hive_alphazero_amd.alpha_net import _frag_major

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
torch.manual_seed(0)
dev = "cuda"
x32 = torch.relu(torch.randn((B, 256, 12, 12), device=dev))
w32 = torch.randn((256, 256, 3, 3), device=dev) * 0.02
ref = F.conv2d(x32, w32, padding=1)

Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32, device=dev)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32, device=dev)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32, device=dev)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3          # us


for dt in (torch.float16, torch.bfloat16):
    x = x32.to(dt).contiguous(memory_format=torch.channels_last)
    U = torch.einsum("ij,kcjl,ml->imkc", G, w32, G).reshape(16, 256, 256).to(dt).contiguous()      # [xi][k][c], transforms in fp32

    def input_transform():
        xp = F.pad(x, (1, 1, 1, 1))                                        # [B,256,14,14]
        t = xp.unfold(2, 4, 2).unfold(3, 4, 2)                             # [B,256,6,6,4,4]
        v = torch.einsum("ij,bcyxjl,ml->imbyxc", Bt.to(dt), t, Bt.to(dt))  # [4,4,B,6,6,256]
        return v.reshape(16, B * 36, 256)

    V = input_transform()

    def gemms():
        return torch.bmm(V, U.transpose(1, 2))                             # [16, B*36, 256(k)]

    M = gemms()

    def output_transform():
        m = M.reshape(4, 4, B, 6, 6, 256)
        y = torch.einsum("ij,jlbyxk,ml->bkyixm", At.to(dt), m, At.to(dt))  # [B,256,6,2,6,2]
        return y.reshape(B, 256, 12, 12)

    y = output_transform()
    t_in, t_g, t_out = timeit(input_transform), timeit(gemms), timeit(output_transform)
    # the direct kernel on the same operands
    xh = x.permute(0, 2, 3, 1).contiguous()                                # [B,12,12,256]
    wp = _frag_major(w32, dev, dt)
    bias = torch.zeros(256, device=dev)
    yd = torch.empty_like(xh)
    DT = _lib.BF16 if dt == torch.bfloat16 else _lib.F16
    direct = lambda: _lib.check(L.hive_nn_conv3x3_dt(P(xh), 256, P(wp), P(bias), None, P(yd), B, 0, DT, None))
    t_d = timeit(direct, 20)
    e_w = float((y.float() - ref).abs().max())
    e_d = float((yd.permute(0, 3, 1, 2).float() - ref).abs().max())
    flop_w, flop_d = 2 * 16 * B * 36 * 256 * 256, 2 * B * 144 * 256 * 2304
    print(f"{str(dt):15s} {B} boards, one 256->256 convolution")
    print(f"   direct MFMA kernel (hive_nn_conv3x3_dt)        {t_d:8.1f} us   {flop_d / t_d / 1e6:6.0f} TFLOP/s   max |d| vs fp32 {e_d:.4f}")
    print(f"   Winograd F(2x2,3x3): input transform           {t_in:8.1f} us   (writes {16 * B * 36 * 256 * 2 / 1e6:.0f} MB)")
    print(f"                        16 GEMMs [{B * 36} x 256 x 256]    {t_g:8.1f} us   {flop_w / t_g / 1e6:6.0f} TFLOP/s")
    print(f"                        output transform          {t_out:8.1f} us   (reads {16 * B * 36 * 256 * 2 / 1e6:.0f} MB)")
    print(f"                        total                     {t_in + t_g + t_out:8.1f} us   max |d| vs fp32 {e_w:.4f}   (max |ref| {float(ref.abs().max()):.2f})")
