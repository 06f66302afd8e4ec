"""Time hive_nn_conv3x3 against F.conv2d (MIOpen) on the residual-tower shape."""
import ctypes, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hive_alphazero_amd as h
from hive_alphazero_amd.alpha_net import _frag_major
L = h.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = torch.randn((B, 12, 12, 256), device="cuda").to(torch.bfloat16)
w = torch.randn((256, 256, 3, 3), device="cuda") * 0.03
bias = torch.randn((256,), device="cuda")
res = torch.randn((B, 12, 12, 256), device="cuda").to(torch.bfloat16)
y = torch.empty_like(x)
wt = _frag_major(w, x.device)
wcl = w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
bb = bias.to(torch.bfloat16)
xn = x.permute(0, 3, 1, 2)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
hip = lambda: L.hive_nn_conv3x3(ctypes.c_void_p(x.data_ptr()), 256, ctypes.c_void_p(wt.data_ptr()), ctypes.c_void_p(bias.data_ptr()), ctypes.c_void_p(res.data_ptr()), ctypes.c_void_p(y.data_ptr()), B, 1, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
mio = lambda: F.relu(F.conv2d(xn, wcl, bb, padding=1) + res.permute(0, 3, 1, 2))
flop = 2.0 * B * 144 * 256 * 2304
th, tm = t(hip), t(mio)
print(f"B={B} hip conv+bias+skip+relu {th:.1f} us = {flop/th/1e6:.0f} TFLOP/s | MIOpen conv + torch epilogue {tm:.1f} us = {flop/tm/1e6:.0f} TFLOP/s")
