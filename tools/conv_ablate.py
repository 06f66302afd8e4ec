"""Timing A/B of hive_nn_conv3x3 builds (build/conv/conv_*.so, -D variants of csrc/hive_nn.hip).  The device's clock moves
by 10-15 % within seconds under this load, so the variants are INTERLEAVED: ROUNDS rounds, in each round every build runs 10
launches; median and minimum per build are reported.  Every build's output is compared with the first one's (builds named
*_no* are ablations: wrong by construction, only timed)."""
import ctypes, glob, os, statistics, sys, torch
B, ROUNDS = 1024, 12
x = torch.randn((B, 12, 12, 256), device="cuda").to(torch.bfloat16)
wt = (torch.randn((9 * 8 * 16 * 64 * 8,), device="cuda") * 0.03).to(torch.bfloat16)
bias = torch.randn((256,), device="cuda")
res = torch.randn((B, 12, 12, 256), device="cuda").to(torch.bfloat16)
libs, times, outs = {}, {}, {}
for so in sorted(glob.glob(sys.argv[1] + "/conv_*.so")):
    L = ctypes.CDLL(so)
    vp = ctypes.c_void_p
    L.hive_nn_conv3x3.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, vp]
    libs[os.path.basename(so)] = L
    times[os.path.basename(so)] = []
def run(L, y):
    L.hive_nn_conv3x3(x.data_ptr(), 256, wt.data_ptr(), bias.data_ptr(), res.data_ptr(), y.data_ptr(), B, 1, torch.cuda.current_stream().cuda_stream)
for name, L in libs.items():
    y = torch.zeros_like(x)
    for _ in range(3): run(L, y)
    torch.cuda.synchronize()
    outs[name] = y
for _ in range(ROUNDS):
    for name, L in libs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run(L, outs[name])
        e1.record(); torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 10 * 1e3)
first = next(n for n in libs if "_no" not in n)
for name in libs:
    same = "" if "_no" in name or name == first else "  identical to %s: %s" % (first, bool(torch.equal(outs[first], outs[name])))
    med, mn = statistics.median(times[name]), min(times[name])
    print(f"{name:22s} median {med:7.1f} us ({2.0 * B * 144 * 256 * 2304 / med / 1e6:5.0f} TFLOP/s)  min {mn:7.1f} us{same}", flush=True)
