import ctypes, glob, os, sys, torch
B = 1024
x = torch.randn((B, 12, 12, 256), device="cuda").to(torch.bfloat16)
wt = (torch.randn((9 * 8 * 16 * 64 * 8,), device="cuda") * 0.03).to(torch.bfloat16)
bias = torch.randn((256,), device="cuda")
res = torch.randn((B, 12, 12, 256), device="cuda").to(torch.bfloat16)
y = torch.empty_like(x)
for so in sorted(glob.glob(sys.argv[1] + "/conv_*.so")):
    L = ctypes.CDLL(so)
    vp = ctypes.c_void_p
    L.hive_nn_conv3x3.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, vp]
    f = lambda: L.hive_nn_conv3x3(x.data_ptr(), 256, wt.data_ptr(), bias.data_ptr(), res.data_ptr(), y.data_ptr(), B, 1, torch.cuda.current_stream().cuda_stream)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(os.path.basename(so), f"{e0.elapsed_time(e1)/20*1e3:.1f} us")
