"""hive_nn_conv3x3_wgrad vs MIOpen's weight-gradient convolution at the training batch, plus timing-only ablations
(builds with -DHIVE_WG_ABL_*: results of those are wrong by construction)."""
import ctypes, glob, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "hive-alphazero_amd", "csrc")
OUT = os.path.join(ROOT, "build", "wg")
VARIANTS = {"base": ["-DHIVE_WG_DEBUG"], "tiles2x2": ["-DHIVE_WG_MT=2"], "tiles2x2_nostage_noatomic": ["-DHIVE_WG_MT=2", "-DHIVE_WG_ABL_NOSTAGE", "-DHIVE_WG_ABL_NOATOMIC"],
            "earlydma": ["-DHIVE_WG_EARLY_DMA"], "noatomic": ["-DHIVE_WG_ABL_NOATOMIC"], "nostage": ["-DHIVE_WG_ABL_NOSTAGE"],
            "nostage_noatomic": ["-DHIVE_WG_ABL_NOSTAGE", "-DHIVE_WG_ABL_NOATOMIC"]}


def build():
    os.makedirs(OUT, exist_ok=True)
    for name, flags in VARIANTS.items():
        so = os.path.join(OUT, f"wg_{name}.so")
        srcs = [os.path.join(SRC, "hive_wgrad.hip"), os.path.join(SRC, "hive_env.hip")]
        if not os.path.exists(so) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in srcs):
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared"] + flags +
                                  ["-o", so] + srcs)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


if __name__ == "__main__":
    build()
    if not torch.cuda.is_available():
        sys.exit(0)
    import statistics
    ROUNDS = 8
    for B in [int(a) for a in sys.argv[1:]] or [512, 1024, 128]:
        x = torch.randn((B, 256, 12, 12), device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = torch.randn((B, 256, 12, 12), device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wl = torch.zeros((256, 256, 3, 3), dtype=torch.bfloat16, device="cuda").contiguous(memory_format=torch.channels_last)
        flop = 2 * 256 * 2304 * B * 144
        dw = torch.empty((3, 3, 256, 256), dtype=torch.float32, device="cuda")
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        runs = {"MIOpen wgrad": lambda: torch.ops.aten.convolution_backward(dy, x, wl, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                                            (False, True, False))}
        for so in sorted(glob.glob(OUT + "/wg_*.so")):
            L = ctypes.CDLL(so)
            L.hive_nn_conv3x3_wgrad.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
            ws = torch.empty((L.hive_nn_wgrad_workspace_floats(),), dtype=torch.float32, device="cuda")
            name = os.path.basename(so)
            runs[name] = (lambda L=L, ws=ws: L.hive_nn_conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, ws.data_ptr(), st))
            if hasattr(L, "hive_nn_wgrad_debug_order"):
                for order in (0, 1):
                    def f(L=L, ws=ws, order=order):
                        L.hive_nn_wgrad_debug_order(order)
                        L.hive_nn_conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, ws.data_ptr(), st)
                        L.hive_nn_wgrad_debug_order(-1)
                    runs[f"{name} workgroup order {order}"] = f
        times = {k: [] for k in runs}
        for r in range(ROUNDS):                    # interleaved: the device's clock drifts by 10 % within seconds
            for k, fn in runs.items():
                times[k].append(timeit(fn, 8))
        print(f"batch {B} (median / min over {ROUNDS} interleaved rounds of 8 launches):")
        for k, v in times.items():
            med = statistics.median(v)
            print(f"   {k:42s} {med:7.1f} / {min(v):7.1f} us = {flop / med / 1e6:5.0f} TFLOP/s", flush=True)
