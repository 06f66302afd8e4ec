"""What dense bf16 MFMA rate does this card SUSTAIN with no memory traffic at all, by accumulator tiles per wave and waves
per SIMD?  (The nominal 2.5 PFLOP/s assumes 2.4 GHz on every SIMD; under MFMA load the chip lowers its clock.)
Runs tools/dev/mfma_peak.hip (built into build/mfma_peak.so); prints TFLOP/s and the in-kernel shader clock."""
import ctypes, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, "build", "mfma_peak.so")
src = os.path.join(ROOT, "tools", "dev", "mfma_peak.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared", "-o", so, src])
if __name__ == "__main__" and torch.cuda.is_available():
    L = ctypes.CDLL(so)
    L.mfma_peak_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    out = torch.zeros(8192 * 256, device="cuda")
    clk = torch.zeros(2048, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream()
    print("| acc tiles / wave | waves / SIMD (launch bound) | resident waves / SIMD | TFLOP/s | of 2.5 PFLOP/s | in-kernel clock GHz |")
    print("|---|---|---|---|---|---|")
    for tiles, wps in ((36, 1), (36, 2), (36, 3), (24, 2), (24, 4), (16, 1), (16, 2), (16, 4), (16, 8), (8, 4), (8, 8)):
        blocks = 256 * wps * 4                      # wps workgroups of 4 waves per CU, 4 rounds
        trips = 40000 // tiles * 4
        args = (out.data_ptr(), clk.data_ptr(), blocks, trips, tiles, wps, s.cuda_stream)
        for _ in range(3):
            assert L.mfma_peak_launch(*args) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 12
        e0.record(s)
        for _ in range(reps):
            L.mfma_peak_launch(*args)
        e1.record(s)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        flop = reps * blocks * 4 * trips * tiles * 16 * 16 * 32 * 2
        c = clk.view(-1, 2)[:1024].double()
        ghz = float((c[:, 0] / c[:, 1]).median().item()) * 0.1
        print(f"| {tiles} | {wps} | {min(wps, 8)} | {flop / ms / 1e9:.0f} | {flop / ms / 1e9 / 2500:.3f} | {ghz:.2f} |  window {ms:.0f} ms", flush=True)
