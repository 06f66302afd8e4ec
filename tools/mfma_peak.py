"""What dense bf16 MFMA rate does this card SUSTAIN with no memory traffic at all?  (The nominal 2.5 PFLOP/s assumes
2.4 GHz on every SIMD.)  Runs tools/dev/mfma_peak.hip (built into build/mfma_peak.so) for ~0.1 s and ~2 s windows."""
import ctypes, os, subprocess, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, "build", "mfma_peak.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared", "-o", so,
                           os.path.join(ROOT, "tools", "dev", "mfma_peak.hip")])
if __name__ == "__main__" and torch.cuda.is_available():
    L = ctypes.CDLL(so)
    L.mfma_peak_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    out = torch.zeros(4096 * 256, device="cuda")
    s = torch.cuda.current_stream()
    for tiles in (36, 16):
        for blocks, trips, reps in ((512, 2000, 5), (1024, 2000, 5), (2048, 4000, 40)):
            args = (out.data_ptr(), blocks, trips, tiles, s.cuda_stream)
            L.mfma_peak_launch(*args)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(reps):
                L.mfma_peak_launch(*args)
            e1.record(s)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            flop = reps * blocks * 4 * trips * tiles * 16 * 16 * 32 * 2
            print(f"tiles/wave {tiles:2d}  blocks {blocks:5d} (x4 waves)  window {ms:8.1f} ms  {flop / ms / 1e9:8.1f} TFLOP/s "
                  f"= {flop / ms / 1e9 / 2500:.3f} of 2.5 PFLOP/s", flush=True)
