"""Print the XCD of the first workgroups of a 1-D grid (tools/dev/xcc_map.hip): is it `index mod 8`?"""
import ctypes, os, subprocess, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, "build", "xcc_map.so")
src = os.path.join(ROOT, "tools", "dev", "xcc_map.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "--offload-arch=gfx950", "-shared", "-o", so, src])
if __name__ == "__main__" and torch.cuda.is_available():
    L = ctypes.CDLL(so)
    L.xcc_map_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    for blocks, threads in ((256, 512), (512, 256), (4096, 704)):
        out = torch.full((blocks,), 99, dtype=torch.int32, device="cuda")
        L.xcc_map_launch(out.data_ptr(), blocks, threads, None)
        torch.cuda.synchronize()
        o = out.cpu().tolist()
        ok = all(v == i % 8 for i, v in enumerate(o))
        print(f"grid {blocks} x {threads}: first 24 workgroups on XCDs {o[:24]}; index mod 8 everywhere: {ok}")
