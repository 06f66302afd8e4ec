import ctypes, os, sys, torch
sys.path.insert(0, "/root/repo")
from hive_alphazero_amd import playout
boards = playout.random_positions(4096, seed=1000)
L = ctypes.CDLL("/root/repo/gpurun_tmp_abl/dbg.so")
L.hive_movegen_launch.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4
mask = torch.empty((4096, 50), dtype=torch.int32, device="cuda"); cnt = torch.empty((4096,), dtype=torch.int32, device="cuda")
L.hive_movegen_launch(boards.data_ptr(), 4096, mask.data_ptr(), cnt.data_ptr(), None, None)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
L.hive_debug_iters(buf)
names = ["queen", "beetle", "spider", "grass", "ant"]
for t in range(5):
    print(names[t], "waves", buf[8 + t], "mean fused-loop iterations (2 flood steps each)", buf[t] / max(1, buf[8 + t]))
