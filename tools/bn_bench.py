"""BatchNorm2d (training mode) + ReLU forward/backward on the tower's activation shape: MIOpen vs torch native kernels."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
def run(label, hip=False):
    from hive_alphazero_amd.alpha_net import bn_act
    bn = torch.nn.BatchNorm2d(256).cuda()
    x = torch.randn((B, 256, 12, 12), device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    def step():
        y = bn_act(x, bn) if hip else torch.relu(bn(x))
        y.backward(torch.ones_like(y))
        x.grad = None
    for _ in range(5): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): step()
    e1.record(); torch.cuda.synchronize()
    print(f"{label:28s} B={B}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us per BN+ReLU fwd+bwd")
run("HIP (csrc/hive_train.hip)", hip=True)
print(f"   (tensor = {B * 144 * 256 * 2 / 1e6:.1f} MB; the HIP passes move 9 tensor-volumes: x | x,y | dy,y,x | dy,y,x,dx)")
run("MIOpen (cudnn.enabled=True)")
torch.backends.cudnn.enabled = False
run("torch native")
