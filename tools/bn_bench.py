"""BatchNorm2d (training mode) + ReLU forward/backward on the tower's activation shape: MIOpen vs torch native kernels."""
import sys, torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
def run(label):
    bn = torch.nn.BatchNorm2d(256).cuda()
    x = torch.randn((B, 256, 12, 12), device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    def step():
        y = torch.relu(bn(x))
        y.backward(torch.ones_like(y))
        x.grad = None
    for _ in range(5): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): step()
    e1.record(); torch.cuda.synchronize()
    print(f"{label:28s} B={B}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us per BN+ReLU fwd+bwd")
run("MIOpen (cudnn.enabled=True)")
torch.backends.cudnn.enabled = False
run("torch native")
