"""Where do the slow training steps come from?  (BENCH_r02: median 17.1 ms, mean 26.3 ms, one step of 106 ms.)
Per-step wall time of Trainer.step after an empty_cache(), beside the caching allocator's counters (device mallocs,
retries) and Python's garbage-collector runs; then the same with the collector frozen / the allocator warm."""
import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_alphazero_amd.alpha_net import ChessNet, Trainer

def run(tag, steps=24, batch=512, freeze_gc=False, empty=True):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = (torch.rand((batch, 56, 12, 12), device="cuda", generator=g) < 0.1).float()
    pi = torch.softmax(torch.randn((batch, 1584), device="cuda", generator=g), 1)
    z = torch.sign(torch.randn((batch,), device="cuda", generator=g))
    torch.manual_seed(0)
    tr = Trainer(ChessNet().cuda(), ddp=False)
    if empty:
        gc.collect(); torch.cuda.empty_cache()
    torch.cuda.synchronize()
    if freeze_gc:
        gc.collect(); gc.freeze(); gc.disable()
    rows = []
    gc_runs = [0]
    def cb(phase, info):
        if phase == "stop": gc_runs[0] += 1
    gc.callbacks.append(cb)
    for i in range(steps):
        s0 = torch.cuda.memory_stats()
        g0 = gc_runs[0]
        t0 = time.perf_counter()
        tr.step(x, pi, z)
        dt = (time.perf_counter() - t0) * 1e3
        s1 = torch.cuda.memory_stats()
        rows.append((i, dt, s1["num_device_alloc"] - s0["num_device_alloc"], s1["num_device_free"] - s0["num_device_free"],
                     s1["num_alloc_retries"] - s0["num_alloc_retries"], gc_runs[0] - g0,
                     s1["reserved_bytes.all.current"] / 2**30))
    gc.callbacks.remove(cb)
    if freeze_gc:
        gc.enable(); gc.unfreeze()
    print(f"## {tag}")
    print("| step | ms | device mallocs | device frees | alloc retries | gc runs | reserved GiB |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %d | %.2f | %d | %d | %d | %d | %.2f |" % r)
    late = sorted(r[1] for r in rows[8:])
    print(f"steps 8..: median {late[len(late)//2]:.2f} ms, mean {sum(late)/len(late):.2f} ms, max {late[-1]:.2f} ms\n", flush=True)
    del tr

run("as bench.py runs it (allocator emptied first)")
run("collector frozen during the steps", freeze_gc=True)
