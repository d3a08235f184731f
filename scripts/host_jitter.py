"""Per-step host enqueue times of a pipelined run (no per-step sync), with the Python cyclic GC on / frozen."""
import gc, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transformer_cnn_hybrid_network_for_video_processing_amd as P
torch.manual_seed(0)
m = P.TransformerCNNHybrid().cuda().train()
opt = P.HybridAdamW(m.parameters(), lr=1e-3)
crit = P.HybridCrossEntropyLoss()
x = torch.rand(8, 16, 3, 224, 224, device="cuda"); y = torch.randint(0, 8, (8,), device="cuda")
def step():
    opt.zero_grad(set_to_none=True)
    loss = crit(m(x), y); loss.backward(); opt.step()
    return loss
def run(n, tag):
    for _ in range(5): step()
    torch.cuda.synchronize()
    ts = []
    t00 = time.perf_counter()
    for _ in range(n):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t00) / n
    s = sorted(ts)
    print(f"{tag}: wall/step {wall*1e3:.3f} ms; host median {s[n//2]*1e3:.3f} p90 {s[int(n*.9)]*1e3:.3f} max {s[-1]*1e3:.3f} ms; steps>3ms: {[ (i, round(t*1e3,1)) for i,t in enumerate(ts) if t>3e-3][:12]}", flush=True)
print("gc counts", gc.get_count(), gc.get_threshold())
run(20, "gc on, 20 steps"); run(100, "gc on, 100 steps"); run(100, "gc on, 100 steps again")
gc.collect(); gc.freeze()
run(100, "gc frozen, 100 steps")
gc.disable()
run(100, "gc disabled, 100 steps")
gc.enable()
