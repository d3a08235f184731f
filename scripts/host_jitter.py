"""Host cost of a training step and how it shows up in short timed windows.
 (a) pure host enqueue time: synchronise before every step, time only the Python/HIP enqueue;  (b) bench.py's protocol
 (sync, K pipelined steps, sync) repeated, to see the spread a 20-step window has."""
import gc, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transformer_cnn_hybrid_network_for_video_processing_amd as P
torch.manual_seed(0)
m = P.TransformerCNNHybrid().cuda().train()
if "--stagewise" in sys.argv:
    m.fuse_model_ops = False
opt = P.HybridAdamW(m.parameters(), lr=1e-3)
crit = P.HybridCrossEntropyLoss()
x = torch.rand(8, 16, 3, 224, 224, device="cuda"); y = torch.randint(0, 8, (8,), device="cuda")
def step():
    opt.zero_grad(set_to_none=True)
    loss = crit(m(x), y); loss.backward(); opt.step()
    return loss
def pure_host(n, tag):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    s = sorted(ts)
    print(f"{tag}: pure host enqueue median {s[n//2]*1e3:.3f} p90 {s[int(n*.9)]*1e3:.3f} max {s[-1]*1e3:.3f} ms", flush=True)
def window(k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
for _ in range(5): step()
pure_host(50, "gc on")
print("20-step windows (gc on):    ", " ".join(f"{window(20):.3f}" for _ in range(8)), flush=True)
print("100-step windows (gc on):   ", " ".join(f"{window(100):.3f}" for _ in range(3)), flush=True)
gc.collect(); gc.freeze()
pure_host(50, "gc frozen")
print("20-step windows (gc frozen):", " ".join(f"{window(20):.3f}" for _ in range(8)), flush=True)
print("100-step windows (frozen):  ", " ".join(f"{window(100):.3f}" for _ in range(3)), flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(50): step()
e1.record(); torch.cuda.synchronize()
print(f"GPU time per step by events over 50 pipelined steps: {e0.elapsed_time(e1)/50:.3f} ms")
