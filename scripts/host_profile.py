"""Host enqueue cost of one training step (the cProfile method of DESIGN.md section 6): the step is enqueued WITHOUT waiting for
the GPU, so the wall time of the Python loop is the host's cost; a synchronised loop gives the GPU-bound step time beside it."""
import argparse, cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transformer_cnn_hybrid_network_for_video_processing_amd as P

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=30); ap.add_argument("--top", type=int, default=25)
a = ap.parse_args()
torch.manual_seed(0)
m = P.TransformerCNNHybrid().cuda().train()
opt = P.HybridAdamW(m.parameters(), lr=1e-3)
crit = P.HybridCrossEntropyLoss()
x = torch.rand(8, 16, 3, 224, 224, device="cuda"); y = torch.randint(0, 8, (8,), device="cuda")

def step():
    opt.zero_grad(set_to_none=True)
    loss = crit(m(x), y)
    loss.backward()
    opt.step()

for _ in range(5):
    step()
torch.cuda.synchronize()
# host-only: enqueue `steps` steps; the queue depth is bounded by the runtime, so keep it short and sync between batches
host = []
for _ in range(a.steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / a.steps
host.sort()
print(f"host enqueue per step: median {host[len(host)//2]*1e3:.3f} ms, min {host[0]*1e3:.3f} ms; pipelined step {wall*1e3:.3f} ms")
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(a.top)
print(s.getvalue()[:6000])
