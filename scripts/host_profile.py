"""Where the host time of a training step goes (synchronised before every step so that only enqueue cost is seen):
 per C entry point (time inside lib.call = ctypes + the library's kernel launches) and a cProfile by own time."""
import argparse, cProfile, collections, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transformer_cnn_hybrid_network_for_video_processing_amd as P
from transformer_cnn_hybrid_network_for_video_processing_amd import _lib

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=30); ap.add_argument("--top", type=int, default=40)
ap.add_argument("--stagewise", action="store_true")
a = ap.parse_args()
torch.manual_seed(0)
m = P.TransformerCNNHybrid().cuda().train()
m.fuse_model_ops = not a.stagewise
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
acc = collections.Counter(); cnt = collections.Counter()
orig = _lib._Lib.call
def timed_call(self, name, *args):
    t0 = time.perf_counter()
    r = orig(self, name, *args)
    acc[name] += time.perf_counter() - t0; cnt[name] += 1
    return r
_lib._Lib.call = timed_call
tot = 0.0
for _ in range(a.steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); tot += time.perf_counter() - t0
torch.cuda.synchronize()
_lib._Lib.call = orig
print(f"host enqueue per step {tot/a.steps*1e3:.3f} ms; inside lib.call {sum(acc.values())/a.steps*1e3:.3f} ms:")
for k, v in acc.most_common():
    print(f"   {k:28s} {v/a.steps*1e6:8.1f} us/step  ({cnt[k]//a.steps} calls)")
pr = cProfile.Profile()
for _ in range(10):
    torch.cuda.synchronize()
    pr.enable(); step(); pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(a.top)
print(s.getvalue()[:9000])
