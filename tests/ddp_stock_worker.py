"""Worker of tests/test_gpu_dp.py::test_stock_ddp_wraps_the_hip_model (launched by torch.distributed.run, 2 ranks sharing cuda:0, gloo).
SURVEY.md section 8b promises that "stock AdamW / DDP / checkpointing work unchanged" on the drop-in module, and section 7.1 step 7 starts
from DDP: so the SAME TransformerCNNHybrid is wrapped in torch's own torch.nn.parallel.DistributedDataParallel (nothing of this repo's dp.py
or graph.py involved), one forward + backward on a per-rank shard, and the gradients DDP leaves in p.grad are compared with
 (a) dp.GradAllReducer's on an identical copy (same weights, same shard), and
 (b) the plain mean of the two ranks' local gradients, gathered by hand.
BatchNorm statistics stay per rank in both (broadcast_buffers=False: the reference has no SyncBN and dp.py does not broadcast buffers
per step).  One stock torch.optim.AdamW step on the DDP model must then leave both ranks with equal parameters."""
import os
import sys

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import transformer_cnn_hybrid_network_for_video_processing_amd as P
from transformer_cnn_hybrid_network_for_video_processing_amd.dp import GradAllReducer

dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
kw = dict(cnn_channels=(32, 64, 128, 256), d_model=128, num_heads=4, num_layers=2, hidden_dim=256, dropout=0.0)


def make():
    torch.manual_seed(0)
    m = P.TransformerCNNHybrid(**kw).to(dev).train()
    for a in m.encoder.attention_layers:
        a.dropoutLayer.p = 0.0
    return m


g = torch.Generator().manual_seed(1000 + rank)
x = torch.rand(2, 4, 3, 64, 64, generator=g).to(dev)
y = torch.randint(0, 8, (2,), generator=g).to(dev)
crit = P.HybridCrossEntropyLoss()

# local gradients (no communication), then their mean over ranks by hand
ml = make()
crit(ml(x), y).backward()
want = []
for p in ml.parameters():
    t = p.grad.detach().clone()
    dist.all_reduce(t)
    want.append(t / world)

# stock DDP
md = DDP(make(), device_ids=[0], broadcast_buffers=False, bucket_cap_mb=1)
loss_d = crit(md(x), y)
loss_d.backward()
torch.cuda.synchronize()
gd = [p.grad.detach().clone() for p in md.module.parameters()]

# this repo's reducer
mr = make()
red = GradAllReducer(mr, bucket_bytes=1 << 20)
crit(mr(x), y).backward()
red.finalize()
torch.cuda.synchronize()
gr = [p.grad.detach().clone() for p in mr.parameters()]


def err(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


names = [n for n, _ in ml.named_parameters()]
e_ddp_mean = max(err(a, b) for a, b in zip(gd, want))
e_ddp_red = max(err(a, b) for a, b in zip(gd, gr))
nonzero = sum(1 for t in want if t.abs().max().item() > 0)

# one stock AdamW step on the DDP-wrapped model: ranks must stay in lock-step
opt = torch.optim.AdamW(md.parameters(), lr=1e-3)
opt.step()
torch.cuda.synchronize()
flat = torch.cat([p.detach().flatten() for p in md.module.parameters()])
others = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(others, flat)
same = all(torch.equal(o, flat) for o in others)
# and the state dict still carries the reference's key names under DDP's "module." prefix
keys_ok = all(k.startswith("module.") for k in md.state_dict()) and [k[7:] for k in md.state_dict()] == list(ml.state_dict())
print(f"DDPW rank {rank}: {len(names)} tensors ({nonzero} non-zero); DDP vs hand-averaged {e_ddp_mean:.2e}; DDP vs GradAllReducer {e_ddp_red:.2e}; "
      f"ranks equal after AdamW {same}; state-dict keys {keys_ok}", flush=True)
ok = e_ddp_mean <= 1e-6 and e_ddp_red <= 1e-6 and same and keys_ok and nonzero == len(names)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
