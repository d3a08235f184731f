"""Arbitrate HIP-fp32 vs the fp32 oracle with an fp64 run of the oracle (config 2): who is closer to the truth, per gradient."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transformer_cnn_hybrid_network_for_video_processing_amd as P
from oracle import hybrid_ref as R
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
ref = R.TransformerCNNHybridRef()
for a in ref.encoder.attention_layers: a.dropoutLayer.p = 0.0
ref.train()
sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
x, y = R.synthetic_batch(B, 16, 224, 224, seed=0)
t = time.time(); R.loss_fn(ref(x), y).backward(); print("fp32 oracle", time.time() - t, flush=True)
g32 = {n: p.grad.clone() for n, p in ref.named_parameters()}
ref64 = R.TransformerCNNHybridRef().double(); ref64.load_state_dict(sd0)
for a in ref64.encoder.attention_layers: a.dropoutLayer.p = 0.0
ref64.train()
t = time.time(); R.loss_fn(ref64(x.double()), y).backward(); print("fp64 oracle", time.time() - t, flush=True)
g64 = {n: p.grad.clone() for n, p in ref64.named_parameters()}
m = P.TransformerCNNHybrid(compute_dtype="fp32"); m.load_state_dict(sd0)
for a in m.encoder.attention_layers: a.dropoutLayer.p = 0.0
m = m.cuda().train()
P.HybridCrossEntropyLoss()(m(x.cuda()), y.cuda()).backward()
gh = {n: p.grad.double().cpu() for n, p in m.named_parameters()}
G = max(g.abs().max().item() for g in g64.values())
for n in g64:
    d = max(g64[n].abs().max().item(), 1e-4 * G)
    print(f"{n:48s} |g|max {g64[n].abs().max().item():.3e}  oracle32-vs-64 {(g32[n].double()-g64[n]).abs().max().item()/d:.2e}   hip32-vs-64 {(gh[n]-g64[n]).abs().max().item()/d:.2e}")
