"""bf16x3 mode (fp32 storage, split-bf16 products): logits / gradient error vs the fp32 oracle on a small batch, next to fp32 and bf16 modes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transformer_cnn_hybrid_network_for_video_processing_amd as P
from oracle import hybrid_ref as R
kw = dict(cnn_channels=(32, 64, 128, 256), d_model=512, num_heads=8, num_layers=2, hidden_dim=2048)
torch.manual_seed(0)
ref = R.TransformerCNNHybridRef(**kw)
for a in ref.encoder.attention_layers: a.dropoutLayer.p = 0.0
x, y = R.synthetic_batch(2, 8, 112, 112)
ref.train()
sd = {k: v.clone() for k, v in ref.state_dict().items() if "num_batches_tracked" not in k}
lr = ref(x); R.loss_fn(lr, y).backward()
gr = {n: p.grad for n, p in ref.named_parameters()}
for mode in ("fp32", "bf16x3", "bf16"):
    hip = P.TransformerCNNHybrid(compute_dtype=mode, **kw)
    hip.load_state_dict(sd, strict=False)
    for a in hip.encoder.attention_layers: a.dropoutLayer.p = 0.0
    hip = hip.cuda().train()
    lh = hip(x.cuda()); P.HybridCrossEntropyLoss()(lh, y.cuda()).backward()
    e = ((lh.detach().cpu() - lr).abs().max() / lr.abs().max()).item()
    G = max(g.abs().max().item() for g in gr.values())
    ge = {n: ((p.grad.cpu() - gr[n]).abs().max() / max(gr[n].abs().max().item(), 1e-4 * G)).item() for n, p in hip.named_parameters()}
    w = max(ge, key=ge.get)
    print(f"{mode:7s} logits max-rel {e:.2e}; worst gradient max-rel {ge[w]:.2e} ({w})", flush=True)
