import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hybrid_ref as R
import transformer_cnn_hybrid_network_for_video_processing_amd as P

def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item(), ((a - b).norm() / b.norm().clamp_min(1e-30)).item()

kw = dict(cnn_channels=(32, 64), d_model=64, num_heads=4, num_layers=2, hidden_dim=128)
for mode in ("fp32", "bf16"):
    torch.manual_seed(0)
    ref = R.TransformerCNNHybridRef(**kw); hip = P.TransformerCNNHybrid(compute_dtype=mode, **kw); hip.load_state_dict(ref.state_dict()); hip.cuda()
    x, y = R.synthetic_batch(2, 4, 64, 64)
    for training in (False, True):
        ref.train(training); hip.train(training)
        for m in (ref, hip):
            for a in m.encoder.attention_layers: a.dropoutLayer.p = 0.0
        ref.zero_grad(); hip.zero_grad()
        lr = ref(x); R.loss_fn(lr, y).backward()
        lh = hip(x.cuda()); P.HybridCrossEntropyLoss()(lh, y.cuda()).backward()
        print(mode, "training", training, "logits", rel(lh, lr))
        hp = dict(hip.named_parameters())
        for n, p in ref.named_parameters():
            print("   %-50s max-rel %.3e  l2-rel %.3e  |ref|max %.3e" % ((n,) + rel(hp[n].grad, p.grad) + (p.grad.abs().max().item(),)))
