"""Where does the bf16 mode's logits error (6.7e-3 at config 2) come from?  Same weights in a bf16 and an fp32 model; the two halves are crossed:
   bf16 backbone -> fp32 temporal part, fp32 backbone -> bf16 temporal part.  Errors are max |logit - logit_fp32| (eval mode, no dropout)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import transformer_cnn_hybrid_network_for_video_processing_amd as P
from transformer_cnn_hybrid_network_for_video_processing_amd import ops
torch.manual_seed(0)
B, T, S = 8, 16, 224
kw = dict(in_channels=3, cnn_channels=(32, 64, 128, 256), d_model=512, num_heads=8, hidden_dim=2048, num_layers=2, num_classes=8, dropout=0.1)
try:
    m32 = P.TransformerCNNHybrid(compute_dtype="fp32", **kw).cuda()
except TypeError as e:
    print("constructor:", e); raise
m16 = P.TransformerCNNHybrid(compute_dtype="bf16", **kw).cuda(); m16.load_state_dict(m32.state_dict())
mx3 = P.TransformerCNNHybrid(compute_dtype="bf16x3", **kw).cuda(); mx3.load_state_dict(m32.state_dict())
x = torch.rand(B, T, 3, S, S, device="cuda")
out = {}
for train in (False, True):
    for m in (m32, m16, mx3):
        m.train(train)
        for mod in m.modules():
            if hasattr(mod, "p") and isinstance(getattr(mod, "p"), float): pass
    # dropout off in both cases: the comparison is about arithmetic
    for m in (m32, m16, mx3): m.encoder.dropout = 0.0
    with torch.no_grad():
        h32, _ = m32.forward_backbone(x); h16, _ = m16.forward_backbone(x)
        ref = m32.forward_temporal(h32, B).float()
        full16 = m16.forward_temporal(h16, B).float()
        h16_as32 = h16.float()
        h32_as16 = h32.bfloat16()
        a = m32.forward_temporal(h16_as32, B).float()          # bf16 backbone, fp32 temporal
        b = m16.forward_temporal(h32_as16, B).float()          # fp32 backbone (rounded once), bf16 temporal
        c = mx3.forward_temporal(h16_as32, B).float()          # bf16 backbone, bf16x3 temporal
    err = lambda t: float((t - ref).abs().max())
    out["train" if train else "eval"] = {"logit_scale": float(ref.abs().max()), "bf16_full": err(full16), "bf16_backbone_fp32_temporal": err(a),
                                         "fp32_backbone_bf16_temporal": err(b), "bf16_backbone_bf16x3_temporal": err(c)}
print(json.dumps(out, indent=1))
