"""bf16-mode forward error against the fp32 CPU oracle on the smoke workload ([2,4,3,64,64]) and a [2,8,3,112,112] batch, over several
seeds: is the round-1 -> round-2 change of the smoke number (4.99e-3 -> 9.12e-3) a systematic loss or the spread of a max-over-16-logits
statistic?  Run once per kernel-selection environment (HYB_S1_GRAM=0, HYB_S1_WAVE=0, HYB_CONV_V2=0, ...): the switches are read once per process."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transformer_cnn_hybrid_network_for_video_processing_amd as P
from oracle import hybrid_ref as R
kw = dict(cnn_channels=(32, 64, 128, 256), d_model=512, num_heads=8, num_layers=2, hidden_dim=2048)
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("HYB_")}, "cases": []}
for (B, T, S) in ((2, 4, 64), (2, 8, 112)):
    errs, errs32 = [], []
    for seed in range(8):
        torch.manual_seed(seed)
        ref = R.TransformerCNNHybridRef(**kw)
        for a in ref.encoder.attention_layers: a.dropoutLayer.p = 0.0
        x, y = R.synthetic_batch(B, T, S, S, seed=seed)
        ref.train()
        sd = {k: v.clone() for k, v in ref.state_dict().items() if "num_batches_tracked" not in k}
        with torch.no_grad(): lr = ref(x)
        for mode, acc in (("bf16", errs), ("fp32", errs32)):
            hip = P.TransformerCNNHybrid(compute_dtype=mode, **kw)
            hip.load_state_dict(sd, strict=False)
            for a in hip.encoder.attention_layers: a.dropoutLayer.p = 0.0
            hip = hip.cuda().train()
            with torch.no_grad(): lh = hip(x.cuda())
            acc.append(((lh.cpu() - lr).abs().max() / lr.abs().max()).item())
    out["cases"].append({"shape": [B, T, 3, S, S], "bf16_logits_rel_err": errs, "bf16_mean": sum(errs) / len(errs), "bf16_max": max(errs),
                         "fp32_max": max(errs32)})
print(json.dumps(out))
