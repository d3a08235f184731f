"""Attention core (QK^T/sqrt(d) -> softmax -> PV, fwd and bwd) at the literal BASELINE shapes and at saturating batch*heads.
Reports time, algorithmic TFLOP/s (4*T^2*d_h per problem fwd, 10*T^2*d_h bwd) and algorithmic GB/s (q,k,v,out + the row statistics;
backward: q,k,v,dout in, dq,dk,dv out).  HYB_ATTN_QROWS=16|32|64 selects the query rows per wave (tile sweep)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_cnn_hybrid_network_for_video_processing_amd._lib import lib

def run(B, S, D, H, reps=20):
    dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
    q, k, v, do = (torch.randn(B, S, D, device=dev).to(torch.bfloat16) for _ in range(4))
    out = torch.empty_like(q); probs = torch.empty(B * H, S, 2, device=dev)      # softmax row statistics
    dq, dk, dv = (torch.empty_like(q) for _ in range(3))
    def fwd(): lib.call("hyb_attention_fwd", 1, q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(), probs.data_ptr(), B, S, D, H, 0.0, 0, st)
    def bwd(): lib.call("hyb_attention_bwd", 1, q.data_ptr(), k.data_ptr(), v.data_ptr(), None, probs.data_ptr(), do.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, S, D, H, 0.0, 0, st)
    res = {}
    for name, fn, fl, by in (("fwd", fwd, 4.0 * S * S * (D // H), 4 * S * (D // H) * 2 + S * 8),
                             ("bwd", bwd, 10.0 * S * S * (D // H), 7 * S * (D // H) * 2 + S * 8)):
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res[name] = dict(ms=round(ms, 4), TFLOPs=round(fl * B * H / ms / 1e9, 2), GBps=round(by * B * H / ms / 1e6, 1))
    return res

rows = []
for label, B, S, D, H in [("config2 literal (B=8,T=16,d=512,h=8)", 8, 16, 512, 8), ("config4 literal (B=8,T=64,d=768,h=8)", 8, 64, 768, 8),
                          ("T=16 d_h=64 saturating (B*H=65536)", 8192, 16, 512, 8), ("T=64 d_h=96 saturating (B*H=16384)", 2048, 64, 768, 8),
                          ("T=64 d_h=64 saturating (B*H=16384)", 2048, 64, 512, 8)]:
    r = run(B, S, D, H)
    rows.append(dict(case=label, problems=B * H, **{f"{k}_{kk}": vv for k, d in r.items() for kk, vv in d.items()}))
    print(rows[-1])
tag = os.environ.get("HYB_ATTN_QROWS", "16")
json.dump(rows, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"attn_microbench_qrows{tag}.json"), "w"), indent=1)
