"""Where do the bf16 HIP path and the bf16-rounded oracle part ways?  Stage-by-stage forward comparison (fp64-accumulating rounded oracle)."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transformer_cnn_hybrid_network_for_video_processing_amd as P
from transformer_cnn_hybrid_network_for_video_processing_amd import ops
from oracle import hybrid_ref as R, hybrid_ref_bf16 as RB
import torch.nn.functional as F
B, T, H = 2, 8, 112
torch.manual_seed(0)
ref = R.TransformerCNNHybridRef().double()
for a in ref.encoder.attention_layers: a.dropoutLayer.p = 0.0
ref.train()
x, y = R.synthetic_batch(B, T, H, H, seed=0)
m = P.TransformerCNNHybrid(compute_dtype="bf16"); m.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in ref.state_dict().items()})
for a in m.encoder.attention_layers: a.dropoutLayer.p = 0.0
m = m.cuda().train()
def cmp(name, got, want):
    got, want = got.double().cpu(), want.double()
    d = (got - want)
    nz = (d != 0).double().mean().item()
    print(f"{name:28s} L2-rel {d.norm().item()/want.norm().item():.3e}  max-rel {d.abs().max().item()/want.abs().max().item():.3e}  differing elements {nz*100:.3f}%", flush=True)
f = x.reshape(B * T, 3, H, H)
fo = f.double()
h = f.cuda()
for i in range(4):
    st = getattr(m, f"encoder{i+1}")
    h = st.forward_nhwc(h, i == 0)
    fo = RB.conv_stage(getattr(ref, f"encoder{i+1}"), f"enc{i+1}", fo, i == 0, True)
    cmp(f"stage {i+1} pooled", ops.nhwc_to_nchw(h, m._dt, fo.shape[1]), fo)
    # restart the HIP path from the oracle's values so that errors do not compound in this report
    h = ops.nchw_to_nhwc(fo.float().cuda(), m._dt, h.shape[3])
feat = RB.rb(fo.mean(dim=(2, 3)))
tok_o = RB._linear(feat, ref.token_proj).reshape(B, T, -1)
tok_h = ops.token(h, m.token_proj.weight, m.token_proj.bias, m._dt).reshape(B, T, -1)
cmp("tokens", tok_h, tok_o)
enc_o = RB._encoder(ref.encoder, tok_o, None)
enc_h = m.encoder.forward_compute(tok_o.float().cuda().bfloat16(), None)
cmp("encoder out (2 layers)", enc_h, enc_o)
# one MHA + pieces
att = ref.encoder.attention_layers[0]
mh = m.encoder.attention_layers[0]
o_o = RB._mha(att, tok_o, None)
o_h = ops.mha(tok_o.float().cuda().bfloat16(), tok_o.float().cuda().bfloat16(), tok_o.float().cuda().bfloat16(), None, mh._params(), m._dt, 8, 0.0, 1)
cmp("mha out (layer 0)", o_h, o_o)
q_o = RB._linear(tok_o, att.query_layer, True)
q_h = torch.empty(B, T, 512, dtype=torch.bfloat16, device="cuda")
from transformer_cnn_hybrid_network_for_video_processing_amd._lib import lib
tb = tok_o.float().cuda().bfloat16().contiguous()
lib.call("hyb_linear_fwd", 1, tb.data_ptr(), 512, mh.query_layer.weight.data_ptr(), mh.query_layer.bias.data_ptr(), q_h.data_ptr(), B * T, 512, 512, 1, torch.cuda.current_stream().cuda_stream)
cmp("q = relu(linear)", q_h, q_o)
