import sys, torch
sys.path.insert(0, "tests")
from oracle import encoder32k_ref as R
from transformer_cnn_hybrid_network_for_video_processing_amd import encoder32k as M

def rel(got, want, floor=1e-12):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return (got - want).abs().max().item() / max(want.abs().max().item(), floor)

params = {k: (v.float().double() if v.is_floating_point() else v) for k, v in R.make_params(seed=1).items()}
model = M.Encoder_32K()
model.load_state_dict({k: v.clone().float() if v.is_floating_point() else v.clone() for k, v in params.items()})
model = model.cuda().train(); model.dropout.p = 0.0
g = torch.Generator().manual_seed(2)
x = torch.rand(4, 3, 64, 64, generator=g, dtype=torch.float64).float().double()
p = {k: (v.clone().requires_grad_() if v.is_floating_point() and "running" not in k else v.clone()) for k, v in params.items()}
want = R.feature_map(p, x, True)
dy = torch.randn(want.shape, generator=g, dtype=torch.float64)
want.backward(dy)
got = model.feature_map(x.float().cuda())
print("forward", rel(got, want))
got.backward(dy.float().cuda())
errs = {n: (rel(q.grad, p[n].grad), float(p[n].grad.abs().max())) for n, q in model.named_parameters()}
for n in sorted(errs, key=lambda k: -errs[k][0])[:25]:
    print(f"{n:40s} rel {errs[n][0]:.2e}  max|grad| {errs[n][1]:.3e}")
# the same comparison with the oracle itself in float32 (how much of the error is fp32 arithmetic?)
p32 = {k: (v.clone().float().requires_grad_() if v.is_floating_point() and "running" not in k else (v.clone().float() if v.is_floating_point() else v.clone())) for k, v in params.items()}
w32 = R.feature_map(p32, x.float(), True)
w32.backward(dy.float())
e32 = {n: rel(p32[n].grad, p[n].grad) for n, q in model.named_parameters()}
print("oracle fp32 vs fp64: forward", rel(w32, want))
for n in sorted(e32, key=lambda k: -e32[k])[:12]:
    print(f"  fp32-oracle {n:40s} rel {e32[n]:.2e}")
eg = {n: rel(q.grad, p32[n].grad) for n, q in model.named_parameters()}
for n in sorted(eg, key=lambda k: -eg[k])[:8]:
    print(f"  gpu vs fp32-oracle {n:40s} rel {eg[n]:.2e}")
