"""FCT on the HIP path, forward and backward: every operator against stock torch on the CPU, every block and
the whole model against the golden vectors captured from the REFERENCE's own classes (tests/golden/g3..g8: outputs, input and
parameter gradients), and a 224 x 224 batch against the CPU oracle (oracle/fct_ref.py, itself pinned by the same goldens in
tests/test_oracle_fct.py).
Tolerances: forward max|got - want| / max|want| <= 1e-3 and gradients <= 1e-2 (north_star's tolerances); the fp32 path measures
~1e-6 / ~1e-5, and the operator tests gate at 1e-5 / 1e-4."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)
from det_init import det_state_dict, digest  # noqa: E402
from oracle import fct_ref as F  # noqa: E402
import transformer_cnn_hybrid_network_for_video_processing_amd  # noqa: E402,F401  (registers torch.ops.hybrid.*)


def P():
    import transformer_cnn_hybrid_network_for_video_processing_amd as pkg
    return pkg


def fct():
    from transformer_cnn_hybrid_network_for_video_processing_amd import fct as m
    return m


def rel(got, want, floor=1e-12):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    return (got - want).abs().max().item() / max(want.abs().max().item(), floor)


def grad_floor(grads):
    """Gradients that are zero in exact arithmetic (the key LayerNorm's bias and the key in-projection bias: a constant added to every
    key leaves the softmax unchanged) hold only rounding noise, ~1e-9: errors are measured against at least 1e-4 of the largest
    gradient of the block."""
    return 1e-4 * max(float(np.abs(np.asarray(v)).max()) for v in grads)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(y):
    return y.permute(0, 3, 1, 2).cpu()


def gold(name):
    return dict(np.load(os.path.join(GOLD, name), allow_pickle=False))


@pytest.mark.parametrize("ci,co,n,h,w,dil,act", [(3, 8, 2, 16, 16, 1, 1), (8, 8, 1, 12, 20, 2, 2), (16, 8, 2, 9, 7, 3, 2), (8, 1, 1, 16, 16, 1, 3),
                                                 (128, 64, 1, 8, 8, 1, 0), (11, 5, 3, 10, 6, 1, 1), (64, 128, 2, 7, 7, 1, 1)])
def test_conv3x3_bias_dilation_activation(ci, co, n, h, w, dil, act):
    torch.manual_seed(ci + co)
    x = torch.randn(n, ci, h, w)
    conv = torch.nn.Conv2d(ci, co, 3, 1, padding="same", dilation=dil)
    want = conv(x)
    want = [want, torch.relu(want), TF.gelu(want), torch.sigmoid(want)][act]
    r = torch.randn_like(want)
    x.requires_grad_(True)
    want = [conv(x), torch.relu(conv(x)), TF.gelu(conv(x)), torch.sigmoid(conv(x))][act]
    (want * r).sum().backward()
    xg, wg, bg = nhwc(x.detach()).requires_grad_(True), conv.weight.detach().cuda().requires_grad_(True), conv.bias.detach().cuda().requires_grad_(True)
    got = torch.ops.hybrid.fct_conv(xg, wg, bg, dil, act)[0]
    assert rel(nchw(got), want) <= 1e-5
    (got * nhwc(r)).sum().backward()
    assert rel(nchw(xg.grad), x.grad) <= 1e-4 and rel(wg.grad, conv.weight.grad) <= 1e-4 and rel(bg.grad, conv.bias.grad) <= 1e-4
    # weights only (first layer of a network: no dx is formed)
    wg.grad = None
    (torch.ops.hybrid.fct_conv(nhwc(x.detach()), wg, None, dil, act)[0] * nhwc(r)).sum().backward()
    assert wg.grad is not None


@pytest.mark.parametrize("c,n,h,w", [(8, 2, 8, 8), (16, 1, 5, 9), (64, 1, 6, 6), (128, 2, 4, 4)])
def test_qkv_projection_and_layernorm(c, n, h, w):
    torch.manual_seed(c)
    att = F.Attention(c, 2)
    for ln in (att.layernorm_q, att.layernorm_k, att.layernorm_v):
        torch.nn.init.normal_(ln.weight, 1.0, 0.2); torch.nn.init.normal_(ln.bias, 0.0, 0.2)
    x = torch.randn(n, c, h, w, requires_grad=True)
    convs, lns = (att.conv_q, att.conv_k, att.conv_v), (att.layernorm_q, att.layernorm_k, att.layernorm_v)
    dev = lambda mods, a: [getattr(m, a).detach().cuda().requires_grad_(True) for m in mods]
    xg = nhwc(x.detach()).requires_grad_(True)
    params = [dev(convs, "weight"), dev(convs, "bias"), dev(lns, "weight"), dev(lns, "bias")]
    got = torch.ops.hybrid.fct_qkv_proj(xg, *params, 1e-5)
    want = [att._project(x, cv, ln) for cv, ln in zip(convs, lns)]
    rs = [torch.randn_like(w_) for w_ in want]
    for g, w_ in zip(got, want):
        assert rel(nchw(g), w_) <= 1e-5
    sum((w_ * r).sum() for w_, r in zip(want, rs)).backward()
    sum((g * nhwc(r)).sum() for g, r in zip(got, rs)).backward()
    assert rel(nchw(xg.grad), x.grad) <= 1e-4
    for i in range(3):
        for got_p, ref_p in ((params[0][i], convs[i].weight), (params[1][i], convs[i].bias), (params[2][i], lns[i].weight), (params[3][i], lns[i].bias)):
            assert rel(got_p.grad, ref_p.grad) <= 1e-4
    # only q used downstream: the missing gradients count as zeros
    xg.grad = None
    torch.ops.hybrid.fct_qkv_proj(xg, *params, 1e-5)[0].sum().backward()
    assert torch.isfinite(xg.grad).all()
    ln = lns[0]
    x2 = x.detach().clone().requires_grad_(True)
    ln.zero_grad()
    want = ln(x2.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
    r = torch.randn_like(want)
    (want * r).sum().backward()
    xg, wg, bg = nhwc(x2.detach()).requires_grad_(True), ln.weight.detach().cuda().requires_grad_(True), ln.bias.detach().cuda().requires_grad_(True)
    y = torch.ops.hybrid.fct_ln(xg, wg, bg, 1e-5)
    assert rel(nchw(y), want) <= 1e-5
    (y * nhwc(r)).sum().backward()
    assert rel(nchw(xg.grad), x2.grad) <= 1e-4 and rel(wg.grad, ln.weight.grad) <= 1e-4 and rel(bg.grad, ln.bias.grad) <= 1e-4


# (.., 1024 / 1100 / 1283 / 1029, 8, 2) and (.., 4096 / 1100 / 1027, 16, 2): heads of 4 and of 8 features with >= 1024 tokens take the 4x4x1-MFMA
# kernels (a query / key per lane; one or two feature quads);
# 1100, 1283 and 1029 are not multiples of 16 or 256: ragged key groups, padding queries, a partial last workgroup
@pytest.mark.parametrize("n,l,c,heads", [(2, 64, 8, 2), (1, 100, 16, 2), (2, 1024, 8, 2), (1, 257, 64, 2), (1, 49, 128, 2), (3, 16, 32, 4), (1, 4096, 16, 2),
                                         (1, 1100, 8, 2), (2, 1283, 8, 2), (1, 1029, 8, 2), (1, 1100, 16, 2), (2, 1027, 16, 2)])
def test_multihead_attention_over_pixel_tokens(n, l, c, heads):
    torch.manual_seed(l + c)
    mha = torch.nn.MultiheadAttention(c, heads, batch_first=True)
    torch.nn.init.normal_(mha.in_proj_bias, 0.0, 0.2); torch.nn.init.normal_(mha.out_proj.bias, 0.0, 0.2)
    q, k, v = ((torch.randn(n, l, c) * 2.0).requires_grad_(True) for _ in range(3))
    want = mha(q, k, v, need_weights=False)[0]
    r = torch.randn_like(want)
    (want * r).sum().backward()
    names = ("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias")
    ref_p = [mha.get_parameter(nm) for nm in names]
    dev_p = [p.detach().cuda().requires_grad_(True) for p in ref_p]
    dq, dk, dv = (t.detach().cuda().requires_grad_(True) for t in (q, k, v))
    got = torch.ops.hybrid.fct_mha(dq, dk, dv, *dev_p, heads)[0]
    assert rel(got, want) <= 2e-5
    (got * r.cuda()).sum().backward()
    for nm, g_, w_ in zip(("q", "k", "v") + names, (dq, dk, dv, *dev_p), (q, k, v, *ref_p)):
        assert rel(g_.grad, w_.grad) <= 2e-4, nm


def test_resample_concat_add_dice_dropout_backward():
    torch.manual_seed(2)
    for mode, fn, shape in ((0, lambda t: TF.max_pool2d(t, 2), (2, 5, 6, 10)), (0, lambda t: TF.max_pool2d(t, 2), (1, 3, 7, 5)),
                            (2, lambda t: TF.interpolate(t, scale_factor=2), (2, 5, 6, 10))):
        x = torch.randn(*shape, requires_grad=True)
        want = fn(x)
        r = torch.randn_like(want)
        (want * r).sum().backward()
        xg = nhwc(x.detach()).requires_grad_(True)
        (torch.ops.hybrid.fct_resample(xg, mode) * nhwc(r)).sum().backward()
        assert rel(nchw(xg.grad), x.grad) <= 1e-6
    # ties in a pooling window: the gradient goes to one element, like torch (first maximum in scan order)
    x = torch.zeros(1, 2, 4, 4, requires_grad=True)
    TF.max_pool2d(x, 2).sum().backward()
    xg = nhwc(x.detach()).requires_grad_(True)
    torch.ops.hybrid.fct_resample(xg, 0).sum().backward()
    assert torch.equal(nchw(xg.grad), x.grad)
    with pytest.raises(NotImplementedError, match="input frames only"):
        torch.ops.hybrid.fct_resample(xg, 1)
    a, b = torch.randn(2, 4, 4, 5, device="cuda", requires_grad=True), torch.randn(2, 4, 4, 3, device="cuda", requires_grad=True)
    r = torch.randn(2, 4, 4, 8, device="cuda")
    (torch.ops.hybrid.fct_concat(a, b) * r).sum().backward()
    assert torch.equal(a.grad, r[..., :5]) and torch.equal(b.grad, r[..., 5:])
    a.grad = None
    (torch.ops.hybrid.fct_add(a, a.detach() * 2) * r[..., :5]).sum().backward()
    assert torch.equal(a.grad, r[..., :5].contiguous())
    g = gold("g7_dice_loss.npz")
    pred = torch.from_numpy(g["pred"]).cuda().requires_grad_(True)
    (P().DiceLoss()(pred, torch.from_numpy(g["true"]).cuda()) * 1.0).backward()
    assert rel(pred.grad, torch.from_numpy(g["dpred"])) <= 1e-5
    big_p, big_t = torch.rand(4, 3, 64, 64, requires_grad=True), (torch.rand(4, 3, 64, 64) > 0.5).float()
    (F.DiceLoss()(big_p, big_t) * 3.0).backward()
    dp = big_p.detach().cuda().requires_grad_(True)
    (P().DiceLoss()(dp, big_t.cuda()) * 3.0).backward()
    assert rel(dp.grad, big_p.grad) <= 1e-5 and torch.count_nonzero(dp.grad[:, 1:]) == 0
    # dropout: keep rate, scaling, same mask in the backward, a new mask per seed
    x = torch.randn(1 << 20, device="cuda").abs_().add_(0.1).requires_grad_(True)
    y = torch.ops.hybrid.fct_dropout(x, 0.3, 1234)
    keep = (y != 0)
    assert abs(keep.float().mean().item() - 0.7) < 3e-3 and torch.allclose(y[keep], x.detach()[keep] / 0.7, rtol=1e-6)
    y.sum().backward()
    assert torch.equal(x.grad != 0, keep) and torch.allclose(x.grad[keep], torch.full_like(x.grad[keep], 1 / 0.7))
    assert torch.equal(torch.ops.hybrid.fct_dropout(x, 0.3, 1234), y) and not torch.equal(torch.ops.hybrid.fct_dropout(x, 0.3, 1235) != 0, keep)
    inc = torch.tensor([1], dtype=torch.int64, device="cuda")
    assert torch.equal(torch.ops.hybrid.fct_dropout(x, 0.3, 1234, inc), torch.ops.hybrid.fct_dropout(x, 0.3, 1235))


def test_resample_concat_add_dice():
    torch.manual_seed(1)
    x = torch.randn(2, 5, 6, 10)
    assert torch.equal(nchw(torch.ops.hybrid.fct_resample(nhwc(x), 0)), TF.max_pool2d(x, 2))
    assert rel(nchw(torch.ops.hybrid.fct_resample(nhwc(x), 1)), TF.avg_pool2d(x, 2, 2)) <= 1e-6
    assert torch.equal(nchw(torch.ops.hybrid.fct_resample(nhwc(x), 2)), TF.interpolate(x, scale_factor=2))
    odd = torch.randn(1, 3, 7, 5)
    assert torch.equal(nchw(torch.ops.hybrid.fct_resample(nhwc(odd), 0)), TF.max_pool2d(odd, 2))
    y = torch.randn(2, 3, 6, 10)
    assert torch.equal(nchw(torch.ops.hybrid.fct_concat(nhwc(x), nhwc(y))), torch.cat([x, y], 1))
    assert torch.equal(nchw(torch.ops.hybrid.fct_add(nhwc(x), nhwc(x * 2))), x + x * 2)
    g = gold("g7_dice_loss.npz")
    loss = P().DiceLoss()(torch.from_numpy(g["pred"]).cuda(), torch.from_numpy(g["true"]).cuda())
    assert abs(loss.item() - float(g["loss"])) < 1e-6 and abs(loss.item() - 0.558098316) < 1e-6
    kat = P().DiceLoss()(torch.tensor([[0.5, 0.5], [1.0, 0.0]]).view(1, 1, 2, 2).cuda(), torch.tensor([[1.0, 0.0], [1.0, 0.0]]).view(1, 1, 2, 2).cuda())
    assert abs(kat.item() - 0.2) < 1e-6
    big_p, big_t = torch.rand(4, 3, 64, 64), (torch.rand(4, 3, 64, 64) > 0.5).float()
    assert abs(P().DiceLoss()(big_p.cuda(), big_t.cuda()).item() - F.DiceLoss()(big_p, big_t).item()) < 1e-6


@pytest.mark.parametrize("name,ctor,extra", [
    ("g3_fct_block_first.npz", lambda: fct().Block_encoder_bottleneck("first", 3, 8, 2, 0), ()),          # the reference's own 5-argument call (make_golden.py)
    ("g3b_fct_block_second.npz", lambda: fct().Block_encoder_bottleneck("second", 8, 16, 2, 0), ("scale_img",)),
    ("g4_fct_attention.npz", lambda: fct().Attention(8, 2), ()),
    ("g4b_fct_transformer.npz", lambda: fct().Transformer(8, 8, 2), ()),
    ("g5_fct_wide_focus.npz", lambda: fct().Wide_Focus(8, 8), ()),
    ("g6_fct_block_decoder.npz", lambda: fct().Block_decoder(16, 8, 2, 0), ("skip",)),
    ("g6b_fct_ds_out.npz", lambda: fct().DS_out(8, 1), ()),
])
def test_reference_block_goldens_forward_and_gradients(name, ctor, extra):
    g = gold(name)
    m = ctor()
    m.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd::")})
    m = m.cuda().eval()
    x = nhwc(torch.from_numpy(g["x"])).requires_grad_(True)
    y = m(x, *[nhwc(torch.from_numpy(g[k])) for k in extra])
    assert rel(nchw(y), torch.from_numpy(g["out"])) <= 1e-3          # the gate; measured ~1e-6
    (y * nhwc(torch.from_numpy(g["r"]))).sum().backward()
    errs = {"dx": rel(nchw(x.grad), torch.from_numpy(g["dx"]))}
    named = dict(m.named_parameters())
    want = {k[6:]: v for k, v in g.items() if k.startswith("grad::")}
    assert {k for k, p in named.items() if p.grad is not None} == set(want)          # the same parameters take part
    fl = grad_floor(want.values())
    for k, v in want.items():
        errs[k] = rel(named[k].grad, torch.from_numpy(v), fl)
    worst = max(errs, key=errs.get)
    print(f"\n[{name}] worst gradient error {errs[worst]:.2e} ({worst})")
    assert errs[worst] <= 1e-2, errs                                  # the gate; measured ~1e-5


def test_reference_whole_model_golden_g8_and_default_init():
    g = gold("g8_fct_full.npz")
    m = P().FCT()
    assert [k for k, _ in m.named_parameters()] == list(g["param_names"])
    m.load_state_dict(det_state_dict(m))
    m = m.cuda().eval()
    with torch.no_grad():
        out = m(torch.from_numpy(g["x"]).cuda())
    assert out.shape == (1, 1, 64, 64)
    e = rel(out, torch.from_numpy(g["out"]))
    print(f"\n[FCT G8] forward max-rel error vs the reference's output: {e:.2e}")
    assert e <= 1e-3
    loss = P().DiceLoss()(out, torch.from_numpy(g["y_true"]).cuda())
    assert abs(loss.item() - float(g["loss"])) <= 1e-5
    # gradients of the Dice loss: the reference's digests (sum, norm, samples) for every parameter, three tensors in full
    out = m(torch.from_numpy(g["x"]).cuda())
    P().DiceLoss()(out, torch.from_numpy(g["y_true"]).cuda()).backward()
    named = dict(m.named_parameters())
    assert sorted(k for k, p in named.items() if p.grad is None) == sorted(g["unused_parameters"])
    worst = 0.0
    fl = grad_floor(g[k][2:] for k in g if k.startswith("gdig::"))
    for k, p in named.items():
        if p.grad is None:
            continue
        want = torch.from_numpy(g["gdig::" + k])
        got = torch.from_numpy(digest(p.grad.cpu()))
        scale = max(want[2:].abs().max().item(), fl)
        e = max(abs(got[1] - want[1]).item() / max(want[1].item(), fl), (got[2:] - want[2:]).abs().max().item() / scale)
        worst = max(worst, e)
        assert e <= 1e-2, (k, e)
    for k in ("block_1.conv1_a.weight", "ds.conv3.weight", "block_5.trans.attention_output.attention.in_proj_weight"):
        e = rel(named[k].grad, torch.from_numpy(g["grad::" + k]))
        worst = max(worst, e)
        assert e <= 1e-2, (k, e)
    print(f"[FCT G8] worst gradient error vs the reference's digests: {worst:.2e}")
    # the survey's own record of the reference: default init under manual_seed(0), rand(1,3,64,64) -> min 0.4851 / max 0.4964
    gs = gold("g8s_fct_default_init.npz")
    torch.manual_seed(0)
    m0 = P().FCT()
    x0 = torch.rand(1, 3, 64, 64)
    with torch.no_grad():
        o0 = m0.cuda().eval()(x0.cuda())
    assert abs(o0.min().item() - float(gs["out_min"])) < 1e-5 and abs(o0.max().item() - float(gs["out_max"])) < 1e-5


def test_frame_folded_clip_at_224_matches_the_oracle():
    """[B*T = 4, 3, 224, 224]: 12 544 pixel tokens per frame in block_1 / block_9 (the long-sequence attention path)."""
    torch.manual_seed(3)
    ref = F.FCT().eval()
    ref.load_state_dict(det_state_dict(ref))
    m = P().FCT()
    m.load_state_dict(ref.state_dict())
    m = m.cuda().eval()
    x = torch.rand(4, 3, 224, 224)
    y_true = (torch.rand(4, 1, 224, 224) > 0.5).float()
    want = ref(x)
    F.DiceLoss()(want, y_true).backward()
    got = m(x.cuda())
    P().DiceLoss()(got, y_true.cuda()).backward()
    e = rel(got, want)
    print(f"\n[FCT 4x224x224] forward max-rel error vs the CPU oracle: {e:.2e}")
    assert e <= 1e-3
    refg = {k: p.grad for k, p in ref.named_parameters() if p.grad is not None}
    fl = grad_floor(v.numpy() for v in refg.values())
    errs = {k: rel(p.grad, refg[k], fl) for k, p in m.named_parameters() if p.grad is not None}
    assert set(errs) == set(refg)
    worst = max(errs, key=errs.get)
    print(f"[FCT 4x224x224] worst gradient error vs the CPU oracle: {errs[worst]:.2e} ({worst})")
    assert errs[worst] <= 1e-2


def test_train_mode_dropout_and_a_training_step():
    """Train mode: the dropouts of FCT.py:115,146,175 are active (outputs differ run to run and from eval), gradients reach all the
    used parameters, and a few AdamW steps on one batch lower the Dice loss."""
    torch.manual_seed(5)
    m = P().FCT().cuda()
    x = torch.rand(2, 3, 64, 64, device="cuda")
    y_true = (torch.rand(2, 1, 64, 64, device="cuda") > 0.5).float()
    with torch.no_grad():
        e0 = m.eval()(x)
        t1, t2 = m.train()(x), m.train()(x)
    assert not torch.equal(t1, t2) and not torch.equal(t1, e0) and torch.equal(m.eval()(x), e0)
    crit = P().DiceLoss()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)                 # the reference's optimiser (FCT.py:312)
    m.train()
    losses = []
    for _ in range(12):
        opt.zero_grad()
        loss = crit(m(x), y_true)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert sum(p.grad is None for p in m.parameters()) == 14 and all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    assert sum(losses[-3:]) / 3 < sum(losses[:3]) / 3, losses


def test_contract_and_loud_failures():
    m = P().FCT()                                                   # zero-argument constructor (FCT.py:302)
    ref = F.FCT()
    assert list(m.state_dict().keys()) == list(ref.state_dict().keys()) and sum(p.numel() for p in m.parameters()) == 2_094_789
    ref.load_state_dict(m.state_dict()); m.load_state_dict(ref.state_dict())           # checkpoints interchange
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.eval()(torch.rand(1, 3, 64, 64))
    m = m.cuda()
    with pytest.raises(RuntimeError, match="multiple of 32"):
        m.eval()(torch.rand(1, 3, 112, 112, device="cuda"))        # the reference fails at FCT.py:181 for this size too
    with pytest.raises(ValueError):
        m.eval()(torch.rand(3, 64, 64, device="cuda"))
    with pytest.raises(AssertionError):
        P().DiceLoss()(torch.rand(1, 1, 8, 8, device="cuda"), torch.rand(1, 1, 8, 4, device="cuda"))     # Metrics.py:15
    # the inner classes take the reference's constructor arguments (FCT.py:25,86,137,168); the ones the reference never uses are ignored
    # like there, a conv geometry the HIP path does not implement is refused instead of computed differently
    pkg = fct()
    import inspect
    for cls in ("Attention", "Transformer", "Block_encoder_bottleneck", "Block_decoder", "Wide_Focus", "DS_out"):
        assert list(inspect.signature(getattr(pkg, cls).__init__).parameters) == list(inspect.signature(getattr(F, cls).__init__).parameters), cls
    pkg.Block_encoder_bottleneck("first", 3, 8, 2, 0.0); pkg.Block_decoder(16, 8, 2, 0.1)
    pkg.Transformer(8, 8, 2, dpr=0.2, proj_drop=0.5, padding_kv="valid"); pkg.Attention(8, 2, 0.5, 3, 1, 1, "valid", "same", False)
    with pytest.raises(TypeError):
        pkg.Block_encoder_bottleneck("first", 3, 8, 2)                 # dpr is a required positional in the reference too
    for kw in (dict(kernel_size=5), dict(stride_kv=2), dict(stride_q=2), dict(padding_q="valid")):
        with pytest.raises(NotImplementedError):
            pkg.Attention(8, 2, **kw)
