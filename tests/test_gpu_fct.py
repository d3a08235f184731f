"""FCT on the HIP path: every forward operator against stock torch on the CPU, every block and the whole
model against the golden vectors captured from the REFERENCE's own classes (tests/golden/g3..g8), and a 224 x 224 batch against
the CPU oracle (oracle/fct_ref.py, itself pinned by the same goldens in tests/test_oracle_fct.py).
Tolerance: max|got - want| / max|want| <= 1e-3 (north_star's forward tolerance; fp32 path: measured ~1e-6)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)
from det_init import det_state_dict  # noqa: E402
from oracle import fct_ref as F  # noqa: E402
import transformer_cnn_hybrid_network_for_video_processing_amd  # noqa: E402,F401  (registers torch.ops.hybrid.*)


def P():
    import transformer_cnn_hybrid_network_for_video_processing_amd as pkg
    return pkg


def fct():
    from transformer_cnn_hybrid_network_for_video_processing_amd import fct as m
    return m


def rel(got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    return (got - want).abs().max().item() / max(want.abs().max().item(), 1e-12)


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
    got = torch.ops.hybrid.fct_conv(nhwc(x), conv.weight.detach().cuda(), conv.bias.detach().cuda(), dil, act)
    assert rel(nchw(got), want) <= 1e-5


@pytest.mark.parametrize("c,n,h,w", [(8, 2, 8, 8), (16, 1, 5, 9), (64, 1, 6, 6), (128, 2, 4, 4)])
def test_qkv_projection_and_layernorm(c, n, h, w):
    torch.manual_seed(c)
    att = F.Attention(c, 2)
    for ln in (att.layernorm_q, att.layernorm_k, att.layernorm_v):
        torch.nn.init.normal_(ln.weight, 1.0, 0.2); torch.nn.init.normal_(ln.bias, 0.0, 0.2)
    x = torch.randn(n, c, h, w)
    convs, lns = (att.conv_q, att.conv_k, att.conv_v), (att.layernorm_q, att.layernorm_k, att.layernorm_v)
    got = torch.ops.hybrid.fct_qkv_proj(nhwc(x), [m.weight.detach().cuda() for m in convs], [m.bias.detach().cuda() for m in convs],
                                        [m.weight.detach().cuda() for m in lns], [m.bias.detach().cuda() for m in lns], 1e-5)
    for g, cv, ln in zip(got, convs, lns):
        assert rel(nchw(g), att._project(x, cv, ln)) <= 1e-5
    ln = lns[0]
    y = torch.ops.hybrid.fct_ln(nhwc(x), ln.weight.detach().cuda(), ln.bias.detach().cuda(), 1e-5)
    assert rel(nchw(y), ln(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)) <= 1e-5


@pytest.mark.parametrize("n,l,c,heads", [(2, 64, 8, 2), (1, 100, 16, 2), (2, 1024, 8, 2), (1, 257, 64, 2), (1, 49, 128, 2), (3, 16, 32, 4), (1, 4096, 16, 2)])
def test_multihead_attention_over_pixel_tokens(n, l, c, heads):
    torch.manual_seed(l + c)
    mha = torch.nn.MultiheadAttention(c, heads, batch_first=True)
    torch.nn.init.normal_(mha.in_proj_bias, 0.0, 0.2); torch.nn.init.normal_(mha.out_proj.bias, 0.0, 0.2)
    q, k, v = (torch.randn(n, l, c) * 2.0 for _ in range(3))
    want = mha(q, k, v, need_weights=False)[0]
    got = torch.ops.hybrid.fct_mha(q.cuda(), k.cuda(), v.cuda(), mha.in_proj_weight.detach().cuda(), mha.in_proj_bias.detach().cuda(),
                                   mha.out_proj.weight.detach().cuda(), mha.out_proj.bias.detach().cuda(), heads)
    assert rel(got, want) <= 2e-5


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
    ("g3_fct_block_first.npz", lambda: fct().Block_encoder_bottleneck("first", 3, 8, 2), ()),
    ("g3b_fct_block_second.npz", lambda: fct().Block_encoder_bottleneck("second", 8, 16, 2), ("scale_img",)),
    ("g4_fct_attention.npz", lambda: fct().Attention(8, 2), ()),
    ("g4b_fct_transformer.npz", lambda: fct().Transformer(8, 8, 2), ()),
    ("g5_fct_wide_focus.npz", lambda: fct().Wide_Focus(8, 8), ()),
    ("g6_fct_block_decoder.npz", lambda: fct().Block_decoder(16, 8, 2), ("skip",)),
    ("g6b_fct_ds_out.npz", lambda: fct().DS_out(8, 1), ()),
])
def test_reference_block_goldens_forward(name, ctor, extra):
    g = gold(name)
    m = ctor()
    m.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd::")})
    m = m.cuda().eval()
    with torch.no_grad():
        y = m(nhwc(torch.from_numpy(g["x"])), *[nhwc(torch.from_numpy(g[k])) for k in extra])
    assert rel(nchw(y), torch.from_numpy(g["out"])) <= 1e-3          # the gate; measured ~1e-6


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
    with torch.no_grad():
        want = ref(x)
        got = m(x.cuda())
    e = rel(got, want)
    print(f"\n[FCT 4x224x224] forward max-rel error vs the CPU oracle: {e:.2e}")
    assert e <= 1e-3


def test_contract_and_loud_failures():
    m = P().FCT()                                                   # zero-argument constructor (FCT.py:302)
    ref = F.FCT()
    assert list(m.state_dict().keys()) == list(ref.state_dict().keys()) and sum(p.numel() for p in m.parameters()) == 2_094_789
    ref.load_state_dict(m.state_dict()); m.load_state_dict(ref.state_dict())           # checkpoints interchange
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.eval()(torch.rand(1, 3, 64, 64))
    m = m.cuda()
    with pytest.raises(NotImplementedError, match="forward-only"):
        m.train()(torch.rand(1, 3, 64, 64, device="cuda"))
    with pytest.raises(RuntimeError, match="multiple of 32"):
        m.eval()(torch.rand(1, 3, 112, 112, device="cuda"))        # the reference fails at FCT.py:181 for this size too
    with pytest.raises(ValueError):
        m.eval()(torch.rand(3, 64, 64, device="cuda"))
    out = m.eval()(torch.rand(1, 3, 64, 64, device="cuda"))        # grad mode on: the forward works, a backward must not pass silently
    with pytest.raises(NotImplementedError, match="forward-only"):
        out.sum().backward()
