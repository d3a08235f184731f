"""CPU tests that pin the oracle (oracle/hybrid_ref.py).

* conv stage: replayed against golden vectors captured from the reference's own
  ``UNet._block`` / ``MaxPool2d`` (tests/golden/make_golden.py).
* MultiheadAttention / TransformerEncoder: the reference ships only CPython-3.8
  bytecode and no tests for them ("parity unpinned" by the reference), so they are
  pinned by hand-computed known answers (an independent numpy restatement written
  from SURVEY.md Appendix A) and a scaled_dot_product_attention cross-check.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import hybrid_ref as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return dict(np.load(os.path.join(GOLD, name), allow_pickle=False))


def test_g1_conv_stage_matches_reference_block():
    g = _load("g1_unet_block_stage.npz")
    st = R.conv_stage(3, 8, "enc1")
    with torch.no_grad():
        st.enc1conv1.weight.copy_(torch.from_numpy(g["conv_weight"]))
        st.enc1norm1.weight.copy_(torch.from_numpy(g["bn_weight"]))
        st.enc1norm1.bias.copy_(torch.from_numpy(g["bn_bias"]))
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    r = torch.from_numpy(g["r"])
    st.train()
    y = st(x)
    (y * r).sum().backward()
    assert torch.allclose(y, torch.from_numpy(g["train_out"]), atol=1e-6)
    assert torch.allclose(x.grad, torch.from_numpy(g["train_dx"]), atol=1e-5)
    assert torch.allclose(st.enc1conv1.weight.grad, torch.from_numpy(g["train_dw"]), atol=1e-4)
    assert torch.allclose(st.enc1norm1.weight.grad, torch.from_numpy(g["train_dgamma"]), atol=1e-4)
    assert torch.allclose(st.enc1norm1.bias.grad, torch.from_numpy(g["train_dbeta"]), atol=1e-4)
    assert torch.allclose(st.enc1norm1.running_mean, torch.from_numpy(g["running_mean1"]), atol=1e-7)
    assert torch.allclose(st.enc1norm1.running_var, torch.from_numpy(g["running_var1"]), atol=1e-7)
    assert int(st.enc1norm1.num_batches_tracked) == int(g["num_batches_tracked1"])
    x.grad = None
    st.zero_grad()
    st.eval()
    y = st(x)
    (y * r).sum().backward()
    assert torch.allclose(y, torch.from_numpy(g["eval_out"]), atol=1e-6)
    assert torch.allclose(x.grad, torch.from_numpy(g["eval_dx"]), atol=1e-5)
    assert torch.allclose(st.enc1conv1.weight.grad, torch.from_numpy(g["eval_dw"]), atol=1e-4)


def test_g2_two_stages_match_reference_unet_encoder():
    g = _load("g2_unet_two_stage.npz")
    m = R.TransformerCNNHybridRef(cnn_channels=(8, 16), d_model=8, num_heads=2, num_layers=1, hidden_dim=8)
    sd = {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd::")}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected                     # reference key names load unchanged
    x = torch.from_numpy(g["x"])
    m.eval()
    with torch.no_grad():
        h = m.encoder2(m.encoder1(x))
    assert torch.allclose(h, torch.from_numpy(g["eval_out"]), atol=1e-6)
    m.train()
    with torch.no_grad():
        h = m.encoder2(m.encoder1(x))
    assert torch.allclose(h, torch.from_numpy(g["train_out"]), atol=1e-5)


# ---------------------------------------------------------------------------
# independent numpy restatement of Appendix A (loops, no torch) for the KATs
# ---------------------------------------------------------------------------
def np_mha(x_q, x_k, x_v, W, b, H, mask=None):
    """W,b: dicts q,k,v,o.  x: [B,S,D].  Follows pyc src L67-89 literally."""
    B, S, D = x_q.shape
    dh = D // H
    relu = lambda a: np.maximum(a, 0.0)
    q = relu(x_q @ W["q"].T + b["q"])
    k = relu(x_k @ W["k"].T + b["k"])
    v = relu(x_v @ W["v"].T + b["v"])
    out = np.zeros((B, S, D))
    if mask is not None:
        mask_rep = np.tile(mask, (H, 1, 1))            # L78: mask.repeat(H,1,1)
    for bb in range(B):
        for h in range(H):
            qs = q[bb, :, h * dh:(h + 1) * dh]
            ks = k[bb, :, h * dh:(h + 1) * dh]
            vs = v[bb, :, h * dh:(h + 1) * dh]
            s = qs @ ks.T / math.sqrt(D)                # L51: sqrt(input_dim)
            if mask is not None:
                s = np.where(mask_rep[bb * H + h] == 0, -1e9, s)   # batch index b*H+h (L32-37)
            s = s - s.max(axis=-1, keepdims=True)
            p = np.exp(s)
            p /= p.sum(axis=-1, keepdims=True)
            out[bb, :, h * dh:(h + 1) * dh] = p @ vs
    return out @ W["o"].T + b["o"]


def np_ln(x, g, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def test_mha_hand_kat_identity_weights():
    """S=2, D=4, H=2, W=I, b=0: answer computed by hand.
    x = [[1,0,2,0],[0,1,0,2]] -> relu(x)=x.  head0 uses dims 0:2, head1 dims 2:4.
    head0: q=k=v=[[1,0],[0,1]], scores = I/sqrt(4) = [[.5,0],[0,.5]]
      softmax row0 = [e^.5, 1]/(e^.5+1) = [a, 1-a], a = 1/(1+e^-.5)
      out0 = [[a,1-a],[1-a,a]]
    head1: q=k=v=[[2,0],[0,2]], scores = [[2,0],[0,2]], c = 1/(1+e^-2)
      out1 = 2*[[c,1-c],[1-c,c]]"""
    m = R.MultiheadAttention(4, 2).eval()
    with torch.no_grad():
        for lin in (m.query_layer, m.key_layer, m.value_layer, m.output_layer):
            lin.weight.copy_(torch.eye(4))
            lin.bias.zero_()
    x = torch.tensor([[[1., 0., 2., 0.], [0., 1., 0., 2.]]])
    y = m(x, x, x)
    a = 1.0 / (1.0 + math.exp(-0.5))
    c = 1.0 / (1.0 + math.exp(-2.0))
    want = torch.tensor([[[a, 1 - a, 2 * c, 2 * (1 - c)], [1 - a, a, 2 * (1 - c), 2 * c]]])
    assert torch.allclose(y, want, atol=1e-6)


@pytest.mark.parametrize("B,S,D,H,use_mask", [(1, 2, 4, 2, False), (2, 5, 8, 2, False), (3, 4, 12, 3, True), (1, 7, 8, 4, True)])
def test_mha_matches_numpy_restatement(B, S, D, H, use_mask):
    torch.manual_seed(3)
    m = R.MultiheadAttention(D, H).double().eval()
    xq, xk, xv = (torch.randn(B, S, D, dtype=torch.float64) for _ in range(3))
    mask = None
    if use_mask:
        mask = (torch.rand(B, S, S) > 0.3).to(torch.float64)
        mask[:, :, 0] = 1          # keep at least one key per row
    W = {n: getattr(m, l).weight.detach().numpy() for n, l in
         (("q", "query_layer"), ("k", "key_layer"), ("v", "value_layer"), ("o", "output_layer"))}
    b = {n: getattr(m, l).bias.detach().numpy() for n, l in
         (("q", "query_layer"), ("k", "key_layer"), ("v", "value_layer"), ("o", "output_layer"))}
    want = np_mha(xq.numpy(), xk.numpy(), xv.numpy(), W, b, H, None if mask is None else mask.numpy())
    got = m(xq, xk, xv, mask).detach().numpy()
    assert np.allclose(got, want, atol=1e-10)


def test_attention_core_equals_sdpa_with_dmodel_scale():
    """With W=I, b=0 the core must equal SDPA(relu(x)) at scale 1/sqrt(D) (quirk Q1+Q2)."""
    torch.manual_seed(4)
    D, H, B, S = 16, 4, 2, 6
    m = R.MultiheadAttention(D, H).eval()
    with torch.no_grad():
        for lin in (m.query_layer, m.key_layer, m.value_layer, m.output_layer):
            lin.weight.copy_(torch.eye(D))
            lin.bias.zero_()
    x = torch.randn(B, S, D)
    xr = F.relu(x).reshape(B, S, H, D // H).transpose(1, 2)
    want = F.scaled_dot_product_attention(xr, xr, xr, scale=1.0 / math.sqrt(D)).transpose(1, 2).reshape(B, S, D)
    assert torch.allclose(m(x, x, x), want, atol=1e-6)


def test_encoder_matches_numpy_restatement_and_quirks():
    torch.manual_seed(5)
    D, Hd, L, H, B, S = 8, 16, 2, 2, 2, 3
    enc = R.TransformerEncoder(D, Hd, L, H, 0.0).double().eval()
    with torch.no_grad():
        for ln in enc.layer_norm:                       # non-trivial affine
            ln.weight.copy_(torch.randn(D).abs() + 0.5)
            ln.bias.copy_(torch.randn(D) * 0.1)
    x = torch.randn(B, S, D, dtype=torch.float64)
    got = enc(x, None).detach().numpy()
    h = x.numpy()
    for i in range(L):
        a = enc.attention_layers[i]
        W = {"q": a.query_layer.weight, "k": a.key_layer.weight, "v": a.value_layer.weight, "o": a.output_layer.weight}
        b = {"q": a.query_layer.bias, "k": a.key_layer.bias, "v": a.value_layer.bias, "o": a.output_layer.bias}
        W = {k: v.detach().numpy() for k, v in W.items()}
        b = {k: v.detach().numpy() for k, v in b.items()}
        g, be = enc.layer_norm[i].weight.detach().numpy(), enc.layer_norm[i].bias.detach().numpy()
        skip1 = h
        h = np_ln(np_mha(h, h, h, W, b, H), g, be) + skip1            # Q3
        skip2 = h
        f = enc.feedforward_layers[i]
        ff = np.maximum(h @ f[0].weight.detach().numpy().T + f[0].bias.detach().numpy(), 0) \
            @ f[2].weight.detach().numpy().T + f[2].bias.detach().numpy()
        h = (np_ln(ff, g, be) + skip2) * math.sqrt(0.5)               # Q7
    assert np.allclose(got, h, atol=1e-10)


def test_encoder_ctor_contract():
    with pytest.raises(ValueError, match="Input dimension must be divisible by number of heads"):
        R.TransformerEncoder(10, 16, 1, 3, 0.0)
    enc = R.TransformerEncoder(8, 16, 2, 2, 0.0)
    keys = set(enc.state_dict().keys())
    for i in range(2):
        for lin in ("query_layer", "key_layer", "value_layer", "output_layer"):
            assert f"attention_layers.{i}.{lin}.weight" in keys and f"attention_layers.{i}.{lin}.bias" in keys
        assert f"feedforward_layers.{i}.0.weight" in keys and f"feedforward_layers.{i}.2.bias" in keys
        assert f"layer_norm.{i}.weight" in keys
    # Q6: dropout inside forward is active even in eval()
    enc = R.TransformerEncoder(8, 16, 1, 2, 0.5).eval()
    x = torch.randn(1, 4, 8)
    assert not torch.equal(enc(x, None), enc(x, None))


def test_composite_config1_plumbing():
    """BASELINE config 1: [1,8,3,112,112] CPU forward, finite loss, param count of config 2."""
    torch.manual_seed(0)
    m = R.TransformerCNNHybridRef()
    assert sum(p.numel() for p in m.parameters()) == 6_827_304
    x, y = R.synthetic_batch(1, 8, 112, 112)
    logits = m(x)
    assert logits.shape == (1, 8)
    loss = R.loss_fn(logits, y)
    assert torch.isfinite(loss)
    loss.backward()
    assert all(p.grad is not None for p in m.parameters())
    m2 = R.TransformerCNNHybridRef()
    m2.load_state_dict(m.state_dict())
    m.eval(); m2.eval()
    with torch.no_grad():
        assert torch.equal(m(x), m2(x))
    assert m(x[:, 0]).shape == (1, 8)             # [B,3,H,W] => T=1
