"""Full-size GPU tests at BASELINE.json's configs 2, 4 and 5.

1. ORACLE PARITY AT FULL SIZE.  The CPU oracle costs ~0.3 s per clip-step on the GPU box's host cores, so a whole
   config-2 step is a few seconds: the HIP path is compared with oracle/hybrid_ref.py on the same seeded weights and clips
   at [8,16,3,224,224] (config 2), [4,64,3,224,224] d=768 hid=3072 (config 4; BASELINE leaves B open, the bench uses 8) and
   [4,16,3,448,448] (config 5), in train mode (BatchNorm batch statistics; attention dropout off so the run is deterministic,
   SURVEY.md section 0.3 decision 4): logits, loss, updated running statistics and EVERY parameter gradient.
     * fp32 mode is the gate: max|got-want| / max|want| <= 1e-3 (north_star's logits tolerance, applied to the gradients too);
     * bf16 mode is (a) REPORTED against the fp32 oracle (printed; bf16's unit round-off 3.9e-3 rules out 1e-3) and
       (b) GATED against the bf16-rounded oracle (oracle/hybrid_ref_bf16.py: same algorithm, rounded to bf16 at the points where
       the kernels store or feed bf16), where summation order is all that is left: relative L2 <= 2e-2 on every gradient.
2. SIZE-INDEPENDENT PROPERTIES (kept from round 1): bit-identical repeat runs, clip independence (what makes batch-of-clips
   data parallelism exact), frame-order invariance (no positional encoding, quirk Q8), the closed-form head-bias gradient and
   batch-split gradient linearity (what the gradient all-reduce relies on).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

CFG2 = dict(B=8, T=16, H=224, kw=dict())                                            # headline config
CFG4 = dict(B=8, T=64, H=224, kw=dict(d_model=768, num_heads=8, hidden_dim=3072))   # long clip: T=64, d=768 (d_head 96)
CFG5 = dict(B=4, T=16, H=448, kw=dict())                                            # high-res frames
CFG4_ORACLE = dict(CFG4, B=4)                                                       # oracle parity: 256 frames of CPU work instead of 512


def P():
    import transformer_cnn_hybrid_network_for_video_processing_amd as pkg
    return pkg


def make(cfg, mode="bf16", seed=0):
    torch.manual_seed(seed)
    m = P().TransformerCNNHybrid(compute_dtype=mode, **cfg["kw"]).cuda()
    for a in m.encoder.attention_layers:
        a.dropoutLayer.p = 0.0
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.rand(cfg["B"], cfg["T"], 3, cfg["H"], cfg["H"], generator=g).cuda()
    y = torch.randint(0, 8, (cfg["B"],), generator=g).cuda()
    return m, x, y


def fwd_bwd(m, x, y):
    m.zero_grad(set_to_none=True)
    logits = m(x)
    loss = P().HybridCrossEntropyLoss()(logits, y)
    loss.backward()
    return logits.detach().clone(), loss.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}


@pytest.mark.parametrize("cfg", [CFG2, CFG4, CFG5], ids=["config2", "config4_T64_d768", "config5_448"])
def test_fullsize_runs_finite_and_deterministic(cfg):
    m, x, y = make(cfg)
    m.train()
    rm0 = m.encoder1.enc1norm1.running_mean.clone()
    l1, loss1, g1 = fwd_bwd(m, x, y)
    l2, loss2, g2 = fwd_bwd(m, x, y)
    assert torch.isfinite(l1).all() and torch.isfinite(loss1)
    assert all(torch.isfinite(v).all() for v in g1.values())
    assert not torch.equal(m.encoder1.enc1norm1.running_mean, rm0)          # train-mode BN updated its running statistics
    assert torch.equal(l1, l2) and torch.equal(loss1, loss2)                # forward is bit-reproducible
    for n in g1:                                                            # every reduction is in a fixed order: no float atomics
        assert torch.equal(g1[n], g2[n]), n


@pytest.mark.parametrize("cfg", [CFG2, CFG4], ids=["config2", "config4_T64_d768"])
def test_clip_independence_and_frame_order_invariance(cfg):
    m, x, y = make(cfg)
    m.eval()
    with torch.no_grad():
        full = m(x)
        alone = torch.cat([m(x[i:i + 1]) for i in range(min(3, cfg["B"]))])
        perm = torch.randperm(cfg["T"], generator=torch.Generator().manual_seed(3)).cuda()
        shuffled = m(x[:, perm])
    scale = full.abs().max().item()
    assert (full[:alone.shape[0]] - alone).abs().max().item() <= 2e-3 * scale       # bf16 activations; same math per clip
    assert (full - shuffled).abs().max().item() <= 2e-2 * scale                      # summation order over T changes rounding


def test_head_bias_gradient_closed_form_and_batch_split_linearity():
    m, x, y = make(CFG2)
    m.eval()                                       # BN with running stats: the loss is a plain mean over clips
    logits, loss, g = fwd_bwd(m, x, y)
    want = (torch.softmax(logits, dim=1) - torch.nn.functional.one_hot(y, 8).float()).mean(0)
    assert torch.allclose(g["head.bias"], want, rtol=1e-4, atol=1e-6)
    _, _, ga = fwd_bwd(m, x[:4], y[:4])
    _, _, gb = fwd_bwd(m, x[4:], y[4:])
    gmax = max(v.abs().max().item() for v in g.values())
    for n in g:
        avg = 0.5 * (ga[n] + gb[n])
        err = (avg - g[n]).abs().max().item()
        assert err <= 3e-2 * max(g[n].abs().max().item(), 1e-3 * gmax), (n, err)


def test_fp32_mode_fullsize_matches_bf16_mode_loosely():
    """The exact-fp32 path at config-2 size: finite, and the bf16 path stays within bf16 rounding of it."""
    m, x, y = make(dict(B=2, T=16, H=224, kw={}), mode="fp32")
    m.eval()
    with torch.no_grad():
        l32 = m(x)
        m.set_compute_dtype("bf16")
        l16 = m(x)
    assert torch.isfinite(l32).all()
    assert (l32 - l16).abs().max().item() <= 3e-2 * l32.abs().max().item()


# ------------------------------------------------------------------------------------------------------------------------
# full-size oracle parity
# ------------------------------------------------------------------------------------------------------------------------
_ORACLE_CACHE = {}


def _oracle(name, cfg):
    """fp32 oracle and bf16-rounded oracle results for one config (computed once, on the host cores)."""
    if name in _ORACLE_CACHE:
        return _ORACLE_CACHE[name]
    from oracle import hybrid_ref as R
    from oracle import hybrid_ref_bf16 as RB
    torch.manual_seed(0)
    ref = R.TransformerCNNHybridRef(**cfg["kw"])
    for a in ref.encoder.attention_layers:
        a.dropoutLayer.p = 0.0
    ref.train()
    state0 = {k: v.clone() for k, v in ref.state_dict().items()}
    x, y = R.synthetic_batch(cfg["B"], cfg["T"], cfg["H"], cfg["H"], seed=0)
    out = {"state0": state0, "x": x, "y": y}
    logits = ref(x)
    loss = R.loss_fn(logits, y)
    loss.backward()
    out["fp32"] = dict(logits=logits.detach().clone(), loss=loss.item(), grads={n: p.grad.clone() for n, p in ref.named_parameters()},
                       running={k: v.clone() for k, v in ref.state_dict().items() if "running_" in k})
    ref.zero_grad()
    logits = RB.forward(ref, x)                                 # reads ref's parameters; does not touch the running statistics
    loss = R.loss_fn(logits, y)
    loss.backward()
    out["bf16r"] = dict(logits=logits.detach().clone(), loss=loss.item(), grads={n: p.grad.clone() for n, p in ref.named_parameters()})
    del ref
    _ORACLE_CACHE[name] = out
    return out


def _hip_step(cfg, mode, orc):
    m = P().TransformerCNNHybrid(compute_dtype=mode, **cfg["kw"])
    m.load_state_dict(orc["state0"])
    for a in m.encoder.attention_layers:
        a.dropoutLayer.p = 0.0
    m = m.cuda().train()
    logits, loss, grads = fwd_bwd(m, orc["x"].cuda(), orc["y"].cuda())
    running = {k: v.detach().cpu() for k, v in m.state_dict().items() if "running_" in k}
    return logits.cpu(), loss.item(), {k: v.cpu() for k, v in grads.items()}, running


def _maxrel(got, want, floor=0.0):
    return (got - want).abs().max().item() / max(want.abs().max().item(), floor, 1e-30)


def _l2rel(got, want, floor=0.0):
    return (got - want).norm().item() / max(want.norm().item(), floor * math.sqrt(want.numel()), 1e-30)


ORACLE_CFGS = [("config2", CFG2), ("config4_B4_T64_d768", CFG4_ORACLE), ("config5_448", CFG5)]


@pytest.mark.parametrize("name,cfg", ORACLE_CFGS, ids=[n for n, _ in ORACLE_CFGS])
def test_fullsize_fp32_mode_matches_the_oracle(name, cfg):
    """The north_star gate at the benchmark's own sizes: logits, loss, running statistics and every parameter gradient within 1e-3."""
    orc = _oracle(name, cfg)
    want = orc["fp32"]
    logits, loss, grads, running = _hip_step(cfg, "fp32", orc)
    e_log = _maxrel(logits, want["logits"])
    G = max(g.abs().max().item() for g in want["grads"].values())
    errs = {n: _maxrel(grads[n], want["grads"][n], floor=1e-4 * G) for n in want["grads"]}
    worst = max(errs, key=errs.get)
    print(f"\n[{name} fp32] logits max-rel {e_log:.2e}; loss {loss:.6f} vs {want['loss']:.6f}; worst grad {worst} {errs[worst]:.2e}")
    assert e_log <= 1e-3
    assert abs(loss - want["loss"]) <= 1e-3 * max(1.0, abs(want["loss"]))
    for n, e in errs.items():
        assert e <= 1e-3, (n, e)
    for k, v in want["running"].items():
        assert _maxrel(running[k], v) <= 1e-4, k


@pytest.mark.parametrize("name,cfg", ORACLE_CFGS, ids=[n for n, _ in ORACLE_CFGS])
def test_fullsize_bf16_mode_against_both_oracles(name, cfg):
    """bf16 (the benchmarked mode): reported against the fp32 oracle, gated against the bf16-rounded oracle."""
    orc = _oracle(name, cfg)
    logits, loss, grads, _ = _hip_step(cfg, "bf16", orc)
    f32, r16 = orc["fp32"], orc["bf16r"]
    G = max(g.abs().max().item() for g in f32["grads"].values())
    rep32 = {n: _l2rel(grads[n], f32["grads"][n], floor=1e-4 * G) for n in grads}
    rep16 = {n: _l2rel(grads[n], r16["grads"][n], floor=1e-4 * G) for n in grads}
    w32, w16 = max(rep32, key=rep32.get), max(rep16, key=rep16.get)
    e32, e16 = _maxrel(logits, f32["logits"]), _maxrel(logits, r16["logits"])
    print(f"\n[{name} bf16] logits max-rel: {e32:.2e} vs fp32 oracle, {e16:.2e} vs bf16-rounded oracle; loss {loss:.6f} "
          f"(fp32 oracle {f32['loss']:.6f}, rounded {r16['loss']:.6f}); worst gradient L2-rel: {rep32[w32]:.2e} ({w32}) vs fp32 oracle, "
          f"{rep16[w16]:.2e} ({w16}) vs rounded oracle; the oracles differ from each other by "
          f"{_maxrel(r16['logits'], f32['logits']):.2e} on the logits")
    assert e32 <= 3e-2                                            # sanity bound only: bf16 cannot meet 1e-3 (reported above)
    assert e16 <= 5e-3, e16
    assert abs(loss - r16["loss"]) <= 5e-3 * max(1.0, abs(r16["loss"]))
    for n, e in rep16.items():
        assert e <= 2e-2, (n, e)
