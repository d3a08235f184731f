"""Full-size GPU tests at BASELINE.json's configs 2, 4 and 5.

1. ORACLE PARITY AT FULL SIZE.  The CPU oracle costs ~0.3 s per clip-step on the GPU box's host cores, so a whole
   config-2 step is a few seconds: the HIP path is compared with oracle/hybrid_ref.py on the same seeded weights and clips
   at [8,16,3,224,224] (config 2), [8,64,3,224,224] d=768 hid=3072 (config 4 at the bench's own batch: 512 frames, ~40 s of host work) and
   [4,16,3,448,448] (config 5), in train mode (BatchNorm batch statistics; attention dropout off so the run is deterministic,
   SURVEY.md section 0.3 decision 4): logits, loss, updated running statistics and EVERY parameter gradient.
     * fp32 mode is the gate: logits and loss within 1e-3 (north_star's tolerance; measured 6e-7); every gradient outside the conv
       stack within 1e-3 of the fp32 oracle; the conv-stack gradients (3-6 M summed products per element behind max-pool / ReLU
       routing, where the fp32 oracle is itself 1e-3..9e-3 away from its own fp64 run) within 5e-3 of the FP64 oracle;
     * bf16 mode is REPORTED against the fp32 oracle and against the bf16-rounded oracle (oracle/hybrid_ref_bf16.py) and gated
       at 1.3 x the distances measured for this build (BF16_MEASURED below: logits 6.5e-3 .. 1.06e-2, gradients 5.1e-2 .. 6.3e-2 relative L2).  End to end the rounded oracle
       cannot be tight -- rounding amplifies any summation-order difference to bf16 noise within a few stages (its docstring) --
       so the TIGHT bf16 gates (2e-3 forward, 2e-2 gradients) live in the per-stage tests of tests/test_gpu_parity.py.
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
    """Three host runs of one config, computed once: the fp32 oracle (the reference's CPU path restated), the same oracle in fp64
    (the arbiter: at 6.4 M positions per conv-gradient element fp32 itself is only good to ~1e-3..1e-2 -- measured: the fp32 oracle's
    enc1conv1.weight gradient is 8.9e-3 away from the fp64 one at config 2, the HIP fp32 path 2.2e-3) and the bf16-rounded oracle
    with fp64 accumulation."""
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

    def run(model, fwd, xin):
        model.zero_grad()
        logits = fwd(xin)
        loss = R.loss_fn(logits, y)
        loss.backward()
        return dict(logits=logits.detach().double(), loss=loss.item(), grads={n: p.grad.double() for n, p in model.named_parameters()})
    out["fp32"] = run(ref, ref, x)
    out["fp32"]["running"] = {k: v.clone() for k, v in ref.state_dict().items() if "running_" in k}
    ref64 = R.TransformerCNNHybridRef(**cfg["kw"]).double()
    ref64.load_state_dict(state0)
    for a in ref64.encoder.attention_layers:
        a.dropoutLayer.p = 0.0
    ref64.train()
    out["fp64"] = run(ref64, ref64, x.double())
    out["bf16r"] = run(ref64, lambda t: RB.forward(ref64, t), x.double())     # reads ref64's parameters, leaves its buffers alone
    del ref, ref64
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
    return logits.double().cpu(), loss.item(), {k: v.double().cpu() for k, v in grads.items()}, running


def _maxrel(got, want, floor=0.0):
    return (got - want).abs().max().item() / max(want.abs().max().item(), floor, 1e-30)


def _l2rel(got, want, floor=0.0):
    return (got - want).norm().item() / max(want.norm().item(), floor * math.sqrt(want.numel()), 1e-30)


ORACLE_CFGS = [("config2", CFG2), ("config4_B8_T64_d768", CFG4), ("config5_448", CFG5)]


@pytest.mark.parametrize("name,cfg", ORACLE_CFGS, ids=[n for n, _ in ORACLE_CFGS])
def test_fullsize_fp32_mode_matches_the_oracle(name, cfg):
    """The north_star gate at the benchmark's own sizes.  Logits and loss: within 1e-3 of the fp32 oracle (measured ~1e-6).
    Parameter gradients: the conv-stack gradients sum 3-6 M products per element through max-pool/ReLU routing, and there two
    correct fp32 implementations differ by more than 1e-3 (the fp32 oracle is 1e-3..9e-3 from its own fp64 run; the HIP fp32 path
    1e-3..4e-3), so those are gated against the fp64 oracle at 5e-3; everything else at 1e-3 against the fp32 oracle."""
    orc = _oracle(name, cfg)
    f32, f64 = orc["fp32"], orc["fp64"]
    logits, loss, grads, running = _hip_step(cfg, "fp32", orc)
    e_log = _maxrel(logits, f32["logits"])
    G = max(g.abs().max().item() for g in f64["grads"].values())
    e_hip = {n: _maxrel(grads[n], f64["grads"][n], floor=1e-4 * G) for n in grads}
    e_ref = {n: _maxrel(f32["grads"][n], f64["grads"][n], floor=1e-4 * G) for n in grads}
    e_two = {n: _maxrel(grads[n], f32["grads"][n], floor=1e-4 * G) for n in grads}
    worst = max(e_hip, key=e_hip.get)
    print(f"\n[{name} fp32] logits max-rel {e_log:.2e} vs fp32 oracle ({_maxrel(logits, f64['logits']):.2e} vs fp64); loss {loss:.6f} vs {f32['loss']:.6f}; "
          f"worst gradient vs fp64 oracle: HIP {e_hip[worst]:.2e} ({worst}; the fp32 oracle itself: {e_ref[worst]:.2e}); "
          f"worst HIP-vs-fp32-oracle {max(e_two.values()):.2e}; gradients beyond 1e-3 of fp64: HIP {sum(v > 1e-3 for v in e_hip.values())}, "
          f"fp32 oracle {sum(v > 1e-3 for v in e_ref.values())} of {len(grads)}")
    assert e_log <= 1e-3
    assert abs(loss - f32["loss"]) <= 1e-3 * max(1.0, abs(f32["loss"]))
    for n in grads:
        if n.startswith("encoder") and not n.startswith("encoder."):          # conv stack (encoder1..4)
            assert e_hip[n] <= 5e-3, (n, e_hip[n], e_ref[n])
        else:
            assert e_two[n] <= 1e-3 and e_hip[n] <= 1e-3, (n, e_two[n], e_hip[n])
    for k, v in f32["running"].items():
        assert _maxrel(running[k], v) <= 1e-4, k


@pytest.mark.parametrize("name,cfg", ORACLE_CFGS, ids=[n for n, _ in ORACLE_CFGS])
def test_fullsize_bf16x3_mode_meets_the_north_star_tolerance(name, cfg):
    """compute_dtype="bf16x3" -- fp32 storage, every contraction product from three bf16 MFMAs on two-term splits (hyb_common.h) -- is
    the fast mode that still meets north_star's 1e-3: logits and loss within 1e-3 of the fp32 oracle (measured ~1e-5), gradients
    outside the conv stack within 1.7e-2 of the fp32 oracle (max-abs over the tensor's max; measured 2.8e-3 / 4.7e-3 / 1.27e-2 at configs
    2 / 4 / 5, the worst being the tiny key-projection gradients, where the softmax backward's dP - delta cancels; the FFN / projection
    weight gradients also sit behind ReLU decisions, and a pre-activation within 1e-5 of zero switches a whole hidden unit's
    contribution), conv-stack gradients within 1.2e-2 of the fp64 oracle (measured 8.1e-3 .. 8.8e-3) (ReLU and arg-max routing; the exact-fp32 mode is gated at 5e-3 there and the fp32 CPU oracle itself is up to
    9e-3 from fp64)."""
    orc = _oracle(name, cfg)
    f32, f64 = orc["fp32"], orc["fp64"]
    logits, loss, grads, running = _hip_step(cfg, "bf16x3", orc)
    e_log = _maxrel(logits, f32["logits"])
    G = max(g.abs().max().item() for g in f64["grads"].values())
    e_hip = {n: _maxrel(grads[n], f64["grads"][n], floor=1e-4 * G) for n in grads}
    e_two = {n: _maxrel(grads[n], f32["grads"][n], floor=1e-4 * G) for n in grads}
    conv = [n for n in grads if n.startswith("encoder") and not n.startswith("encoder.")]
    rest = [n for n in grads if n not in conv]
    wc, wr = max(conv, key=lambda n: e_hip[n]), max(rest, key=lambda n: e_two[n])
    print(f"\n[{name} bf16x3] logits max-rel {e_log:.2e} vs fp32 oracle; loss {loss:.6f} vs {f32['loss']:.6f}; worst conv-stack gradient vs fp64 "
          f"{e_hip[wc]:.2e} ({wc}); worst other gradient vs fp32 oracle {e_two[wr]:.2e} ({wr})")
    assert e_log <= 1e-3
    assert abs(loss - f32["loss"]) <= 1e-3 * max(1.0, abs(f32["loss"]))
    for n in conv:
        assert e_hip[n] <= 1.2e-2, (n, e_hip[n])          # measured 8.1e-3 / 8.7e-3 / 8.8e-3 (configs 2 / 4 / 5)
    for n in rest:
        assert e_two[n] <= 1.7e-2, (n, e_two[n])          # measured 2.8e-3 / 4.7e-3 / 1.27e-2
    for k, v in f32["running"].items():
        assert _maxrel(running[k], v) <= 1e-4, k


@pytest.mark.parametrize("name,cfg", ORACLE_CFGS, ids=[n for n, _ in ORACLE_CFGS])
def test_fullsize_mixed_mode_meets_the_north_star_tolerance(name, cfg):
    """compute_dtype="mixed": the conv stages in bf16 (the benchmarked kernels), token projection + temporal encoder + head in bf16x3.  The bf16
    mode's logits error is the temporal half's (scripts/exp/logit_error_split.py), so this mode's logits and loss are within north_star's
    1e-3 of the fp32 oracle at nearly the bf16 mode's speed.  Gradients: the conv stack is the bf16 mode's (gated like it, L2-rel against the
    bf16-rounded oracle at MIXED_CONV x the bf16 mode's own measurement); the temporal half's gradients are exact-arithmetic gradients of a forward
    whose INPUT (the pooled map) carries bf16 rounding: L2-rel against the fp32 oracle, gated at MIXED_REST."""
    orc = _oracle(name, cfg)
    f32, r16 = orc["fp32"], orc["bf16r"]
    logits, loss, grads, running = _hip_step(cfg, "mixed", orc)
    e_log = _maxrel(logits, f32["logits"])
    G = max(g.abs().max().item() for g in f32["grads"].values())
    conv = [n for n in grads if n.startswith("encoder") and not n.startswith("encoder.")]
    rest = [n for n in grads if n not in conv]
    e_conv = {n: _l2rel(grads[n], r16["grads"][n], floor=1e-4 * G) for n in conv}
    e_rest = {n: _l2rel(grads[n], f32["grads"][n], floor=1e-4 * G) for n in rest}
    wc, wr = max(e_conv, key=e_conv.get), max(e_rest, key=e_rest.get)
    print(f"\n[{name} mixed] logits max-rel {e_log:.2e} vs fp32 oracle; loss {loss:.6f} vs {f32['loss']:.6f}; worst conv-stack gradient L2-rel vs the "
          f"bf16-rounded oracle {e_conv[wc]:.2e} ({wc}); worst temporal-half gradient L2-rel vs the fp32 oracle {e_rest[wr]:.2e} ({wr})")
    assert e_log <= 1e-3
    assert abs(loss - f32["loss"]) <= 1e-3 * max(1.0, abs(f32["loss"]))
    for n in conv:
        assert e_conv[n] <= MIXED_CONV * BF16_MEASURED[name][2], (n, e_conv[n])       # measured 6.6e-2 / 6.8e-2 (configs 2 / 4)
    for n in rest:
        assert e_rest[n] <= MIXED_REST, (n, e_rest[n])                                # measured 6.2e-3 / 3.3e-3
    for k, v in f32["running"].items():
        assert _maxrel(running[k], v) <= 2e-3, k          # BatchNorm statistics of bf16 conv stages (fp32 accumulators, bf16 inputs)


# (the rounded oracle also rounds the temporal half, which this mode does not: its conv-stack distance is a little above the bf16 mode's own)
MIXED_CONV, MIXED_REST = 1.6, 1.5e-2


@pytest.mark.parametrize("name,cfg", ORACLE_CFGS, ids=[n for n, _ in ORACLE_CFGS])
def test_fullsize_bf16_mode_against_both_oracles(name, cfg):
    """bf16 (the benchmarked mode): reported against the fp32 oracle and the bf16-rounded oracle; loose end-to-end gates (module docstring)."""
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
    print("   per-gradient L2-rel vs rounded oracle: " + ", ".join(f"{n.replace('encoder', 'e').replace('.weight', '.w').replace('.bias', '.b')} {v:.1e}"
                                                               for n, v in rep16.items() if v > 5e-3))
    # gates = 1.3 x what this build measures at this config (the runs are bit-reproducible): (logits vs fp32 oracle, logits vs rounded
    # oracle, worst gradient L2-rel vs rounded oracle).  bf16 cannot meet north_star's 1e-3 -- compute_dtype="bf16x3" does (test above).
    m32, m16, mg = BF16_MEASURED[name]
    assert e32 <= 1.3 * m32, e32
    assert e16 <= 1.3 * m16, e16
    assert abs(loss - r16["loss"]) <= 5e-3 * max(1.0, abs(r16["loss"]))
    for n, e in rep16.items():
        assert e <= 1.3 * mg, (n, e)


# round 3, measured on MI355X (gpurun_out/r3_fullsize2.log; profiles/r03_bf16_error_table.txt)
BF16_MEASURED = {"config2": (9.75e-3, 8.32e-3, 5.45e-2), "config4_B8_T64_d768": (1.06e-2, 6.42e-3, 5.07e-2), "config5_448": (6.53e-3, 4.48e-3, 6.25e-2)}
