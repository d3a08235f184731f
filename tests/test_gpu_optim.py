"""Fused AdamW against torch.optim.AdamW (the reference harness's optimizer, Model.py:153): same update within
fp32 round-off over several steps, odd sizes and unaligned views included; state-dict interchange."""
import pytest
import torch

import transformer_cnn_hybrid_network_for_video_processing_amd as P

pytestmark = pytest.mark.gpu


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(512, 512), (2048, 512), (8,), (3,), (1000,), (4097,), (32, 3, 3, 3), (1,)]
    ps = [torch.randn(s, generator=g).cuda() for s in shapes]
    base = torch.randn(1030, generator=g).cuda()
    ps.append(base[1:1026])                               # 4-byte aligned only: exercises the scalar path
    return ps


@pytest.mark.parametrize("wd,lr", [(1e-2, 1e-3), (0.0, 3e-4), (0.1, 1e-2)])
def test_adamw_matches_torch(wd, lr):
    init = _params(0)
    a = [torch.nn.Parameter(t.clone()) for t in init]
    b = [torch.nn.Parameter(t.clone()) for t in init]
    oa = torch.optim.AdamW(a, lr=lr, weight_decay=wd)
    ob = P.HybridAdamW(b, lr=lr, weight_decay=wd)
    g = torch.Generator().manual_seed(1)
    for step in range(5):
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g).cuda() * (10.0 ** (step - 2))
            pa.grad = gr.clone(); pb.grad = gr.clone()
        oa.step(); ob.step()
        for pa, pb in zip(a, b):
            torch.testing.assert_close(pb.data, pa.data, rtol=2e-6, atol=2e-7)
            torch.testing.assert_close(ob.state[pb]["exp_avg"], oa.state[pa]["exp_avg"], rtol=2e-6, atol=1e-12)
            torch.testing.assert_close(ob.state[pb]["exp_avg_sq"], oa.state[pa]["exp_avg_sq"], rtol=2e-6, atol=1e-12)


def test_adamw_on_the_model_and_state_dict_round_trip():
    torch.manual_seed(0)
    kw = dict(cnn_channels=(32, 64), d_model=64, num_heads=4, num_layers=1, hidden_dim=128, dropout=0.0)
    m1, m2 = P.TransformerCNNHybrid(**kw).cuda().eval(), P.TransformerCNNHybrid(**kw).cuda().eval()   # eval: no dropout streams to align
    m2.load_state_dict(m1.state_dict())
    o1, o2 = torch.optim.AdamW(m1.parameters(), lr=1e-3), P.HybridAdamW(m2.parameters(), lr=1e-3)
    x = torch.rand(2, 4, 3, 32, 32, device="cuda"); y = torch.tensor([1, 3], device="cuda")
    for _ in range(2):
        for m, o in ((m1, o1), (m2, o2)):
            o.zero_grad(set_to_none=True)
            P.HybridCrossEntropyLoss()(m(x), y).backward()
            o.step()
    for (n, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        torch.testing.assert_close(p2, p1, rtol=1e-3, atol=2e-5, msg=n)        # Adam's m/sqrt(v) amplifies the 1e-7 LayerNorm-atomics noise where v ~ 0
    sd = o2.state_dict()
    o3 = P.HybridAdamW(m2.parameters(), lr=1e-3)
    o3.load_state_dict(sd)
    assert int(o3.state[next(iter(m2.parameters()))]["step"]) == 2


def test_adamw_reload_into_the_same_optimizer_midrun():
    """load_state_dict after a step replaces the moment tensors; the cached device-pointer tables must follow
    (they used to keep pointing at the old, freed moments).  step, save, step, roll back, step == torch doing the same."""
    import copy
    init = _params(3)
    a = [torch.nn.Parameter(t.clone()) for t in init]
    b = [torch.nn.Parameter(t.clone()) for t in init]
    oa, ob = torch.optim.AdamW(a, lr=1e-3), P.HybridAdamW(b, lr=1e-3)
    g = torch.Generator().manual_seed(5)

    def grads():
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g).cuda()
            pa.grad = gr.clone(); pb.grad = gr.clone()
    grads(); oa.step(); ob.step()
    sa, sb = copy.deepcopy(oa.state_dict()), copy.deepcopy(ob.state_dict())
    wa, wb = [p.detach().clone() for p in a], [p.detach().clone() for p in b]
    grads(); oa.step(); ob.step()
    oa.load_state_dict(sa); ob.load_state_dict(sb)                    # roll both back to after step 1
    with torch.no_grad():
        for p, w in zip(a, wa):
            p.copy_(w)
        for p, w in zip(b, wb):
            p.copy_(w)
    junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(8)]     # recycle freed blocks with poison
    del junk
    grads(); oa.step(); ob.step()
    for pa, pb in zip(a, b):
        torch.testing.assert_close(pb.data, pa.data, rtol=2e-6, atol=2e-7)
        torch.testing.assert_close(ob.state[pb]["exp_avg"], oa.state[pa]["exp_avg"], rtol=2e-6, atol=1e-12)
        assert int(ob.state[pb]["step"]) == 2


def test_adamw_device_step_counter_advanced_by_the_launch_itself():
    """hyb_adamw_step(advance=1): the step number is state['step'] + *counter, and the launch adds 1 to the counter after every workgroup
    has read it -- same parameters, bit for bit, as the host-stepped optimizer, and the counter counts the steps."""
    init = _params(3)
    a = [torch.nn.Parameter(t.clone()) for t in init]
    b = [torch.nn.Parameter(t.clone()) for t in init]
    oa = P.HybridAdamW(a, lr=1e-3)
    ob = P.HybridAdamW(b, lr=1e-3)
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    ob.set_step_counter(counter, advance=True)
    g = torch.Generator().manual_seed(4)
    for step in range(6):
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g).cuda()
            pa.grad = gr.clone(); pb.grad = gr.clone()
        oa.step(); ob.step()
        assert int(counter.item()) == step + 1
        for pa, pb in zip(a, b):
            # the device forms the bias corrections with its own pow(): equal to the host's to the last bit or one ulp of the step size
            torch.testing.assert_close(pb.data, pa.data, rtol=1e-6, atol=1e-7)
    ob.set_step_counter(None)
    two = P.HybridAdamW([{"params": b[:2]}, {"params": b[2:]}], lr=1e-3)
    two.set_step_counter(counter, advance=True)
    with pytest.raises(RuntimeError, match="one parameter group"):
        two.step()


def test_adamw_more_tensors_than_one_launch_holds_with_an_advancing_counter():
    """100 tensors = two launches of the 80-entry tensor table: both read the same step number, only the last one advances the counter."""
    g = torch.Generator().manual_seed(7)
    init = [torch.randn(int(n), generator=g).cuda() for n in torch.randint(1, 5000, (100,), generator=g)]
    a = [torch.nn.Parameter(t.clone()) for t in init]
    b = [torch.nn.Parameter(t.clone()) for t in init]
    oa = torch.optim.AdamW(a, lr=1e-3)
    ob = P.HybridAdamW(b, lr=1e-3)
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    ob.set_step_counter(counter, advance=True)
    for step in range(3):
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g).cuda()
            pa.grad = gr.clone(); pb.grad = gr.clone()
        oa.step(); ob.step()
        assert int(counter.item()) == step + 1
        for pa, pb in zip(a, b):
            torch.testing.assert_close(pb.data, pa.data, rtol=2e-6, atol=2e-7)


def test_two_advancing_optimizers_on_two_streams_do_not_share_a_ticket():
    """the advancing launch used to count its finished workgroups in ONE process-global device word, so two advancing launches in
    flight on one device (two models on two streams) could bump a counter early or never.  The ticket now belongs to the optimizer: two
    optimizers, each with its own counter, stepping concurrently on two streams, must each count exactly their own steps and end with the
    parameters of a host-stepped optimizer fed the same gradients."""
    g = torch.Generator().manual_seed(11)
    sizes = [300_000, 70_001, 4096, 1_000_003]                     # ~340 workgroups per launch: the two launches overlap on the device
    init = [torch.randn(n, generator=g).cuda() for n in sizes]
    K = 25
    grads = [[torch.randn(n, generator=g).cuda() for n in sizes] for _ in range(3)]
    ref = [torch.nn.Parameter(t.clone()) for t in init]
    oref = P.HybridAdamW(ref, lr=1e-3)
    for k in range(K):
        for p, gr in zip(ref, grads[k % 3]):
            p.grad = gr
        oref.step()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    models, opts, counters = [], [], []
    for _ in range(2):
        ps = [torch.nn.Parameter(t.clone()) for t in init]
        o = P.HybridAdamW(ps, lr=1e-3)
        c = torch.zeros(1, dtype=torch.int64, device="cuda")
        o.set_step_counter(c, advance=True)
        models.append(ps); opts.append(o); counters.append(c)
    assert opts[0]._ticket.data_ptr() != opts[1]._ticket.data_ptr()
    torch.cuda.synchronize()
    for k in range(K):
        for ps, o, st in zip(models, opts, streams):
            with torch.cuda.stream(st):
                for p, gr in zip(ps, grads[k % 3]):
                    p.grad = gr
                o.step()
    torch.cuda.synchronize()
    for ps, o, c in zip(models, opts, counters):
        assert int(c.item()) == K
        assert int(o._ticket.item()) == 0
        for p, r in zip(ps, ref):
            torch.testing.assert_close(p.data, r.data, rtol=1e-6, atol=1e-7)
