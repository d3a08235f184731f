"""Bit-exact GPU tests of the MFMA contraction kernels on integer-valued data.

With small integer operands every product and partial sum is exactly representable (bf16 holds integers up to 256,
fp32 up to 2^24), so the HIP result must EQUAL a CPU fp32 reference computed with stock torch ops -- for both the bf16
(v_mfma_f32_16x16x32_bf16, ds_read_b64_tr_b16 paths) and the fp32 (v_mfma_f32_16x16x4_f32) builds of every kernel.
This pins fragment layouts, LDS swizzles, tile masking and the weight packing independently of rounding noise
(cdna_hip_programming.md section 3: "check the map with exact integer data", asymmetric operands)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {"fp32": (0, torch.float32), "bf16": (1, torch.bfloat16)}


def L():
    from transformer_cnn_hybrid_network_for_video_processing_amd._lib import lib
    return lib


def st():
    return torch.cuda.current_stream().cuda_stream


def pad32(c):
    return (c + 31) // 32 * 32


def sparse_int(shape, gen, lo, hi, p):
    v = torch.randint(lo, hi + 1, shape, generator=gen).float()
    return v * (torch.rand(shape, generator=gen) < p).float()


def to_nhwc(x, code, tdt, cp):
    N, C, H, W = x.shape
    out = torch.empty(N, H, W, cp, dtype=tdt, device="cuda")
    L().call("hyb_nchw_to_nhwc", code, x.cuda().contiguous().data_ptr(), out.data_ptr(), N, C, H, W, cp, st())
    return out


def to_nchw(x, code, C):
    N, H, W, cp = x.shape
    out = torch.empty(N, C, H, W, dtype=torch.float32, device="cuda")
    L().call("hyb_nhwc_to_nchw", code, x.data_ptr(), out.data_ptr(), N, C, H, W, cp, st())
    return out


@pytest.mark.parametrize("mode", ["bf16", "fp32"])
@pytest.mark.parametrize("ci,co,n,h,w", [(32, 32, 2, 16, 32), (32, 64, 2, 19, 37), (64, 128, 2, 16, 16), (128, 256, 1, 12, 20),
                                         (256, 128, 1, 9, 17), (96, 96, 1, 8, 16), (64, 32, 1, 33, 35), (40, 72, 1, 8, 8),
                                         (3, 32, 2, 16, 32), (3, 64, 1, 21, 45), (2, 32, 1, 8, 8), (1, 96, 1, 5, 5)])
def test_conv3x3_fwd_and_stats_exact(mode, ci, co, n, h, w):
    code, tdt = DT[mode]
    g = torch.Generator().manual_seed(ci * 1000 + co)
    x = sparse_int((n, ci, h, w), g, 0, 2, 0.5)
    wt = sparse_int((co, ci, 3, 3), g, -2, 2, 0.15 if ci > 3 else 0.6)
    want = F.conv2d(x, wt, padding=1)
    assert want.abs().max() <= 256
    first = ci <= 3
    cip, cop = (0 if first else pad32(ci)), pad32(co)
    wp = torch.empty(L().query("hyb_conv_packed_elems", int(first), cip, cop), dtype=tdt, device="cuda")
    wd = wt.cuda().contiguous()
    L().call("hyb_conv_pack_weight", code, 2 if first else 0, wd.data_ptr(), wp.data_ptr(), co, ci, cop, cip, st())
    xin = x.cuda().contiguous() if first else to_nhwc(x, code, tdt, cip)
    y = torch.full((n, h, w, cop), 7.0, dtype=tdt, device="cuda")
    stats = torch.zeros(2, cop, device="cuda")
    part = torch.empty(L().query("hyb_conv_stats_workspace", cop), dtype=torch.uint8, device="cuda")
    L().call("hyb_conv3x3_fwd", code, int(first), xin.data_ptr(), wp.data_ptr(), y.data_ptr(), stats.data_ptr(), part.data_ptr(), n, h, w, ci, cip, cop, st())
    got = to_nchw(y, code, co).cpu()
    assert torch.equal(got, want)
    if cop > co:
        assert torch.all(y[..., co:] == 0)
    assert torch.equal(stats[0, :co].cpu(), want.sum(dim=(0, 2, 3)))
    assert torch.equal(stats[1, :co].cpu(), (want * want).sum(dim=(0, 2, 3)))


@pytest.mark.parametrize("mode", ["bf16", "fp32"])
@pytest.mark.parametrize("ci,co,n,h,w", [(32, 64, 2, 16, 16), (64, 32, 1, 19, 21), (128, 64, 1, 9, 9), (32, 32, 1, 8, 16)])
def test_conv3x3_dgrad_exact(mode, ci, co, n, h, w):
    """dgrad = conv3x3 with mode-1 packed (transposed, tap-flipped) weights; compared with autograd of F.conv2d."""
    code, tdt = DT[mode]
    g = torch.Generator().manual_seed(7)
    wt = sparse_int((co, ci, 3, 3), g, -2, 2, 0.15)
    dy = sparse_int((n, co, h, w), g, -1, 1, 0.4)
    x = torch.zeros(n, ci, h, w, requires_grad=True)
    F.conv2d(x, wt, padding=1).backward(dy)
    want = x.grad
    assert want.abs().max() <= 256
    cip, cop = pad32(ci), pad32(co)
    wpd = torch.empty(cip * 9 * cop, dtype=tdt, device="cuda")
    wd = wt.cuda().contiguous()
    L().call("hyb_conv_pack_weight", code, 1, wd.data_ptr(), wpd.data_ptr(), co, ci, cop, cip, st())
    dyn = to_nhwc(dy, code, tdt, cop)
    dx = torch.empty(n, h, w, cip, dtype=tdt, device="cuda")
    L().call("hyb_conv3x3_fwd", code, 0, dyn.data_ptr(), wpd.data_ptr(), dx.data_ptr(), None, None, n, h, w, co, cop, cip, st())
    assert torch.equal(to_nchw(dx, code, ci).cpu(), want)


@pytest.mark.parametrize("mode", ["bf16", "fp32"])
@pytest.mark.parametrize("ci,co,n,h,w", [(32, 64, 2, 16, 16), (64, 128, 2, 11, 23), (128, 256, 1, 14, 14), (96, 96, 1, 8, 16),
                                         (32, 32, 3, 9, 9), (40, 72, 1, 8, 8), (3, 32, 2, 16, 32), (3, 64, 1, 21, 45), (1, 32, 1, 7, 5),
                                         (64, 64, 3, 30, 58), (128, 64, 2, 56, 56), (64, 192, 5, 17, 29), (32, 128, 3, 30, 58), (96, 64, 2, 20, 33)])
def test_conv3x3_wgrad_exact(mode, ci, co, n, h, w):
    code, tdt = DT[mode]
    g = torch.Generator().manual_seed(11)
    x = sparse_int((n, ci, h, w), g, 0, 2, 0.5)
    dy = sparse_int((n, co, h, w), g, -2, 2, 0.3)
    wt = torch.zeros(co, ci, 3, 3, requires_grad=True)
    F.conv2d(x, wt, padding=1).backward(dy)
    want = wt.grad
    first = ci <= 3
    cip, cop = (0 if first else pad32(ci)), pad32(co)
    xin = x.cuda().contiguous() if first else to_nhwc(x, code, tdt, cip)
    dyn = to_nhwc(dy, code, tdt, cop)
    dw = torch.full((co, ci, 3, 3), 3.0, device="cuda")
    nb = L().query("hyb_conv3x3_wgrad_workspace", int(first), n, h, w, cip, cop)
    assert nb > 0
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    L().call("hyb_conv3x3_wgrad", code, int(first), xin.data_ptr(), dyn.data_ptr(), dw.data_ptr(), n, h, w, ci, cip, co, cop, ws.data_ptr(), nb, st())
    assert torch.equal(dw.cpu(), want)


def _fuzz_shapes(seed, count, cis, cos):
    import random
    rnd = random.Random(seed)
    out = []
    for _ in range(count):
        ci, co = rnd.choice(cis), rnd.choice(cos)
        n = rnd.choice([1, 2, 3, 5])
        h = rnd.choice([1, 3, 4, 7, 8, 9, 27, 28, 29, 31, 57])
        w = rnd.choice([1, 2, 5, 27, 28, 29, 55, 56, 57, 60, 85])
        out.append((ci, co, n, h, w))
    return out


@pytest.mark.parametrize("ci,co,n,h,w", _fuzz_shapes(101, 10, [32, 64, 128], [32, 64, 128, 256]))
def test_conv3x3_fwd_stats_dgrad_exact_fuzz(ci, co, n, h, w):
    """Seeded shape fuzz of the asynchronous bf16 conv kernel (tile edges, 1-pixel images, several tiles per workgroup, all channel
    configurations): forward + statistics, and dgrad through the same kernel with mode-1 weights."""
    code, tdt = DT["bf16"]
    g = torch.Generator().manual_seed(ci * 7 + co * 3 + h * 11 + w)
    x = sparse_int((n, ci, h, w), g, 0, 2, 0.5)
    wt = sparse_int((co, ci, 3, 3), g, -2, 2, 0.15)
    xr = x.clone().requires_grad_(True)
    want = F.conv2d(xr, wt, padding=1)
    dy = sparse_int((n, co, h, w), g, -1, 1, 0.4)
    want.backward(dy)
    assert want.abs().max() <= 256 and xr.grad.abs().max() <= 256
    cip, cop = pad32(ci), pad32(co)
    wp = torch.empty(cop * 9 * cip, dtype=tdt, device="cuda")
    wpd = torch.empty(cip * 9 * cop, dtype=tdt, device="cuda")
    wd = wt.cuda().contiguous()
    L().call("hyb_conv_pack_weight", code, 0, wd.data_ptr(), wp.data_ptr(), co, ci, cop, cip, st())
    L().call("hyb_conv_pack_weight", code, 1, wd.data_ptr(), wpd.data_ptr(), co, ci, cop, cip, st())
    xin = to_nhwc(x, code, tdt, cip)
    y = torch.full((n, h, w, cop), 7.0, dtype=tdt, device="cuda")
    stats = torch.zeros(2, cop, device="cuda")
    part = torch.empty(L().query("hyb_conv_stats_workspace", cop), dtype=torch.uint8, device="cuda")
    L().call("hyb_conv3x3_fwd", code, 0, xin.data_ptr(), wp.data_ptr(), y.data_ptr(), stats.data_ptr(), part.data_ptr(), n, h, w, ci, cip, cop, st())
    assert torch.equal(to_nchw(y, code, co).cpu(), want.detach())
    assert torch.equal(stats[0, :co].cpu(), want.detach().sum(dim=(0, 2, 3)))
    assert torch.equal(stats[1, :co].cpu(), (want.detach() * want.detach()).sum(dim=(0, 2, 3)))
    dyn = to_nhwc(dy, code, tdt, cop)
    dx = torch.empty(n, h, w, cip, dtype=tdt, device="cuda")
    L().call("hyb_conv3x3_fwd", code, 0, dyn.data_ptr(), wpd.data_ptr(), dx.data_ptr(), None, None, n, h, w, co, cop, cip, st())
    assert torch.equal(to_nchw(dx, code, ci).cpu(), xr.grad)


@pytest.mark.parametrize("ci,co,n,h,w", _fuzz_shapes(202, 10, [32, 64, 96, 128], [64, 128, 192]))
def test_conv3x3_wgrad_exact_fuzz(ci, co, n, h, w):
    """Seeded shape fuzz of the warp-specialised bf16 weight-gradient kernel (both channel-block widths, tile edges, odd tile counts)."""
    code, tdt = DT["bf16"]
    g = torch.Generator().manual_seed(ci * 5 + co + h * 13 + w)
    x = sparse_int((n, ci, h, w), g, 0, 2, 0.5)
    dy = sparse_int((n, co, h, w), g, -2, 2, 0.3)
    wt = torch.zeros(co, ci, 3, 3, requires_grad=True)
    F.conv2d(x, wt, padding=1).backward(dy)
    cip, cop = pad32(ci), pad32(co)
    xin = to_nhwc(x, code, tdt, cip)
    dyn = to_nhwc(dy, code, tdt, cop)
    dw = torch.full((co, ci, 3, 3), 3.0, device="cuda")
    nb = L().query("hyb_conv3x3_wgrad_workspace", 0, n, h, w, cip, cop)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    L().call("hyb_conv3x3_wgrad", code, 0, xin.data_ptr(), dyn.data_ptr(), dw.data_ptr(), n, h, w, ci, cip, co, cop, ws.data_ptr(), nb, st())
    assert torch.equal(dw.cpu(), wt.grad)


@pytest.mark.parametrize("mode", ["bf16", "fp32"])
@pytest.mark.parametrize("M,N,K,relu", [(128, 512, 512, 1), (128, 2048, 512, 1), (128, 512, 2048, 0), (5, 24, 40, 0), (37, 72, 256, 1),
                                        (512, 768, 768, 0), (128, 512, 256, 0)])
def test_linear_fwd_bwd_exact(mode, M, N, K, relu):
    code, tdt = DT[mode]
    g = torch.Generator().manual_seed(M + N + K)
    x = sparse_int((M, K), g, -2, 2, 0.2)
    W = sparse_int((N, K), g, -2, 2, 0.1)
    b = sparse_int((N,), g, -3, 3, 0.5)
    dy = sparse_int((M, N), g, -1, 1, 0.1)
    xr, Wr, br = x.clone().requires_grad_(True), W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_want = F.linear(xr, Wr, br)
    if relu:
        y_want = F.relu(y_want)
    y_want.backward(dy)
    assert y_want.abs().max() <= 256 and xr.grad.abs().max() <= 256
    xd, Wd, bd, dyd = x.cuda().to(tdt), W.cuda(), b.cuda(), dy.cuda().to(tdt)
    y = torch.empty(M, N, dtype=tdt, device="cuda")
    L().call("hyb_linear_fwd", code, xd.data_ptr(), K, Wd.data_ptr(), bd.data_ptr(), y.data_ptr(), M, N, K, relu, st())
    assert torch.equal(y.float().cpu(), y_want.detach())
    dx = torch.empty(M, K, dtype=tdt, device="cuda")
    dW = torch.empty(N, K, device="cuda")
    db = torch.empty(N, device="cuda")
    ws = torch.empty(M * N * 4, dtype=torch.uint8, device="cuda")
    L().call("hyb_linear_bwd", code, xd.data_ptr(), K, Wd.data_ptr(), y.data_ptr(), dyd.data_ptr(), dx.data_ptr(), 0, dW.data_ptr(), db.data_ptr(),
             M, N, K, relu, ws.data_ptr(), ws.numel(), st())
    assert torch.equal(dx.float().cpu(), xr.grad)
    assert torch.equal(dW.cpu(), Wr.grad)
    assert torch.equal(db.cpu(), br.grad)
    # accumulate_dx adds into dx
    L().call("hyb_linear_bwd", code, xd.data_ptr(), K, Wd.data_ptr(), y.data_ptr(), dyd.data_ptr(), dx.data_ptr(), 1, None, None,
             M, N, K, relu, ws.data_ptr(), ws.numel(), st())
    assert torch.equal(dx.float().cpu(), 2 * xr.grad)


@pytest.mark.parametrize("mode", ["bf16", "fp32"])
def test_attention_scores_exact_with_uniform_values(mode):
    """With V constant along the key axis the output must equal that constant whatever the softmax weights are
    (rows of P sum to 1): checks P.V tiling, head merge indexing and padding for ragged S."""
    code, tdt = DT[mode]
    for B, S, D, H in [(2, 16, 64, 4), (3, 5, 48, 3), (1, 33, 128, 2), (2, 64, 192, 2)]:
        g = torch.Generator().manual_seed(S)
        q = torch.randn(B, S, D, generator=g).cuda().to(tdt)
        k = torch.randn(B, S, D, generator=g).cuda().to(tdt)
        vrow = torch.randint(-8, 9, (B, 1, D), generator=g).float()
        v = vrow.expand(B, S, D).contiguous().cuda().to(tdt)
        out = torch.empty(B, S, D, dtype=tdt, device="cuda")
        stats = torch.empty(B * H, S, 2, device="cuda")                 # (row max, row sum of exp) per query
        L().call("hyb_attention_fwd", code, q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(), stats.data_ptr(), B, S, D, H, 0.0, 0, st())
        sc = torch.einsum("bhqd,bhkd->bhqk", q.float().view(B, S, H, D // H).transpose(1, 2), k.float().view(B, S, H, D // H).transpose(1, 2)) / D ** 0.5
        assert torch.allclose(stats[..., 0], sc.amax(-1).reshape(B * H, S), atol=1e-4, rtol=1e-4)
        assert torch.allclose(stats[..., 1], torch.exp(sc - sc.amax(-1, keepdim=True)).sum(-1).reshape(B * H, S), rtol=1e-3)
        assert torch.allclose(out.float().cpu(), vrow.expand(B, S, D), atol=0.07 if mode == "bf16" else 1e-4)
