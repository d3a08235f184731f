"""bf16-ROUNDED CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as oracle/hybrid_ref.py: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

SURVEY.md section 7.2: the bf16 fast mode is to be reported "against both the fp32 oracle and a bf16-rounded oracle (same
rounding points)".  This file is that second oracle: the SAME algorithm as oracle/hybrid_ref.py (it runs on the parameters
of a ``TransformerCNNHybridRef`` instance and follows the same reference lines -- UNet.py:58-60 + :13 for the conv stage,
TransformerEncoder.pyc src L49-126 for the temporal blocks), computed in fp32 on the host, but with every value the HIP
path STORES or feeds to a matrix core in bf16 rounded to bf16 at that point, forward and backward:

  * matrix-core operands: activations, weights (``rv``: value rounded, gradient untouched -- weight gradients are fp32);
  * stored activations: pooled maps, raw conv outputs of stages >= 2 (statistics come from the fp32 accumulators BEFORE
    the rounding, like the conv epilogue does), frame features, tokens, q/k/v, dropped-out attention weights, attention
    output, o, x1, hmid, f, layer outputs (``rb``: value and incoming gradient rounded);
  * stored activation gradients that have no stored forward twin: d(raw conv output) of stages >= 2 and d(q k^T) (``rg``).

Accumulation stays fp32 (the MFMA accumulates in fp32; products of two bf16 numbers are exact in fp32), so what is left
between this oracle and the bf16 HIP path is summation order, a few double roundings where the kernels accumulate into a
stored bf16 tensor, and the rare 1-ulp flips those cause.  PER STAGE that is 3e-5..5e-4 relative L2 (0.01 % of the elements
one ulp apart; tokens, q/k/v and the attention block come out bit-identical -- scripts/diag_bf16_stages.py), two orders of
magnitude below bf16's distance from the fp32 oracle, so the per-stage bf16 tests (tests/test_gpu_parity.py) gate forward
and backward tightly against this oracle.  END TO END the agreement is NOT tight and cannot be: rounding is discontinuous,
an L2 discrepancy eps << u = 2^-8 becomes sqrt(eps * u) after the next rounding point (a fraction eps/u of the elements
flips by a whole ulp), so any two implementations with identical rounding points but different summation order drift to
bf16-noise-level differences after a handful of stages (measured: logits 5-7e-3 apart at full size, the same order as
either one's distance from the fp32 oracle).  The full-size tests report both distances (tests/test_gpu_fullsize.py).
Dropout is not modelled: parity runs use p = 0 (decision 4).  The functions follow the dtype of the
model they are given: on a ``.double()`` copy of the oracle the accumulation is fp64 (rounding points unchanged), which takes
the host's own fp32 summation error out of the comparison (tests/test_gpu_fullsize.py).
"""
import math

import torch
import torch.nn.functional as F


class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, value, grad):
        ctx.grad = grad
        return x.bfloat16().to(x.dtype) if value else x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return (g.bfloat16().to(g.dtype) if ctx.grad else g), None, None


def rv(x):
    return _Round.apply(x, True, False)


def rg(x):
    return _Round.apply(x, False, True)


def rb(x):
    return _Round.apply(x, True, True)


def _conv_stage(seq, name, x, first, training):
    """UNet.py:58-60 + UNet.py:13 with the HIP path's rounding points.  Running statistics are NOT updated here (the fp32
    oracle checks those)."""
    conv, bn = getattr(seq, name + "conv1"), getattr(seq, name + "norm1")
    y32 = F.conv2d(rv(x) if first else x, rv(conv.weight), None, padding=1)
    if not first:
        y32 = rg(y32)                                   # the dense gradient of the raw conv output is stored once, in bf16
    if training or not bn.track_running_stats:
        mean = y32.mean(dim=(0, 2, 3))
        var = y32.var(dim=(0, 2, 3), unbiased=False)
    else:
        mean, var = bn.running_mean, bn.running_var
    invstd = torch.rsqrt(var + bn.eps)
    scale = bn.weight * invstd
    shift = bn.bias - mean * scale
    y = y32 if first else rv(y32)                       # stage 1 never materialises its conv output (recomputed in fp32 registers)
    z = torch.relu(y * scale[None, :, None, None] + shift[None, :, None, None])
    return rb(F.max_pool2d(z, 2, 2))


def _linear(x, lin, relu=False):
    y = F.linear(x, rv(lin.weight), lin.bias)
    return rb(torch.relu(y) if relu else y)


def _mha(att, x, mask, k_in=None, v_in=None):
    """MultiheadAttention.forward(q, k, v, mask) (k = v = q unless given), src L67-89 (attention core L49-62, head split L22-45)."""
    B, S, D = x.shape
    H = att.num_heads
    dh = D // H
    q = _linear(x, att.query_layer, True)
    k = _linear(x if k_in is None else k_in, att.key_layer, True)
    v = _linear(x if v_in is None else v_in, att.value_layer, True)

    def split(t):
        return t.reshape(B, S, H, dh).permute(0, 2, 1, 3).reshape(B * H, S, dh)
    q, k, v = split(q), split(k), split(v)
    dot = rg(torch.matmul(q, k.transpose(-2, -1))) / math.sqrt(att.input_dim)
    if mask is not None:
        dot = dot.masked_fill(mask.repeat(H, 1, 1) == 0, -1e9)
    w = rv(torch.softmax(dot, dim=-1))                  # the probabilities are kept in fp32; the P.V operand is bf16
    a = rb(torch.matmul(w, v))
    a = a.reshape(B, H, S, dh).permute(0, 2, 1, 3).reshape(B, S, D)
    return _linear(a, att.output_layer)


def _encoder(enc, x, mask):
    """TransformerEncoder.forward, src L110-126 (dropout p must be 0)."""
    assert enc.dropout == 0.0
    for i in range(enc.num_layers):
        att, ff, ln = enc.attention_layers[i], enc.feedforward_layers[i], enc.layer_norm[i]
        assert not (att.training and att.dropoutLayer.p > 0.0)
        skip1 = x
        x = rb(ln(_mha(att, x, mask)) + skip1)
        skip2 = x
        f = _linear(_linear(x, ff[0], True), ff[2])
        x = rb((ln(f) + skip2) * math.sqrt(0.5))
    return x


def forward(ref, x, mask=None):
    """Logits of ``ref`` (a hybrid_ref.TransformerCNNHybridRef) on clips x with the bf16 path's rounding points."""
    if x.dim() == 4:
        x = x.unsqueeze(1)
    B, T = x.shape[:2]
    f = x.reshape(B * T, *x.shape[2:])
    for i in range(ref.num_stages):
        f = _conv_stage(getattr(ref, f"encoder{i + 1}"), f"enc{i + 1}", f, i == 0, ref.training)
    feat = rb(f.mean(dim=(2, 3)))
    tok = _linear(feat, ref.token_proj).reshape(B, T, -1)
    enc = _encoder(ref.encoder, tok, mask)
    return F.linear(enc.mean(dim=1), ref.head.weight, ref.head.bias)       # the head runs in fp32 on the stored bf16 tokens


def conv_stage(seq, name, x, first, training):
    return _conv_stage(seq, name, x, first, training)


def encoder(enc, x, mask):
    return _encoder(enc, rb(x), mask)


def mha(att, q, k, v, mask):
    return _mha(att, rb(q), mask, rb(k), rb(v))
