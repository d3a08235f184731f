"""Drop-in nn.Modules for the hot path.  Class names, constructor/forward signatures and state-dict
keys follow the reference:

* ``MultiheadAttention(input_dim, num_heads)`` / ``forward(q, k, v, mask=None)``
  and ``TransformerEncoder(input_dim, hidden_dim, num_layers, num_heads, dropout)`` / ``forward(input, mask)``
  -- ``__pycache__/TransformerEncoder.cpython-38.pyc`` src L6-L126.
* the conv stage is the first Conv+BN+ReLU triple of ``UNet._block`` (UNet.py:54-66) + ``MaxPool2d(2,2)``
  (UNet.py:13) with the reference's key names ``encoder{i}.enc{i}conv1.weight``, ``encoder{i}.enc{i}norm1.*``.
* ``TransformerCNNHybrid()`` is zero-argument constructible and is called as ``model(x)`` exactly like the
  reference harnesses call their models (Model.py:27,56 / FCT.py:302,330).

Parameters are ordinary fp32 ``nn.Parameter``s, so stock AdamW, checkpointing and gradient
all-reduce work unchanged.  All compute runs in the HIP library through the ``torch.ops.hybrid.*`` custom operators
(ops.py); CPU tensors raise RuntimeError.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn

from . import ops


def split_compute_dtype(name):
    """-> (backbone dtype, temporal dtype).  'mixed' = ('bf16', 'bf16x3'): the conv stages (99.7 % of the FLOPs) in bf16, the token projection
    + temporal encoder + head in fp32 storage with split-bf16 MFMA products -- the bf16 mode's logits error comes from the temporal half
    (measured at config 2, eval: 9.0e-3 all bf16, 5.1e-4 with only the backbone in bf16; scripts/exp/logit_error_split.py), so this mode
    keeps north_star's 1e-3 at nearly the bf16 mode's speed.  A pair ('bf16', 'fp32') etc. is taken as given."""
    if isinstance(name, (tuple, list)) and len(name) == 2:
        return name[0], name[1]
    if isinstance(name, str) and name == "mixed":
        return "bf16", "bf16x3"
    return name, name


class _ComputeDtypeMixin:
    def set_compute_dtype(self, name):
        """'bf16' (default: bf16 storage + bf16 MFMA, fp32 accumulate), 'bf16x3' (fp32 storage, three bf16 MFMAs per product) or 'fp32'
        (exact fp32 MFMA; parity gate).  TransformerCNNHybrid also takes 'mixed' / a (backbone, temporal) pair (split_compute_dtype)."""
        code = ops.dtype_code(name)
        for m in self.modules():
            if isinstance(m, _ComputeDtypeMixin):
                m._dt = code
        return self


class ConvBNReLUPool(nn.Sequential, _ComputeDtypeMixin):
    """Conv2d(3x3, pad 1, bias=False) -> BatchNorm2d -> ReLU -> MaxPool2d(2, 2) as ONE fused HIP stage.

    The child modules exist to own the parameters/buffers under the reference's key names
    (``{name}conv1.weight``, ``{name}norm1.{weight,bias,running_mean,running_var,num_batches_tracked}``);
    their own forward()s are never called."""

    def __init__(self, in_channels, features, name, compute_dtype="bf16"):
        super().__init__(OrderedDict([
            (name + "conv1", nn.Conv2d(in_channels, features, kernel_size=3, padding=1, bias=False)),
            (name + "norm1", nn.BatchNorm2d(num_features=features)),
        ]))
        self._conv = name + "conv1"
        self._norm = name + "norm1"
        self.in_channels, self.features = in_channels, features
        self._dt = ops.dtype_code(compute_dtype)

    def forward_nhwc(self, x, first, commit=None):
        """x: NCHW fp32 frames when ``first`` else NHWC compute-dtype activations. Returns NHWC.  ``commit``: see ops.convstage."""
        conv, bn = getattr(self, self._conv), getattr(self, self._norm)
        if bn.momentum is None:
            raise RuntimeError("cumulative-average BatchNorm (momentum=None) is not supported")
        training = self.training or not bn.track_running_stats or bn.running_mean is None
        return ops.convstage(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                             training, bn.momentum, bn.eps, self._dt, first, commit)

    def forward(self, x):
        """Standalone use: [N,C,H,W] fp32 in, [N,features,H/2,W/2] fp32 out."""
        if x.dim() != 4:
            raise ValueError("expected [N,C,H,W]")
        if self.in_channels <= 4:
            h = self.forward_nhwc(x.float(), True)
        else:
            h = self.forward_nhwc(ops.nchw_to_nhwc(x, self._dt, ops.pad_channels(self.in_channels)), False)
        return ops.nhwc_to_nchw(h, self._dt, self.features)


class MultiheadAttention(nn.Module, _ComputeDtypeMixin):
    def __init__(self, input_dim, num_heads, compute_dtype="bf16"):                 # src L7-19
        super().__init__()
        self.input_dim = input_dim
        self.num_heads = num_heads
        self.query_layer = nn.Linear(input_dim, input_dim)
        self.key_layer = nn.Linear(input_dim, input_dim)
        self.value_layer = nn.Linear(input_dim, input_dim)
        self.output_layer = nn.Linear(input_dim, input_dim)
        self.activation = nn.ReLU()
        self.softmax = nn.Softmax(dim=-1)
        self.dropoutLayer = nn.Dropout(0.1)
        self._dt = ops.dtype_code(compute_dtype)

    def _attn_p(self):
        return self.dropoutLayer.p if self.training else 0.0                         # quirk Q5

    def _params(self):
        return [self.query_layer.weight, self.query_layer.bias, self.key_layer.weight, self.key_layer.bias,
                self.value_layer.weight, self.value_layer.bias, self.output_layer.weight, self.output_layer.bias]

    def forward(self, q, k, v, mask=None):                                           # src L67-89
        if self.input_dim % self.num_heads != 0:
            raise ValueError("input_dim must be divisible by num_heads")
        out = ops.mha(ops.to_compute(q, self._dt), ops.to_compute(k, self._dt), ops.to_compute(v, self._dt), mask, self._params(),
                      self._dt, self.num_heads, self._attn_p(), ops.next_seed())
        return ops.to_f32(out, self._dt)


class TransformerEncoder(nn.Module, _ComputeDtypeMixin):
    def __init__(self, input_dim, hidden_dim, num_layers, num_heads, dropout, compute_dtype="bf16"):   # src L94-108
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.num_heads = num_heads
        self.dropout = dropout
        if input_dim % num_heads != 0:
            raise ValueError(
                f"Input dimension must be divisible by number of heads. Here, Input dimension = {input_dim}"
                f" is not divisible by number of heads = {num_heads}")
        self.attention_layers = nn.ModuleList(
            [MultiheadAttention(input_dim, num_heads, compute_dtype) for _ in range(num_layers)])
        self.feedforward_layers = nn.ModuleList(
            [nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, input_dim))
             for _ in range(num_layers)])
        self.layer_norm = nn.ModuleList([nn.LayerNorm(input_dim) for _ in range(num_layers)])
        self._dt = ops.dtype_code(compute_dtype)

    def _flat_params(self):
        ps = []
        for i in range(self.num_layers):
            ps += self.attention_layers[i]._params()
            ff = self.feedforward_layers[i]
            ps += [ff[0].weight, ff[0].bias, ff[2].weight, ff[2].bias, self.layer_norm[i].weight, self.layer_norm[i].bias]
        return ps

    def forward_compute(self, x, mask):
        """x already in the compute dtype ([B,S,D]); returns the compute dtype."""
        attn_p = self.attention_layers[0]._attn_p() if self.num_layers else 0.0
        return ops.encoder(x, mask, self._flat_params(), self._dt, self.hidden_dim, self.num_layers, self.num_heads, attn_p,
                           float(self.dropout), ops.next_seed())                                  # Q6: the per-layer dropout is always active

    def forward(self, input, mask):                                                  # src L110-126
        return ops.to_f32(self.forward_compute(ops.to_compute(input, self._dt), mask), self._dt)


class HybridCrossEntropyLoss(nn.Module):
    """Mean cross-entropy over the batch (the composite's own loss), one HIP kernel each way."""

    def forward(self, logits, target):
        return ops.cross_entropy(logits, target)


class TransformerCNNHybrid(nn.Module, _ComputeDtypeMixin):
    """CNN backbone + temporal Transformer encoder over clips [B,T,3,H,W] (or [B,3,H,W] => T=1) -> logits [B,num_classes].

    Defaults are BASELINE config 2: cnn_channels (32,64,128,256) (the UNet(init_features=32) encoder widths,
    UNet.py:8-18), d_model 512, 8 heads, 2 layers, hidden 2048, 8 classes, dropout 0."""

    def __init__(self, in_channels=3, cnn_channels=(32, 64, 128, 256), d_model=512, num_heads=8, num_layers=2, hidden_dim=2048,
                 num_classes=8, dropout=0.0, compute_dtype="bf16"):
        super().__init__()
        chans = (in_channels,) + tuple(cnn_channels)
        if chans[-1] % 8 != 0:
            # the frame-token projection reads 16-byte rows of its [d_model, C] weight and of the pooled features
            raise ValueError(f"TransformerCNNHybrid on the MI355X HIP path needs cnn_channels[-1] to be a multiple of 8 (got {chans[-1]}); "
                             "there is no CPU fallback")
        self.num_stages = len(cnn_channels)
        cb, ct = split_compute_dtype(compute_dtype)
        for i in range(self.num_stages):
            setattr(self, f"encoder{i + 1}", ConvBNReLUPool(chans[i], chans[i + 1], f"enc{i + 1}", cb))
        self.in_channels = in_channels
        self.token_proj = nn.Linear(chans[-1], d_model)
        self.encoder = TransformerEncoder(d_model, hidden_dim, num_layers, num_heads, dropout, ct)
        self.head = nn.Linear(d_model, num_classes)
        self._dt, self._dt_t = ops.dtype_code(cb), ops.dtype_code(ct)            # conv stages / token projection + encoder + head
        self.fuse_model_ops = True        # hybrid::backbone + hybrid::temporal (one C call each way) instead of one operator per stage

    def set_compute_dtype(self, name):
        cb, ct = split_compute_dtype(name)
        for i in range(self.num_stages):
            getattr(self, f"encoder{i + 1}").set_compute_dtype(cb)
        self.encoder.set_compute_dtype(ct)
        self._dt, self._dt_t = ops.dtype_code(cb), ops.dtype_code(ct)
        return self

    def _temporal_dt(self):
        """dtype code of the fused temporal operators: bf16 conv stages in front of an fp32-storage temporal part hand their bf16 map over
        as it is (HYB_H_BF16: the global-average-pool kernels convert, no cast launches)."""
        if ops.torch_dtype(self._dt) == torch.bfloat16 and ops.torch_dtype(self._dt_t) == torch.float32:
            return self._dt_t | ops.HYB_H_BF16
        return self._dt_t

    def _to_temporal(self, h):
        """The last pooled map in the temporal part's storage type (unfused path / other pairs: a cast launch each way)."""
        tb, tt = ops.torch_dtype(self._dt), ops.torch_dtype(self._dt_t)
        if tb == tt:
            return h
        h32 = h if tb == torch.float32 else ops.to_f32(h, self._dt)
        return h32 if tt == torch.float32 else ops.to_compute(h32, self._dt_t)

    def _fused(self):
        """hybrid::backbone / hybrid::temporal apply when the model has the plain reference structure."""
        stages = [getattr(self, f"encoder{i + 1}") for i in range(self.num_stages)]
        bn0 = getattr(stages[0], stages[0]._norm)
        return (self.fuse_model_ops and self.in_channels <= 4 and self.encoder.num_layers > 0 and self.token_proj.bias is not None
                and self.head.bias is not None
                and all(getattr(s, s._norm).track_running_stats and getattr(s, s._norm).running_mean is not None
                        and getattr(s, s._norm).momentum == bn0.momentum and getattr(s, s._norm).eps == bn0.eps
                        and s.training == self.training for s in stages))

    def forward_backbone(self, x):
        """Clips [B,T,C,H,W] (or frames [B,C,H,W] => T=1) -> (last pooled map, NHWC compute dtype [B*T, H/2^S, W/2^S, Cp], B)."""
        if x.dim() == 4:
            x = x.unsqueeze(1)
        if x.dim() != 5:
            raise ValueError("expected a clip tensor [B,T,C,H,W] or a frame batch [B,C,H,W]")
        if not x.is_cuda:
            raise RuntimeError("TransformerCNNHybrid runs on the MI355X HIP path only: move the model and input to 'cuda' "
                               "(there is no CPU fallback)")
        B, T = x.shape[:2]
        f = x.reshape(B * T, *x.shape[2:]).float()                  # frames folded into the batch axis
        stages = [getattr(self, f"encoder{i + 1}") for i in range(self.num_stages)]
        if self._fused():
            # model-level operator: the same kernels as the stage operators below, chained in C (one call)
            return ops.backbone(f, [(getattr(s, s._conv).weight, getattr(s, s._norm)) for s in stages], self.training, self._dt), B
        commit = []                                                  # running-statistics write-back of all stages: one multi-tensor copy
        if self.in_channels <= 4:
            h = self.encoder1.forward_nhwc(f, True, commit)
        else:
            h = self.encoder1.forward_nhwc(ops.nchw_to_nhwc(f, self._dt, ops.pad_channels(self.in_channels)), False, commit)
        for i in range(1, self.num_stages):
            h = stages[i].forward_nhwc(h, False, commit)
        ops.commit_running_stats(commit)
        return h, B

    def forward_temporal(self, h, B, mask=None):
        """Last pooled map -> frame tokens -> temporal encoder -> head: logits [B, num_classes]."""
        enc = self.encoder
        if self._fused():
            tdt = self._temporal_dt()
            return ops.temporal(h if tdt & ops.HYB_H_BF16 else self._to_temporal(h), self.token_proj.weight, self.token_proj.bias, enc._flat_params(),
                                self.head.weight, self.head.bias, mask, B, tdt, enc.hidden_dim, enc.num_layers, enc.num_heads, enc.attention_layers[0]._attn_p(), float(enc.dropout),
                                ops.next_seed())
        h = self._to_temporal(h)
        tok = ops.token(h, self.token_proj.weight, self.token_proj.bias, self._dt_t).reshape(B, h.shape[0] // B, -1)
        return ops.head(enc.forward_compute(tok, mask), self.head.weight, self.head.bias, self._dt_t)

    def forward_temporal_loss(self, h, B, target, mask=None):
        """Last pooled map + class indices [B] -> (mean cross-entropy loss, logits): ``HybridCrossEntropyLoss()(forward_temporal(h, B, mask), target)``
        with the loss inside the temporal part's own launches (hybrid::temporal_ce; same bits).  Plain-structure models only (``_fused()``)."""
        enc = self.encoder
        if not self._fused():
            logits = self.forward_temporal(h, B, mask)
            return ops.cross_entropy(logits, target), logits
        tdt = self._temporal_dt()
        return ops.temporal_ce(h if tdt & ops.HYB_H_BF16 else self._to_temporal(h), self.token_proj.weight, self.token_proj.bias, enc._flat_params(),
                               self.head.weight, self.head.bias, mask, target, B, tdt, enc.hidden_dim, enc.num_layers, enc.num_heads, enc.attention_layers[0]._attn_p(), float(enc.dropout),
                               ops.next_seed())

    def backbone_parameters(self):
        return [p for i in range(self.num_stages) for p in getattr(self, f"encoder{i + 1}").parameters()]

    def temporal_parameters(self):
        return list(self.token_proj.parameters()) + list(self.encoder.parameters()) + list(self.head.parameters())

    def forward(self, x, mask=None):
        h, B = self.forward_backbone(x)
        return self.forward_temporal(h, B, mask)
