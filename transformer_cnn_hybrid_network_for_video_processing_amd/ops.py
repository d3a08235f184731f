"""The hot path as custom torch operators: ``torch.ops.hybrid.*`` over the C ABI of include/hybrid_hip.h.

Every operator enqueues HIP kernels on torch's current stream through ctypes; tensors only provide device memory.
Forward operators return the tensors their backward needs as extra outputs; each backward is itself an operator
(``hybrid::*_bwd``), so the whole path is visible to the dispatcher, has fake (meta) implementations for shape inference
(``torch.library.register_fake``) and passes ``torch.library.opcheck``.  There is no eager/CPU implementation behind them:
a CPU tensor raises.

Registration uses torch.library's low-level interface (``Library.define`` / ``Library.impl`` + ``register_fake``) with a
hand-written ``torch.autograd.Function`` on the Autograd dispatch key of every differentiable operator, NOT the
``custom_op`` / ``register_autograd`` decorators: those were measured first (round 2) and cost 120 us per forward call and
430 us per forward+backward pair for an operator with a 28-tensor parameter list (generic pytree flattening, schema
re-checks, a dynamo-disable wrapper per call) against 36 / 230 us for this form -- with ~1.6 ms of GPU work per step that
difference decides whether the step is GPU-bound.  What the dispatcher sees is the same: a schema, a backend kernel, a
fake kernel and an autograd kernel per operator.

    hybrid::convstage        Conv3x3 -> BatchNorm2d -> ReLU -> MaxPool2d(2,2)   (UNet.py:58-60 + UNet.py:13)
    hybrid::token            global average pool + Linear(C, d)                  (composite's own glue)
    hybrid::encoder          TransformerEncoder.forward, all layers              (TransformerEncoder.pyc src L110-126)
    hybrid::mha              MultiheadAttention.forward                          (src L67-89)
    hybrid::head             mean over T + Linear(d, classes)                    (composite's own)
    hybrid::cross_entropy    mean cross-entropy                                  (composite's own)
    hybrid::cast, hybrid::nchw_to_nhwc, hybrid::nhwc_to_nchw                     layout / dtype glue for standalone module use
"""
import functools
import os
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from ._lib import HYB_BF16, HYB_F32, HYB_F32X3, HYB_H_BF16, lib, ptr_array

_TORCH_DTYPE = {HYB_F32: torch.float32, HYB_BF16: torch.bfloat16, HYB_F32X3: torch.float32}
_LIB = torch.library.Library("hybrid", "DEF")
_below_autograd = torch._C._AutoDispatchBelowAutograd


def _define(name, schema, impl, fake, autograd=None):
    """One operator: schema, backend kernel (every device: CPU tensors get the "no CPU fallback" error from _require_cuda),
    fake kernel, and -- for differentiable operators -- the Autograd-key kernel wrapping a torch.autograd.Function."""
    _LIB.define(name + schema)
    _LIB.impl(name, impl, "CompositeExplicitAutograd")
    torch.library.register_fake("hybrid::" + name, fake)
    if autograd is not None:
        _LIB.impl(name, autograd, "Autograd")
_SEED_COUNTER = [0]
_SEED_MASK = 0x7FFFFFFFFFFFFFFF          # operator schemas carry ints as int64


def dtype_code(name):
    if isinstance(name, int) and not isinstance(name, bool) and name in (HYB_F32, HYB_BF16, HYB_F32X3):
        return name
    if isinstance(name, str) and name == "bf16x3":       # fp32 storage, every contraction product from three bf16 MFMAs (hyb_common.h)
        return HYB_F32X3
    if isinstance(name, str) and name in ("fp32", "float32") or name is torch.float32:
        return HYB_F32
    if isinstance(name, str) and name in ("bf16", "bfloat16") or name is torch.bfloat16:
        return HYB_BF16
    raise ValueError(f"compute dtype must be 'bf16', 'bf16x3' or 'fp32', got {name!r}")


def torch_dtype(code):
    return _TORCH_DTYPE[code]


def pad_channels(c):
    return (c + 31) // 32 * 32


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("the MI355X HIP path needs tensors on a cuda (ROCm) device; there is no CPU fallback")


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


@functools.lru_cache(maxsize=None)
def _query(name, *args):
    """Size queries are pure host functions of their integer arguments: ask the library once per shape.  A tuple argument is
    passed as a C int array."""
    import ctypes
    return lib.query(name, *[(ctypes.c_int * len(a))(*a) if isinstance(a, tuple) else a for a in args])


def _rank():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank()
    return int(os.environ.get("RANK", "0"))


def next_seed():
    """Dropout seed of the next stochastic operator call: deterministic under torch.manual_seed, different on every rank of
    a data-parallel job (ranks share the weight seed but must not share dropout masks), no device sync."""
    _SEED_COUNTER[0] += 1
    return ((torch.initial_seed() * 0x9E3779B97F4A7C15 + _SEED_COUNTER[0] * 0xD1B54A32D192ED03 + _rank() * 0xA24BAED4963EE407)
            & _SEED_MASK)


_STEP_COUNTER = [None]


def set_step_counter(counter):
    """Register (or clear, with None) the DEVICE step counter (int64 [1]) that stochastic kernels add to their by-value dropout
    seed when they run.  Replayed hipGraphs (graph.GraphedTrainStep) re-issue the captured launches with the captured seed;
    advancing this counter once per replay gives every step its own masks.  Eager runs leave it unset."""
    if counter is not None and not (counter.is_cuda and counter.dtype == torch.int64 and counter.numel() == 1):
        raise ValueError("the step counter must be a cuda int64 tensor with one element")
    _STEP_COUNTER[0] = counter


def step_counter():
    return _STEP_COUNTER[0]


def _opt_ptr(t):
    return t.data_ptr() if t is not None else None


def check_mask(mask, B, S, device):
    """The reference applies ``masked_fill(mask.repeat(H,1,1) == 0, -1e9)`` to scores [B*H,S,S] (src L54-55, L77-78): the mask
    must have B leading entries and broadcast to [S,S].  Returns fp32 [B,S,S] contiguous on the device, or raises like torch."""
    if mask is None:
        return None
    if not isinstance(mask, torch.Tensor):
        raise TypeError("mask must be a tensor or None")
    if mask.device != device:
        raise RuntimeError(f"mask is on {mask.device} but the input is on {device} (the reference raises for mixed devices too)")
    if mask.dim() != 3 or mask.shape[0] != B or mask.shape[1] not in (1, S) or mask.shape[2] not in (1, S):
        raise RuntimeError(f"mask of shape {tuple(mask.shape)} does not broadcast against attention scores [B*H,S,S] with B={B}, S={S} "
                           "(expected [B,S,S])")
    return mask.expand(B, S, S).to(torch.float32).contiguous()


# ---------------------------------------------------------------------------------------------
# layout / cast glue (standalone module use; the fused model path needs neither)
# ---------------------------------------------------------------------------------------------
def nchw_to_nhwc_op(x: Tensor, dt: int, cp: int) -> Tensor:
    _require_cuda(x)
    x = x.contiguous().float()
    N, C, H, W = x.shape
    out = torch.empty(N, H, W, cp, dtype=_TORCH_DTYPE[dt], device=x.device)
    lib.call("hyb_nchw_to_nhwc", dt, x.data_ptr(), out.data_ptr(), N, C, H, W, cp, _stream())
    return out


def nchw_to_nhwc_fake(x, dt, cp):
    N, C, H, W = x.shape
    return x.new_empty((N, H, W, cp), dtype=_TORCH_DTYPE[dt])


def nhwc_to_nchw_op(x: Tensor, dt: int, C: int) -> Tensor:
    _require_cuda(x)
    x = x.contiguous()
    N, H, W, cp = x.shape
    out = torch.empty(N, C, H, W, dtype=torch.float32, device=x.device)
    lib.call("hyb_nhwc_to_nchw", dt, x.data_ptr(), out.data_ptr(), N, C, H, W, cp, _stream())
    return out


def nhwc_to_nchw_fake(x, dt, C):
    N, H, W, cp = x.shape
    return x.new_empty((N, C, H, W), dtype=torch.float32)












def nchw_to_nhwc(x, dt, cp):
    return torch.ops.hybrid.nchw_to_nhwc(x, dt, cp)


def nhwc_to_nchw(x, dt, C):
    return torch.ops.hybrid.nhwc_to_nchw(x, dt, C)


def cast_op(x: Tensor, dt: int, to_t: bool) -> Tensor:
    """fp32 <-> T with the library's own cast kernels (differentiable)."""
    _require_cuda(x)
    x = x.contiguous()
    if to_t:
        x = x.float()
        out = torch.empty(x.shape, dtype=_TORCH_DTYPE[dt], device=x.device)
        lib.call("hyb_cast_from_f32", dt, x.data_ptr(), out.data_ptr(), x.numel(), _stream())
    else:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        lib.call("hyb_cast_to_f32", dt, x.data_ptr(), out.data_ptr(), x.numel(), _stream())
    return out


def cast_fake(x, dt, to_t):
    return x.new_empty(x.shape, dtype=_TORCH_DTYPE[dt] if to_t else torch.float32)






def to_compute(x, dt):
    return torch.ops.hybrid.cast(x, dt, True)


def to_f32(x, dt):
    return torch.ops.hybrid.cast(x, dt, False)


# ---------------------------------------------------------------------------------------------
# conv stage: Conv3x3 -> BN -> ReLU -> MaxPool  (UNet.py:58-60, UNet.py:13)
# ---------------------------------------------------------------------------------------------
def _convstage_dims(x, weight, first):
    Co, Ci = weight.shape[0], weight.shape[1]
    Cop = pad_channels(Co)
    if first:
        N, _, H, W = x.shape
        Cip = 0
    else:
        N, H, W, Cip = x.shape
    return N, H, W, Ci, Cip, Co, Cop


def convstage_op(x: Tensor, weight: Tensor, gamma: Tensor, beta: Tensor, running_mean: Optional[Tensor], running_var: Optional[Tensor],
                 training: bool, momentum: float, eps: float, dt: int, first: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> (pooled, y_raw, scale_shift, mean_invstd, packed_bwd, running_out).  FUNCTIONAL, like aten's
    _native_batch_norm_legit_functional: the running statistics are read-only inputs and, in training mode, their updated values
    come back in running_out [2, Co] (empty in eval mode / without running statistics) for the module to commit to its buffers.
    y_raw .. packed_bwd are what the backward needs.  x: NCHW fp32 frames when `first`, else NHWC activations of the compute dtype
    with padded channels."""
    _require_cuda(x, weight)
    x = x.contiguous()
    N, H, W, Ci, Cip, Co, Cop = _convstage_dims(x, weight, first)
    if first and Ci > 4:
        raise RuntimeError("first-stage kernel supports in_channels <= 4")
    if H < 2 or W < 2:
        raise RuntimeError(f"conv stage needs H, W >= 2 (got {H}x{W})")
    dev = x.device
    tdt = _TORCH_DTYPE[dt]
    track = running_mean is not None and running_var is not None
    if not track:
        if not training:
            raise RuntimeError("eval-mode BatchNorm needs running statistics")
        running_mean, running_var = torch.zeros(Co, device=dev), torch.ones(Co, device=dev)    # track_running_stats=False: batch statistics only
    running_out = torch.empty((2, Co) if training else (0,), dtype=torch.float32, device=dev)
    # stage 1 recomputes its conv in backward: no full-resolution buffer is kept
    y_raw = torch.empty(0 if first else (N, H, W, Cop), dtype=tdt, device=dev)
    pooled = torch.empty(N, H // 2, W // 2, Cop, dtype=tdt, device=dev)
    scale_shift = torch.empty(2, Cop, dtype=torch.float32, device=dev)
    mean_invstd = torch.empty(2, Cop, dtype=torch.float32, device=dev)
    ws = _ws(_query("hyb_convstage_fwd_workspace", dt, int(first), Cip, Cop), dev)
    packed_bwd = torch.empty(_query("hyb_convstage_packed_bwd_elems", int(first), Cip, Cop), dtype=tdt, device=dev)
    lib.call("hyb_convstage_fwd", dt, int(first), x.data_ptr(), weight.contiguous().data_ptr(), gamma.contiguous().data_ptr(),
             beta.contiguous().data_ptr(), running_mean.contiguous().data_ptr(), running_var.contiguous().data_ptr(), None,
             int(training), float(momentum), float(eps), N, H, W, Ci, Cip, Co, Cop, None if first else y_raw.data_ptr(), pooled.data_ptr(),
             scale_shift.data_ptr(), mean_invstd.data_ptr(), packed_bwd.data_ptr(), running_out.data_ptr() if training else None,
             ws.data_ptr(), ws.numel(), _stream())
    if training and not track:
        running_out = running_out.new_empty((0,))
    return pooled, y_raw, scale_shift, mean_invstd, packed_bwd, running_out


def convstage_fake(x, weight, gamma, beta, running_mean, running_var, training, momentum, eps, dt, first):
    N, H, W, Ci, Cip, Co, Cop = _convstage_dims(x, weight, first)
    tdt = _TORCH_DTYPE[dt]
    track = running_mean is not None and running_var is not None
    return (x.new_empty((N, H // 2, W // 2, Cop), dtype=tdt),
            x.new_empty((0,) if first else (N, H, W, Cop), dtype=tdt),
            x.new_empty((2, Cop), dtype=torch.float32), x.new_empty((2, Cop), dtype=torch.float32),
            x.new_empty((_query("hyb_convstage_packed_bwd_elems", int(first), Cip, Cop),), dtype=tdt),
            x.new_empty((2, Co) if training and track else (0,), dtype=torch.float32))


def convstage_bwd_op(dpooled: Tensor, x: Tensor, y_raw: Tensor, pooled: Optional[Tensor], weight: Tensor, gamma: Tensor, scale_shift: Tensor,
                     mean_invstd: Tensor, packed_bwd: Tensor, training: bool, dt: int, first: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """-> (dx, dweight, dgamma, dbeta); dx is an empty placeholder for the first stage (the clip tensor gets no gradient)."""
    _require_cuda(dpooled, x)
    N, H, W, Ci, Cip, Co, Cop = _convstage_dims(x, weight, first)
    dev = x.device
    dpooled = dpooled.contiguous()
    dx = torch.empty(0, dtype=_TORCH_DTYPE[dt], device=dev) if first else torch.empty(N, H, W, Cip, dtype=_TORCH_DTYPE[dt], device=dev)
    dw = torch.empty_like(weight, memory_format=torch.contiguous_format)
    dgamma = torch.empty(Co, dtype=torch.float32, device=dev)
    dbeta = torch.empty(Co, dtype=torch.float32, device=dev)
    ws = _ws(_query("hyb_convstage_bwd_workspace", dt, int(first), N, H, W, Cip, Cop), dev)
    lib.call("hyb_convstage_bwd", dt, int(first), dpooled.data_ptr(), x.data_ptr(), None if first else y_raw.data_ptr(),
             None if (first or pooled is None) else pooled.contiguous().data_ptr(), weight.contiguous().data_ptr(),
             gamma.contiguous().data_ptr(), scale_shift.data_ptr(), mean_invstd.data_ptr(), int(training), N, H, W, Ci, Cip, Co, Cop,
             None if first else dx.data_ptr(), dw.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
             packed_bwd.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    return dx, dw, dgamma, dbeta


def convstage_bwd_fake(dpooled, x, y_raw, pooled, weight, gamma, scale_shift, mean_invstd, packed_bwd, training, dt, first):
    N, H, W, Ci, Cip, Co, Cop = _convstage_dims(x, weight, first)
    return (x.new_empty((0,) if first else (N, H, W, Cip), dtype=_TORCH_DTYPE[dt]), torch.empty_like(weight, memory_format=torch.contiguous_format),
            x.new_empty((Co,), dtype=torch.float32), x.new_empty((Co,), dtype=torch.float32))






def convstage(x, weight, gamma, beta, running_mean, running_var, num_batches_tracked, training, momentum, eps, dt, first, commit=None):
    """nn.BatchNorm2d semantics on top of the functional operator: in training mode the updated running statistics are written
    back to the module's buffers and num_batches_tracked advances.  `commit` (a list) defers that write-back: the caller collects
    (running_mean, running_var, num_batches_tracked, running_out) of several stages and commits them with ONE multi-tensor copy
    (commit_running_stats) instead of three small launches per stage."""
    out = torch.ops.hybrid.convstage(x, weight, gamma, beta, running_mean, running_var, bool(training), float(momentum), float(eps),
                                     int(dt), bool(first))
    if training and running_mean is not None and running_var is not None:
        entry = (running_mean, running_var, num_batches_tracked, out[5])
        if commit is None:
            commit_running_stats([entry])
        else:
            commit.append(entry)
    return out[0]


@torch.no_grad()
def commit_running_stats(entries):
    dst, src, nbts = [], [], []
    for rm, rv, nbt, ro in entries:
        dst += [rm, rv]
        src += [ro[0], ro[1]]
        if nbt is not None:
            nbts.append(nbt)
    if dst:
        torch._foreach_copy_(dst, src)
    if nbts:
        torch._foreach_add_(nbts, 1)


# ---------------------------------------------------------------------------------------------
# frame token: global average pool + Linear(C, d)
# ---------------------------------------------------------------------------------------------
def token_op(x: Tensor, weight: Tensor, bias: Optional[Tensor], dt: int) -> Tuple[Tensor, Tensor]:
    """x [N,Hh,Ww,Cp] -> (tok [N,d], feat [N,Cp] saved)."""
    _require_cuda(x, weight)
    x = x.contiguous()
    N, Hh, Ww, Cp = x.shape
    d, C = weight.shape
    dev = x.device
    feat = torch.empty(N, Cp, dtype=_TORCH_DTYPE[dt], device=dev)
    tok = torch.empty(N, d, dtype=_TORCH_DTYPE[dt], device=dev)
    st = _stream()
    lib.call("hyb_gap_fwd", dt, x.data_ptr(), feat.data_ptr(), N, Hh * Ww, Cp, st)
    lib.call("hyb_linear_fwd", dt, feat.data_ptr(), Cp, weight.contiguous().data_ptr(), bias.contiguous().data_ptr() if bias is not None else None,
             tok.data_ptr(), N, d, C, 0, st)
    return tok, feat


def token_fake(x, weight, bias, dt):
    N, Hh, Ww, Cp = x.shape
    return x.new_empty((N, weight.shape[0]), dtype=_TORCH_DTYPE[dt]), x.new_empty((N, Cp), dtype=_TORCH_DTYPE[dt])


def token_bwd_op(dtok: Tensor, feat: Tensor, weight: Tensor, Hh: int, Ww: int, has_bias: bool, dt: int) -> Tuple[Tensor, Tensor, Tensor]:
    _require_cuda(dtok, feat)
    N, Cp = feat.shape
    d, C = weight.shape
    dev = feat.device
    dtok = dtok.contiguous()
    dfeat = torch.zeros(N, Cp, dtype=_TORCH_DTYPE[dt], device=dev)     # padded channels stay zero
    dw = torch.empty_like(weight, memory_format=torch.contiguous_format)
    db = torch.empty(d if has_bias else 0, dtype=torch.float32, device=dev)
    st = _stream()
    lib.call("hyb_linear_bwd", dt, feat.data_ptr(), Cp, weight.contiguous().data_ptr(), None, dtok.data_ptr(), dfeat.data_ptr(), 0, dw.data_ptr(),
             db.data_ptr() if has_bias else None, N, d, C, 0, None, 0, st)
    dx = torch.empty(N, Hh, Ww, Cp, dtype=_TORCH_DTYPE[dt], device=dev)
    lib.call("hyb_gap_bwd", dt, dfeat.data_ptr(), dx.data_ptr(), N, Hh * Ww, Cp, st)
    return dx, dw, db


def token_bwd_fake(dtok, feat, weight, Hh, Ww, has_bias, dt):
    N, Cp = feat.shape
    return (feat.new_empty((N, Hh, Ww, Cp)), torch.empty_like(weight, memory_format=torch.contiguous_format),
            feat.new_empty((weight.shape[0] if has_bias else 0,), dtype=torch.float32))






def token(x, weight, bias, dt):
    return torch.ops.hybrid.token(x, weight, bias, dt)[0]


# ---------------------------------------------------------------------------------------------
# TransformerEncoder.forward (all layers) -- TransformerEncoder.pyc src L110-126
# ---------------------------------------------------------------------------------------------
def _check_attention_limits(S, D, H):
    # (no limit on S: up to 64 tokens per clip the register-resident kernels, beyond that the online-softmax kernels -- hyb_attention_long_*;
    # the reference's attention(), TransformerEncoder.pyc src L49-62, has none either)
    if D % H != 0 or (D // H) % 8 != 0 or D // H > 128:
        raise RuntimeError(f"temporal attention kernel supports head widths that are multiples of 8 up to 128 (got {D}/{H})")


def encoder_op(x: Tensor, mask: Optional[Tensor], params: Sequence[Tensor], dt: int, hid: int, L: int, H: int, attn_p: float,
               layer_p: float, seed: int, seed_inc: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """x [B,S,D] T, mask fp32 [B,S,S] or None, params = L*14 fp32 tensors (order: include/hybrid_hip.h) -> (out, saved blob)."""
    _require_cuda(x, *params)
    x = x.contiguous()
    B, S, D = x.shape
    _check_attention_limits(S, D, H)
    dev = x.device
    ps = [p.contiguous() for p in params]
    saved = _ws(_query("hyb_encoder_saved_bytes", dt, B, S, D, hid, L, H), dev)
    out = torch.empty(B, S, D, dtype=_TORCH_DTYPE[dt], device=dev)
    lib.call("hyb_encoder_fwd", dt, x.data_ptr(), _opt_ptr(mask), ptr_array([p.data_ptr() for p in ps]), out.data_ptr(), saved.data_ptr(),
             B, S, D, hid, L, H, float(attn_p), float(layer_p), seed, _opt_ptr(seed_inc), _stream())
    return out, saved


def encoder_fake(x, mask, params, dt, hid, L, H, attn_p, layer_p, seed, seed_inc=None):
    B, S, D = x.shape
    return torch.empty_like(x, memory_format=torch.contiguous_format), x.new_empty((max(_query("hyb_encoder_saved_bytes", dt, B, S, D, hid, L, H), 256),),
                                                                                    dtype=torch.uint8)


def encoder_bwd_op(dout: Tensor, mask: Optional[Tensor], params: Sequence[Tensor], saved: Tensor, dt: int, hid: int, L: int, H: int,
                   attn_p: float, layer_p: float, seed: int, seed_inc: Optional[Tensor] = None) -> List[Tensor]:
    """-> [dx, dparam_0, ..., dparam_{14L-1}]"""
    _require_cuda(dout, saved)
    dout = dout.contiguous()
    B, S, D = dout.shape
    dev = dout.device
    ps = [p.contiguous() for p in params]
    grads = [torch.empty_like(p) for p in ps]
    dx = torch.empty(B, S, D, dtype=_TORCH_DTYPE[dt], device=dev)
    ws = _ws(_query("hyb_encoder_workspace_bytes", dt, B, S, D, hid, L, H), dev)
    lib.call("hyb_encoder_bwd", dt, dout.data_ptr(), _opt_ptr(mask), ptr_array([p.data_ptr() for p in ps]),
             ptr_array([g.data_ptr() for g in grads]), saved.data_ptr(), dx.data_ptr(), B, S, D, hid, L, H, float(attn_p),
             float(layer_p), seed, _opt_ptr(seed_inc), ws.data_ptr(), ws.numel(), _stream())
    return [dx] + grads


def encoder_bwd_fake(dout, mask, params, saved, dt, hid, L, H, attn_p, layer_p, seed, seed_inc=None):
    return [torch.empty_like(dout, memory_format=torch.contiguous_format)] + [torch.empty_like(p, memory_format=torch.contiguous_format) for p in params]






def encoder(x, mask, params, dt, hid, L, H, attn_p, layer_p, seed):
    B, S, _ = x.shape
    return torch.ops.hybrid.encoder(x, check_mask(mask, B, S, x.device), list(params), dt, hid, L, H, float(attn_p), float(layer_p), seed,
                                    step_counter())[0]


# ---------------------------------------------------------------------------------------------
# standalone MultiheadAttention.forward(q, k, v, mask) -- src L67-89
# ---------------------------------------------------------------------------------------------
def mha_op(q_in: Tensor, k_in: Tensor, v_in: Tensor, mask: Optional[Tensor], params: Sequence[Tensor], dt: int, H: int, p_drop: float,
           seed: int) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """params = Wq,bq,Wk,bk,Wv,bv,Wo,bo (fp32) -> (out, q, k, v, a, probs); all but `out` saved for backward."""
    _require_cuda(q_in, *params)
    q_in, k_in, v_in = q_in.contiguous(), k_in.contiguous(), v_in.contiguous()
    B, S, D = q_in.shape
    _check_attention_limits(S, D, H)
    dev, tdt, M = q_in.device, _TORCH_DTYPE[dt], B * S
    ps = [t.contiguous() for t in params]
    q, k, v, a, out = (torch.empty(B, S, D, dtype=tdt, device=dev) for _ in range(5))
    probs = torch.empty(B * H, S, 2, dtype=torch.float32, device=dev)       # softmax row statistics (max, sum)
    st = _stream()
    for src, W_, b_, dst in ((q_in, ps[0], ps[1], q), (k_in, ps[2], ps[3], k), (v_in, ps[4], ps[5], v)):
        lib.call("hyb_linear_fwd", dt, src.data_ptr(), D, W_.data_ptr(), b_.data_ptr(), dst.data_ptr(), M, D, D, 1, st)
    if S > 64:           # probs then holds the log-sum-exp per (clip, head, query) in its first B*H*S floats
        ws = _ws(_query("hyb_attention_long_workspace", dt, B, S, D, H), dev)
        lib.call("hyb_attention_long_fwd", dt, q.data_ptr(), k.data_ptr(), v.data_ptr(), D, _opt_ptr(mask), a.data_ptr(), probs.data_ptr(), B, S, D, H,
                 float(p_drop), seed, None, ws.data_ptr(), ws.numel(), st)
    else:
        lib.call("hyb_attention_fwd", dt, q.data_ptr(), k.data_ptr(), v.data_ptr(), _opt_ptr(mask), a.data_ptr(),
                 probs.data_ptr(), B, S, D, H, float(p_drop), seed, st)
    lib.call("hyb_linear_fwd", dt, a.data_ptr(), D, ps[6].data_ptr(), ps[7].data_ptr(), out.data_ptr(), M, D, D, 0, st)
    return out, q, k, v, a, probs


def mha_fake(q_in, k_in, v_in, mask, params, dt, H, p_drop, seed):
    B, S, D = q_in.shape
    e = lambda: q_in.new_empty((B, S, D), dtype=_TORCH_DTYPE[dt])
    return e(), e(), e(), e(), e(), q_in.new_empty((B * H, S, 2), dtype=torch.float32)


def mha_bwd_op(dout: Tensor, q_in: Tensor, k_in: Tensor, v_in: Tensor, mask: Optional[Tensor], q: Tensor, k: Tensor, v: Tensor, a: Tensor,
               probs: Tensor, params: Sequence[Tensor], dt: int, H: int, p_drop: float, seed: int) -> List[Tensor]:
    """-> [dq_in, dk_in, dv_in, dWq, dbq, dWk, dbk, dWv, dbv, dWo, dbo]"""
    _require_cuda(dout, q)
    B, S, D = q.shape
    dev, tdt, M = q.device, _TORCH_DTYPE[dt], B * S
    dout = dout.contiguous()
    ps = [t.contiguous() for t in params]
    st = _stream()
    grads = [torch.empty_like(p) for p in ps]
    da, dq, dk, dv, dqi, dki, dvi = (torch.empty(B, S, D, dtype=tdt, device=dev) for _ in range(7))
    ws = _ws(M * D * 4, dev)
    lib.call("hyb_linear_bwd", dt, a.data_ptr(), D, ps[6].data_ptr(), None, dout.data_ptr(), da.data_ptr(), 0, grads[6].data_ptr(),
             grads[7].data_ptr(), M, D, D, 0, None, 0, st)
    if S > 64:
        lws = _ws(_query("hyb_attention_long_workspace", dt, B, S, D, H), dev)
        lib.call("hyb_attention_long_bwd", dt, q.data_ptr(), k.data_ptr(), v.data_ptr(), D, _opt_ptr(mask), a.data_ptr(), probs.data_ptr(), da.data_ptr(),
                 dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), D, B, S, D, H, float(p_drop), seed, None, lws.data_ptr(), lws.numel(), st)
    else:
        lib.call("hyb_attention_bwd", dt, q.data_ptr(), k.data_ptr(), v.data_ptr(), _opt_ptr(mask), probs.data_ptr(), da.data_ptr(), dq.data_ptr(),
                 dk.data_ptr(), dv.data_ptr(), B, S, D, H, float(p_drop), seed, st)
    for src, y, dy, dsrc, iw in ((q_in, q, dq, dqi, 0), (k_in, k, dk, dki, 2), (v_in, v, dv, dvi, 4)):
        lib.call("hyb_linear_bwd", dt, src.contiguous().data_ptr(), D, ps[iw].data_ptr(), y.data_ptr(), dy.data_ptr(), dsrc.data_ptr(), 0,
                 grads[iw].data_ptr(), grads[iw + 1].data_ptr(), M, D, D, 1, ws.data_ptr(), ws.numel(), st)
    return [dqi, dki, dvi] + grads


def mha_bwd_fake(dout, q_in, k_in, v_in, mask, q, k, v, a, probs, params, dt, H, p_drop, seed):
    return [torch.empty_like(q) for _ in range(3)] + [torch.empty_like(p, memory_format=torch.contiguous_format) for p in params]






def mha(q, k, v, mask, params, dt, H, p_drop, seed):
    B, S, _ = q.shape
    return torch.ops.hybrid.mha(q, k, v, check_mask(mask, B, S, q.device), list(params), dt, H, float(p_drop), seed)[0]


# ---------------------------------------------------------------------------------------------
# head (mean over T + Linear) and cross-entropy
# ---------------------------------------------------------------------------------------------
def head_op(x: Tensor, weight: Tensor, bias: Optional[Tensor], dt: int) -> Tensor:
    _require_cuda(x, weight)
    x = x.contiguous()
    B, S, D = x.shape
    C = weight.shape[0]
    logits = torch.empty(B, C, dtype=torch.float32, device=x.device)
    lib.call("hyb_head_fwd", dt, x.data_ptr(), weight.contiguous().data_ptr(), bias.contiguous().data_ptr() if bias is not None else None,
             logits.data_ptr(), B, S, D, C, _stream())
    return logits


def head_fake(x, weight, bias, dt):
    return x.new_empty((x.shape[0], weight.shape[0]), dtype=torch.float32)


def head_bwd_op(dlogits: Tensor, x: Tensor, weight: Tensor, has_bias: bool, dt: int) -> Tuple[Tensor, Tensor, Tensor]:
    _require_cuda(dlogits, x)
    B, S, D = x.shape
    C = weight.shape[0]
    dlogits = dlogits.contiguous().float()
    dx = torch.empty_like(x)
    dw = torch.empty_like(weight, memory_format=torch.contiguous_format)
    db = torch.empty(C if has_bias else 0, dtype=torch.float32, device=x.device)
    lib.call("hyb_head_bwd", dt, x.data_ptr(), weight.contiguous().data_ptr(), dlogits.data_ptr(), dx.data_ptr(), dw.data_ptr(),
             db.data_ptr() if has_bias else None, B, S, D, C, _stream())
    return dx, dw, db


def head_bwd_fake(dlogits, x, weight, has_bias, dt):
    return (torch.empty_like(x), torch.empty_like(weight, memory_format=torch.contiguous_format),
            x.new_empty((weight.shape[0] if has_bias else 0,), dtype=torch.float32))






def head(x, weight, bias, dt):
    return torch.ops.hybrid.head(x, weight, bias, dt)


def cross_entropy_op(logits: Tensor, target: Tensor) -> Tensor:
    _require_cuda(logits, target)
    if logits.dim() != 2 or target.dim() != 1 or target.shape[0] != logits.shape[0]:
        raise ValueError(f"expected logits [B,C] and class indices [B], got {tuple(logits.shape)} and {tuple(target.shape)}")
    logits = logits.contiguous().float()
    target = target.contiguous().to(torch.int64)
    B, C = logits.shape
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    lib.call("hyb_cross_entropy_fwd", logits.data_ptr(), target.data_ptr(), loss.data_ptr(), B, C, _stream())
    return loss


def cross_entropy_fake(logits, target):
    return logits.new_empty((), dtype=torch.float32)


def cross_entropy_bwd_op(dloss: Tensor, logits: Tensor, target: Tensor) -> Tensor:
    _require_cuda(dloss, logits)
    logits = logits.contiguous().float()
    target = target.contiguous().to(torch.int64)
    B, C = logits.shape
    dl = dloss.contiguous().float().reshape(1)
    dlogits = torch.empty_like(logits)
    lib.call("hyb_cross_entropy_bwd", logits.data_ptr(), target.data_ptr(), dl.data_ptr(), dlogits.data_ptr(), B, C, _stream())
    return dlogits


def cross_entropy_bwd_fake(dloss, logits, target):
    return logits.new_empty(logits.shape, dtype=torch.float32)






def cross_entropy(logits, target):
    return torch.ops.hybrid.cross_entropy(logits, target)


# ---------------------------------------------------------------------------------------------
# model-level operators: the whole CNN backbone / the whole temporal part in one call each way (hyb_backbone_*, hyb_temporal_*).
# Same kernels as the stage operators above, chained in C: a training step is three operator calls each way, which keeps the
# host (Python dispatch ~50 us per operator call) off the critical path.
# ---------------------------------------------------------------------------------------------
import ctypes as _ct


def _int_array(vals):
    return (_ct.c_int * len(vals))(*vals)


def _backbone_geometry(x, weights):
    N, Ci, H, W = x.shape
    chans = [Ci] + [w.shape[0] for w in weights]
    dims = []
    h, w_ = H, W
    for s in range(len(weights)):
        dims.append((h, w_, 0 if s == 0 else pad_channels(chans[s]), pad_channels(chans[s + 1])))
        h, w_ = h // 2, w_ // 2
    return N, H, W, chans, dims


def backbone_op(x: Tensor, weights: Sequence[Tensor], gammas: Sequence[Tensor], betas: Sequence[Tensor], running_means: Sequence[Tensor],
                running_vars: Sequence[Tensor], training: bool, momentum: float, eps: float, dt: int) -> List[Tensor]:
    """All conv stages of the backbone on NCHW fp32 frames x [N, C_in <= 4, H, W].
    -> [pooled_S, then per stage s: y_raw_s, pooled_s (s < S-1 only), scale_shift_s, mean_invstd_s, packed_bwd_s, running_out_s]
    (functional BatchNorm: running_out_s [2, C_s] holds the updated statistics in training mode, empty otherwise)."""
    return _backbone_impl(x, weights, gammas, betas, running_means, running_vars, None, training, momentum, eps, dt)


def backbone_inplace_op(x: Tensor, weights: Sequence[Tensor], gammas: Sequence[Tensor], betas: Sequence[Tensor], running_means: Sequence[Tensor],
                        running_vars: Sequence[Tensor], num_batches_tracked: Sequence[Tensor], training: bool, momentum: float, eps: float,
                        dt: int) -> List[Tensor]:
    """hybrid::backbone with nn.BatchNorm2d's own buffer semantics: in training mode running_means / running_vars are updated in place by
    the statistics kernels and every num_batches_tracked advances by one (no write-back launches); same outputs, running_out_s empty."""
    return _backbone_impl(x, weights, gammas, betas, running_means, running_vars, list(num_batches_tracked), training, momentum, eps, dt)


def _backbone_impl(x, weights, gammas, betas, running_means, running_vars, nbts, training, momentum, eps, dt):
    inplace = nbts is not None
    _require_cuda(x, *weights)
    if inplace:
        for t in list(running_means) + list(running_vars):
            if not (t.is_contiguous() and t.dtype == torch.float32):
                raise RuntimeError("hybrid::backbone_ updates contiguous fp32 running statistics in place")
        for t in nbts:
            if not (t.is_cuda and t.dtype == torch.int64 and t.numel() == 1):
                raise RuntimeError("hybrid::backbone_: num_batches_tracked must be cuda int64 scalars")
    x = x.contiguous()
    S = len(weights)
    N, H, W, chans, dims = _backbone_geometry(x, weights)
    if chans[0] > 4:
        raise RuntimeError("hybrid::backbone reads NCHW frames with C_in <= 4")
    if dims[-1][0] < 2 or dims[-1][1] < 2:
        raise RuntimeError(f"frames of {H}x{W} are too small for {S} conv stages (each needs H, W >= 2)")
    dev, tdt = x.device, _TORCH_DTYPE[dt]
    per_stage, pooled = [], None
    params, outs = [], []
    for s in range(S):
        h, w_, Cip, Cop = dims[s]
        Co = chans[s + 1]
        # (stage 1 keeps no raw conv output: its slot carries the pooling / ReLU routing codes for the backward, where the path writes them)
        y_raw = torch.empty(_query("hyb_convstage_route_elems", dt, N, h, w_, Cop) if s == 0 else (N, h, w_, Cop), dtype=tdt, device=dev)
        pooled = torch.empty(N, h // 2, w_ // 2, Cop, dtype=tdt, device=dev)
        ss = torch.empty(2, Cop, dtype=torch.float32, device=dev)
        mi = torch.empty(2, Cop, dtype=torch.float32, device=dev)
        pk = torch.empty(_query("hyb_convstage_packed_bwd_elems", int(s == 0), Cip, Cop), dtype=tdt, device=dev)
        ro = torch.empty((2, Co) if training and not inplace else (0,), dtype=torch.float32, device=dev)
        per_stage.append((y_raw, pooled, ss, mi, pk, ro))
        params += [weights[s].contiguous().data_ptr(), gammas[s].contiguous().data_ptr(), betas[s].contiguous().data_ptr(),
                   running_means[s].contiguous().data_ptr(), running_vars[s].contiguous().data_ptr()]
        outs += [y_raw.data_ptr() if y_raw.numel() else None, pooled.data_ptr(), ss.data_ptr(), mi.data_ptr(), pk.data_ptr(),
                 ro.data_ptr() if training and not inplace else None]
    ch = _int_array(chans)
    ws = _ws(_query("hyb_backbone_fwd_workspace", dt, S, tuple(chans)), dev)
    lib.call("hyb_backbone_fwd", dt, S, ch, x.data_ptr(), ptr_array(params), ptr_array([t.data_ptr() for t in nbts]) if inplace else None,
             int(training), float(momentum), float(eps), N, H, W, ptr_array(outs), ws.data_ptr(), ws.numel(), _stream())
    res = [pooled]
    for s, (y_raw, p, ss, mi, pk, ro) in enumerate(per_stage):
        res += [y_raw] + ([p] if s < S - 1 else []) + [ss, mi, pk, ro]
    return res


def backbone_inplace_fake(x, weights, gammas, betas, running_means, running_vars, num_batches_tracked, training, momentum, eps, dt):
    return backbone_fake(x, weights, gammas, betas, running_means, running_vars, training, momentum, eps, dt, functional=False)


def backbone_fake(x, weights, gammas, betas, running_means, running_vars, training, momentum, eps, dt, functional=True):
    S = len(weights)
    N, H, W, chans, dims = _backbone_geometry(x, weights)
    tdt = _TORCH_DTYPE[dt]
    res, last = [], None
    for s in range(S):
        h, w_, Cip, Cop = dims[s]
        last = x.new_empty((N, h // 2, w_ // 2, Cop), dtype=tdt)
        res += [x.new_empty((_query("hyb_convstage_route_elems", dt, N, h, w_, Cop),) if s == 0 else (N, h, w_, Cop), dtype=tdt)] + ([last] if s < S - 1 else []) + [
            x.new_empty((2, Cop), dtype=torch.float32), x.new_empty((2, Cop), dtype=torch.float32),
            x.new_empty((_query("hyb_convstage_packed_bwd_elems", int(s == 0), Cip, Cop),), dtype=tdt),
            x.new_empty((2, chans[s + 1]) if training and functional else (0,), dtype=torch.float32)]
    return [last] + res


def _backbone_unpack(res, S):
    """[pooled_S, ...] -> per stage (y_raw, pooled, scale_shift, mean_invstd, packed_bwd, running_out)."""
    out, i = [], 1
    for s in range(S):
        y_raw = res[i]; i += 1
        if s < S - 1:
            p = res[i]; i += 1
        else:
            p = res[0]
        out.append((y_raw, p, res[i], res[i + 1], res[i + 2], res[i + 3]))
        i += 4
    return out


def backbone_bwd_op(dpooled: Tensor, pooled: Tensor, x: Tensor, weights: Sequence[Tensor], gammas: Sequence[Tensor], saved: Sequence[Tensor],
                    training: bool, dt: int) -> List[Tensor]:
    """saved: per stage (y_raw, stage input [placeholder for stage 0], scale_shift, mean_invstd, packed_bwd).
    -> per stage (dweight, dgamma, dbeta), flattened."""
    _require_cuda(dpooled, x)
    S = len(weights)
    N, H, W, chans, dims = _backbone_geometry(x, weights)
    dev = x.device
    dpooled = dpooled.contiguous()
    grads, gptr, pptr, sptr = [], [], [], []
    for s in range(S):
        dw = torch.empty_like(weights[s], memory_format=torch.contiguous_format)
        dg = torch.empty(chans[s + 1], dtype=torch.float32, device=dev)
        db = torch.empty(chans[s + 1], dtype=torch.float32, device=dev)
        grads += [dw, dg, db]
        gptr += [dw.data_ptr(), dg.data_ptr(), db.data_ptr()]
        pptr += [weights[s].contiguous().data_ptr(), gammas[s].contiguous().data_ptr()]
        sv = saved[5 * s:5 * s + 5]
        sptr += [sv[0].data_ptr() if sv[0].numel() else None, None if s == 0 else sv[1].data_ptr(), sv[2].data_ptr(), sv[3].data_ptr(), sv[4].data_ptr()]
    ws = _ws(_query("hyb_backbone_bwd_workspace", dt, S, tuple(chans), N, H, W), dev)
    lib.call("hyb_backbone_bwd", dt, S, _int_array(chans), dpooled.data_ptr(), pooled.contiguous().data_ptr(), x.data_ptr(), ptr_array(pptr), ptr_array(sptr), int(training),
             N, H, W, ptr_array(gptr), ws.data_ptr(), ws.numel(), _stream())
    return grads


def backbone_bwd_fake(dpooled, pooled, x, weights, gammas, saved, training, dt):
    res = []
    for w in weights:
        res += [torch.empty_like(w, memory_format=torch.contiguous_format), x.new_empty((w.shape[0],), dtype=torch.float32),
                x.new_empty((w.shape[0],), dtype=torch.float32)]
    return res






def backbone(x, stages, training, dt):
    """stages: list of (conv.weight, bn) pairs.  Applies nn.BatchNorm2d's buffer updates after the functional operator."""
    bns = [bn for _, bn in stages]
    if training and all(bn.num_batches_tracked is not None and bn.num_batches_tracked.is_cuda and bn.running_mean.is_contiguous()
                        and bn.running_var.is_contiguous() for bn in bns):
        # the statistics kernels update the modules' buffers themselves (nn.BatchNorm2d's in-place semantics): no write-back launches
        return torch.ops.hybrid.backbone_(x, [w for w, _ in stages], [bn.weight for bn in bns], [bn.bias for bn in bns],
                                          [bn.running_mean for bn in bns], [bn.running_var for bn in bns],
                                          [bn.num_batches_tracked for bn in bns], True, float(bns[0].momentum), float(bns[0].eps), int(dt))[0]
    res = torch.ops.hybrid.backbone(x, [w for w, _ in stages], [bn.weight for bn in bns], [bn.bias for bn in bns],
                                    [bn.running_mean for bn in bns], [bn.running_var for bn in bns], bool(training), float(bns[0].momentum),
                                    float(bns[0].eps), int(dt))
    if training:
        st = _backbone_unpack(res, len(stages))
        commit_running_stats([(bn.running_mean, bn.running_var, bn.num_batches_tracked, st[s][5]) for s, bn in enumerate(bns)])
    return res[0]


def _check_h_dtype(h, dt):
    """dt | HYB_H_BF16 (fp32 / bf16x3 temporal part behind bf16 conv stages): the pooled map is bf16, the global-average-pool kernels convert."""
    want = torch.bfloat16 if dt & HYB_H_BF16 else _TORCH_DTYPE[dt & 0xff]
    if dt & HYB_H_BF16 and _TORCH_DTYPE[dt & 0xff] != torch.float32:
        raise ValueError("HYB_H_BF16 goes with an fp32-storage temporal part ('fp32' / 'bf16x3')")
    if h.dtype != want:
        raise TypeError(f"the pooled map must be {want} for this compute dtype, got {h.dtype}")


def temporal_op(h: Tensor, token_w: Tensor, token_b: Tensor, enc_params: Sequence[Tensor], head_w: Tensor, head_b: Tensor, mask: Optional[Tensor],
                B: int, dt: int, hid: int, L: int, H: int, attn_p: float, layer_p: float, seed: int,
                seed_inc: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """h [B*S, Hh, Ww, Cp] T (last pooled map) -> (logits [B, classes] fp32, feat, enc_saved, enc_out); the last three are saved for backward."""
    _require_cuda(h, token_w, head_w, *enc_params)
    _check_h_dtype(h, dt)
    h = h.contiguous()
    N, Hh, Ww, Cp = h.shape
    S = N // B
    D, C = token_w.shape
    classes = head_w.shape[0]
    _check_attention_limits(S, D, H)
    dev, tdt = h.device, _TORCH_DTYPE[dt & 0xff]
    feat = torch.empty(N, Cp, dtype=tdt, device=dev)
    tok = torch.empty(B, S, D, dtype=tdt, device=dev)
    enc_out = torch.empty(B, S, D, dtype=tdt, device=dev)
    saved = _ws(_query("hyb_encoder_saved_bytes", dt & 0xff, B, S, D, hid, L, H), dev)
    logits = torch.empty(B, classes, dtype=torch.float32, device=dev)
    ps = [p.contiguous() for p in enc_params]
    lib.call("hyb_temporal_fwd", dt, h.data_ptr(), token_w.contiguous().data_ptr(), token_b.contiguous().data_ptr(), ptr_array([p.data_ptr() for p in ps]),
             head_w.contiguous().data_ptr(), head_b.contiguous().data_ptr(), _opt_ptr(mask), feat.data_ptr(), tok.data_ptr(), saved.data_ptr(),
             enc_out.data_ptr(), logits.data_ptr(), B, S, Hh * Ww, C, Cp, D, hid, L, H, classes, float(attn_p), float(layer_p), seed,
             _opt_ptr(seed_inc), _stream())
    return logits, feat, saved, enc_out


def temporal_fake(h, token_w, token_b, enc_params, head_w, head_b, mask, B, dt, hid, L, H, attn_p, layer_p, seed, seed_inc=None):
    N, Hh, Ww, Cp = h.shape
    S, D = N // B, token_w.shape[0]
    tdt = _TORCH_DTYPE[dt & 0xff]
    return (h.new_empty((B, head_w.shape[0]), dtype=torch.float32), h.new_empty((N, Cp), dtype=tdt),
            h.new_empty((max(_query("hyb_encoder_saved_bytes", dt & 0xff, B, S, D, hid, L, H), 256),), dtype=torch.uint8), h.new_empty((B, S, D), dtype=tdt))


def temporal_bwd_op(dlogits: Tensor, token_w: Tensor, enc_params: Sequence[Tensor], head_w: Tensor, mask: Optional[Tensor], feat: Tensor,
                    saved: Tensor, enc_out: Tensor, Hh: int, Ww: int, dt: int, hid: int, L: int, H: int, attn_p: float, layer_p: float,
                    seed: int, seed_inc: Optional[Tensor] = None) -> List[Tensor]:
    """-> [dh, dtoken_w, dtoken_b, dhead_w, dhead_b, denc_param_0, ...]"""
    _require_cuda(dlogits, feat)
    B, S, D = enc_out.shape
    N, Cp = feat.shape
    C = token_w.shape[1]
    classes = head_w.shape[0]
    dev, tdt = feat.device, _TORCH_DTYPE[dt & 0xff]
    dlogits = dlogits.contiguous().float()
    ps = [p.contiguous() for p in enc_params]
    grads = [torch.empty_like(p) for p in ps]
    dh = torch.empty(N, Hh, Ww, Cp, dtype=torch.bfloat16 if dt & HYB_H_BF16 else tdt, device=dev)
    dtw = torch.empty_like(token_w, memory_format=torch.contiguous_format)
    dtb = torch.empty(D, dtype=torch.float32, device=dev)
    dhw = torch.empty_like(head_w, memory_format=torch.contiguous_format)
    dhb = torch.empty(classes, dtype=torch.float32, device=dev)
    ws = _ws(_query("hyb_temporal_bwd_workspace", dt & 0xff, B, S, Hh * Ww, Cp, D, hid, L, H), dev)
    lib.call("hyb_temporal_bwd", dt, dlogits.data_ptr(), token_w.contiguous().data_ptr(), ptr_array([p.data_ptr() for p in ps]),
             head_w.contiguous().data_ptr(), _opt_ptr(mask), feat.data_ptr(), saved.data_ptr(), enc_out.data_ptr(), dtw.data_ptr(), dtb.data_ptr(),
             ptr_array([g.data_ptr() for g in grads]), dhw.data_ptr(), dhb.data_ptr(), dh.data_ptr(), B, S, Hh * Ww, C, Cp, D, hid, L, H, classes,
             float(attn_p), float(layer_p), seed, _opt_ptr(seed_inc), ws.data_ptr(), ws.numel(), _stream())
    return [dh, dtw, dtb, dhw, dhb] + grads


def temporal_bwd_fake(dlogits, token_w, enc_params, head_w, mask, feat, saved, enc_out, Hh, Ww, dt, hid, L, H, attn_p, layer_p, seed, seed_inc=None):
    N, Cp = feat.shape
    c = lambda t: torch.empty_like(t, memory_format=torch.contiguous_format)
    return [feat.new_empty((N, Hh, Ww, Cp), dtype=torch.bfloat16 if dt & HYB_H_BF16 else feat.dtype), c(token_w), feat.new_empty((token_w.shape[0],), dtype=torch.float32), c(head_w),
            feat.new_empty((head_w.shape[0],), dtype=torch.float32)] + [c(p) for p in enc_params]






_CE_SCRATCH = {}


def _ce_scratch(B, device):
    """The B + 1 floats hyb_temporal_ce_fwd keeps between its workgroups (per-clip loss terms + a ticket word that every call leaves zero):
    one zero-initialised buffer per (device, stream, B), so calls that share it are stream-ordered."""
    key = (device.index, _stream(), B)
    t = _CE_SCRATCH.get(key)
    if t is None:
        t = _CE_SCRATCH[key] = torch.zeros(B + 1, dtype=torch.float32, device=device)
    return t


def temporal_ce_op(h: Tensor, token_w: Tensor, token_b: Tensor, enc_params: Sequence[Tensor], head_w: Tensor, head_b: Tensor, mask: Optional[Tensor],
                   target: Tensor, B: int, dt: int, hid: int, L: int, H: int, attn_p: float, layer_p: float, seed: int,
                   seed_inc: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """hybrid::temporal + hybrid::cross_entropy in the same launches (hyb_temporal_ce_fwd): -> (loss [], logits, feat, enc_saved, enc_out)."""
    _require_cuda(h, token_w, head_w, target, *enc_params)
    _check_h_dtype(h, dt)
    h = h.contiguous()
    N, Hh, Ww, Cp = h.shape
    S = N // B
    D, C = token_w.shape
    classes = head_w.shape[0]
    _check_attention_limits(S, D, H)
    if target.dim() != 1 or target.shape[0] != B:
        raise ValueError(f"expected class indices [B={B}], got {tuple(target.shape)}")
    target = target.contiguous().to(torch.int64)
    dev, tdt = h.device, _TORCH_DTYPE[dt & 0xff]
    feat = torch.empty(N, Cp, dtype=tdt, device=dev)
    tok = torch.empty(B, S, D, dtype=tdt, device=dev)
    enc_out = torch.empty(B, S, D, dtype=tdt, device=dev)
    saved = _ws(_query("hyb_encoder_saved_bytes", dt & 0xff, B, S, D, hid, L, H), dev)
    logits = torch.empty(B, classes, dtype=torch.float32, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    ps = [p.contiguous() for p in enc_params]
    lib.call("hyb_temporal_ce_fwd", dt, h.data_ptr(), token_w.contiguous().data_ptr(), token_b.contiguous().data_ptr(), ptr_array([p.data_ptr() for p in ps]),
             head_w.contiguous().data_ptr(), head_b.contiguous().data_ptr(), _opt_ptr(mask), target.data_ptr(), feat.data_ptr(), tok.data_ptr(),
             saved.data_ptr(), enc_out.data_ptr(), logits.data_ptr(), loss.data_ptr(), _ce_scratch(B, dev).data_ptr(), B, S, Hh * Ww, C, Cp, D, hid, L, H,
             classes, float(attn_p), float(layer_p), seed, _opt_ptr(seed_inc), _stream())
    return loss, logits, feat, saved, enc_out


def temporal_ce_fake(h, token_w, token_b, enc_params, head_w, head_b, mask, target, B, dt, hid, L, H, attn_p, layer_p, seed, seed_inc=None):
    return (h.new_empty((), dtype=torch.float32),) + temporal_fake(h, token_w, token_b, enc_params, head_w, head_b, mask, B, dt, hid, L, H, attn_p, layer_p, seed)


def temporal_ce_bwd_op(dloss: Tensor, logits: Tensor, target: Tensor, token_w: Tensor, enc_params: Sequence[Tensor], head_w: Tensor,
                       mask: Optional[Tensor], feat: Tensor, saved: Tensor, enc_out: Tensor, Hh: int, Ww: int, dt: int, hid: int, L: int, H: int,
                       attn_p: float, layer_p: float, seed: int, seed_inc: Optional[Tensor] = None) -> List[Tensor]:
    """-> [dh, dtoken_w, dtoken_b, dhead_w, dhead_b, denc_param_0, ...]: hybrid::cross_entropy_bwd + hybrid::temporal_bwd in the same launches."""
    _require_cuda(dloss, logits, feat)
    B, S, D = enc_out.shape
    N, Cp = feat.shape
    C = token_w.shape[1]
    classes = head_w.shape[0]
    dev, tdt = feat.device, _TORCH_DTYPE[dt & 0xff]
    dl = dloss.contiguous().float().reshape(1)
    ps = [p.contiguous() for p in enc_params]
    grads = [torch.empty_like(p) for p in ps]
    dh = torch.empty(N, Hh, Ww, Cp, dtype=torch.bfloat16 if dt & HYB_H_BF16 else tdt, device=dev)
    dtw = torch.empty_like(token_w, memory_format=torch.contiguous_format)
    dtb = torch.empty(D, dtype=torch.float32, device=dev)
    dhw = torch.empty_like(head_w, memory_format=torch.contiguous_format)
    dhb = torch.empty(classes, dtype=torch.float32, device=dev)
    ws = _ws(_query("hyb_temporal_bwd_workspace", dt & 0xff, B, S, Hh * Ww, Cp, D, hid, L, H), dev)
    lib.call("hyb_temporal_ce_bwd", dt, dl.data_ptr(), logits.contiguous().data_ptr(), target.contiguous().data_ptr(), token_w.contiguous().data_ptr(),
             ptr_array([p.data_ptr() for p in ps]), head_w.contiguous().data_ptr(), _opt_ptr(mask), feat.data_ptr(), saved.data_ptr(), enc_out.data_ptr(),
             dtw.data_ptr(), dtb.data_ptr(), ptr_array([g.data_ptr() for g in grads]), dhw.data_ptr(), dhb.data_ptr(), dh.data_ptr(), B, S, Hh * Ww, C, Cp,
             D, hid, L, H, classes, float(attn_p), float(layer_p), seed, _opt_ptr(seed_inc), ws.data_ptr(), ws.numel(), _stream())
    return [dh, dtw, dtb, dhw, dhb] + grads


def temporal_ce_bwd_fake(dloss, logits, target, token_w, enc_params, head_w, mask, feat, saved, enc_out, Hh, Ww, dt, hid, L, H, attn_p, layer_p, seed,
                         seed_inc=None):
    return temporal_bwd_fake(logits, token_w, enc_params, head_w, mask, feat, saved, enc_out, Hh, Ww, dt, hid, L, H, attn_p, layer_p, seed)


def temporal_ce(h, token_w, token_b, enc_params, head_w, head_b, mask, target, B, dt, hid, L, H, attn_p, layer_p, seed):
    """-> (loss, logits): the temporal part and the mean cross-entropy loss as one operator (the loss rides in the temporal part's last launch,
    its backward in the backward's first)."""
    S = h.shape[0] // B
    r = torch.ops.hybrid.temporal_ce(h, token_w, token_b, list(enc_params), head_w, head_b, check_mask(mask, B, S, h.device), target, B, dt, hid, L, H,
                                     float(attn_p), float(layer_p), seed, step_counter())
    return r[0], r[1]


def temporal(h, token_w, token_b, enc_params, head_w, head_b, mask, B, dt, hid, L, H, attn_p, layer_p, seed):
    S = h.shape[0] // B
    return torch.ops.hybrid.temporal(h, token_w, token_b, list(enc_params), head_w, head_b, check_mask(mask, B, S, h.device), B, dt, hid, L, H,
                                     float(attn_p), float(layer_p), seed, step_counter())[0]


# ---------------------------------------------------------------------------------------------
# Autograd-key kernels: one torch.autograd.Function per differentiable operator.  forward() re-enters the operator below the
# Autograd key (the backend kernel above), saves what the backward operator needs, and marks the outputs that exist only to be
# saved as non-differentiable; backward() is one call of the matching hybrid::*_bwd operator.
# ---------------------------------------------------------------------------------------------
_CLIP_GRAD_MSG = ("the first conv stage does not compute a gradient for its input (the clip tensor): pass clips with "
                  "requires_grad=False (training data never needs one); a silent None would be wrong")


class _NchwToNhwcFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dt, cp):
        ctx.set_materialize_grads(False)
        ctx.cfg = (dt, x.shape[1])
        with _below_autograd():
            return torch.ops.hybrid.nchw_to_nhwc(x, dt, cp)

    @staticmethod
    def backward(ctx, g):
        return torch.ops.hybrid.nhwc_to_nchw(g, *ctx.cfg), None, None


class _NhwcToNchwFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dt, C):
        ctx.set_materialize_grads(False)
        ctx.cfg = (dt, x.shape[3])
        with _below_autograd():
            return torch.ops.hybrid.nhwc_to_nchw(x, dt, C)

    @staticmethod
    def backward(ctx, g):
        return torch.ops.hybrid.nchw_to_nhwc(g, *ctx.cfg), None, None


class _CastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dt, to_t):
        ctx.set_materialize_grads(False)
        ctx.cfg = (dt, not to_t)
        with _below_autograd():
            return torch.ops.hybrid.cast(x, dt, to_t)

    @staticmethod
    def backward(ctx, g):
        return torch.ops.hybrid.cast(g, *ctx.cfg), None, None


class _ConvStageFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, rm, rv, training, momentum, eps, dt, first):
        ctx.set_materialize_grads(False)
        if first and ctx.needs_input_grad[0]:
            raise RuntimeError(_CLIP_GRAD_MSG)
        with _below_autograd():
            out = torch.ops.hybrid.convstage(x, weight, gamma, beta, rm, rv, training, momentum, eps, dt, first)
        ctx.save_for_backward(x, out[1], weight, gamma, out[2], out[3], out[4], out[0])
        ctx.cfg = (training, dt, first)
        ctx.mark_non_differentiable(*out[1:])
        return out

    @staticmethod
    def backward(ctx, dpooled, *unused):
        x, y_raw, weight, gamma, ss, mi, pk, pooled = ctx.saved_tensors
        training, dt, first = ctx.cfg
        dx, dw, dgamma, dbeta = torch.ops.hybrid.convstage_bwd(dpooled, x, y_raw, pooled, weight, gamma, ss, mi, pk, training, dt, first)
        return (None if first else dx, dw, dgamma, dbeta) + (None,) * 7


class _TokenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, dt):
        ctx.set_materialize_grads(False)
        with _below_autograd():
            tok, feat = torch.ops.hybrid.token(x, weight, bias, dt)
        ctx.save_for_backward(feat, weight)
        ctx.cfg = (x.shape[1], x.shape[2], bias is not None, dt)
        ctx.mark_non_differentiable(feat)
        return tok, feat

    @staticmethod
    def backward(ctx, dtok, unused):
        feat, weight = ctx.saved_tensors
        Hh, Ww, has_bias, dt = ctx.cfg
        dx, dw, db = torch.ops.hybrid.token_bwd(dtok, feat, weight, Hh, Ww, has_bias, dt)
        return dx, dw, (db if has_bias else None), None


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask, dt, hid, L, H, attn_p, layer_p, seed, seed_inc, *params):
        ctx.set_materialize_grads(False)
        with _below_autograd():
            out, saved = torch.ops.hybrid.encoder(x, mask, params, dt, hid, L, H, attn_p, layer_p, seed, seed_inc)
        ctx.save_for_backward(saved, *params) if mask is None else ctx.save_for_backward(saved, mask, *params)
        ctx.seed_inc = seed_inc                     # an int64 counter, not part of the autograd graph
        ctx.cfg = (mask is not None, dt, hid, L, H, attn_p, layer_p, seed)
        ctx.mark_non_differentiable(saved)
        return out, saved

    @staticmethod
    def backward(ctx, dout, unused):
        has_mask, dt, hid, L, H, attn_p, layer_p, seed = ctx.cfg
        saved, *rest = ctx.saved_tensors
        mask = rest.pop(0) if has_mask else None
        res = torch.ops.hybrid.encoder_bwd(dout, mask, rest, saved, dt, hid, L, H, attn_p, layer_p, seed, ctx.seed_inc)
        return (res[0],) + (None,) * 9 + tuple(res[1:])


class _MhaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q_in, k_in, v_in, mask, dt, H, p_drop, seed, *params):
        ctx.set_materialize_grads(False)
        with _below_autograd():
            out = torch.ops.hybrid.mha(q_in, k_in, v_in, mask, params, dt, H, p_drop, seed)
        if mask is None:
            ctx.save_for_backward(q_in, k_in, v_in, *out[1:], *params)
        else:
            ctx.save_for_backward(mask, q_in, k_in, v_in, *out[1:], *params)
        ctx.cfg = (mask is not None, dt, H, p_drop, seed)
        ctx.mark_non_differentiable(*out[1:])
        return out

    @staticmethod
    def backward(ctx, dout, *unused):
        t = list(ctx.saved_tensors)
        mask = t.pop(0) if ctx.cfg[0] else None
        q_in, k_in, v_in, q, k, v, a, probs, *ps = t
        res = torch.ops.hybrid.mha_bwd(dout, q_in, k_in, v_in, mask, q, k, v, a, probs, ps, *ctx.cfg[1:])
        return (res[0], res[1], res[2]) + (None,) * 5 + tuple(res[3:])


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, dt):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (bias is not None, dt)
        with _below_autograd():
            return torch.ops.hybrid.head(x, weight, bias, dt)

    @staticmethod
    def backward(ctx, dlogits):
        x, weight = ctx.saved_tensors
        has_bias, dt = ctx.cfg
        dx, dw, db = torch.ops.hybrid.head_bwd(dlogits, x.contiguous(), weight, has_bias, dt)
        return dx, dw, (db if has_bias else None), None


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(logits, target)
        with _below_autograd():
            return torch.ops.hybrid.cross_entropy(logits, target)

    @staticmethod
    def backward(ctx, dloss):
        logits, target = ctx.saved_tensors
        return torch.ops.hybrid.cross_entropy_bwd(dloss, logits, target), None


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, S, training, momentum, eps, dt, *tensors):
        ctx.set_materialize_grads(False)
        if ctx.needs_input_grad[0]:
            raise RuntimeError(_CLIP_GRAD_MSG)
        weights, gammas, betas, rms, rvs = (tensors[i * S:(i + 1) * S] for i in range(5))
        nbts = tensors[5 * S:]                             # hybrid::backbone_ only
        ctx.n_extra = len(tensors) - 3 * S
        with _below_autograd():
            if nbts:                                       # (the buffers carry no gradient: nothing to tell autograd about their update)
                res = torch.ops.hybrid.backbone_(x, weights, gammas, betas, rms, rvs, nbts, training, momentum, eps, dt)
            else:
                res = torch.ops.hybrid.backbone(x, weights, gammas, betas, rms, rvs, training, momentum, eps, dt)
        st = _backbone_unpack(res, S)
        saved = []
        for s in range(S):
            saved += [st[s][0], (st[s - 1][1] if s > 0 else st[s][0]), st[s][2], st[s][3], st[s][4]]
        ctx.save_for_backward(x, res[0], *weights, *gammas, *saved)
        ctx.cfg = (S, training, dt)
        ctx.mark_non_differentiable(*res[1:])
        return tuple(res)

    @staticmethod
    def backward(ctx, dpooled, *unused):
        S, training, dt = ctx.cfg
        t = ctx.saved_tensors
        g = torch.ops.hybrid.backbone_bwd(dpooled, t[1], t[0], t[2:2 + S], t[2 + S:2 + 2 * S], t[2 + 2 * S:], training, dt)
        return (None,) * 6 + tuple(g[0::3]) + tuple(g[1::3]) + tuple(g[2::3]) + (None,) * ctx.n_extra


class _TemporalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, token_w, token_b, head_w, head_b, mask, B, dt, hid, L, H, attn_p, layer_p, seed, seed_inc, *enc_params):
        ctx.set_materialize_grads(False)
        with _below_autograd():
            logits, feat, saved, enc_out = torch.ops.hybrid.temporal(h, token_w, token_b, enc_params, head_w, head_b, mask, B, dt, hid, L, H,
                                                                      attn_p, layer_p, seed, seed_inc)
        ctx.seed_inc = seed_inc
        if mask is None:
            ctx.save_for_backward(token_w, head_w, feat, saved, enc_out, *enc_params)
        else:
            ctx.save_for_backward(token_w, head_w, feat, saved, enc_out, mask, *enc_params)
        ctx.cfg = (mask is not None, h.shape[1], h.shape[2], dt, hid, L, H, attn_p, layer_p, seed)
        ctx.mark_non_differentiable(feat, saved, enc_out)
        return logits, feat, saved, enc_out

    @staticmethod
    def backward(ctx, dlogits, *unused):
        has_mask, Hh, Ww, dt, hid, L, H, attn_p, layer_p, seed = ctx.cfg
        token_w, head_w, feat, saved, enc_out, *rest = ctx.saved_tensors
        mask = rest.pop(0) if has_mask else None
        g = torch.ops.hybrid.temporal_bwd(dlogits, token_w, rest, head_w, mask, feat, saved, enc_out, Hh, Ww, dt, hid, L, H, attn_p, layer_p, seed,
                                          ctx.seed_inc)
        return (g[0], g[1], g[2], g[3], g[4]) + (None,) * 10 + tuple(g[5:])


class _TemporalCeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, token_w, token_b, head_w, head_b, mask, target, B, dt, hid, L, H, attn_p, layer_p, seed, seed_inc, *enc_params):
        ctx.set_materialize_grads(False)
        with _below_autograd():
            loss, logits, feat, saved, enc_out = torch.ops.hybrid.temporal_ce(h, token_w, token_b, enc_params, head_w, head_b, mask, target, B, dt, hid,
                                                                              L, H, attn_p, layer_p, seed, seed_inc)
        ctx.seed_inc = seed_inc
        if mask is None:
            ctx.save_for_backward(token_w, head_w, feat, saved, enc_out, logits, target, *enc_params)
        else:
            ctx.save_for_backward(token_w, head_w, feat, saved, enc_out, logits, target, mask, *enc_params)
        ctx.cfg = (mask is not None, h.shape[1], h.shape[2], dt, hid, L, H, attn_p, layer_p, seed)
        ctx.mark_non_differentiable(logits, feat, saved, enc_out)      # (logits: an output for the caller's metrics; the objective is the loss)
        return loss, logits, feat, saved, enc_out

    @staticmethod
    def backward(ctx, dloss, *unused):
        has_mask, Hh, Ww, dt, hid, L, H, attn_p, layer_p, seed = ctx.cfg
        token_w, head_w, feat, saved, enc_out, logits, target, *rest = ctx.saved_tensors
        mask = rest.pop(0) if has_mask else None
        g = torch.ops.hybrid.temporal_ce_bwd(dloss, logits, target, token_w, rest, head_w, mask, feat, saved, enc_out, Hh, Ww, dt, hid, L, H, attn_p,
                                             layer_p, seed, ctx.seed_inc)
        return (g[0], g[1], g[2], g[3], g[4]) + (None,) * 11 + tuple(g[5:])


_define("nchw_to_nhwc", "(Tensor x, int dt, int cp) -> Tensor", nchw_to_nhwc_op, nchw_to_nhwc_fake, _NchwToNhwcFn.apply)
_define("nhwc_to_nchw", "(Tensor x, int dt, int C) -> Tensor", nhwc_to_nchw_op, nhwc_to_nchw_fake, _NhwcToNchwFn.apply)
_define("cast", "(Tensor x, int dt, bool to_t) -> Tensor", cast_op, cast_fake, _CastFn.apply)
_define("convstage", "(Tensor x, Tensor weight, Tensor gamma, Tensor beta, Tensor? running_mean, Tensor? running_var, bool training, float momentum, "
        "float eps, int dt, bool first) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)", convstage_op, convstage_fake, _ConvStageFn.apply)
_define("convstage_bwd", "(Tensor dpooled, Tensor x, Tensor y_raw, Tensor? pooled, Tensor weight, Tensor gamma, Tensor scale_shift, Tensor mean_invstd, "
        "Tensor packed_bwd, bool training, int dt, bool first) -> (Tensor, Tensor, Tensor, Tensor)", convstage_bwd_op, convstage_bwd_fake)
_define("token", "(Tensor x, Tensor weight, Tensor? bias, int dt) -> (Tensor, Tensor)", token_op, token_fake, _TokenFn.apply)
_define("token_bwd", "(Tensor dtok, Tensor feat, Tensor weight, int Hh, int Ww, bool has_bias, int dt) -> (Tensor, Tensor, Tensor)", token_bwd_op,
        token_bwd_fake)
_define("encoder", "(Tensor x, Tensor? mask, Tensor[] params, int dt, int hid, int L, int H, float attn_p, float layer_p, int seed, "
        "Tensor? seed_inc=None) -> (Tensor, Tensor)", encoder_op, encoder_fake,
        lambda x, mask, params, dt, hid, L, H, attn_p, layer_p, seed, seed_inc=None: _EncoderFn.apply(x, mask, dt, hid, L, H, attn_p, layer_p, seed,
                                                                                                    seed_inc, *params))
_define("encoder_bwd", "(Tensor dout, Tensor? mask, Tensor[] params, Tensor saved, int dt, int hid, int L, int H, float attn_p, float layer_p, "
        "int seed, Tensor? seed_inc=None) -> Tensor[]", encoder_bwd_op, encoder_bwd_fake)
_define("mha", "(Tensor q_in, Tensor k_in, Tensor v_in, Tensor? mask, Tensor[] params, int dt, int H, float p_drop, int seed) -> "
        "(Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)", mha_op, mha_fake,
        lambda q, k, v, mask, params, dt, H, p_drop, seed: _MhaFn.apply(q, k, v, mask, dt, H, p_drop, seed, *params))
_define("mha_bwd", "(Tensor dout, Tensor q_in, Tensor k_in, Tensor v_in, Tensor? mask, Tensor q, Tensor k, Tensor v, Tensor a, Tensor probs, Tensor[] params, "
        "int dt, int H, float p_drop, int seed) -> Tensor[]", mha_bwd_op, mha_bwd_fake)
_define("head", "(Tensor x, Tensor weight, Tensor? bias, int dt) -> Tensor", head_op, head_fake, _HeadFn.apply)
_define("head_bwd", "(Tensor dlogits, Tensor x, Tensor weight, bool has_bias, int dt) -> (Tensor, Tensor, Tensor)", head_bwd_op, head_bwd_fake)
_define("cross_entropy", "(Tensor logits, Tensor target) -> Tensor", cross_entropy_op, cross_entropy_fake, _CrossEntropyFn.apply)
_define("cross_entropy_bwd", "(Tensor dloss, Tensor logits, Tensor target) -> Tensor", cross_entropy_bwd_op, cross_entropy_bwd_fake)
_define("backbone", "(Tensor x, Tensor[] weights, Tensor[] gammas, Tensor[] betas, Tensor[] running_means, Tensor[] running_vars, bool training, "
        "float momentum, float eps, int dt) -> Tensor[]", backbone_op, backbone_fake,
        lambda x, ws, gs, bs, rms, rvs, training, momentum, eps, dt: list(_BackboneFn.apply(x, len(ws), training, momentum, eps, dt, *ws, *gs, *bs,
                                                                                          *rms, *rvs)))
_define("backbone_", "(Tensor x, Tensor[] weights, Tensor[] gammas, Tensor[] betas, Tensor(a!)[] running_means, Tensor(b!)[] running_vars, "
        "Tensor(c!)[] num_batches_tracked, bool training, float momentum, float eps, int dt) -> Tensor[]", backbone_inplace_op, backbone_inplace_fake,
        lambda x, ws, gs, bs, rms, rvs, nbts, training, momentum, eps, dt: list(_BackboneFn.apply(x, len(ws), training, momentum, eps, dt, *ws, *gs,
                                                                                                *bs, *rms, *rvs, *nbts)))
_define("backbone_bwd", "(Tensor dpooled, Tensor pooled, Tensor x, Tensor[] weights, Tensor[] gammas, Tensor[] saved, bool training, int dt) -> Tensor[]",
        backbone_bwd_op, backbone_bwd_fake)
_define("temporal", "(Tensor h, Tensor token_w, Tensor token_b, Tensor[] enc_params, Tensor head_w, Tensor head_b, Tensor? mask, int B, int dt, "
        "int hid, int L, int H, float attn_p, float layer_p, int seed, Tensor? seed_inc=None) -> (Tensor, Tensor, Tensor, Tensor)", temporal_op,
        temporal_fake,
        lambda h, tw, tb, ps, hw, hb, mask, B, dt, hid, L, H, attn_p, layer_p, seed, seed_inc=None: _TemporalFn.apply(
            h, tw, tb, hw, hb, mask, B, dt, hid, L, H, attn_p, layer_p, seed, seed_inc, *ps))
_define("temporal_bwd", "(Tensor dlogits, Tensor token_w, Tensor[] enc_params, Tensor head_w, Tensor? mask, Tensor feat, Tensor saved, "
        "Tensor enc_out, int Hh, int Ww, int dt, int hid, int L, int H, float attn_p, float layer_p, int seed, Tensor? seed_inc=None) -> Tensor[]",
        temporal_bwd_op, temporal_bwd_fake)
_define("temporal_ce", "(Tensor h, Tensor token_w, Tensor token_b, Tensor[] enc_params, Tensor head_w, Tensor head_b, Tensor? mask, Tensor target, int B, "
        "int dt, int hid, int L, int H, float attn_p, float layer_p, int seed, Tensor? seed_inc=None) -> (Tensor, Tensor, Tensor, Tensor, Tensor)",
        temporal_ce_op, temporal_ce_fake,
        lambda h, tw, tb, ps, hw, hb, mask, target, B, dt, hid, L, H, attn_p, layer_p, seed, seed_inc=None: _TemporalCeFn.apply(
            h, tw, tb, hw, hb, mask, target, B, dt, hid, L, H, attn_p, layer_p, seed, seed_inc, *ps))
_define("temporal_ce_bwd", "(Tensor dloss, Tensor logits, Tensor target, Tensor token_w, Tensor[] enc_params, Tensor head_w, Tensor? mask, Tensor feat, "
        "Tensor saved, Tensor enc_out, int Hh, int Ww, int dt, int hid, int L, int H, float attn_p, float layer_p, int seed, Tensor? seed_inc=None) "
        "-> Tensor[]", temporal_ce_bwd_op, temporal_ce_bwd_fake)


# ---------------------------------------------------------------------------------------------
# FCT operators (FCT.py:24-254, Metrics.py:5-22), forward and backward.  NHWC fp32 tensors with the true
# channel count.  Same registration scheme as above: backend + fake kernels for every operator, an autograd Function on the
# Autograd key of the differentiable ones, each backward a hybrid::*_bwd operator.
# ---------------------------------------------------------------------------------------------
ACT_NONE, ACT_RELU, ACT_GELU, ACT_SIGMOID = 0, 1, 2, 3


def _f32c(t):
    return t.contiguous().float()


def _e0(ref):
    return torch.empty(0, dtype=torch.float32, device=ref.device)


def fct_conv_op(x: Tensor, weight: Tensor, bias: Optional[Tensor], dilation: int, act: int) -> Tuple[Tensor, Tensor]:
    """-> (y, z): z is the pre-activation, kept only for GELU (its backward needs it), else empty."""
    _require_cuda(x, weight)
    x = _f32c(x)
    N, H, W, Ci = x.shape
    Co = weight.shape[0]
    y = torch.empty(N, H, W, Co, dtype=torch.float32, device=x.device)
    z = torch.empty_like(y) if act == ACT_GELU else _e0(x)
    ws = _ws(_query("hyb_fct_conv_workspace", N, H, W, Ci, Co), x.device)
    lib.call("hyb_fct_conv_fwd", x.data_ptr(), _f32c(weight).data_ptr(), _f32c(bias).data_ptr() if bias is not None else None, y.data_ptr(),
             z.data_ptr() if act == ACT_GELU else None, N, H, W, Ci, Co, dilation, act, ws.data_ptr(), ws.numel(), _stream())
    return y, z


def fct_conv_fake(x, weight, bias, dilation, act):
    N, H, W, _ = x.shape
    y = x.new_empty((N, H, W, weight.shape[0]), dtype=torch.float32)
    return y, (torch.empty_like(y) if act == ACT_GELU else x.new_empty((0,), dtype=torch.float32))


def fct_conv_bwd_op(dy: Tensor, x: Tensor, weight: Tensor, saved: Tensor, has_bias: bool, need_dx: bool, dilation: int,
                    act: int) -> Tuple[Tensor, Tensor, Tensor]:
    _require_cuda(dy, x)
    dy, x = _f32c(dy), _f32c(x)
    N, H, W, Ci = x.shape
    Co = weight.shape[0]
    dx = torch.empty_like(x) if need_dx else _e0(x)
    dw = torch.empty_like(weight, memory_format=torch.contiguous_format, dtype=torch.float32)
    db = torch.empty(Co if has_bias else 0, dtype=torch.float32, device=x.device)
    ws = _ws(_query("hyb_fct_conv_bwd_workspace", N, H, W, Ci, Co), x.device)
    lib.call("hyb_fct_conv_bwd", dy.data_ptr(), x.data_ptr(), _f32c(weight).data_ptr(), saved.data_ptr() if act != ACT_NONE else None,
             dx.data_ptr() if need_dx else None, dw.data_ptr(), db.data_ptr() if has_bias else None, N, H, W, Ci, Co, dilation, act,
             ws.data_ptr(), ws.numel(), _stream())
    return dx, dw, db


def fct_conv_bwd_fake(dy, x, weight, saved, has_bias, need_dx, dilation, act):
    return ((torch.empty_like(x) if need_dx else x.new_empty((0,))), torch.empty_like(weight, memory_format=torch.contiguous_format),
            x.new_empty((weight.shape[0] if has_bias else 0,)))


class _FctConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, dilation, act):
        ctx.set_materialize_grads(False)
        with _below_autograd():
            y, z = torch.ops.hybrid.fct_conv(x, weight, bias, dilation, act)
        ctx.save_for_backward(x, weight, z if act == ACT_GELU else y)
        ctx.cfg = (bias is not None, dilation, act)
        ctx.mark_non_differentiable(z)
        return y, z

    @staticmethod
    def backward(ctx, dy, unused):
        x, weight, saved = ctx.saved_tensors
        has_bias, dilation, act = ctx.cfg
        dx, dw, db = torch.ops.hybrid.fct_conv_bwd(dy, x, weight, saved, has_bias, ctx.needs_input_grad[0], dilation, act)
        return (dx if ctx.needs_input_grad[0] else None), dw, (db if has_bias else None), None, None


def fct_qkv_proj_op(x: Tensor, weights: Sequence[Tensor], biases: Sequence[Tensor], ln_weights: Sequence[Tensor], ln_biases: Sequence[Tensor],
                    eps: float) -> Tuple[Tensor, Tensor, Tensor]:
    _require_cuda(x, *weights)
    x = _f32c(x)
    N, H, W, C = x.shape
    q, k, v = (torch.empty_like(x) for _ in range(3))
    ws_, bs_, gs_, be_ = ([_f32c(t) for t in lst] for lst in (weights, biases, ln_weights, ln_biases))
    lib.call("hyb_fct_qkv_proj_fwd", x.data_ptr(), ptr_array([t.data_ptr() for t in ws_]), ptr_array([t.data_ptr() for t in bs_]),
             ptr_array([t.data_ptr() for t in gs_]), ptr_array([t.data_ptr() for t in be_]), q.data_ptr(), k.data_ptr(), v.data_ptr(),
             N, H, W, C, float(eps), _stream())
    return q, k, v


def fct_qkv_proj_fake(x, weights, biases, ln_weights, ln_biases, eps):
    e = lambda: x.new_empty(x.shape, dtype=torch.float32)
    return e(), e(), e()


def fct_qkv_proj_bwd_op(x: Tensor, weights: Sequence[Tensor], biases: Sequence[Tensor], ln_weights: Sequence[Tensor], dqs: Sequence[Tensor],
                        eps: float) -> List[Tensor]:
    """-> [dx, dw_q, dw_k, dw_v, db_q, db_k, db_v, dg_q, dg_k, dg_v, dbeta_q, dbeta_k, dbeta_v]"""
    _require_cuda(x, *dqs)
    x = _f32c(x)
    N, H, W, C = x.shape
    dev = x.device
    ws_, bs_, gs_, dq_ = ([_f32c(t) for t in lst] for lst in (weights, biases, ln_weights, dqs))
    dx = torch.empty_like(x)
    dws = [torch.empty(C, 1, 3, 3, dtype=torch.float32, device=dev) for _ in range(3)]
    dbs, dgs, dbetas = ([torch.empty(C, dtype=torch.float32, device=dev) for _ in range(3)] for _ in range(3))
    ws = _ws(_query("hyb_fct_qkv_proj_bwd_workspace", N, H, W, C), dev)
    pa = lambda lst: ptr_array([t.data_ptr() for t in lst])
    lib.call("hyb_fct_qkv_proj_bwd", x.data_ptr(), pa(ws_), pa(bs_), pa(gs_), pa(dq_), dx.data_ptr(), pa(dws), pa(dbs), pa(dgs), pa(dbetas),
             N, H, W, C, float(eps), ws.data_ptr(), ws.numel(), _stream())
    return [dx] + dws + dbs + dgs + dbetas


def fct_qkv_proj_bwd_fake(x, weights, biases, ln_weights, dqs, eps):
    C = x.shape[-1]
    v = lambda: x.new_empty((C,), dtype=torch.float32)
    return [torch.empty_like(x)] + [x.new_empty((C, 1, 3, 3), dtype=torch.float32) for _ in range(3)] + [v() for _ in range(9)]


class _FctQkvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps, *t):                               # t = 3 weights, 3 biases, 3 LN weights, 3 LN biases
        ctx.set_materialize_grads(False)
        with _below_autograd():
            out = torch.ops.hybrid.fct_qkv_proj(x, t[0:3], t[3:6], t[6:9], t[9:12], eps)
        ctx.save_for_backward(x, *t[0:9])
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, dq, dk, dv):
        x, *t = ctx.saved_tensors
        z = lambda g: g if g is not None else torch.zeros_like(x)
        r = torch.ops.hybrid.fct_qkv_proj_bwd(x, t[0:3], t[3:6], t[6:9], [z(dq), z(dk), z(dv)], ctx.eps)
        return (r[0], None) + tuple(r[1:])


def fct_ln_op(x: Tensor, weight: Tensor, bias: Tensor, eps: float) -> Tensor:
    _require_cuda(x, weight)
    x = _f32c(x)
    y = torch.empty_like(x)
    C = x.shape[-1]
    lib.call("hyb_fct_ln_fwd", x.data_ptr(), _f32c(weight).data_ptr(), _f32c(bias).data_ptr(), y.data_ptr(), x.numel() // C, C, float(eps), _stream())
    return y


def fct_ln_fake(x, weight, bias, eps):
    return x.new_empty(x.shape, dtype=torch.float32)


def fct_ln_bwd_op(dy: Tensor, x: Tensor, weight: Tensor, eps: float) -> Tuple[Tensor, Tensor, Tensor]:
    _require_cuda(dy, x)
    dy, x = _f32c(dy), _f32c(x)
    C = x.shape[-1]
    P = x.numel() // C
    dx = torch.empty_like(x)
    dg, db = (torch.empty(C, dtype=torch.float32, device=x.device) for _ in range(2))
    ws = _ws(_query("hyb_fct_ln_bwd_workspace", P, C), x.device)
    lib.call("hyb_fct_ln_bwd", dy.data_ptr(), x.data_ptr(), _f32c(weight).data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), P, C, float(eps),
             ws.data_ptr(), ws.numel(), _stream())
    return dx, dg, db


def fct_ln_bwd_fake(dy, x, weight, eps):
    C = x.shape[-1]
    return torch.empty_like(x), x.new_empty((C,), dtype=torch.float32), x.new_empty((C,), dtype=torch.float32)


class _FctLnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, weight)
        ctx.eps = eps
        with _below_autograd():
            return torch.ops.hybrid.fct_ln(x, weight, bias, eps)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx, dg, db = torch.ops.hybrid.fct_ln_bwd(dy, x, weight, ctx.eps)
        return dx, dg, db, None


def fct_mha_op(q: Tensor, k: Tensor, v: Tensor, in_w: Tensor, in_b: Optional[Tensor], out_w: Tensor, out_b: Optional[Tensor],
               heads: int) -> Tuple[Tensor, Tensor]:
    """q, k, v [N, L, C] (pixel tokens) -> (nn.MultiheadAttention output [N, L, C], blob saved for backward)."""
    _require_cuda(q, in_w)
    q, k, v = _f32c(q), _f32c(k), _f32c(v)
    N, L, C = q.shape
    out = torch.empty_like(q)
    saved = _ws(_query("hyb_fct_mha_saved_bytes", N, L, C, heads), q.device)
    ws = _ws(_query("hyb_fct_mha_workspace", N, L, C, heads), q.device)
    lib.call("hyb_fct_mha_fwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), _f32c(in_w).data_ptr(), _f32c(in_b).data_ptr() if in_b is not None else None,
             _f32c(out_w).data_ptr(), _f32c(out_b).data_ptr() if out_b is not None else None, out.data_ptr(), saved.data_ptr(), N, L, C, heads,
             ws.data_ptr(), ws.numel(), _stream())
    return out, saved


def fct_mha_fake(q, k, v, in_w, in_b, out_w, out_b, heads):
    N, L, C = q.shape
    return q.new_empty(q.shape, dtype=torch.float32), q.new_empty((max(_query("hyb_fct_mha_saved_bytes", N, L, C, heads), 256),), dtype=torch.uint8)


def fct_mha_bwd_op(dout: Tensor, q: Tensor, k: Tensor, v: Tensor, in_w: Tensor, out_w: Tensor, saved: Tensor, heads: int, has_in_b: bool,
                   has_out_b: bool) -> List[Tensor]:
    """-> [dq, dk, dv, din_w, din_b, dout_w, dout_b]"""
    _require_cuda(dout, q)
    dout, q, k, v = _f32c(dout), _f32c(q), _f32c(k), _f32c(v)
    N, L, C = q.shape
    dev = q.device
    dq, dk, dv = (torch.empty_like(q) for _ in range(3))
    din_w = torch.empty(3 * C, C, dtype=torch.float32, device=dev)
    din_b = torch.empty(3 * C if has_in_b else 0, dtype=torch.float32, device=dev)
    dout_w = torch.empty(C, C, dtype=torch.float32, device=dev)
    dout_b = torch.empty(C if has_out_b else 0, dtype=torch.float32, device=dev)
    ws = _ws(_query("hyb_fct_mha_bwd_workspace", N, L, C, heads), dev)
    lib.call("hyb_fct_mha_bwd", dout.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), _f32c(in_w).data_ptr(), _f32c(out_w).data_ptr(),
             saved.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), din_w.data_ptr(), din_b.data_ptr() if has_in_b else None,
             dout_w.data_ptr(), dout_b.data_ptr() if has_out_b else None, N, L, C, heads, ws.data_ptr(), ws.numel(), _stream())
    return [dq, dk, dv, din_w, din_b, dout_w, dout_b]


def fct_mha_bwd_fake(dout, q, k, v, in_w, out_w, saved, heads, has_in_b, has_out_b):
    C = q.shape[-1]
    f = lambda *s: q.new_empty(s, dtype=torch.float32)
    return [torch.empty_like(q), torch.empty_like(q), torch.empty_like(q), f(3 * C, C), f(3 * C if has_in_b else 0), f(C, C), f(C if has_out_b else 0)]


class _FctMhaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, in_w, in_b, out_w, out_b, heads):
        ctx.set_materialize_grads(False)
        with _below_autograd():
            out, saved = torch.ops.hybrid.fct_mha(q, k, v, in_w, in_b, out_w, out_b, heads)
        ctx.save_for_backward(q, k, v, in_w, out_w, saved)
        ctx.cfg = (heads, in_b is not None, out_b is not None)
        ctx.mark_non_differentiable(saved)
        return out, saved

    @staticmethod
    def backward(ctx, dout, unused):
        q, k, v, in_w, out_w, saved = ctx.saved_tensors
        heads, hib, hob = ctx.cfg
        r = torch.ops.hybrid.fct_mha_bwd(dout, q, k, v, in_w, out_w, saved, heads, hib, hob)
        return r[0], r[1], r[2], r[3], (r[4] if hib else None), r[5], (r[6] if hob else None), None


def fct_add_op(a: Tensor, b: Tensor) -> Tensor:
    _require_cuda(a, b)
    a, b = _f32c(a), _f32c(b)
    if a.shape != b.shape:
        raise RuntimeError(f"shapes differ: {tuple(a.shape)} vs {tuple(b.shape)}")
    y = torch.empty_like(a)
    lib.call("hyb_fct_add", a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _stream())
    return y


def fct_add_fake(a, b):
    return a.new_empty(a.shape, dtype=torch.float32)


class _FctAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.set_materialize_grads(False)
        with _below_autograd():
            return torch.ops.hybrid.fct_add(a, b)

    @staticmethod
    def backward(ctx, g):
        return g, g


def fct_resample_op(x: Tensor, mode: int) -> Tensor:
    """mode 0 MaxPool2d(2), 1 AvgPool2d(2,2), 2 nearest Upsample x2; NHWC."""
    _require_cuda(x)
    x = _f32c(x)
    N, H, W, C = x.shape
    y = torch.empty((N, 2 * H, 2 * W, C) if mode == 2 else (N, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
    lib.call("hyb_fct_resample", mode, x.data_ptr(), y.data_ptr(), N, H, W, C, _stream())
    return y


def fct_resample_fake(x, mode):
    N, H, W, C = x.shape
    return x.new_empty((N, 2 * H, 2 * W, C) if mode == 2 else (N, H // 2, W // 2, C), dtype=torch.float32)


def fct_resample_bwd_op(dy: Tensor, x: Tensor, mode: int) -> Tensor:
    _require_cuda(dy, x)
    dy, x = _f32c(dy), _f32c(x)
    N, H, W, C = x.shape
    dx = torch.empty_like(x)
    lib.call("hyb_fct_resample_bwd", mode, dy.data_ptr(), x.data_ptr(), dx.data_ptr(), N, H, W, C, _stream())
    return dx


def fct_resample_bwd_fake(dy, x, mode):
    return torch.empty_like(x)


class _FctResampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        ctx.set_materialize_grads(False)
        if mode == 1 and ctx.needs_input_grad[0]:
            raise NotImplementedError("AvgPool2d is applied to the input frames only (FCT.py:238-240): no backward")
        ctx.save_for_backward(x)
        ctx.mode = mode
        with _below_autograd():
            return torch.ops.hybrid.fct_resample(x, mode)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return torch.ops.hybrid.fct_resample_bwd(dy, x, ctx.mode), None


def fct_concat_op(a: Tensor, b: Tensor) -> Tensor:
    _require_cuda(a, b)
    a, b = _f32c(a), _f32c(b)
    if a.shape[:-1] != b.shape[:-1]:
        raise RuntimeError(f"Sizes of tensors must match except in the channel dimension: {tuple(a.shape)} vs {tuple(b.shape)}")
    y = torch.empty(*a.shape[:-1], a.shape[-1] + b.shape[-1], dtype=torch.float32, device=a.device)
    lib.call("hyb_fct_concat", a.data_ptr(), a.shape[-1], b.data_ptr(), b.shape[-1], y.data_ptr(), a.numel() // a.shape[-1], _stream())
    return y


def fct_concat_fake(a, b):
    return a.new_empty((*a.shape[:-1], a.shape[-1] + b.shape[-1]), dtype=torch.float32)


def fct_concat_bwd_op(dy: Tensor, Ca: int, Cb: int) -> Tuple[Tensor, Tensor]:
    _require_cuda(dy)
    dy = _f32c(dy)
    da = torch.empty(*dy.shape[:-1], Ca, dtype=torch.float32, device=dy.device)
    db = torch.empty(*dy.shape[:-1], Cb, dtype=torch.float32, device=dy.device)
    lib.call("hyb_fct_concat_bwd", dy.data_ptr(), da.data_ptr(), Ca, db.data_ptr(), Cb, dy.numel() // (Ca + Cb), _stream())
    return da, db


def fct_concat_bwd_fake(dy, Ca, Cb):
    return dy.new_empty((*dy.shape[:-1], Ca)), dy.new_empty((*dy.shape[:-1], Cb))


class _FctConcatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.set_materialize_grads(False)
        ctx.cfg = (a.shape[-1], b.shape[-1])
        with _below_autograd():
            return torch.ops.hybrid.fct_concat(a, b)

    @staticmethod
    def backward(ctx, dy):
        return torch.ops.hybrid.fct_concat_bwd(dy, *ctx.cfg)


def fct_dropout_op(x: Tensor, p: float, seed: int, seed_inc: Optional[Tensor] = None) -> Tensor:
    _require_cuda(x)
    x = _f32c(x)
    y = torch.empty_like(x)
    lib.call("hyb_fct_dropout", x.data_ptr(), y.data_ptr(), x.numel(), float(p), seed, _opt_ptr(seed_inc), _stream())
    return y


def fct_dropout_fake(x, p, seed, seed_inc=None):
    return x.new_empty(x.shape, dtype=torch.float32)


class _FctDropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, seed_inc):
        ctx.set_materialize_grads(False)
        ctx.cfg = (p, seed)
        ctx.seed_inc = seed_inc
        with _below_autograd():
            return torch.ops.hybrid.fct_dropout(x, p, seed, seed_inc)

    @staticmethod
    def backward(ctx, dy):
        return torch.ops.hybrid.fct_dropout(dy, *ctx.cfg, ctx.seed_inc), None, None, None


def dice_loss_op(pred: Tensor, true: Tensor, smooth: float) -> Tensor:
    """DiceLoss (Metrics.py:5-22) on channel 0 of NCHW tensors."""
    _require_cuda(pred, true)
    if pred.shape != true.shape:
        raise AssertionError("y_pred and y_true sizes differ")          # the reference asserts (Metrics.py:15)
    pred, true = _f32c(pred), _f32c(true)
    N, C = pred.shape[0], pred.shape[1]
    loss = torch.empty((), dtype=torch.float32, device=pred.device)
    ws = _ws(_query("hyb_dice_workspace"), pred.device)
    lib.call("hyb_dice_fwd", pred.data_ptr(), true.data_ptr(), loss.data_ptr(), N, C, pred.numel() // (N * C), float(smooth), ws.data_ptr(),
             ws.numel(), _stream())
    return loss


def dice_loss_fake(pred, true, smooth):
    return pred.new_empty((), dtype=torch.float32)


def dice_loss_bwd_op(dloss: Tensor, pred: Tensor, true: Tensor, smooth: float) -> Tensor:
    _require_cuda(dloss, pred)
    pred, true = _f32c(pred), _f32c(true)
    N, C = pred.shape[0], pred.shape[1]
    dpred = torch.empty_like(pred)
    ws = _ws(4096, pred.device)
    lib.call("hyb_dice_bwd", pred.data_ptr(), true.data_ptr(), _f32c(dloss).reshape(1).data_ptr(), dpred.data_ptr(), N, C, pred.numel() // (N * C),
             float(smooth), ws.data_ptr(), ws.numel(), _stream())
    return dpred


def dice_loss_bwd_fake(dloss, pred, true, smooth):
    return torch.empty_like(pred)


class _DiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, true, smooth):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(pred, true)
        ctx.smooth = smooth
        with _below_autograd():
            return torch.ops.hybrid.dice_loss(pred, true, smooth)

    @staticmethod
    def backward(ctx, dloss):
        pred, true = ctx.saved_tensors
        return torch.ops.hybrid.dice_loss_bwd(dloss, pred, true, ctx.smooth), None, None


# ---------------------------------------------------------------------------------------------
# ResNet-bottleneck backbone `Encoder_32K`: general Conv2d, BatchNorm2d (+ residual + ReLU), Dropout2d.  NHWC fp32.
# ---------------------------------------------------------------------------------------------
def _conv_out(n, k, stride, pad, dilation):
    return (n + 2 * pad - dilation * (k - 1) - 1) // stride + 1


def conv2d_op(x: Tensor, weight: Tensor, bias: Optional[Tensor], stride: int, padding: int, dilation: int, act: int) -> Tuple[Tensor, Tensor]:
    """nn.Conv2d(Ci, Co, k, stride, padding, dilation) on NHWC x, weight [Co,Ci,k,k] -> (y [N,Ho,Wo,Co], z): z is the pre-activation,
    kept only for GELU."""
    _require_cuda(x, weight)
    x = _f32c(x)
    N, H, W, Ci = x.shape
    Co, Ci_w, k, k2 = weight.shape
    if Ci_w != Ci or k != k2:
        raise RuntimeError(f"conv2d: weight {tuple(weight.shape)} does not match input channels {Ci} (square kernels only)")
    Ho, Wo = _conv_out(H, k, stride, padding, dilation), _conv_out(W, k, stride, padding, dilation)
    if Ho < 1 or Wo < 1:
        raise RuntimeError(f"conv2d: kernel {k} (dilation {dilation}) larger than the padded input {H}x{W}")
    y = torch.empty(N, Ho, Wo, Co, dtype=torch.float32, device=x.device)
    z = torch.empty_like(y) if act == ACT_GELU else _e0(x)
    ws = _ws(_query("hyb_conv2d_workspace", N, H, W, Ci, Co, k, stride, padding, dilation), x.device)
    lib.call("hyb_conv2d_fwd", x.data_ptr(), _f32c(weight).data_ptr(), _f32c(bias).data_ptr() if bias is not None else None, y.data_ptr(),
             z.data_ptr() if act == ACT_GELU else None, N, H, W, Ci, Co, k, stride, padding, dilation, act, ws.data_ptr(), ws.numel(), _stream())
    return y, z


def conv2d_fake(x, weight, bias, stride, padding, dilation, act):
    N, H, W, _ = x.shape
    k = weight.shape[2]
    y = x.new_empty((N, _conv_out(H, k, stride, padding, dilation), _conv_out(W, k, stride, padding, dilation), weight.shape[0]), dtype=torch.float32)
    return y, (torch.empty_like(y) if act == ACT_GELU else x.new_empty((0,), dtype=torch.float32))


def conv2d_bwd_op(dy: Tensor, x: Tensor, weight: Tensor, saved: Tensor, has_bias: bool, need_dx: bool, stride: int, padding: int, dilation: int,
                  act: int) -> Tuple[Tensor, Tensor, Tensor]:
    _require_cuda(dy, x)
    dy, x = _f32c(dy), _f32c(x)
    N, H, W, Ci = x.shape
    Co, k = weight.shape[0], weight.shape[2]
    dx = torch.empty_like(x) if need_dx else _e0(x)
    dw = torch.empty_like(weight, memory_format=torch.contiguous_format, dtype=torch.float32)
    db = torch.empty(Co if has_bias else 0, dtype=torch.float32, device=x.device)
    ws = _ws(_query("hyb_conv2d_bwd_workspace", N, H, W, Ci, Co, k, stride, padding, dilation), x.device)
    lib.call("hyb_conv2d_bwd", dy.data_ptr(), x.data_ptr(), _f32c(weight).data_ptr(), saved.data_ptr() if act != ACT_NONE else None,
             dx.data_ptr() if need_dx else None, dw.data_ptr(), db.data_ptr() if has_bias else None, N, H, W, Ci, Co, k, stride, padding, dilation, act,
             ws.data_ptr(), ws.numel(), _stream())
    return dx, dw, db


def conv2d_bwd_fake(dy, x, weight, saved, has_bias, need_dx, stride, padding, dilation, act):
    return ((torch.empty_like(x) if need_dx else x.new_empty((0,))), torch.empty_like(weight, memory_format=torch.contiguous_format),
            x.new_empty((weight.shape[0] if has_bias else 0,)))


class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, act):
        ctx.set_materialize_grads(False)
        with _below_autograd():
            y, z = torch.ops.hybrid.conv2d(x, weight, bias, stride, padding, dilation, act)
        ctx.save_for_backward(x, weight, z if act == ACT_GELU else y if act != ACT_NONE else _e0(x))
        ctx.cfg = (bias is not None, stride, padding, dilation, act)
        ctx.mark_non_differentiable(z)
        return y, z

    @staticmethod
    def backward(ctx, dy, unused):
        x, weight, saved = ctx.saved_tensors
        has_bias, stride, padding, dilation, act = ctx.cfg
        dx, dw, db = torch.ops.hybrid.conv2d_bwd(dy, x, weight, saved, has_bias, ctx.needs_input_grad[0], stride, padding, dilation, act)
        return (dx if ctx.needs_input_grad[0] else None), dw, (db if has_bias else None), None, None, None, None


def _bn_check(x, C):
    if x.dim() != 4 or C % 4 != 0 or C < 4 or C > 1024 or 256 % (C // 4) != 0:
        raise RuntimeError(f"bn2d: NHWC input with C in {{8, 16, 32, 64, 128, 256, 512, 1024}} (or 4) expected, got {tuple(x.shape)}")


def bn2d_op(x: Tensor, weight: Tensor, bias: Tensor, residual: Optional[Tensor], running_mean: Optional[Tensor], running_var: Optional[Tensor],
            training: bool, momentum: float, eps: float, relu: bool) -> Tuple[Tensor, Tensor]:
    """nn.BatchNorm2d on NHWC x (+ residual) (+ ReLU) -> (y, coef [4,C] = a, b, mean, invstd).  In training mode the running
    statistics are updated IN PLACE (they are module buffers, not differentiable)."""
    _require_cuda(x, weight, bias)
    x = _f32c(x)
    C = x.shape[-1]
    _bn_check(x, C)
    if not training and (running_mean is None or running_var is None):
        raise RuntimeError("bn2d: eval mode needs running statistics")
    if residual is not None:
        residual = _f32c(residual)
        if residual.shape != x.shape:
            raise RuntimeError(f"bn2d: residual {tuple(residual.shape)} vs input {tuple(x.shape)}")
    P = x.numel() // C
    y = torch.empty_like(x)
    coef = torch.empty(4, C, dtype=torch.float32, device=x.device)
    ws = _ws(_query("hyb_bn2d_workspace", P, C), x.device)
    lib.call("hyb_bn2d_fwd", x.data_ptr(), _f32c(weight).data_ptr(), _f32c(bias).data_ptr(), _opt_ptr(residual), y.data_ptr(), coef.data_ptr(),
             _opt_ptr(running_mean), _opt_ptr(running_var), P, C, float(eps), float(momentum), int(training), int(relu), ws.data_ptr(), ws.numel(),
             _stream())
    return y, coef


def bn2d_fake(x, weight, bias, residual, running_mean, running_var, training, momentum, eps, relu):
    return x.new_empty(x.shape, dtype=torch.float32), x.new_empty((4, x.shape[-1]), dtype=torch.float32)


def bn2d_bwd_op(dy: Tensor, x: Tensor, y: Tensor, weight: Tensor, coef: Tensor, training: bool, relu: bool,
                has_residual: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """-> (dx, dresidual (empty unless has_residual), dweight, dbias)"""
    _require_cuda(dy, x)
    dy, x = _f32c(dy), _f32c(x)
    C = x.shape[-1]
    _bn_check(x, C)
    P = x.numel() // C
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if has_residual else _e0(x)
    dg = torch.empty(C, dtype=torch.float32, device=x.device)
    db = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = _ws(_query("hyb_bn2d_workspace", P, C), x.device)
    lib.call("hyb_bn2d_bwd", dy.data_ptr(), x.data_ptr(), y.data_ptr() if (relu and has_residual) else None, _f32c(weight).data_ptr(), coef.data_ptr(), dx.data_ptr(),
             dres.data_ptr() if has_residual else None, dg.data_ptr(), db.data_ptr(), P, C, int(training), int(relu), ws.data_ptr(), ws.numel(),
             _stream())
    return dx, dres, dg, db


def bn2d_bwd_fake(dy, x, y, weight, coef, training, relu, has_residual):
    C = x.shape[-1]
    return torch.empty_like(x), (torch.empty_like(x) if has_residual else x.new_empty((0,))), x.new_empty((C,)), x.new_empty((C,))


class _Bn2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, training, momentum, eps, relu):
        ctx.set_materialize_grads(False)
        with _below_autograd():
            y, coef = torch.ops.hybrid.bn2d(x, weight, bias, residual, running_mean, running_var, training, momentum, eps, relu)
        ctx.save_for_backward(x, weight, coef, y if (relu and residual is not None) else _e0(x))     # without a residual the mask is recomputed from x
        ctx.cfg = (training, relu, residual is not None)
        ctx.mark_non_differentiable(coef)
        return y, coef

    @staticmethod
    def backward(ctx, dy, unused):
        x, weight, coef, y = ctx.saved_tensors
        training, relu, has_res = ctx.cfg
        dx, dres, dg, db = torch.ops.hybrid.bn2d_bwd(dy, x, y, weight, coef, training, relu, has_res)
        return dx, dg, db, (dres if has_res else None), None, None, None, None, None, None


def dropout2d_op(x: Tensor, p: float, seed: int, seed_inc: Optional[Tensor] = None) -> Tensor:
    """nn.Dropout2d(p) in train mode on NHWC x: whole (image, channel) planes are dropped."""
    _require_cuda(x)
    x = _f32c(x)
    N, H, W, C = x.shape
    y = torch.empty_like(x)
    lib.call("hyb_dropout2d", x.data_ptr(), y.data_ptr(), N, H * W, C, float(p), seed, _opt_ptr(seed_inc), _stream())
    return y


class _Dropout2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, seed_inc):
        ctx.set_materialize_grads(False)
        ctx.cfg = (p, seed)
        ctx.seed_inc = seed_inc
        with _below_autograd():
            return torch.ops.hybrid.dropout2d(x, p, seed, seed_inc)

    @staticmethod
    def backward(ctx, dy):
        return torch.ops.hybrid.dropout2d(dy, *ctx.cfg, ctx.seed_inc), None, None, None


_define("conv2d", "(Tensor x, Tensor weight, Tensor? bias, int stride, int padding, int dilation, int act) -> (Tensor, Tensor)", conv2d_op, conv2d_fake,
        _Conv2dFn.apply)
_define("conv2d_bwd", "(Tensor dy, Tensor x, Tensor weight, Tensor saved, bool has_bias, bool need_dx, int stride, int padding, int dilation, int act) "
        "-> (Tensor, Tensor, Tensor)", conv2d_bwd_op, conv2d_bwd_fake)
_define("bn2d", "(Tensor x, Tensor weight, Tensor bias, Tensor? residual, Tensor(a!)? running_mean, Tensor(b!)? running_var, bool training, "
        "float momentum, float eps, bool relu) -> (Tensor, Tensor)", bn2d_op, bn2d_fake, _Bn2dFn.apply)
_define("bn2d_bwd", "(Tensor dy, Tensor x, Tensor y, Tensor weight, Tensor coef, bool training, bool relu, bool has_residual) "
        "-> (Tensor, Tensor, Tensor, Tensor)", bn2d_bwd_op, bn2d_bwd_fake)
_define("dropout2d", "(Tensor x, float p, int seed, Tensor? seed_inc=None) -> Tensor", dropout2d_op, fct_dropout_fake,
        lambda x, p, seed, seed_inc=None: _Dropout2dFn.apply(x, p, seed, seed_inc))
_define("fct_conv", "(Tensor x, Tensor weight, Tensor? bias, int dilation, int act) -> (Tensor, Tensor)", fct_conv_op, fct_conv_fake, _FctConvFn.apply)
_define("fct_conv_bwd", "(Tensor dy, Tensor x, Tensor weight, Tensor saved, bool has_bias, bool need_dx, int dilation, int act) -> (Tensor, Tensor, Tensor)",
        fct_conv_bwd_op, fct_conv_bwd_fake)
_define("fct_qkv_proj", "(Tensor x, Tensor[] weights, Tensor[] biases, Tensor[] ln_weights, Tensor[] ln_biases, float eps) -> (Tensor, Tensor, Tensor)",
        fct_qkv_proj_op, fct_qkv_proj_fake, lambda x, ws, bs, gs, be, eps: _FctQkvFn.apply(x, eps, *ws, *bs, *gs, *be))
_define("fct_qkv_proj_bwd", "(Tensor x, Tensor[] weights, Tensor[] biases, Tensor[] ln_weights, Tensor[] dqs, float eps) -> Tensor[]",
        fct_qkv_proj_bwd_op, fct_qkv_proj_bwd_fake)
_define("fct_ln", "(Tensor x, Tensor weight, Tensor bias, float eps) -> Tensor", fct_ln_op, fct_ln_fake, _FctLnFn.apply)
_define("fct_ln_bwd", "(Tensor dy, Tensor x, Tensor weight, float eps) -> (Tensor, Tensor, Tensor)", fct_ln_bwd_op, fct_ln_bwd_fake)
_define("fct_mha", "(Tensor q, Tensor k, Tensor v, Tensor in_w, Tensor? in_b, Tensor out_w, Tensor? out_b, int heads) -> (Tensor, Tensor)", fct_mha_op,
        fct_mha_fake, _FctMhaFn.apply)
_define("fct_mha_bwd", "(Tensor dout, Tensor q, Tensor k, Tensor v, Tensor in_w, Tensor out_w, Tensor saved, int heads, bool has_in_b, bool has_out_b) "
        "-> Tensor[]", fct_mha_bwd_op, fct_mha_bwd_fake)
_define("fct_add", "(Tensor a, Tensor b) -> Tensor", fct_add_op, fct_add_fake, _FctAddFn.apply)
_define("fct_resample", "(Tensor x, int mode) -> Tensor", fct_resample_op, fct_resample_fake, _FctResampleFn.apply)
_define("fct_resample_bwd", "(Tensor dy, Tensor x, int mode) -> Tensor", fct_resample_bwd_op, fct_resample_bwd_fake)
_define("fct_concat", "(Tensor a, Tensor b) -> Tensor", fct_concat_op, fct_concat_fake, _FctConcatFn.apply)
_define("fct_concat_bwd", "(Tensor dy, int Ca, int Cb) -> (Tensor, Tensor)", fct_concat_bwd_op, fct_concat_bwd_fake)
_define("fct_dropout", "(Tensor x, float p, int seed, Tensor? seed_inc=None) -> Tensor", fct_dropout_op, fct_dropout_fake,
        lambda x, p, seed, seed_inc=None: _FctDropoutFn.apply(x, p, seed, seed_inc))
_define("dice_loss", "(Tensor pred, Tensor true, float smooth) -> Tensor", dice_loss_op, dice_loss_fake, _DiceFn.apply)
_define("dice_loss_bwd", "(Tensor dloss, Tensor pred, Tensor true, float smooth) -> Tensor", dice_loss_bwd_op, dice_loss_bwd_fake)
