"""The hot path as custom torch operators: ``torch.ops.hybrid.*`` (torch.library.custom_op + register_autograd +
register_fake) over the C ABI of include/hybrid_hip.h.

Every operator enqueues HIP kernels on torch's current stream through ctypes; tensors only provide device memory.
Forward operators return the tensors their backward needs as extra outputs (saved by ``setup_context``); each backward is
itself an operator (``hybrid::*_bwd``), so the whole path is visible to the dispatcher, has fake (meta) implementations for
shape inference and passes ``torch.library.opcheck``.  There is no eager/CPU implementation behind them: a CPU tensor raises.

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

from ._lib import HYB_BF16, HYB_F32, lib, ptr_array

_TORCH_DTYPE = {HYB_F32: torch.float32, HYB_BF16: torch.bfloat16}
_SEED_COUNTER = [0]
_SEED_MASK = 0x7FFFFFFFFFFFFFFF          # operator schemas carry ints as int64


def dtype_code(name):
    if isinstance(name, int) and not isinstance(name, bool) and name in (HYB_F32, HYB_BF16):
        return name
    if isinstance(name, str) and name in ("fp32", "float32") or name is torch.float32:
        return HYB_F32
    if isinstance(name, str) and name in ("bf16", "bfloat16") or name is torch.bfloat16:
        return HYB_BF16
    raise ValueError(f"compute dtype must be 'bf16' or 'fp32', got {name!r}")


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
    """Size queries are pure host functions of their integer arguments: ask the library once per shape."""
    return lib.query(name, *args)


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
@torch.library.custom_op("hybrid::nchw_to_nhwc", mutates_args=())
def nchw_to_nhwc_op(x: Tensor, dt: int, cp: int) -> Tensor:
    _require_cuda(x)
    x = x.contiguous().float()
    N, C, H, W = x.shape
    out = torch.empty(N, H, W, cp, dtype=_TORCH_DTYPE[dt], device=x.device)
    lib.call("hyb_nchw_to_nhwc", dt, x.data_ptr(), out.data_ptr(), N, C, H, W, cp, _stream())
    return out


@nchw_to_nhwc_op.register_fake
def _(x, dt, cp):
    N, C, H, W = x.shape
    return x.new_empty((N, H, W, cp), dtype=_TORCH_DTYPE[dt])


@torch.library.custom_op("hybrid::nhwc_to_nchw", mutates_args=())
def nhwc_to_nchw_op(x: Tensor, dt: int, C: int) -> Tensor:
    _require_cuda(x)
    x = x.contiguous()
    N, H, W, cp = x.shape
    out = torch.empty(N, C, H, W, dtype=torch.float32, device=x.device)
    lib.call("hyb_nhwc_to_nchw", dt, x.data_ptr(), out.data_ptr(), N, C, H, W, cp, _stream())
    return out


@nhwc_to_nchw_op.register_fake
def _(x, dt, C):
    N, H, W, cp = x.shape
    return x.new_empty((N, C, H, W), dtype=torch.float32)


def _nchw_to_nhwc_setup(ctx, inputs, output):
    ctx.dt, ctx.C = inputs[1], inputs[0].shape[1]


def _nchw_to_nhwc_bwd(ctx, g):
    return torch.ops.hybrid.nhwc_to_nchw(g, ctx.dt, ctx.C), None, None


def _nhwc_to_nchw_setup(ctx, inputs, output):
    ctx.dt, ctx.cp = inputs[1], inputs[0].shape[3]


def _nhwc_to_nchw_bwd(ctx, g):
    return torch.ops.hybrid.nchw_to_nhwc(g, ctx.dt, ctx.cp), None, None


nchw_to_nhwc_op.register_autograd(_nchw_to_nhwc_bwd, setup_context=_nchw_to_nhwc_setup)
nhwc_to_nchw_op.register_autograd(_nhwc_to_nchw_bwd, setup_context=_nhwc_to_nchw_setup)


def nchw_to_nhwc(x, dt, cp):
    return torch.ops.hybrid.nchw_to_nhwc(x, dt, cp)


def nhwc_to_nchw(x, dt, C):
    return torch.ops.hybrid.nhwc_to_nchw(x, dt, C)


@torch.library.custom_op("hybrid::cast", mutates_args=())
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


@cast_op.register_fake
def _(x, dt, to_t):
    return x.new_empty(x.shape, dtype=_TORCH_DTYPE[dt] if to_t else torch.float32)


def _cast_setup(ctx, inputs, output):
    ctx.dt, ctx.to_t = inputs[1], inputs[2]


def _cast_bwd(ctx, g):
    return torch.ops.hybrid.cast(g, ctx.dt, not ctx.to_t), None, None


cast_op.register_autograd(_cast_bwd, setup_context=_cast_setup)


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


@torch.library.custom_op("hybrid::convstage", mutates_args=())
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


@convstage_op.register_fake
def _(x, weight, gamma, beta, running_mean, running_var, training, momentum, eps, dt, first):
    N, H, W, Ci, Cip, Co, Cop = _convstage_dims(x, weight, first)
    tdt = _TORCH_DTYPE[dt]
    track = running_mean is not None and running_var is not None
    return (x.new_empty((N, H // 2, W // 2, Cop), dtype=tdt),
            x.new_empty((0,) if first else (N, H, W, Cop), dtype=tdt),
            x.new_empty((2, Cop), dtype=torch.float32), x.new_empty((2, Cop), dtype=torch.float32),
            x.new_empty((_query("hyb_convstage_packed_bwd_elems", int(first), Cip, Cop),), dtype=tdt),
            x.new_empty((2, Co) if training and track else (0,), dtype=torch.float32))


@torch.library.custom_op("hybrid::convstage_bwd", mutates_args=())
def convstage_bwd_op(dpooled: Tensor, x: Tensor, y_raw: Tensor, weight: Tensor, gamma: Tensor, scale_shift: Tensor, mean_invstd: Tensor,
                     packed_bwd: Tensor, training: bool, dt: int, first: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
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
    lib.call("hyb_convstage_bwd", dt, int(first), dpooled.data_ptr(), x.data_ptr(), None if first else y_raw.data_ptr(), weight.contiguous().data_ptr(),
             gamma.contiguous().data_ptr(), scale_shift.data_ptr(), mean_invstd.data_ptr(), int(training), N, H, W, Ci, Cip, Co, Cop,
             None if first else dx.data_ptr(), dw.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
             packed_bwd.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    return dx, dw, dgamma, dbeta


@convstage_bwd_op.register_fake
def _(dpooled, x, y_raw, weight, gamma, scale_shift, mean_invstd, packed_bwd, training, dt, first):
    N, H, W, Ci, Cip, Co, Cop = _convstage_dims(x, weight, first)
    return (x.new_empty((0,) if first else (N, H, W, Cip), dtype=_TORCH_DTYPE[dt]), torch.empty_like(weight, memory_format=torch.contiguous_format),
            x.new_empty((Co,), dtype=torch.float32), x.new_empty((Co,), dtype=torch.float32))


def _convstage_setup(ctx, inputs, output):
    x, weight, gamma, beta, rm, rv, training, momentum, eps, dt, first = inputs
    if first and ctx.needs_input_grad[0]:
        raise RuntimeError("the first conv stage does not compute a gradient for its input (the clip tensor): pass clips with "
                           "requires_grad=False (training data never needs one); a silent None would be wrong")
    pooled, y_raw, scale_shift, mean_invstd, packed_bwd, running_out = output
    ctx.mark_non_differentiable(running_out)
    ctx.save_for_backward(x, y_raw, weight, gamma, scale_shift, mean_invstd, packed_bwd)
    ctx.cfg = (bool(training), dt, bool(first))
    ctx.set_materialize_grads(False)


def _convstage_bwd(ctx, dpooled, *unused):
    x, y_raw, weight, gamma, scale_shift, mean_invstd, packed_bwd = ctx.saved_tensors
    training, dt, first = ctx.cfg
    if dpooled is None:
        return (None,) * 11
    dx, dw, dgamma, dbeta = torch.ops.hybrid.convstage_bwd(dpooled, x, y_raw, weight.detach(), gamma.detach(), scale_shift, mean_invstd,
                                                           packed_bwd, training, dt, first)
    return (None if first else dx, dw, dgamma, dbeta) + (None,) * 7


convstage_op.register_autograd(_convstage_bwd, setup_context=_convstage_setup)


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
@torch.library.custom_op("hybrid::token", mutates_args=())
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


@token_op.register_fake
def _(x, weight, bias, dt):
    N, Hh, Ww, Cp = x.shape
    return x.new_empty((N, weight.shape[0]), dtype=_TORCH_DTYPE[dt]), x.new_empty((N, Cp), dtype=_TORCH_DTYPE[dt])


@torch.library.custom_op("hybrid::token_bwd", mutates_args=())
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


@token_bwd_op.register_fake
def _(dtok, feat, weight, Hh, Ww, has_bias, dt):
    N, Cp = feat.shape
    return (feat.new_empty((N, Hh, Ww, Cp)), torch.empty_like(weight, memory_format=torch.contiguous_format),
            feat.new_empty((weight.shape[0] if has_bias else 0,), dtype=torch.float32))


def _token_setup(ctx, inputs, output):
    x, weight, bias, dt = inputs
    ctx.save_for_backward(output[1], weight)
    ctx.cfg = (x.shape[1], x.shape[2], bias is not None, dt)
    ctx.set_materialize_grads(False)


def _token_bwd(ctx, dtok, dfeat_unused):
    feat, weight = ctx.saved_tensors
    Hh, Ww, has_bias, dt = ctx.cfg
    if dtok is None:
        return None, None, None, None
    dx, dw, db = torch.ops.hybrid.token_bwd(dtok, feat, weight.detach(), Hh, Ww, has_bias, dt)
    return dx, dw, (db if has_bias else None), None


token_op.register_autograd(_token_bwd, setup_context=_token_setup)


def token(x, weight, bias, dt):
    return torch.ops.hybrid.token(x, weight, bias, dt)[0]


# ---------------------------------------------------------------------------------------------
# TransformerEncoder.forward (all layers) -- TransformerEncoder.pyc src L110-126
# ---------------------------------------------------------------------------------------------
def _check_attention_limits(S, D, H):
    if S > 64:
        raise RuntimeError(f"temporal attention kernel supports T <= 64 tokens per clip (got {S})")
    if D % H != 0 or (D // H) % 8 != 0 or D // H > 128:
        raise RuntimeError(f"temporal attention kernel supports head widths that are multiples of 8 up to 128 (got {D}/{H})")


@torch.library.custom_op("hybrid::encoder", mutates_args=())
def encoder_op(x: Tensor, mask: Optional[Tensor], params: Sequence[Tensor], dt: int, hid: int, L: int, H: int, attn_p: float,
               layer_p: float, seed: int) -> Tuple[Tensor, Tensor]:
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
             B, S, D, hid, L, H, float(attn_p), float(layer_p), seed, _stream())
    return out, saved


@encoder_op.register_fake
def _(x, mask, params, dt, hid, L, H, attn_p, layer_p, seed):
    B, S, D = x.shape
    return torch.empty_like(x, memory_format=torch.contiguous_format), x.new_empty((max(_query("hyb_encoder_saved_bytes", dt, B, S, D, hid, L, H), 256),),
                                                                                    dtype=torch.uint8)


@torch.library.custom_op("hybrid::encoder_bwd", mutates_args=())
def encoder_bwd_op(dout: Tensor, mask: Optional[Tensor], params: Sequence[Tensor], saved: Tensor, dt: int, hid: int, L: int, H: int,
                   attn_p: float, layer_p: float, seed: int) -> List[Tensor]:
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
             float(layer_p), seed, ws.data_ptr(), ws.numel(), _stream())
    return [dx] + grads


@encoder_bwd_op.register_fake
def _(dout, mask, params, saved, dt, hid, L, H, attn_p, layer_p, seed):
    return [torch.empty_like(dout, memory_format=torch.contiguous_format)] + [torch.empty_like(p, memory_format=torch.contiguous_format) for p in params]


def _encoder_setup(ctx, inputs, output):
    x, mask, params, dt, hid, L, H, attn_p, layer_p, seed = inputs
    ctx.save_for_backward(output[1], *([mask] if mask is not None else []), *params)
    ctx.cfg = (mask is not None, dt, hid, L, H, attn_p, layer_p, seed)
    ctx.set_materialize_grads(False)


def _encoder_bwd(ctx, dout, dsaved_unused):
    has_mask, dt, hid, L, H, attn_p, layer_p, seed = ctx.cfg
    saved, *rest = ctx.saved_tensors
    mask = rest.pop(0) if has_mask else None
    if dout is None:
        return (None, None, [None] * len(rest)) + (None,) * 7
    res = torch.ops.hybrid.encoder_bwd(dout, mask, [p.detach() for p in rest], saved, dt, hid, L, H, attn_p, layer_p, seed)
    return (res[0], None, list(res[1:])) + (None,) * 7


encoder_op.register_autograd(_encoder_bwd, setup_context=_encoder_setup)


def encoder(x, mask, params, dt, hid, L, H, attn_p, layer_p, seed):
    B, S, _ = x.shape
    return torch.ops.hybrid.encoder(x, check_mask(mask, B, S, x.device), list(params), dt, hid, L, H, float(attn_p), float(layer_p), seed)[0]


# ---------------------------------------------------------------------------------------------
# standalone MultiheadAttention.forward(q, k, v, mask) -- src L67-89
# ---------------------------------------------------------------------------------------------
@torch.library.custom_op("hybrid::mha", mutates_args=())
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
    probs = torch.empty(B * H, S, S, dtype=torch.float32, device=dev)
    st = _stream()
    for src, W_, b_, dst in ((q_in, ps[0], ps[1], q), (k_in, ps[2], ps[3], k), (v_in, ps[4], ps[5], v)):
        lib.call("hyb_linear_fwd", dt, src.data_ptr(), D, W_.data_ptr(), b_.data_ptr(), dst.data_ptr(), M, D, D, 1, st)
    lib.call("hyb_attention_fwd", dt, q.data_ptr(), k.data_ptr(), v.data_ptr(), _opt_ptr(mask), a.data_ptr(),
             probs.data_ptr(), B, S, D, H, float(p_drop), seed, st)
    lib.call("hyb_linear_fwd", dt, a.data_ptr(), D, ps[6].data_ptr(), ps[7].data_ptr(), out.data_ptr(), M, D, D, 0, st)
    return out, q, k, v, a, probs


@mha_op.register_fake
def _(q_in, k_in, v_in, mask, params, dt, H, p_drop, seed):
    B, S, D = q_in.shape
    e = lambda: q_in.new_empty((B, S, D), dtype=_TORCH_DTYPE[dt])
    return e(), e(), e(), e(), e(), q_in.new_empty((B * H, S, S), dtype=torch.float32)


@torch.library.custom_op("hybrid::mha_bwd", mutates_args=())
def mha_bwd_op(dout: Tensor, q_in: Tensor, k_in: Tensor, v_in: Tensor, q: Tensor, k: Tensor, v: Tensor, a: Tensor, probs: Tensor,
               params: Sequence[Tensor], dt: int, H: int, p_drop: float, seed: int) -> List[Tensor]:
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
    lib.call("hyb_attention_bwd", dt, q.data_ptr(), k.data_ptr(), v.data_ptr(), probs.data_ptr(), da.data_ptr(), dq.data_ptr(),
             dk.data_ptr(), dv.data_ptr(), B, S, D, H, float(p_drop), seed, st)
    for src, y, dy, dsrc, iw in ((q_in, q, dq, dqi, 0), (k_in, k, dk, dki, 2), (v_in, v, dv, dvi, 4)):
        lib.call("hyb_linear_bwd", dt, src.contiguous().data_ptr(), D, ps[iw].data_ptr(), y.data_ptr(), dy.data_ptr(), dsrc.data_ptr(), 0,
                 grads[iw].data_ptr(), grads[iw + 1].data_ptr(), M, D, D, 1, ws.data_ptr(), ws.numel(), st)
    return [dqi, dki, dvi] + grads


@mha_bwd_op.register_fake
def _(dout, q_in, k_in, v_in, q, k, v, a, probs, params, dt, H, p_drop, seed):
    return [torch.empty_like(q) for _ in range(3)] + [torch.empty_like(p, memory_format=torch.contiguous_format) for p in params]


def _mha_setup(ctx, inputs, output):
    q_in, k_in, v_in, mask, params, dt, H, p_drop, seed = inputs
    out, q, k, v, a, probs = output
    ctx.save_for_backward(q_in, k_in, v_in, q, k, v, a, probs, *params)
    ctx.cfg = (dt, H, p_drop, seed)
    ctx.set_materialize_grads(False)


def _mha_bwd(ctx, dout, *unused):
    q_in, k_in, v_in, q, k, v, a, probs, *ps = ctx.saved_tensors
    dt, H, p_drop, seed = ctx.cfg
    if dout is None:
        return (None, None, None, None, [None] * len(ps), None, None, None, None)
    res = torch.ops.hybrid.mha_bwd(dout, q_in, k_in, v_in, q, k, v, a, probs, [p.detach() for p in ps], dt, H, p_drop, seed)
    return (res[0], res[1], res[2], None, list(res[3:]), None, None, None, None)


mha_op.register_autograd(_mha_bwd, setup_context=_mha_setup)


def mha(q, k, v, mask, params, dt, H, p_drop, seed):
    B, S, _ = q.shape
    return torch.ops.hybrid.mha(q, k, v, check_mask(mask, B, S, q.device), list(params), dt, H, float(p_drop), seed)[0]


# ---------------------------------------------------------------------------------------------
# head (mean over T + Linear) and cross-entropy
# ---------------------------------------------------------------------------------------------
@torch.library.custom_op("hybrid::head", mutates_args=())
def head_op(x: Tensor, weight: Tensor, bias: Optional[Tensor], dt: int) -> Tensor:
    _require_cuda(x, weight)
    x = x.contiguous()
    B, S, D = x.shape
    C = weight.shape[0]
    logits = torch.empty(B, C, dtype=torch.float32, device=x.device)
    lib.call("hyb_head_fwd", dt, x.data_ptr(), weight.contiguous().data_ptr(), bias.contiguous().data_ptr() if bias is not None else None,
             logits.data_ptr(), B, S, D, C, _stream())
    return logits


@head_op.register_fake
def _(x, weight, bias, dt):
    return x.new_empty((x.shape[0], weight.shape[0]), dtype=torch.float32)


@torch.library.custom_op("hybrid::head_bwd", mutates_args=())
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


@head_bwd_op.register_fake
def _(dlogits, x, weight, has_bias, dt):
    return (torch.empty_like(x), torch.empty_like(weight, memory_format=torch.contiguous_format),
            x.new_empty((weight.shape[0] if has_bias else 0,), dtype=torch.float32))


def _head_setup(ctx, inputs, output):
    x, weight, bias, dt = inputs
    ctx.save_for_backward(x.contiguous(), weight)
    ctx.cfg = (bias is not None, dt)


def _head_bwd(ctx, dlogits):
    x, weight = ctx.saved_tensors
    has_bias, dt = ctx.cfg
    dx, dw, db = torch.ops.hybrid.head_bwd(dlogits, x, weight.detach(), has_bias, dt)
    return dx, dw, (db if has_bias else None), None


head_op.register_autograd(_head_bwd, setup_context=_head_setup)


def head(x, weight, bias, dt):
    return torch.ops.hybrid.head(x, weight, bias, dt)


@torch.library.custom_op("hybrid::cross_entropy", mutates_args=())
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


@cross_entropy_op.register_fake
def _(logits, target):
    return logits.new_empty((), dtype=torch.float32)


@torch.library.custom_op("hybrid::cross_entropy_bwd", mutates_args=())
def cross_entropy_bwd_op(dloss: Tensor, logits: Tensor, target: Tensor) -> Tensor:
    _require_cuda(dloss, logits)
    logits = logits.contiguous().float()
    target = target.contiguous().to(torch.int64)
    B, C = logits.shape
    dl = dloss.contiguous().float().reshape(1)
    dlogits = torch.empty_like(logits)
    lib.call("hyb_cross_entropy_bwd", logits.data_ptr(), target.data_ptr(), dl.data_ptr(), dlogits.data_ptr(), B, C, _stream())
    return dlogits


@cross_entropy_bwd_op.register_fake
def _(dloss, logits, target):
    return logits.new_empty(logits.shape, dtype=torch.float32)


def _ce_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _ce_bwd(ctx, dloss):
    logits, target = ctx.saved_tensors
    return torch.ops.hybrid.cross_entropy_bwd(dloss, logits, target), None


cross_entropy_op.register_autograd(_ce_bwd, setup_context=_ce_setup)


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


@torch.library.custom_op("hybrid::backbone", mutates_args=())
def backbone_op(x: Tensor, weights: Sequence[Tensor], gammas: Sequence[Tensor], betas: Sequence[Tensor], running_means: Sequence[Tensor],
                running_vars: Sequence[Tensor], training: bool, momentum: float, eps: float, dt: int) -> List[Tensor]:
    """All conv stages of the backbone on NCHW fp32 frames x [N, C_in <= 4, H, W].
    -> [pooled_S, then per stage s: y_raw_s, pooled_s (s < S-1 only), scale_shift_s, mean_invstd_s, packed_bwd_s, running_out_s]
    (functional BatchNorm: running_out_s [2, C_s] holds the updated statistics in training mode, empty otherwise)."""
    _require_cuda(x, *weights)
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
        y_raw = torch.empty(0 if s == 0 else (N, h, w_, Cop), dtype=tdt, device=dev)
        pooled = torch.empty(N, h // 2, w_ // 2, Cop, dtype=tdt, device=dev)
        ss = torch.empty(2, Cop, dtype=torch.float32, device=dev)
        mi = torch.empty(2, Cop, dtype=torch.float32, device=dev)
        pk = torch.empty(_query("hyb_convstage_packed_bwd_elems", int(s == 0), Cip, Cop), dtype=tdt, device=dev)
        ro = torch.empty((2, Co) if training else (0,), dtype=torch.float32, device=dev)
        per_stage.append((y_raw, pooled, ss, mi, pk, ro))
        params += [weights[s].contiguous().data_ptr(), gammas[s].contiguous().data_ptr(), betas[s].contiguous().data_ptr(),
                   running_means[s].contiguous().data_ptr(), running_vars[s].contiguous().data_ptr()]
        outs += [None if s == 0 else y_raw.data_ptr(), pooled.data_ptr(), ss.data_ptr(), mi.data_ptr(), pk.data_ptr(), ro.data_ptr() if training else None]
    ch = _int_array(chans)
    ws = _ws(_query("hyb_backbone_fwd_workspace", dt, S, tuple(chans)), dev)
    lib.call("hyb_backbone_fwd", dt, S, ch, x.data_ptr(), ptr_array(params), int(training), float(momentum), float(eps), N, H, W,
             ptr_array(outs), ws.data_ptr(), ws.numel(), _stream())
    res = [pooled]
    for s, (y_raw, p, ss, mi, pk, ro) in enumerate(per_stage):
        res += [y_raw] + ([p] if s < S - 1 else []) + [ss, mi, pk, ro]
    return res


@backbone_op.register_fake
def _(x, weights, gammas, betas, running_means, running_vars, training, momentum, eps, dt):
    S = len(weights)
    N, H, W, chans, dims = _backbone_geometry(x, weights)
    tdt = _TORCH_DTYPE[dt]
    res, last = [], None
    for s in range(S):
        h, w_, Cip, Cop = dims[s]
        last = x.new_empty((N, h // 2, w_ // 2, Cop), dtype=tdt)
        res += [x.new_empty((0,) if s == 0 else (N, h, w_, Cop), dtype=tdt)] + ([last] if s < S - 1 else []) + [
            x.new_empty((2, Cop), dtype=torch.float32), x.new_empty((2, Cop), dtype=torch.float32),
            x.new_empty((_query("hyb_convstage_packed_bwd_elems", int(s == 0), Cip, Cop),), dtype=tdt),
            x.new_empty((2, chans[s + 1]) if training else (0,), dtype=torch.float32)]
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


@torch.library.custom_op("hybrid::backbone_bwd", mutates_args=())
def backbone_bwd_op(dpooled: Tensor, x: Tensor, weights: Sequence[Tensor], gammas: Sequence[Tensor], saved: Sequence[Tensor], training: bool,
                    dt: int) -> List[Tensor]:
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
        sptr += [None if s == 0 else sv[0].data_ptr(), None if s == 0 else sv[1].data_ptr(), sv[2].data_ptr(), sv[3].data_ptr(), sv[4].data_ptr()]
    ws = _ws(_query("hyb_backbone_bwd_workspace", dt, S, tuple(chans), N, H, W), dev)
    lib.call("hyb_backbone_bwd", dt, S, _int_array(chans), dpooled.data_ptr(), x.data_ptr(), ptr_array(pptr), ptr_array(sptr), int(training),
             N, H, W, ptr_array(gptr), ws.data_ptr(), ws.numel(), _stream())
    return grads


@backbone_bwd_op.register_fake
def _(dpooled, x, weights, gammas, saved, training, dt):
    res = []
    for w in weights:
        res += [torch.empty_like(w, memory_format=torch.contiguous_format), x.new_empty((w.shape[0],), dtype=torch.float32),
                x.new_empty((w.shape[0],), dtype=torch.float32)]
    return res


def _backbone_setup(ctx, inputs, output):
    x, weights, gammas, betas, rms, rvs, training, momentum, eps, dt = inputs
    if ctx.needs_input_grad[0]:
        raise RuntimeError("the first conv stage does not compute a gradient for its input (the clip tensor): pass clips with "
                           "requires_grad=False (training data never needs one); a silent None would be wrong")
    S = len(weights)
    st = _backbone_unpack(output, S)
    saved = []
    for s in range(S):
        y_raw, p, ss, mi, pk, ro = st[s]
        saved += [y_raw, (st[s - 1][1] if s > 0 else y_raw), ss, mi, pk]
    ctx.save_for_backward(x, *weights, *gammas, *saved)
    ctx.cfg = (S, bool(training), dt)
    ctx.set_materialize_grads(False)


def _backbone_bwd(ctx, grads_out):
    S, training, dt = ctx.cfg
    t = ctx.saved_tensors
    x, weights, gammas, saved = t[0], list(t[1:1 + S]), list(t[1 + S:1 + 2 * S]), list(t[1 + 2 * S:])
    dpooled = grads_out[0]
    none = [None] * S
    if dpooled is None:
        return (None, none, none, none, none, none, None, None, None, None)
    g = torch.ops.hybrid.backbone_bwd(dpooled, x, [w.detach() for w in weights], [v.detach() for v in gammas], saved, training, dt)
    return (None, [g[3 * s] for s in range(S)], [g[3 * s + 1] for s in range(S)], [g[3 * s + 2] for s in range(S)], none, none, None, None, None, None)


backbone_op.register_autograd(_backbone_bwd, setup_context=_backbone_setup)


def backbone(x, stages, training, dt):
    """stages: list of (conv.weight, bn) pairs.  Applies nn.BatchNorm2d's buffer updates after the functional operator."""
    bns = [bn for _, bn in stages]
    res = torch.ops.hybrid.backbone(x, [w for w, _ in stages], [bn.weight for bn in bns], [bn.bias for bn in bns],
                                    [bn.running_mean for bn in bns], [bn.running_var for bn in bns], bool(training), float(bns[0].momentum),
                                    float(bns[0].eps), int(dt))
    if training:
        st = _backbone_unpack(res, len(stages))
        commit_running_stats([(bn.running_mean, bn.running_var, bn.num_batches_tracked, st[s][5]) for s, bn in enumerate(bns)])
    return res[0]


@torch.library.custom_op("hybrid::temporal", mutates_args=())
def temporal_op(h: Tensor, token_w: Tensor, token_b: Tensor, enc_params: Sequence[Tensor], head_w: Tensor, head_b: Tensor, mask: Optional[Tensor],
                B: int, dt: int, hid: int, L: int, H: int, attn_p: float, layer_p: float, seed: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """h [B*S, Hh, Ww, Cp] T (last pooled map) -> (logits [B, classes] fp32, feat, enc_saved, enc_out); the last three are saved for backward."""
    _require_cuda(h, token_w, head_w, *enc_params)
    h = h.contiguous()
    N, Hh, Ww, Cp = h.shape
    S = N // B
    D, C = token_w.shape
    classes = head_w.shape[0]
    _check_attention_limits(S, D, H)
    dev, tdt = h.device, _TORCH_DTYPE[dt]
    feat = torch.empty(N, Cp, dtype=tdt, device=dev)
    tok = torch.empty(B, S, D, dtype=tdt, device=dev)
    enc_out = torch.empty(B, S, D, dtype=tdt, device=dev)
    saved = _ws(_query("hyb_encoder_saved_bytes", dt, B, S, D, hid, L, H), dev)
    logits = torch.empty(B, classes, dtype=torch.float32, device=dev)
    ps = [p.contiguous() for p in enc_params]
    lib.call("hyb_temporal_fwd", dt, h.data_ptr(), token_w.contiguous().data_ptr(), token_b.contiguous().data_ptr(), ptr_array([p.data_ptr() for p in ps]),
             head_w.contiguous().data_ptr(), head_b.contiguous().data_ptr(), _opt_ptr(mask), feat.data_ptr(), tok.data_ptr(), saved.data_ptr(),
             enc_out.data_ptr(), logits.data_ptr(), B, S, Hh * Ww, C, Cp, D, hid, L, H, classes, float(attn_p), float(layer_p), seed, _stream())
    return logits, feat, saved, enc_out


@temporal_op.register_fake
def _(h, token_w, token_b, enc_params, head_w, head_b, mask, B, dt, hid, L, H, attn_p, layer_p, seed):
    N, Hh, Ww, Cp = h.shape
    S, D = N // B, token_w.shape[0]
    tdt = _TORCH_DTYPE[dt]
    return (h.new_empty((B, head_w.shape[0]), dtype=torch.float32), h.new_empty((N, Cp), dtype=tdt),
            h.new_empty((max(_query("hyb_encoder_saved_bytes", dt, B, S, D, hid, L, H), 256),), dtype=torch.uint8), h.new_empty((B, S, D), dtype=tdt))


@torch.library.custom_op("hybrid::temporal_bwd", mutates_args=())
def temporal_bwd_op(dlogits: Tensor, token_w: Tensor, enc_params: Sequence[Tensor], head_w: Tensor, mask: Optional[Tensor], feat: Tensor,
                    saved: Tensor, enc_out: Tensor, Hh: int, Ww: int, dt: int, hid: int, L: int, H: int, attn_p: float, layer_p: float,
                    seed: int) -> List[Tensor]:
    """-> [dh, dtoken_w, dtoken_b, dhead_w, dhead_b, denc_param_0, ...]"""
    _require_cuda(dlogits, feat)
    B, S, D = enc_out.shape
    N, Cp = feat.shape
    C = token_w.shape[1]
    classes = head_w.shape[0]
    dev, tdt = feat.device, _TORCH_DTYPE[dt]
    dlogits = dlogits.contiguous().float()
    ps = [p.contiguous() for p in enc_params]
    grads = [torch.empty_like(p) for p in ps]
    dh = torch.empty(N, Hh, Ww, Cp, dtype=tdt, device=dev)
    dtw = torch.empty_like(token_w, memory_format=torch.contiguous_format)
    dtb = torch.empty(D, dtype=torch.float32, device=dev)
    dhw = torch.empty_like(head_w, memory_format=torch.contiguous_format)
    dhb = torch.empty(classes, dtype=torch.float32, device=dev)
    ws = _ws(_query("hyb_temporal_bwd_workspace", dt, B, S, Hh * Ww, Cp, D, hid, L, H), dev)
    lib.call("hyb_temporal_bwd", dt, dlogits.data_ptr(), token_w.contiguous().data_ptr(), ptr_array([p.data_ptr() for p in ps]),
             head_w.contiguous().data_ptr(), _opt_ptr(mask), feat.data_ptr(), saved.data_ptr(), enc_out.data_ptr(), dtw.data_ptr(), dtb.data_ptr(),
             ptr_array([g.data_ptr() for g in grads]), dhw.data_ptr(), dhb.data_ptr(), dh.data_ptr(), B, S, Hh * Ww, C, Cp, D, hid, L, H, classes,
             float(attn_p), float(layer_p), seed, ws.data_ptr(), ws.numel(), _stream())
    return [dh, dtw, dtb, dhw, dhb] + grads


@temporal_bwd_op.register_fake
def _(dlogits, token_w, enc_params, head_w, mask, feat, saved, enc_out, Hh, Ww, dt, hid, L, H, attn_p, layer_p, seed):
    N, Cp = feat.shape
    c = lambda t: torch.empty_like(t, memory_format=torch.contiguous_format)
    return [feat.new_empty((N, Hh, Ww, Cp)), c(token_w), feat.new_empty((token_w.shape[0],), dtype=torch.float32), c(head_w),
            feat.new_empty((head_w.shape[0],), dtype=torch.float32)] + [c(p) for p in enc_params]


def _temporal_setup(ctx, inputs, output):
    h, token_w, token_b, enc_params, head_w, head_b, mask, B, dt, hid, L, H, attn_p, layer_p, seed = inputs
    logits, feat, saved, enc_out = output
    ctx.save_for_backward(token_w, head_w, feat, saved, enc_out, *([mask] if mask is not None else []), *enc_params)
    ctx.cfg = (mask is not None, h.shape[1], h.shape[2], dt, hid, L, H, attn_p, layer_p, seed)
    ctx.set_materialize_grads(False)


def _temporal_bwd(ctx, dlogits, *unused):
    has_mask, Hh, Ww, dt, hid, L, H, attn_p, layer_p, seed = ctx.cfg
    token_w, head_w, feat, saved, enc_out, *rest = ctx.saved_tensors
    mask = rest.pop(0) if has_mask else None
    if dlogits is None:
        return (None, None, None, [None] * len(rest), None, None) + (None,) * 9
    g = torch.ops.hybrid.temporal_bwd(dlogits, token_w.detach(), [p.detach() for p in rest], head_w.detach(), mask, feat, saved, enc_out, Hh, Ww,
                                      dt, hid, L, H, attn_p, layer_p, seed)
    return (g[0], g[1], g[2], list(g[5:]), g[3], g[4]) + (None,) * 9


temporal_op.register_autograd(_temporal_bwd, setup_context=_temporal_setup)


def temporal(h, token_w, token_b, enc_params, head_w, head_b, mask, B, dt, hid, L, H, attn_p, layer_p, seed):
    S = h.shape[0] // B
    return torch.ops.hybrid.temporal(h, token_w, token_b, list(enc_params), head_w, head_b, check_mask(mask, B, S, h.device), B, dt, hid, L, H,
                                     float(attn_p), float(layer_p), seed)[0]
