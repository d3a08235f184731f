"""torch.autograd.Function wrappers around the C ABI (include/hybrid_hip.h).

Every function here enqueues HIP kernels on torch's current stream through ctypes; tensors only
provide device memory.  There is no eager/CPU implementation behind them: a CPU tensor raises."""
import torch

from ._lib import HYB_BF16, HYB_F32, lib, ptr_array

_TORCH_DTYPE = {HYB_F32: torch.float32, HYB_BF16: torch.bfloat16}
_SEED_COUNTER = [0]


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


def next_seed():
    """Deterministic under torch.manual_seed; no device sync."""
    _SEED_COUNTER[0] += 1
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _SEED_COUNTER[0] * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF


# ---------------------------------------------------------------------------------------------
# layout / cast
# ---------------------------------------------------------------------------------------------
class _NchwToNhwc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dt, cp):
        _require_cuda(x)
        x = x.contiguous().float()
        N, C, H, W = x.shape
        out = torch.empty(N, H, W, cp, dtype=_TORCH_DTYPE[dt], device=x.device)
        lib.call("hyb_nchw_to_nhwc", dt, x.data_ptr(), out.data_ptr(), N, C, H, W, cp, _stream())
        ctx.dt, ctx.C = dt, C
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        N, H, W, cp = g.shape
        out = torch.empty(N, ctx.C, H, W, dtype=torch.float32, device=g.device)
        lib.call("hyb_nhwc_to_nchw", ctx.dt, g.data_ptr(), out.data_ptr(), N, ctx.C, H, W, cp, _stream())
        return out, None, None


class _NhwcToNchw(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dt, C):
        _require_cuda(x)
        x = x.contiguous()
        N, H, W, cp = x.shape
        out = torch.empty(N, C, H, W, dtype=torch.float32, device=x.device)
        lib.call("hyb_nhwc_to_nchw", dt, x.data_ptr(), out.data_ptr(), N, C, H, W, cp, _stream())
        ctx.dt, ctx.cp = dt, cp
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().float()
        N, C, H, W = g.shape
        out = torch.empty(N, H, W, ctx.cp, dtype=_TORCH_DTYPE[ctx.dt], device=g.device)
        lib.call("hyb_nchw_to_nhwc", ctx.dt, g.data_ptr(), out.data_ptr(), N, C, H, W, ctx.cp, _stream())
        return out, None, None


def nchw_to_nhwc(x, dt, cp):
    return _NchwToNhwc.apply(x, dt, cp)


def nhwc_to_nchw(x, dt, C):
    return _NhwcToNchw.apply(x, dt, C)


class _Cast(torch.autograd.Function):
    """fp32 <-> T with the library's own cast kernels (differentiable)."""

    @staticmethod
    def forward(ctx, x, dt, to_t):
        _require_cuda(x)
        x = x.contiguous()
        ctx.dt, ctx.to_t = dt, to_t
        if to_t:
            out = torch.empty(x.shape, dtype=_TORCH_DTYPE[dt], device=x.device)
            lib.call("hyb_cast_from_f32", dt, x.float().data_ptr(), out.data_ptr(), x.numel(), _stream())
        else:
            out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            lib.call("hyb_cast_to_f32", dt, x.data_ptr(), out.data_ptr(), x.numel(), _stream())
        return out

    @staticmethod
    def backward(ctx, g):
        return _Cast.apply(g, ctx.dt, not ctx.to_t), None, None


def to_compute(x, dt):
    return _Cast.apply(x, dt, True)


def to_f32(x, dt):
    return _Cast.apply(x, dt, False)


# ---------------------------------------------------------------------------------------------
# conv stage: Conv3x3 -> BN -> ReLU -> MaxPool  (UNet.py:58-60, UNet.py:13)
# ---------------------------------------------------------------------------------------------
class ConvStageFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, dt, first):
        _require_cuda(x, weight)
        x = x.contiguous()
        Co, Ci = weight.shape[0], weight.shape[1]
        Cop = pad_channels(Co)
        if first:
            N, _, H, W = x.shape
            Cip = 0
            if Ci > 4:
                raise RuntimeError('first-stage kernel supports in_channels <= 4')
        else:
            N, H, W, Cip = x.shape
        if H < 2 or W < 2:
            raise RuntimeError(f"conv stage needs H, W >= 2 (got {H}x{W})")
        dev = x.device
        tdt = _TORCH_DTYPE[dt]
        # stage 1 recomputes its conv in backward: no full-resolution buffer is kept
        y_raw = torch.empty(8, dtype=tdt, device=dev) if first else torch.empty(N, H, W, Cop, dtype=tdt, device=dev)
        pooled = torch.empty(N, H // 2, W // 2, Cop, dtype=tdt, device=dev)
        scale_shift = torch.empty(2, Cop, dtype=torch.float32, device=dev)
        mean_invstd = torch.empty(2, Cop, dtype=torch.float32, device=dev)
        ws_bytes = lib.query("hyb_convstage_fwd_workspace", dt, int(first), Cip, Cop)
        ws = _ws(ws_bytes, dev)
        w = weight.detach().contiguous()
        packed_bwd = torch.empty(lib.query("hyb_convstage_packed_bwd_elems", int(first), Cip, Cop), dtype=tdt, device=dev)
        lib.call("hyb_convstage_fwd", dt, int(first), x.data_ptr(), w.data_ptr(), gamma.detach().contiguous().data_ptr(),
                 beta.detach().contiguous().data_ptr(), running_mean.data_ptr(), running_var.data_ptr(),
                 nbt.data_ptr() if nbt is not None else None, int(training), float(momentum), float(eps),
                 N, H, W, Ci, Cip, Co, Cop, y_raw.data_ptr(), pooled.data_ptr(), scale_shift.data_ptr(), mean_invstd.data_ptr(),
                 packed_bwd.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
        ctx.save_for_backward(x, y_raw, scale_shift, mean_invstd, w, gamma.detach().contiguous(), packed_bwd)
        ctx.cfg = (dt, int(first), int(training), N, H, W, Ci, Cip, Co, Cop)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        x, y_raw, scale_shift, mean_invstd, w, gamma, packed_bwd = ctx.saved_tensors
        dt, first, training, N, H, W, Ci, Cip, Co, Cop = ctx.cfg
        dev = x.device
        dpooled = dpooled.contiguous()
        dx = None if first else torch.empty(N, H, W, Cip, dtype=_TORCH_DTYPE[dt], device=dev)
        dw = torch.empty_like(w)
        dgamma = torch.empty(Co, dtype=torch.float32, device=dev)
        dbeta = torch.empty(Co, dtype=torch.float32, device=dev)
        ws_bytes = lib.query("hyb_convstage_bwd_workspace", dt, first, N, H, W, Cip, Cop)
        ws = _ws(ws_bytes, dev)
        lib.call("hyb_convstage_bwd", dt, first, dpooled.data_ptr(), x.data_ptr(), y_raw.data_ptr(), w.data_ptr(), gamma.data_ptr(),
                 scale_shift.data_ptr(), mean_invstd.data_ptr(), training, N, H, W, Ci, Cip, Co, Cop,
                 dx.data_ptr() if dx is not None else None, dw.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                 packed_bwd.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
        return dx, dw, dgamma, dbeta, None, None, None, None, None, None, None, None


# ---------------------------------------------------------------------------------------------
# frame token: global average pool + Linear(C, d)
# ---------------------------------------------------------------------------------------------
class TokenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, dt):
        _require_cuda(x, weight)
        x = x.contiguous()
        N, Hh, Ww, Cp = x.shape
        d, C = weight.shape
        dev = x.device
        feat = torch.empty(N, Cp, dtype=_TORCH_DTYPE[dt], device=dev)
        tok = torch.empty(N, d, dtype=_TORCH_DTYPE[dt], device=dev)
        w = weight.detach().contiguous()
        b = bias.detach().contiguous() if bias is not None else None
        lib.call("hyb_gap_fwd", dt, x.data_ptr(), feat.data_ptr(), N, Hh * Ww, Cp, _stream())
        lib.call("hyb_linear_fwd", dt, feat.data_ptr(), Cp, w.data_ptr(), b.data_ptr() if b is not None else None, tok.data_ptr(),
                 N, d, C, 0, _stream())
        ctx.save_for_backward(feat, w)
        ctx.cfg = (dt, N, Hh, Ww, Cp, d, C, bias is not None)
        return tok

    @staticmethod
    def backward(ctx, dtok):
        feat, w = ctx.saved_tensors
        dt, N, Hh, Ww, Cp, d, C, has_bias = ctx.cfg
        dev = feat.device
        dtok = dtok.contiguous()
        dfeat = torch.zeros(N, Cp, dtype=_TORCH_DTYPE[dt], device=dev)     # padded channels stay zero
        dw = torch.empty_like(w)
        db = torch.empty(d, dtype=torch.float32, device=dev) if has_bias else None
        lib.call("hyb_linear_bwd", dt, feat.data_ptr(), Cp, w.data_ptr(), None, dtok.data_ptr(), dfeat.data_ptr(), 0, dw.data_ptr(),
                 db.data_ptr() if db is not None else None, N, d, C, 0, None, 0, _stream())
        dx = torch.empty(N, Hh, Ww, Cp, dtype=_TORCH_DTYPE[dt], device=dev)
        lib.call("hyb_gap_bwd", dt, dfeat.data_ptr(), dx.data_ptr(), N, Hh * Ww, Cp, _stream())
        return dx, dw, db, None


# ---------------------------------------------------------------------------------------------
# TransformerEncoder.forward (all layers) -- TransformerEncoder.pyc src L110-126
# ---------------------------------------------------------------------------------------------
class EncoderFn(torch.autograd.Function):
    """args: x [B,S,D] T, mask (fp32 [B,S,S] or None), cfg tuple, then L*14 fp32 parameters."""

    @staticmethod
    def forward(ctx, x, mask, cfg, *params):
        dt, Hid, L, H, attn_p, layer_p, seed = cfg
        _require_cuda(x, *params)
        x = x.contiguous()
        B, S, D = x.shape
        dev = x.device
        if S > 64:
            raise RuntimeError(f"temporal attention kernel supports T <= 64 tokens per clip (got {S})")
        ps = [p.detach().contiguous() for p in params]
        m = mask.contiguous().float() if mask is not None else None
        saved = _ws(lib.query("hyb_encoder_saved_bytes", dt, B, S, D, Hid, L, H), dev)
        out = torch.empty(B, S, D, dtype=_TORCH_DTYPE[dt], device=dev)
        lib.call("hyb_encoder_fwd", dt, x.data_ptr(), m.data_ptr() if m is not None else None, ptr_array([p.data_ptr() for p in ps]),
                 out.data_ptr(), saved.data_ptr(), B, S, D, Hid, L, H, float(attn_p), float(layer_p), seed, _stream())
        ctx.save_for_backward(saved, m if m is not None else torch.empty(0, device=dev), *ps)
        ctx.cfg = (dt, B, S, D, Hid, L, H, attn_p, layer_p, seed, mask is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        saved, m, *ps = ctx.saved_tensors
        dt, B, S, D, Hid, L, H, attn_p, layer_p, seed, has_mask = ctx.cfg
        dev = saved.device
        dout = dout.contiguous()
        grads = [torch.empty_like(p) for p in ps]
        dx = torch.empty(B, S, D, dtype=_TORCH_DTYPE[dt], device=dev)
        ws = _ws(lib.query("hyb_encoder_workspace_bytes", dt, B, S, D, Hid, L, H), dev)
        lib.call("hyb_encoder_bwd", dt, dout.data_ptr(), m.data_ptr() if has_mask else None, ptr_array([p.data_ptr() for p in ps]),
                 ptr_array([g.data_ptr() for g in grads]), saved.data_ptr(), dx.data_ptr(), B, S, D, Hid, L, H, float(attn_p),
                 float(layer_p), seed, ws.data_ptr(), ws.numel(), _stream())
        return (dx, None, None, *grads)


# ---------------------------------------------------------------------------------------------
# standalone MultiheadAttention.forward(q, k, v, mask) -- src L67-89
# ---------------------------------------------------------------------------------------------
class MultiheadAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q_in, k_in, v_in, mask, cfg, Wq, bq, Wk, bk, Wv, bv, Wo, bo):
        dt, H, p_drop, seed = cfg
        _require_cuda(q_in, Wq)
        q_in, k_in, v_in = q_in.contiguous(), k_in.contiguous(), v_in.contiguous()
        B, S, D = q_in.shape
        if S > 64:
            raise RuntimeError(f"temporal attention kernel supports T <= 64 tokens per clip (got {S})")
        dev, tdt, M = q_in.device, _TORCH_DTYPE[dt], B * S
        ps = [t.detach().contiguous() for t in (Wq, bq, Wk, bk, Wv, bv, Wo, bo)]
        q, k, v, a, out = (torch.empty(B, S, D, dtype=tdt, device=dev) for _ in range(5))
        probs = torch.empty(B * H, S, S, dtype=torch.float32, device=dev)
        m = mask.contiguous().float() if mask is not None else None
        st = _stream()
        for src, W_, b_, dst in ((q_in, ps[0], ps[1], q), (k_in, ps[2], ps[3], k), (v_in, ps[4], ps[5], v)):
            lib.call("hyb_linear_fwd", dt, src.data_ptr(), D, W_.data_ptr(), b_.data_ptr(), dst.data_ptr(), M, D, D, 1, st)
        lib.call("hyb_attention_fwd", dt, q.data_ptr(), k.data_ptr(), v.data_ptr(), m.data_ptr() if m is not None else None, a.data_ptr(),
                 probs.data_ptr(), B, S, D, H, float(p_drop), seed, st)
        lib.call("hyb_linear_fwd", dt, a.data_ptr(), D, ps[6].data_ptr(), ps[7].data_ptr(), out.data_ptr(), M, D, D, 0, st)
        ctx.save_for_backward(q_in, k_in, v_in, q, k, v, a, probs, *ps)
        ctx.cfg = (dt, B, S, D, H, p_drop, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        q_in, k_in, v_in, q, k, v, a, probs, *ps = ctx.saved_tensors
        dt, B, S, D, H, p_drop, seed = ctx.cfg
        dev, tdt, M = q.device, _TORCH_DTYPE[dt], B * S
        dout = dout.contiguous()
        st = _stream()
        grads = [torch.empty_like(p) for p in ps]
        da, dq, dk, dv, dqi, dki, dvi = (torch.empty(B, S, D, dtype=tdt, device=dev) for _ in range(7))
        ws = _ws(M * D * 4, dev)
        lib.call("hyb_linear_bwd", dt, a.data_ptr(), D, ps[6].data_ptr(), None, dout.data_ptr(), da.data_ptr(), 0, grads[6].data_ptr(),
                 grads[7].data_ptr(), M, D, D, 0, None, 0, st)
        lib.call("hyb_attention_bwd", dt, q.data_ptr(), k.data_ptr(), v.data_ptr(), probs.data_ptr(), da.data_ptr(), dq.data_ptr(),
                 dk.data_ptr(), dv.data_ptr(), B, S, D, H, float(p_drop), seed, st)
        for src, y, dy, dsrc, iw in ((q_in, q, dq, dqi, 0), (k_in, k, dk, dki, 2), (v_in, v, dv, dvi, 4)):
            lib.call("hyb_linear_bwd", dt, src.data_ptr(), D, ps[iw].data_ptr(), y.data_ptr(), dy.data_ptr(), dsrc.data_ptr(), 0,
                     grads[iw].data_ptr(), grads[iw + 1].data_ptr(), M, D, D, 1, ws.data_ptr(), ws.numel(), st)
        return (dqi, dki, dvi, None, None, *grads)


# ---------------------------------------------------------------------------------------------
# head (mean over T + Linear) and cross-entropy
# ---------------------------------------------------------------------------------------------
class HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, dt):
        _require_cuda(x, weight)
        x = x.contiguous()
        B, S, D = x.shape
        C = weight.shape[0]
        w = weight.detach().contiguous()
        b = bias.detach().contiguous() if bias is not None else None
        logits = torch.empty(B, C, dtype=torch.float32, device=x.device)
        lib.call("hyb_head_fwd", dt, x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None, logits.data_ptr(), B, S, D, C,
                 _stream())
        ctx.save_for_backward(x, w)
        ctx.cfg = (dt, B, S, D, C, bias is not None)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        x, w = ctx.saved_tensors
        dt, B, S, D, C, has_bias = ctx.cfg
        dlogits = dlogits.contiguous().float()
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.empty(C, dtype=torch.float32, device=x.device) if has_bias else None
        lib.call("hyb_head_bwd", dt, x.data_ptr(), w.data_ptr(), dlogits.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                 db.data_ptr() if db is not None else None, B, S, D, C, _stream())
        return dx, dw, db, None


class CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        _require_cuda(logits, target)
        logits = logits.contiguous().float()
        target = target.contiguous().to(torch.int64)
        B, C = logits.shape
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        lib.call("hyb_cross_entropy_fwd", logits.data_ptr(), target.data_ptr(), loss.data_ptr(), B, C, _stream())
        ctx.save_for_backward(logits, target)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        logits, target = ctx.saved_tensors
        B, C = logits.shape
        dl = dloss.contiguous().float().reshape(1)
        dlogits = torch.empty_like(logits)
        lib.call("hyb_cross_entropy_bwd", logits.data_ptr(), target.data_ptr(), dl.data_ptr(), dlogits.data_ptr(), B, C, _stream())
        return dlogits, None
