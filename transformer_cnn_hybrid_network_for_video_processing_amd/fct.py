"""Drop-in ``FCT`` (the reference's "Fully Convolutional Transformer", FCT.py:210-254) and ``DiceLoss`` (Metrics.py:5-22) on the
MI355X HIP path -- SURVEY.md section 8f-1, the first "next" row.

Contract kept from the reference harness (FCT.py:302,330,333): ``FCT()`` takes no arguments, ``model(x)`` maps ``[B,3,H,W]`` fp32
(H = W a multiple of 32) to a sigmoid mask ``[B,1,H,W]``, ``DiceLoss()(y_pred, y_true)`` is a scalar, and the module / state-dict
names are the reference's (``block_1.conv1_a.weight`` .. ``ds.conv3.bias``; the submodules below own the parameters under those
names and are never called themselves), so checkpoints written by the reference harness (FCT.py:366-373) load unchanged.
Clips are fed frame-folded, ``[B*T,3,H,W]`` (north_star: "folded over B*T").

Forward and backward run on the GPU in exact fp32 (fp32-input MFMA); train mode applies the reference's dropouts (0.3 after the
conv pairs, FCT.py:146,175; 0.1 inside Wide_Focus, FCT.py:115) with the library's counter-based masks.  There is no CPU fallback.
"""
import torch
import torch.nn as nn

from . import ops


def _conv(c, x, act, dilation=1):
    return torch.ops.hybrid.fct_conv(x, c.weight, c.bias, dilation, act)[0]


def _drop(x, p, training):
    """nn.Dropout(p) (FCT.py:115,146,175): identity in eval mode."""
    return torch.ops.hybrid.fct_dropout(x, p, ops.next_seed(), ops.step_counter()) if training and p > 0.0 else x


def _only_fct_geometry(**given):
    """The reference's constructors take conv geometry arguments that FCT() itself never varies (FCT.py:223-231 passes none of them).
    The HIP path implements that one geometry; anything else is refused loudly instead of computed differently."""
    want = dict(kernel_size=3, stride_kv=1, stride_q=1, padding_q="same")
    for k, v in given.items():
        if v != want[k]:
            raise NotImplementedError(f"{k}={v!r}: the HIP path implements the geometry FCT() uses ({k}={want[k]!r}, FCT.py:25,86)")


class Attention(nn.Module):                                       # signature and parameter names of FCT.py:24-39
    def __init__(self, channels, num_heads, proj_drop=0.0, kernel_size=3, stride_kv=1, stride_q=1, padding_kv="same", padding_q="same",
                 attention_bias=True):
        super().__init__()
        # proj_drop and padding_kv are accepted and unused, exactly as in the reference (FCT.py:30 stores proj_drop and never applies it;
        # conv_k / conv_v are built with stride_kv in the padding position, FCT.py:33,35, so padding_kv never reaches a layer)
        _only_fct_geometry(kernel_size=kernel_size, stride_kv=stride_kv, stride_q=stride_q, padding_q=padding_q)
        self.stride_kv, self.stride_q, self.proj_drop = stride_kv, stride_q, proj_drop
        self.num_heads = num_heads
        self.conv_q = nn.Conv2d(channels, channels, 3, 1, 1, bias=attention_bias, groups=channels)
        self.layernorm_q = nn.LayerNorm(channels, eps=1e-5)
        self.conv_k = nn.Conv2d(channels, channels, 3, 1, 1, bias=attention_bias, groups=channels)
        self.layernorm_k = nn.LayerNorm(channels, eps=1e-5)
        self.conv_v = nn.Conv2d(channels, channels, 3, 1, 1, bias=attention_bias, groups=channels)
        self.layernorm_v = nn.LayerNorm(channels, eps=1e-5)
        self.attention = nn.MultiheadAttention(embed_dim=channels, bias=attention_bias, batch_first=True, num_heads=num_heads)

    def forward(self, x):                                         # x NHWC; FCT.py:41-79 as two fused operators
        N, H, W, C = x.shape
        convs, lns = (self.conv_q, self.conv_k, self.conv_v), (self.layernorm_q, self.layernorm_k, self.layernorm_v)
        q, k, v = torch.ops.hybrid.fct_qkv_proj(x, [c.weight for c in convs], [c.bias for c in convs], [l.weight for l in lns],
                                                [l.bias for l in lns], lns[0].eps)
        a = self.attention
        out = torch.ops.hybrid.fct_mha(q.view(N, H * W, C), k.view(N, H * W, C), v.view(N, H * W, C), a.in_proj_weight, a.in_proj_bias,
                                       a.out_proj.weight, a.out_proj.bias, self.num_heads)[0]
        return out.view(N, H, W, C)


class Wide_Focus(nn.Module):                                      # FCT.py:107-115
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, padding="same")
        self.conv2 = nn.Conv2d(in_channels, out_channels, 3, 1, padding="same", dilation=2)
        self.conv3 = nn.Conv2d(in_channels, out_channels, 3, 1, padding="same", dilation=3)
        self.conv4 = nn.Conv2d(in_channels, out_channels, 3, 1, padding="same")

    def forward(self, x):                                         # FCT.py:117-132
        t = self.training
        x1, x2, x3 = (_drop(_conv(c, x, ops.ACT_GELU, d), 0.1, t) for c, d in ((self.conv1, 1), (self.conv2, 2), (self.conv3, 3)))
        added = torch.ops.hybrid.fct_add(torch.ops.hybrid.fct_add(x1, x2), x3)
        return _drop(_conv(self.conv4, added, ops.ACT_GELU, 1), 0.1, t)


class Transformer(nn.Module):                                     # FCT.py:84-91
    def __init__(self, in_channels, out_channels, num_heads, dpr=None, proj_drop=0.0, attention_bias=True, padding_q="same", padding_kv="same",
                 stride_kv=1, stride_q=1):
        super().__init__()
        # dpr (a stochastic-depth rate) is accepted and unused, as in the reference (FCT.py:86-91 never reads it)
        self.attention_output = Attention(channels=in_channels, num_heads=num_heads, proj_drop=proj_drop, padding_q=padding_q, padding_kv=padding_kv,
                                          stride_kv=stride_kv, stride_q=stride_q, attention_bias=attention_bias)
        self.conv1 = nn.Conv2d(out_channels, out_channels, 3, 1, padding="same")
        self.layernorm = nn.LayerNorm(out_channels, eps=1e-5)
        self.wide_focus = Wide_Focus(out_channels, out_channels)

    def forward(self, x):                                         # FCT.py:93-102
        x2 = torch.ops.hybrid.fct_add(_conv(self.conv1, self.attention_output(x), ops.ACT_NONE), x)
        x3 = torch.ops.hybrid.fct_ln(x2, self.layernorm.weight, self.layernorm.bias, self.layernorm.eps)
        return torch.ops.hybrid.fct_add(x2, self.wide_focus(x3))


class Block_encoder_bottleneck(nn.Module):                        # FCT.py:136-147
    def __init__(self, blk, in_channels, out_channels, att_heads, dpr):        # five positionals, as FCT.py:137 (dpr only travels to Transformer)
        super().__init__()
        self.blk = blk
        self.conv1_a = nn.Conv2d(in_channels, out_channels, 3, 1, padding="same")
        self.conv1_b = nn.Conv2d(3, in_channels, 3, 1, padding="same")
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, padding="same")
        self.conv3 = nn.Conv2d(out_channels, out_channels, 3, 1, padding="same")
        self.trans = Transformer(in_channels=out_channels, out_channels=out_channels, num_heads=att_heads, dpr=dpr)

    def forward(self, x, scale_img=None):                         # FCT.py:149-162
        if self.blk in ("first", "bottleneck"):
            x1 = _conv(self.conv2, _conv(self.conv1_a, x, ops.ACT_RELU), ops.ACT_RELU)
        else:
            x1 = torch.ops.hybrid.fct_concat(_conv(self.conv1_b, scale_img, ops.ACT_RELU), x)
            x1 = _conv(self.conv3, _conv(self.conv2, x1, ops.ACT_RELU), ops.ACT_RELU)
        return self.trans(torch.ops.hybrid.fct_resample(_drop(x1, 0.3, self.training), 0))


class Block_decoder(nn.Module):                                   # FCT.py:167-175
    def __init__(self, in_channels, out_channels, att_heads, dpr):             # FCT.py:168
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, padding="same")
        self.conv2 = nn.Conv2d(out_channels * 2, out_channels, 3, 1, padding="same")
        self.conv3 = nn.Conv2d(out_channels, out_channels, 3, 1, padding="same")
        self.trans = Transformer(in_channels=out_channels, out_channels=out_channels, num_heads=att_heads, dpr=dpr)

    def forward(self, x, skip):                                   # FCT.py:177-186
        x1 = _conv(self.conv1, torch.ops.hybrid.fct_resample(x, 2), ops.ACT_RELU)
        x1 = torch.ops.hybrid.fct_concat(skip, x1)
        return self.trans(_drop(_conv(self.conv3, _conv(self.conv2, x1, ops.ACT_RELU), ops.ACT_RELU), 0.3, self.training))


class DS_out(nn.Module):                                          # FCT.py:191-198
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, in_channels, 3, 1, padding="same")
        self.conv2 = nn.Conv2d(in_channels, in_channels, 3, 1, padding="same")
        self.conv3 = nn.Conv2d(in_channels, out_channels, 3, 1, padding="same")

    def forward(self, x):                                         # FCT.py:200-206
        x1 = _conv(self.conv2, _conv(self.conv1, torch.ops.hybrid.fct_resample(x, 2), ops.ACT_RELU), ops.ACT_RELU)
        return _conv(self.conv3, x1, ops.ACT_SIGMOID)


class FCT(nn.Module):
    """``FCT()``: zero-argument constructor like the reference (FCT.py:211-233: 2 heads, filters 8-16-32-64-128-64-32-16-8)."""

    def __init__(self):
        super().__init__()
        f, h = (8, 16, 32, 64, 128, 64, 32, 16, 8), 2
        dpr = [0.0] * len(f)                                                      # np.linspace(0, stochastic_depth_rate = 0.0, blocks), FCT.py:218-219
        self.block_1 = Block_encoder_bottleneck("first", 3, f[0], h, dpr[0])
        self.block_2 = Block_encoder_bottleneck("second", f[0], f[1], h, dpr[1])
        self.block_3 = Block_encoder_bottleneck("third", f[1], f[2], h, dpr[2])
        self.block_4 = Block_encoder_bottleneck("fourth", f[2], f[3], h, dpr[3])
        self.block_5 = Block_encoder_bottleneck("bottleneck", f[3], f[4], h, dpr[4])
        self.block_6 = Block_decoder(f[4], f[5], h, dpr[5])
        self.block_7 = Block_decoder(f[5], f[6], h, dpr[6])
        self.block_8 = Block_decoder(f[6], f[7], h, dpr[7])
        self.block_9 = Block_decoder(f[7], f[8], h, dpr[8])
        self.ds = DS_out(f[8], 1)

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected frames [B,3,H,W] (clips folded over B*T)")
        if not x.is_cuda:
            raise RuntimeError("FCT runs on the MI355X HIP path only: move the model and input to 'cuda' (there is no CPU fallback)")
        if x.shape[2] % 32 != 0 or x.shape[3] != x.shape[2]:
            # the reference fails at its skip concat (FCT.py:181) / its square-map view (FCT.py:77) for any other size
            raise RuntimeError(f"FCT needs square inputs with H = W a multiple of 32 (got {tuple(x.shape[2:])})")
        B, _, H, W = x.shape
        x = ops.nchw_to_nhwc(x, ops.HYB_F32, 3)                                   # NHWC fp32, 3 channels
        s2 = torch.ops.hybrid.fct_resample(x, 1)                                  # multi-scale input pyramid, FCT.py:238-240
        s3 = torch.ops.hybrid.fct_resample(s2, 1)
        s4 = torch.ops.hybrid.fct_resample(s3, 1)
        x1 = self.block_1(x)
        x2 = self.block_2(x1, s2)
        x3 = self.block_3(x2, s3)
        x4 = self.block_4(x3, s4)
        y = self.block_5(x4)
        y = self.block_6(y, x4)
        y = self.block_7(y, x3)
        y = self.block_8(y, x2)
        y = self.block_9(y, x1)
        return self.ds(y).reshape(B, 1, H, W)                                     # one channel: NHWC and NCHW coincide


class DiceLoss(nn.Module):
    """Metrics.py:5-22 (``num_classes`` is accepted and unused there too)."""

    def __init__(self, num_classes=8):
        super().__init__()
        self.smooth = 1.0

    def forward(self, y_pred, y_true):
        return torch.ops.hybrid.dice_loss(y_pred, y_true, self.smooth)
