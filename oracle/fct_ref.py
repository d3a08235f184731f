"""CPU oracle for the "next" row FCT -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as
oracle/hybrid_ref.py: only tests/ and bench.py's cpu_baseline leg may import it).

A stock ``torch.nn`` fp32 restatement of the reference's "Fully Convolutional Transformer" (FCT.py:24-254) and of ``DiceLoss``
(Metrics.py:5-22), with the reference's module / state-dict names.  PINNED by golden vectors captured from the reference's own
classes in the build container (tests/golden/make_golden.py -> g3..g8 ``.npz``; tests/test_oracle_fct.py replays every one of
them on this restatement: outputs and gradients).

Reference behaviour kept on purpose: ``conv_k`` / ``conv_v`` receive ``stride_kv`` as their *padding*
positional (FCT.py:33,35; = 1, i.e. "same" for a 3x3); ``Attention``'s ``self.dropout``, ``proj_drop``, ``padding_kv`` and
``dpr`` are accepted and unused (FCT.py:39,78,25,86); 14 parameters never receive a gradient (``block_1.conv1_b/conv3``,
``block_2-4.conv1_a``, ``block_5.conv1_b/conv3``; FCT.py:140-143 vs the branch at FCT.py:150-159); H = W must be a multiple of 32.
"""
import torch
import torch.nn as nn


class Attention(nn.Module):                                                          # FCT.py:24-79
    def __init__(self, channels, num_heads, proj_drop=0.0, kernel_size=3, stride_kv=1, stride_q=1, padding_kv="same", padding_q="same",
                 attention_bias=True):
        super().__init__()
        self.num_heads = num_heads
        self.conv_q = nn.Conv2d(channels, channels, kernel_size, stride_q, padding_q, bias=attention_bias, groups=channels)      # :31
        self.layernorm_q = nn.LayerNorm(channels, eps=1e-5)                                                                       # :32
        self.conv_k = nn.Conv2d(channels, channels, kernel_size, stride_kv, stride_kv, bias=attention_bias, groups=channels)     # :33
        self.layernorm_k = nn.LayerNorm(channels, eps=1e-5)
        self.conv_v = nn.Conv2d(channels, channels, kernel_size, stride_kv, stride_kv, bias=attention_bias, groups=channels)     # :35
        self.layernorm_v = nn.LayerNorm(channels, eps=1e-5)
        self.attention = nn.MultiheadAttention(embed_dim=channels, bias=attention_bias, batch_first=True, num_heads=num_heads)    # :37
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout()                                                                                               # :39 (unused)

    def _project(self, x, conv, ln):                                                 # FCT.py:41-57
        return ln(self.relu(conv(x)).permute(0, 2, 3, 1)).permute(0, 3, 1, 2)

    def forward(self, x):                                                            # FCT.py:67-79
        B, C, H, W = x.shape
        q, k, v = (self._project(x, c, l) for c, l in ((self.conv_q, self.layernorm_q), (self.conv_k, self.layernorm_k),
                                                        (self.conv_v, self.layernorm_v)))
        q, k, v = (t.reshape(B, C, H * W).permute(0, 2, 1) for t in (q, k, v))                                                    # :69-74
        a = self.attention(query=q, value=v, key=k, need_weights=False)[0]                                                        # :75
        return a.permute(0, 2, 1).reshape(B, C, H, W)                                                                             # :76-77 (square maps)


class Wide_Focus(nn.Module):                                                         # FCT.py:107-132
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, padding="same")
        self.conv2 = nn.Conv2d(in_channels, out_channels, 3, 1, padding="same", dilation=2)
        self.conv3 = nn.Conv2d(in_channels, out_channels, 3, 1, padding="same", dilation=3)
        self.conv4 = nn.Conv2d(in_channels, out_channels, 3, 1, padding="same")
        self.gelu = nn.GELU()
        self.dropout = nn.Dropout(0.1)

    def forward(self, x):
        added = sum(self.dropout(self.gelu(c(x))) for c in (self.conv1, self.conv2, self.conv3))                                  # :118-128
        return self.dropout(self.gelu(self.conv4(added)))                                                                         # :129-132


class Transformer(nn.Module):                                                        # FCT.py:84-102
    def __init__(self, in_channels, out_channels, num_heads, dpr=None, proj_drop=0.0, attention_bias=True, padding_q="same", padding_kv="same",
                 stride_kv=1, stride_q=1):
        super().__init__()
        self.attention_output = Attention(channels=in_channels, num_heads=num_heads, proj_drop=proj_drop, padding_q=padding_q,
                                          padding_kv=padding_kv, stride_kv=stride_kv, stride_q=stride_q, attention_bias=attention_bias)
        self.conv1 = nn.Conv2d(out_channels, out_channels, 3, 1, padding="same")
        self.layernorm = nn.LayerNorm(self.conv1.out_channels, eps=1e-5)
        self.wide_focus = Wide_Focus(out_channels, out_channels)

    def forward(self, x):
        x2 = self.conv1(self.attention_output(x)) + x                                                                             # :94-96
        x3 = self.layernorm(x2.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)                                                           # :97-99
        return x2 + self.wide_focus(x3)                                                                                           # :100-102


class Block_encoder_bottleneck(nn.Module):                                           # FCT.py:136-162
    def __init__(self, blk, in_channels, out_channels, att_heads, dpr):
        super().__init__()
        self.blk = blk
        self.conv1_a = nn.Conv2d(in_channels, out_channels, 3, 1, padding="same")
        self.conv1_b = nn.Conv2d(3, in_channels, 3, 1, padding="same")
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, padding="same")
        self.conv3 = nn.Conv2d(out_channels, out_channels, 3, 1, padding="same")
        self.trans = Transformer(in_channels=out_channels, out_channels=out_channels, num_heads=att_heads, dpr=dpr)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(p=0.3)
        self.maxpool = nn.MaxPool2d(kernel_size=(2, 2))

    def forward(self, x, scale_img="none"):
        if self.blk in ("first", "bottleneck"):                                                                                   # :150-154
            x1 = self.relu(self.conv2(self.relu(self.conv1_a(x))))
        else:                                                                                                                     # :155-160
            x1 = torch.cat([self.relu(self.conv1_b(scale_img)), x], dim=1)
            x1 = self.relu(self.conv3(self.relu(self.conv2(x1))))
        return self.trans(self.maxpool(self.dropout(x1)))


class Block_decoder(nn.Module):                                                      # FCT.py:167-186
    def __init__(self, in_channels, out_channels, att_heads, dpr):
        super().__init__()
        self.upsample = nn.Upsample(scale_factor=2)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, padding="same")
        self.conv2 = nn.Conv2d(out_channels * 2, out_channels, 3, 1, padding="same")
        self.conv3 = nn.Conv2d(out_channels, out_channels, 3, 1, padding="same")
        self.trans = Transformer(in_channels=out_channels, out_channels=out_channels, num_heads=att_heads, dpr=dpr)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(0.3)

    def forward(self, x, skip):
        x1 = self.relu(self.conv1(self.upsample(x)))                                                                              # :178-179
        x1 = torch.cat((skip, x1), dim=1)                                                                                         # :180
        x1 = self.dropout(self.relu(self.conv3(self.relu(self.conv2(x1)))))                                                       # :181-183
        return self.trans(x1)


class DS_out(nn.Module):                                                             # FCT.py:191-206
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.upsample = nn.Upsample(scale_factor=2)
        self.conv1 = nn.Conv2d(in_channels, in_channels, 3, 1, padding="same")
        self.conv2 = nn.Conv2d(in_channels, in_channels, 3, 1, padding="same")
        self.conv3 = nn.Conv2d(in_channels, out_channels, 3, 1, padding="same")
        self.relu = nn.ReLU()
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        x1 = self.relu(self.conv2(self.relu(self.conv1(self.upsample(x)))))
        return self.sigmoid(self.conv3(x1))


class FCT(nn.Module):                                                                # FCT.py:210-254
    FILTERS = (8, 16, 32, 64, 128, 64, 32, 16, 8)                                                                                 # :215
    HEADS = 2                                                                                                                     # :214

    def __init__(self):
        super().__init__()
        f, h = self.FILTERS, self.HEADS
        self.scale_img = nn.AvgPool2d(2, 2)
        self.block_1 = Block_encoder_bottleneck("first", 3, f[0], h, 0.0)
        self.block_2 = Block_encoder_bottleneck("second", f[0], f[1], h, 0.0)
        self.block_3 = Block_encoder_bottleneck("third", f[1], f[2], h, 0.0)
        self.block_4 = Block_encoder_bottleneck("fourth", f[2], f[3], h, 0.0)
        self.block_5 = Block_encoder_bottleneck("bottleneck", f[3], f[4], h, 0.0)
        self.block_6 = Block_decoder(f[4], f[5], h, 0.0)
        self.block_7 = Block_decoder(f[5], f[6], h, 0.0)
        self.block_8 = Block_decoder(f[6], f[7], h, 0.0)
        self.block_9 = Block_decoder(f[7], f[8], h, 0.0)
        self.ds = DS_out(f[8], 1)

    def forward(self, x):
        s2 = self.scale_img(x); s3 = self.scale_img(s2); s4 = self.scale_img(s3)                                                  # :238-240
        x1 = self.block_1(x)
        x2 = self.block_2(x1, s2)
        x3 = self.block_3(x2, s3)
        x4 = self.block_4(x3, s4)
        y = self.block_5(x4)
        y = self.block_6(y, x4)
        y = self.block_7(y, x3)
        y = self.block_8(y, x2)
        y = self.block_9(y, x1)
        return self.ds(y)                                                                                                         # :252


class DiceLoss(nn.Module):                                                           # Metrics.py:5-22
    def __init__(self, num_classes=8):
        super().__init__()
        self.smooth = 1.0

    def forward(self, y_pred, y_true):
        assert y_pred.size() == y_true.size()
        p = y_pred[:, 0].contiguous().view(-1)
        t = y_true[:, 0].contiguous().view(-1)
        inter = (p * t).sum()
        return 1.0 - (2.0 * inter + self.smooth) / (p.sum() + t.sum() + self.smooth)
