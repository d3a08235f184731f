"""``Bottleneck`` / ``Encoder_32K``: the reference's ResNet-bottleneck backbone on the MI355X HIP path -- SURVEY.md section 8f-3.

Only CPython-3.8 bytecode of this model ships with the reference (``__pycache__/AE_256_32K.cpython-38.pyc``; it was read as data
with ``marshal`` and an opcode table, never executed).  What the bytecode fixes, by source line of the embedded line table:

* ``Bottleneck(in_channels, out_channels, stride=1, downsample=None)``, ``expansion = 4`` (src L21-33): ``conv1`` 1x1 no bias,
  ``bn1``, ``conv2`` 3x3 stride ``stride`` padding 1 no bias, ``bn2``, ``conv3`` 1x1 -> 4*out no bias, ``bn3``, ``relu``;
  forward (src L35-53): ``out = relu(bn1(conv1(x)))``, ``relu(bn2(conv2))``, ``bn3(conv3)``, ``residual = downsample(x)`` if there is
  one, ``out += residual``, ``relu``.
* ``Encoder_32K(block, layers)`` (src L60-94): ``inplanes = 64``; ``conv1 = Conv2d(3, 64, 7, stride 2, padding 3, bias=False)``, ``bn1``;
  ``layer1 = _make_layer(block, 64, layers[0])``, ``layer2 = _make_layer(block, 128, layers[1], stride=2)``; ``conv2..conv5 =
  Conv2d(512->128->64->16->8, 3, 1, 1)`` (bias on) each with a ``BatchNorm2d``; ``relu``; ``dropout = Dropout2d(0.3)``;
  ``scale_img = AvgPool2d(2, 2)`` (never called).  ``_make_layer`` (src L96-106) is torchvision's: a ``Sequential(Conv2d(inplanes,
  planes*4, 1, stride, bias=False), BatchNorm2d)`` down-sample when the stride is not 1 or the widths differ.
  forward (src L108-137): stem -> layer1 -> dropout -> layer2 -> four conv+bn+relu -> dropout -> ``view(B, -1)`` -> ``view(B, 8, 4096)``:
  a 256x256 frame becomes 8 tokens of 4096 features (``Autoencoder32K`` builds it as ``Encoder_32K(Bottleneck, [3, 4])``, src L205-213).

Module and state-dict names are the reference's (``conv1.weight``, ``layer1.0.downsample.0.weight``, ``bn5.running_var`` ..), so a
checkpoint written by its ``train`` loads unchanged.  The submodules own the parameters and are never called: every step runs as a
``torch.ops.hybrid`` operator (conv2d, bn2d with the residual add and ReLU fused, dropout2d) on NHWC fp32 arrays, forward and
backward in exact fp32.  Clips are fed frame-folded, ``[B*T, 3, 256, 256]``.  There is no CPU fallback.  Parity: the tests compare
against a CPU restatement of the same bytecode on torch's own operators; nothing executable of the reference exists for this model,
so it is "parity unpinned" (DESIGN.md section 12).
"""
import torch
import torch.nn as nn

from . import ops


def _conv(c, x):
    return torch.ops.hybrid.conv2d(x, c.weight, c.bias, c.stride[0], c.padding[0], c.dilation[0], ops.ACT_NONE)[0]


def _bn(b, x, relu, residual=None):
    if b.training and b.track_running_stats and b.num_batches_tracked is not None:
        b.num_batches_tracked.add_(1)                                # nn.BatchNorm2d bookkeeping (momentum is a constant here)
    use_batch = b.training or b.running_mean is None
    return torch.ops.hybrid.bn2d(x, b.weight, b.bias, residual, b.running_mean, b.running_var, use_batch, b.momentum, b.eps, relu)[0]


class Bottleneck(nn.Module):                                           # AE_256_32K src L21-53
    expansion = 4

    def __init__(self, in_channels, out_channels, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.conv3 = nn.Conv2d(out_channels, out_channels * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(out_channels * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):                                             # x NHWC fp32
        residual = x
        out = _bn(self.bn1, _conv(self.conv1, x), True)
        out = _bn(self.bn2, _conv(self.conv2, out), True)
        out = _conv(self.conv3, out)
        if self.downsample is not None:
            residual = _bn(self.downsample[1], _conv(self.downsample[0], x), False)
        return _bn(self.bn3, out, True, residual)                     # bn3 -> += residual -> relu in one pass


class Encoder_32K(nn.Module):                                          # AE_256_32K src L58-137
    def __init__(self, block=Bottleneck, layers=(3, 4)):
        self.inplanes = 64
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.conv2 = nn.Conv2d(512, 128, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(128)
        self.conv3 = nn.Conv2d(128, 64, 3, 1, 1)
        self.bn3 = nn.BatchNorm2d(64)
        self.conv4 = nn.Conv2d(64, 16, 3, 1, 1)
        self.bn4 = nn.BatchNorm2d(16)
        self.conv5 = nn.Conv2d(16, 8, 3, 1, 1)
        self.bn5 = nn.BatchNorm2d(8)
        self.relu = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout2d(0.3)
        self.scale_img = nn.AvgPool2d(2, 2)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def _drop(self, x):
        p = self.dropout.p
        return torch.ops.hybrid.dropout2d(x, p, ops.next_seed(), ops.step_counter()) if self.training and p > 0.0 else x

    def feature_map(self, x):
        """[B,3,H,W] -> the final 8-channel map [B,8,H/4,W/4] (NCHW, what the reference's two ``view`` calls flatten)."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected frames [B,3,H,W] (clips folded over B*T)")
        if not x.is_cuda:
            raise RuntimeError("Encoder_32K runs on the MI355X HIP path only: move the model and input to 'cuda' (there is no CPU fallback)")
        x = ops.nchw_to_nhwc(x, ops.HYB_F32, 3)
        x = _bn(self.bn1, _conv(self.conv1, x), True)
        for blk in self.layer1:
            x = blk(x)
        x = self._drop(x)
        for blk in self.layer2:
            x = blk(x)
        for conv, bn in ((self.conv2, self.bn2), (self.conv3, self.bn3), (self.conv4, self.bn4), (self.conv5, self.bn5)):
            x = _bn(bn, _conv(conv, x), True)
        x = self._drop(x)
        return ops.nhwc_to_nchw(x, ops.HYB_F32, 8)

    def forward(self, x):
        y = self.feature_map(x)
        B = y.shape[0]
        if y[0].numel() != 8 * 4096:
            # the reference's x.view(B, 8, 4096) (src L118-119) raises for anything but 256x256 frames
            raise RuntimeError(f"shape '[{B}, 8, 4096]' is invalid for input of size {y.numel()}")
        return y.view(B, -1).view(B, 8, 4096)
