"""CPU oracle for the "next" row 8f-3, the ResNet-bottleneck backbone ``Encoder_32K`` -- TEST INFRASTRUCTURE, NOT PRODUCT CODE
(same rules as oracle/hybrid_ref.py: only tests/, __graft_entry__.smoke() and benchmark CPU-baseline legs may import it).

PARITY UNPINNED.  The reference ships this model as CPython-3.8 bytecode only (``__pycache__/AE_256_32K.cpython-38.pyc``): it
cannot be imported by the 3.10 interpreter here and is never executed; no test, fixture or stored output of the reference covers
it.  This file restates what the bytecode's constants, names and call sequence fix (read as data: ``marshal`` + opcode table),
functionally, on torch's own CPU operators, over a flat ``{state-dict name: tensor}`` dictionary:

* ``Bottleneck`` (src L21-53): 1x1 conv (no bias) -> BN -> ReLU -> 3x3 conv stride s padding 1 (no bias) -> BN -> ReLU -> 1x1 conv to
  4x width (no bias) -> BN; ``+= downsample(x)`` or ``+= x``; ReLU.
* ``Encoder_32K.__init__`` (src L60-94) and ``_make_layer`` (src L96-106): stem Conv2d(3, 64, 7, 2, 3, bias=False) + BN; layer1 = 3
  bottlenecks of width 64 (the first with a 1x1 stride-1 down-sample 64 -> 256); layer2 = 4 bottlenecks of width 128 (the first with
  stride 2 and a 1x1 stride-2 down-sample 256 -> 512); Conv2d(512,128,3,1,1), Conv2d(128,64,3,1,1), Conv2d(64,16,3,1,1),
  Conv2d(16,8,3,1,1) -- these four WITH bias (positional ``(in, out, 3, 1, 1)``) -- each followed by BatchNorm2d; Dropout2d(0.3).
* ``Encoder_32K.forward`` (src L108-137): relu(bn1(conv1)) -> layer1 -> dropout -> layer2 -> relu(bn_i(conv_i)), i = 2..5 -> dropout
  -> view(B, -1) -> view(B, 8, 4096).
BatchNorm2d defaults: eps 1e-5, momentum 0.1, affine, batch statistics in train mode.
"""
import torch
import torch.nn.functional as F

LAYERS = (("layer1", 64, 3, 1), ("layer2", 128, 4, 2))          # name, planes, blocks, stride of the first block (src L66-67)
TAIL = (("conv2", "bn2", 512, 128), ("conv3", "bn3", 128, 64), ("conv4", "bn4", 64, 16), ("conv5", "bn5", 16, 8))   # src L70-88
EXPANSION = 4                                                    # src L22
DROP_P = 0.3                                                     # src L92


def param_shapes(layers=(3, 4)):
    """Ordered {state-dict name: shape} exactly as ``Encoder_32K(Bottleneck, layers).state_dict()`` lists them."""
    out = {}

    def bn(prefix, c):
        out[prefix + ".weight"] = (c,); out[prefix + ".bias"] = (c,)
        out[prefix + ".running_mean"] = (c,); out[prefix + ".running_var"] = (c,); out[prefix + ".num_batches_tracked"] = ()

    out["conv1.weight"] = (64, 3, 7, 7); bn("bn1", 64)
    inplanes = 64
    for (name, planes, _, stride), blocks in zip(LAYERS, layers):
        for b in range(blocks):
            s = stride if b == 0 else 1
            p = f"{name}.{b}"
            out[p + ".conv1.weight"] = (planes, inplanes, 1, 1); bn(p + ".bn1", planes)
            out[p + ".conv2.weight"] = (planes, planes, 3, 3); bn(p + ".bn2", planes)
            out[p + ".conv3.weight"] = (planes * EXPANSION, planes, 1, 1); bn(p + ".bn3", planes * EXPANSION)
            if b == 0 and (s != 1 or inplanes != planes * EXPANSION):
                out[p + ".downsample.0.weight"] = (planes * EXPANSION, inplanes, 1, 1); bn(p + ".downsample.1", planes * EXPANSION)
            inplanes = planes * EXPANSION
    for conv, bnn, ci, co in TAIL:
        out[conv + ".weight"] = (co, ci, 3, 3); out[conv + ".bias"] = (co,); bn(bnn, co)
    return out


def make_params(seed=0, dtype=torch.float64, layers=(3, 4)):
    """Random parameters of the right shapes (gamma around 1, beta / biases small: values a trained net could hold)."""
    g = torch.Generator().manual_seed(seed)
    params = {}
    for name, shape in param_shapes(layers).items():
        if name.endswith("num_batches_tracked"):
            params[name] = torch.zeros((), dtype=torch.int64)
        elif name.endswith("running_mean"):
            params[name] = torch.zeros(shape, dtype=dtype)
        elif name.endswith("running_var"):
            params[name] = torch.ones(shape, dtype=dtype)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            params[name] = (torch.randn(shape, generator=g, dtype=torch.float64) * (2.0 / fan_in) ** 0.5).to(dtype)
        elif ".bn" in name or name.startswith("bn") or ".downsample.1" in name:
            base = 1.0 if name.endswith("weight") else 0.0
            params[name] = (base + 0.2 * torch.randn(shape, generator=g, dtype=torch.float64)).to(dtype)
        else:
            params[name] = (0.1 * torch.randn(shape, generator=g, dtype=torch.float64)).to(dtype)
    return params


def _bn(p, prefix, x, training, momentum=0.1, eps=1e-5):
    return F.batch_norm(x, p[prefix + ".running_mean"], p[prefix + ".running_var"], p[prefix + ".weight"], p[prefix + ".bias"], training, momentum, eps)


def bottleneck(p, prefix, x, stride, training):                  # src L35-53
    out = F.relu(_bn(p, prefix + ".bn1", F.conv2d(x, p[prefix + ".conv1.weight"]), training))
    out = F.relu(_bn(p, prefix + ".bn2", F.conv2d(out, p[prefix + ".conv2.weight"], stride=stride, padding=1), training))
    out = _bn(p, prefix + ".bn3", F.conv2d(out, p[prefix + ".conv3.weight"]), training)
    residual = x
    if prefix + ".downsample.0.weight" in p:
        residual = _bn(p, prefix + ".downsample.1", F.conv2d(x, p[prefix + ".downsample.0.weight"], stride=stride), training)
    return F.relu(out + residual)


def feature_map(p, x, training=True, layers=(3, 4), drop_masks=None):
    """x [B,3,H,W] -> [B,8,H/4,W/4].  ``drop_masks``: None (no dropout: eval mode, or a train-mode run with p = 0), or the two
    multiplier planes [B,256,1,1] and [B,8,1,1] (0 or 1/(1-p)) of the two Dropout2d calls.  Running statistics in ``p`` are updated
    in place in training mode, like the modules do."""
    x = F.relu(_bn(p, "bn1", F.conv2d(x, p["conv1.weight"], stride=2, padding=3), training))                 # src L108
    for (name, _, _, stride), blocks in zip(LAYERS, layers):
        for b in range(blocks):
            x = bottleneck(p, f"{name}.{b}", x, stride if b == 0 else 1, training)
        if name == "layer1" and drop_masks is not None:                                                       # src L113
            x = x * drop_masks[0]
    for conv, bnn, _, _ in TAIL:                                                                              # src L120-135
        x = F.relu(_bn(p, bnn, F.conv2d(x, p[conv + ".weight"], p[conv + ".bias"], stride=1, padding=1), training))
    if drop_masks is not None:                                                                                # src L136
        x = x * drop_masks[1]
    return x


def forward(p, x, training=True, layers=(3, 4), drop_masks=None):
    y = feature_map(p, x, training, layers, drop_masks)
    return y.view(y.shape[0], -1).view(y.shape[0], 8, 4096)                                                   # src L118-119 (256x256 frames only)
