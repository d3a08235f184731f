"""Deterministic, name-keyed parameter values shared by tests/golden/make_golden.py (which writes them into the reference's FCT
before recording G8) and the tests (which write the same values into this repo's FCT and into oracle/fct_ref.py): the 2.1 M
weights of the whole-model fixture then need not be stored, only the outputs and gradient digests."""
import hashlib

import torch


def det_param(name, shape):
    """Value for the parameter called `name`: N(0, 1/fan_in) for weights (fan-in = product of the trailing dims), N(1, 0.1) for
    LayerNorm scales, N(0, 0.05) for biases -- magnitudes close to torch's default initialisation, so activations stay O(1)."""
    seed = int.from_bytes(hashlib.sha256(name.encode()).digest()[:6], "little")
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(tuple(shape), generator=g)
    if name.endswith("bias"):
        return t * 0.05
    if "layernorm" in name and name.endswith("weight"):
        return 1.0 + 0.1 * t
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    return t / max(fan_in, 1) ** 0.5


def det_state_dict(module):
    return {k: det_param(k, v.shape) for k, v in module.state_dict().items()}


def digest(t, n=16):
    """(sum, L2 norm, first n elements, n elements spread over the tensor) of a tensor -- enough to pin a gradient without storing it."""
    f = t.detach().double().flatten()
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return torch.cat([f.sum().view(1), f.norm().view(1), f[:n], f[idx]]).numpy()
