"""Generate golden vectors from the REFERENCE's own importable classes.

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_golden.py

It imports ``UNet`` from the reference (UNet.py is importable as-is),
drives the reference's Conv+BN+ReLU unit (``UNet._block``, UNet.py:54-66) and its
``MaxPool2d(2,2)`` (UNet.py:13) on seeded inputs, and stores inputs, parameters,
outputs and gradients as plain float32 arrays in ``.npz`` files (data only -- no
reference source text is stored).  tests/test_oracle.py replays them against
``oracle/hybrid_ref.py``; the GPU parity tests replay them against the HIP path.
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("HYB_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def np_(t):
    return t.detach().cpu().numpy().copy()


def g1_block_stage(UNet):
    """G1: first triple of UNet._block(3, 8, 'enc1') + the pool of UNet.py:13,
    train mode (batch statistics, running-stat update) and eval mode, with grads."""
    torch.manual_seed(0)
    blk = UNet._block(3, 8, name="enc1")
    pool = torch.nn.MaxPool2d(kernel_size=2, stride=2)      # UNet.py:13
    triple = blk[:3]                                        # enc1conv1, enc1norm1, enc1relu1
    # non-trivial affine so gamma/beta gradients and sign handling are exercised
    with torch.no_grad():
        triple[1].weight.copy_(torch.randn(8) * 0.5 + 0.2)  # includes negative gammas
        triple[1].bias.copy_(torch.randn(8) * 0.3)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(2, 3, 16, 16, generator=g, requires_grad=True)
    r = torch.randn(2, 8, 8, 8, generator=g)
    out = {"x": np_(x), "r": np_(r),
           "conv_weight": np_(triple[0].weight), "bn_weight": np_(triple[1].weight), "bn_bias": np_(triple[1].bias),
           "running_mean0": np_(triple[1].running_mean), "running_var0": np_(triple[1].running_var)}
    blk.train()
    y = pool(triple(x))
    (y * r).sum().backward()
    out.update({"train_out": np_(y), "train_dx": np_(x.grad), "train_dw": np_(triple[0].weight.grad),
                "train_dgamma": np_(triple[1].weight.grad), "train_dbeta": np_(triple[1].bias.grad),
                "running_mean1": np_(triple[1].running_mean), "running_var1": np_(triple[1].running_var),
                "num_batches_tracked1": np.array(int(triple[1].num_batches_tracked))})
    x.grad = None
    blk.zero_grad()
    blk.eval()
    y = pool(triple(x))
    (y * r).sum().backward()
    out.update({"eval_out": np_(y), "eval_dx": np_(x.grad), "eval_dw": np_(triple[0].weight.grad),
                "eval_dgamma": np_(triple[1].weight.grad), "eval_dbeta": np_(triple[1].bias.grad)})
    return out


def g2_two_stage(UNet):
    """G2: reference UNet(3,1,8) encoder path restricted to what the composite uses:
    encoder1[:3] -> pool1 -> encoder2[:3] -> pool2 (eval mode after one train step
    so running statistics are not the init values)."""
    torch.manual_seed(0)
    net = UNet(in_channels=3, out_channels=1, init_features=8)
    g = torch.Generator().manual_seed(2)
    x = torch.rand(2, 3, 32, 32, generator=g)
    net.train()
    with torch.no_grad():
        h = net.pool1(net.encoder1[:3](x))
        h = net.pool2(net.encoder2[:3](h))
    train_out = np_(h)
    net.eval()
    with torch.no_grad():
        h = net.pool1(net.encoder1[:3](x))
        h = net.pool2(net.encoder2[:3](h))
    sd = {k: np_(v) for k, v in net.state_dict().items()
          if k.startswith(("encoder1.enc1conv1", "encoder1.enc1norm1", "encoder2.enc2conv1", "encoder2.enc2norm1"))}
    out = {"x": np_(x), "train_out": train_out, "eval_out": np_(h)}
    out.update({"sd::" + k: v for k, v in sd.items()})
    return out


def main():
    sys.path.insert(0, REF)
    from UNet import UNet          # the reference class (never copied into this repo)
    np.savez_compressed(os.path.join(HERE, "g1_unet_block_stage.npz"), **g1_block_stage(UNet))
    np.savez_compressed(os.path.join(HERE, "g2_unet_two_stage.npz"), **g2_two_stage(UNet))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
