"""Generate golden vectors from the REFERENCE's own importable classes.

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_golden.py

It imports ``UNet`` from the reference (UNet.py is importable as-is),
drives the reference's Conv+BN+ReLU unit (``UNet._block``, UNet.py:54-66) and its
``MaxPool2d(2,2)`` (UNet.py:13) on seeded inputs, and -- for the "next" row FCT -- loads the
class definitions of FCT.py:24-254 and Metrics.py:5-39 by line range and drives every FCT block and the whole
model (G3-G8).  Inputs, parameters, outputs and gradients are stored as plain arrays in ``.npz``
files (data only -- no reference source text is stored).  tests/test_oracle.py replays them against
``oracle/hybrid_ref.py``; the GPU parity tests replay them against the HIP path.
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("HYB_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


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


def load_reference_classes():
    """The reference's FCT.py / Metrics.py cannot be imported as modules (FCT.py:4 needs torchsummary and FCT.py:410-412 starts
    training at import; Metrics.py:3 needs pytorch_msssim) but their class DEFINITIONS load when only those line ranges are
    executed with {torch, nn, np} in scope.  The source text is read, executed here and dropped: nothing of it
    is stored in this repository."""
    ns = {"torch": torch, "nn": torch.nn, "np": np}
    lines = open(os.path.join(REF, "FCT.py")).read().split("\n")
    exec(compile("\n".join(lines[23:254]), "reference FCT.py:24-254", "exec"), ns)           # Attention .. FCT
    lines = open(os.path.join(REF, "Metrics.py")).read().split("\n")
    exec(compile("\n".join(lines[4:39]), "reference Metrics.py:5-39", "exec"), ns)            # DiceLoss, JaccardScore
    return ns


def _sd(m):
    return {"sd::" + k: np_(v) for k, v in m.state_dict().items()}


def _run_block(m, x, extra=()):
    """eval-mode forward + gradients of (out * r).sum() w.r.t. the input and every parameter that takes part."""
    m.eval()
    x = x.clone().requires_grad_(True)
    y = m(x, *extra)
    g = torch.Generator().manual_seed(7)
    r = torch.randn(y.shape, generator=g)
    (y * r).sum().backward()
    out = {"x": np_(x), "r": np_(r), "out": np_(y), "dx": np_(x.grad)}
    out.update({"grad::" + k: np_(p.grad) for k, p in m.named_parameters() if p.grad is not None})
    out.update(_sd(m))
    return out


def g3_to_g8(ns):
    """G3-G8 of SURVEY.md section 8c: per-block vectors of the reference's FCT (FCT.py:24-254) and DiceLoss (Metrics.py:5-22)."""
    res = {}
    g = torch.Generator().manual_seed(11)
    torch.manual_seed(0)
    res["g3_fct_block_first"] = _run_block(ns["Block_encoder_bottleneck"]("first", 3, 8, 2, 0), torch.rand(1, 3, 32, 32, generator=g))
    torch.manual_seed(0)
    blk = ns["Block_encoder_bottleneck"]("second", 8, 16, 2, 0)
    scale = torch.rand(1, 3, 16, 16, generator=g)
    d = _run_block(blk, torch.rand(1, 8, 16, 16, generator=g), (scale,))
    d["scale_img"] = np_(scale)
    res["g3b_fct_block_second"] = d
    torch.manual_seed(0)
    res["g4_fct_attention"] = _run_block(ns["Attention"](8, 2), torch.rand(1, 8, 8, 8, generator=g))
    torch.manual_seed(0)
    res["g4b_fct_transformer"] = _run_block(ns["Transformer"](8, 8, 2), torch.rand(1, 8, 8, 8, generator=g))
    torch.manual_seed(0)
    res["g5_fct_wide_focus"] = _run_block(ns["Wide_Focus"](8, 8), torch.rand(2, 8, 12, 12, generator=g))
    torch.manual_seed(0)
    dec = ns["Block_decoder"](16, 8, 2, 0)
    skip = torch.rand(1, 8, 16, 16, generator=g)
    d = _run_block(dec, torch.rand(1, 16, 8, 8, generator=g), (skip,))
    d["skip"] = np_(skip)
    res["g6_fct_block_decoder"] = d
    torch.manual_seed(0)
    res["g6b_fct_ds_out"] = _run_block(ns["DS_out"](8, 1), torch.rand(1, 8, 8, 8, generator=g))
    # G7: DiceLoss on seeded data (0.558098316 with this seed) + the 2x2 known answer (0.2) + its gradient
    torch.manual_seed(0)
    pred = torch.rand(2, 1, 8, 8).requires_grad_(True)
    true = (torch.rand(2, 1, 8, 8) > 0.5).float()
    loss = ns["DiceLoss"]()(pred, true)
    loss.backward()
    kat = ns["DiceLoss"]()(torch.tensor([[0.5, 0.5], [1.0, 0.0]]).view(1, 1, 2, 2), torch.tensor([[1.0, 0.0], [1.0, 0.0]]).view(1, 1, 2, 2))
    res["g7_dice_loss"] = {"pred": np_(pred), "true": np_(true), "loss": np.array(loss.item()), "dpred": np_(pred.grad), "kat_loss": np.array(kat.item())}
    # G8: the whole reference model, eval mode, with the gradient of the Dice loss.  Its 2.1 M parameters are set from the name-keyed
    # formula of det_init.py (the tests regenerate them), gradients are stored as digests (sum, norm, 32 sampled elements) -- the
    # first conv's and the last conv's gradients in full.  G8s: the survey's own check (default init under manual_seed(0): min/max only).
    from det_init import det_state_dict, digest
    torch.manual_seed(0)
    m = ns["FCT"]().eval()
    x = torch.rand(1, 3, 64, 64)
    with torch.no_grad():
        o0 = m(x)
    res["g8s_fct_default_init"] = {"out_min": np.array(float(o0.min())), "out_max": np.array(float(o0.max())), "out_mean": np.array(float(o0.mean()))}
    m.load_state_dict(det_state_dict(m))
    y_true = (torch.rand(1, 1, 64, 64) > 0.5).float()
    out = m(x)
    loss = ns["DiceLoss"]()(out, y_true)
    loss.backward()
    d = {"x": np_(x), "y_true": np_(y_true), "out": np_(out), "loss": np.array(loss.item()),
         "param_names": np.array([k for k, _ in m.named_parameters()]), "param_numel": np.array([p.numel() for p in m.parameters()])}
    d.update({"gdig::" + k: digest(p.grad) for k, p in m.named_parameters() if p.grad is not None})
    for k in ("block_1.conv1_a.weight", "ds.conv3.weight", "block_5.trans.attention_output.attention.in_proj_weight"):
        d["grad::" + k] = np_(dict(m.named_parameters())[k].grad)
    d["unused_parameters"] = np.array([k for k, p in m.named_parameters() if p.grad is None])
    res["g8_fct_full"] = d
    print("G7 loss", res["g7_dice_loss"]["loss"], "kat", res["g7_dice_loss"]["kat_loss"], "G8s (default init) out min/max", float(o0.min()),
          float(o0.max()), "G8 out min/max", float(out.min()), float(out.max()), "loss", loss.item(), "params",
          sum(p.numel() for p in m.parameters()), "without grad:", len(d["unused_parameters"]))
    return res


def main():
    sys.path.insert(0, REF)
    from UNet import UNet          # the reference class (never copied into this repo)
    np.savez_compressed(os.path.join(HERE, "g1_unet_block_stage.npz"), **g1_block_stage(UNet))
    np.savez_compressed(os.path.join(HERE, "g2_unet_two_stage.npz"), **g2_two_stage(UNet))
    for name, d in g3_to_g8(load_reference_classes()).items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
