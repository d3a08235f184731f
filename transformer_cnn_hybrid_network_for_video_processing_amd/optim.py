"""Fused AdamW: the optimizer step that follows the hot path in the reference harnesses
(`optim.AdamW(model.parameters(), lr)`, Model.py:153 / FCT.py:305), as ONE HIP launch over all parameter tensors
(`hyb_adamw_step`).  Same constructor defaults and update rule as `torch.optim.AdamW` (betas (0.9, 0.999), eps 1e-8,
weight_decay 1e-2, amsgrad/maximize off); state-dict keys (`step`, `exp_avg`, `exp_avg_sq`) follow torch's so checkpoints
interchange.  CUDA fp32 parameters only -- there is no CPU fallback."""
import ctypes

import torch

from ._lib import lib, ptr_array
from .ops import _stream


class HybridAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}            # per group: cached pointer tables of the tensors whose addresses never change

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("HybridAdamW: contiguous fp32 CUDA parameters only (no CPU fallback)")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
            steps = {int(self.state[p]["step"]) for p in ps}
            if len(steps) != 1:
                raise RuntimeError("HybridAdamW: parameters of one group must share the step count")
            key = tuple(id(p) for p in ps)
            tab = self._tables.get(gi)
            if tab is None or tab[0] != key or any(p.data_ptr() != a for p, a in zip(ps, tab[1])):
                addrs = [p.data_ptr() for p in ps]
                tab = (key, addrs, ptr_array(addrs), ptr_array([self.state[p]["exp_avg"].data_ptr() for p in ps]),
                       ptr_array([self.state[p]["exp_avg_sq"].data_ptr() for p in ps]),
                       (ctypes.c_longlong * len(ps))(*[p.numel() for p in ps]))
                self._tables[gi] = tab
            grads = []
            for p in ps:
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous()
                grads.append(g)
            b1, b2 = group["betas"]
            lib.call("hyb_adamw_step", len(ps), tab[2], ptr_array([g.data_ptr() for g in grads]), tab[3], tab[4], tab[5],
                     float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), steps.pop(), _stream())
        return loss
