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
        self._step_counter = None    # device int64 [1]: the kernel uses step + counter (captured launches, graph.GraphedTrainStep)
        self._advance = False
        self._ticket = None          # device uint32 [1], this optimizer's own: the advancing launch counts its finished workgroups there

    def set_step_counter(self, counter, advance=False):
        """With a device counter the step number used by the kernel is state['step'] + counter, read on the device: one captured
        launch then serves every replay of a hipGraph.  advance=True: step() also adds 1 to the counter (inside the AdamW launch, after
        every workgroup has read it) -- the caller then needs no separate `counter += 1` launch; every parameter must then be in
        ONE group, so that one launch ends the step."""
        self._step_counter = counter
        self._advance = bool(advance) and counter is not None
        if self._advance and (self._ticket is None or self._ticket.device != counter.device):
            self._ticket = torch.zeros(1, dtype=torch.int32, device=counter.device)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables.clear()
        for st in self.state.values():              # torch casts loaded state to the parameter's dtype/device; keep the layout the kernel needs
            for k in ("exp_avg", "exp_avg_sq"):
                if k in st:
                    st[k] = st[k].to(torch.float32).contiguous()
            if "step" in st and torch.is_tensor(st["step"]):
                st["step"] = int(st["step"].item())

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, "_tables"):
            self._tables.clear()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        live = [gi for gi, group in enumerate(self.param_groups) if any(p.grad is not None for p in group["params"])]
        if self._advance and len(live) != 1:
            raise RuntimeError("HybridAdamW: an advancing step counter needs exactly one parameter group with gradients")
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
                if self._step_counter is None:
                    st["step"] += 1
            # with a device counter the Python-side step stays put and the kernel uses (step + 1) + counter
            steps = {int(self.state[p]["step"]) + (1 if self._step_counter is not None else 0) for p in ps}
            if len(steps) != 1:
                raise RuntimeError("HybridAdamW: parameters of one group must share the step count")
            # the cached pointer tables are valid only while every parameter AND both of its moment tensors stay where they are:
            # load_state_dict() / a rollback replaces the moments with new allocations (the old ones may already be freed)
            addrs = [(p.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr()) for p in ps]
            tab = self._tables.get(gi)
            if tab is None or tab[1] != addrs:
                tab = (None, addrs, ptr_array([a[0] for a in addrs]), ptr_array([a[1] for a in addrs]), ptr_array([a[2] for a in addrs]),
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
                     float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), steps.pop(),
                     self._step_counter.data_ptr() if self._step_counter is not None else None,
                     self._ticket.data_ptr() if self._advance else None, _stream())
        return loss
