"""The training step as replayed hipGraphs: zero_grad -> forward -> cross-entropy -> backward -> (gradient all-reduce) -> AdamW.

Why: the step is ~70 small-to-medium kernel launches; issued eagerly they cost the host 1.4 ms of Python + launch time for ~1.6 ms
of GPU work (scripts/host_jitter.py), so any kernel speed-up beyond ~15 % -- and any slower host, e.g. 8 ranks sharing one
machine -- would leave the GPU waiting.  Captured once (torch.cuda.graphs = HIP stream capture) the step is one graph launch (three with data parallelism).

The reference has nothing comparable (single process, eager); this is host-side scheduling only: the
kernels, their order and their arithmetic are exactly those of the eager step, so results are bit-identical to eager for the
same dropout seeds.

What makes capture legal here
  * every buffer an operator needs is a torch allocation made during capture (graph-private pool): nothing is allocated or
    freed by the library, and no call synchronises;
  * step-dependent scalars live on the device: the kernels add a device counter to their by-value dropout seed, and AdamW forms
    its bias corrections from state['step'] + counter on the device (hyb_*'s seed_inc / step_inc arguments).  The counter is
    advanced inside the last graph, so every replay is a new step with new masks;
  * BatchNorm running statistics are updated in place by the captured statistics kernels (hybrid::backbone_).

Data parallelism (world > 1): the backward pass is captured in two pieces so that the gradient all-reduce of the temporal part
(25 of the 27 MB) runs -- eagerly, on the collective's own stream, outside any graph, so any torch.distributed backend works --
while the CNN backbone's backward graph executes:

    graph A: forward, loss, backward of head + encoder + token projection  -> their gradients in the "temporal" bucket
    all_reduce(temporal bucket, async)                                      [RCCL / xGMI]
    graph B: backward of the conv stages                                   -> their gradients in the "backbone" bucket
    all_reduce(backbone bucket, async); wait both
    graph C: AdamW over all parameters (gradients = views of the buckets), step counter += 1

With one rank there is nothing to exchange and no bucket: the gradient tensors produced under capture live at fixed addresses in
the graphs' private pool, so they are bound to ``p.grad`` and AdamW reads them in place (saves the two pack copies, ~25 us); and the
whole step is one graph (A, B and C captured back to back: every graph launch leaves the GPU idle for ~8 us, scripts/step_trace.py).
"""
import os

import torch
import torch.distributed as dist

from . import ops


class GraphedTrainStep:
    def __init__(self, model, criterion, optimizer, x, y, mask=None, process_group=None, warmup=3):
        if not (hasattr(model, "forward_backbone") and hasattr(model, "forward_temporal")):
            raise TypeError("GraphedTrainStep drives a TransformerCNNHybrid")
        if not hasattr(optimizer, "set_step_counter"):
            raise TypeError("GraphedTrainStep needs HybridAdamW (its step number lives on the device)")
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        from .modules import HybridCrossEntropyLoss
        self._fused_loss = (type(criterion) is HybridCrossEntropyLoss and hasattr(model, "forward_temporal_loss")
                            and os.environ.get("HYB_FUSED_LOSS", "1") != "0")       # (=0: A/B, the criterion as its own two launches)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        dev = x.device
        self.x, self.y = x.clone(), y.clone()                 # static inputs: copy new batches in with load()
        self.mask = mask.clone() if mask is not None else None
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self._one = None
        self.t_params = [p for p in model.temporal_parameters() if p.requires_grad]
        self.b_params = [p for p in model.backbone_parameters() if p.requires_grad]
        if len(self.t_params) + len(self.b_params) != sum(1 for p in model.parameters() if p.requires_grad):
            raise RuntimeError("model has trainable parameters outside its backbone and temporal parts")
        self.t_bucket, self.t_views = self._bucket(self.t_params) if self.world > 1 else (None, None)
        self.b_bucket, self.b_views = self._bucket(self.b_params) if self.world > 1 else (None, None)
        self._avg = self.world > 1 and dist.get_backend(process_group) == "nccl" and hasattr(dist.ReduceOp, "AVG")
        if self.world > 1:                                     # same start on every rank
            with torch.no_grad():
                for t in list(model.parameters()) + list(model.buffers()):
                    dist.broadcast(t.data, src=0, group=process_group)

        ops.set_step_counter(self.counter)
        # one parameter group: the AdamW launch itself advances the counter (no separate `counter += 1` launch per step)
        self._advancing = sum(1 for g in optimizer.param_groups if any(p.requires_grad for p in g["params"])) == 1
        optimizer.set_step_counter(self.counter, advance=self._advancing)
        if self.world > 1:
            for p, v in zip(self.t_params + self.b_params, self.t_views + self.b_views):
                p.grad = v                                     # AdamW reads the (all-reduced) buckets in place
        # warm-up on a side stream (lazy initialisation, allocator steady state), then capture
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._piece_a(); self._reduce_wait(self._reduce(self.t_bucket)); self._piece_b(); self._reduce_wait(self._reduce(self.b_bucket))
                self._piece_c()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self._warm_steps = max(1, warmup)
        self.ga, self.gb, self.gc_ = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.ga):
            self._piece_a()
        with torch.cuda.graph(self.gb, pool=self.ga.pool()):
            self._piece_b()
        self.gs = None
        params = self.t_params + self.b_params
        if self.world == 1 and os.environ.get("HYB_GRAPH_SINGLE", "1") != "0":      # (=0: A/B switch, the three-graph form)
            # one rank: nothing happens between the pieces, so the whole step is ALSO captured as one graph (every graph launch leaves the
            # GPU idle for ~8 us: two launches fewer per step).  Graphs A + B stay for fwd_bwd(); the two captures have their own gradient
            # tensors, and p.grad is re-bound to the set the last call wrote.
            self._grads_fb, self._loss_fb = [p.grad for p in params], self.loss
            self.gs = torch.cuda.CUDAGraph()
            # (its own memory pool: in A/B's pool the loss and gradient tensors step() returns could sit in blocks that A/B use as scratch,
            # and a later fwd_bwd() would overwrite them -- ADVICE r3)
            with torch.cuda.graph(self.gs):
                self._piece_a(); self._piece_b(); self._piece_c()
            self._grads_step, self._loss_step = [p.grad for p in params], self.loss
            self._bound = "step"
        else:
            with torch.cuda.graph(self.gc_, pool=self.ga.pool()):
                self._piece_c()
        # the captures above did not execute: the state is still "after the warm-up steps"

    def _bind(self, which):
        if self.gs is not None and self._bound != which:
            for p, g in zip(self.t_params + self.b_params, self._grads_step if which == "step" else self._grads_fb):
                p.grad = g
            self._bound = which

    @staticmethod
    def _bucket(params):
        total = sum(p.numel() for p in params)
        flat = torch.zeros(max(total, 1), dtype=torch.float32, device=params[0].device if params else "cuda")
        views, off = [], 0
        for p in params:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        return flat, views

    # ---- the three pieces (identical code runs eagerly in the warm-up and under capture) -------------------------------
    def _piece_a(self, bind=True):
        h, B = self.model.forward_backbone(self.x)
        if self._fused_loss:                                   # the loss rides in the temporal part's last launch, its backward in the backward's first
            loss, logits = self.model.forward_temporal_loss(h, B, self.y, self.mask)
        else:
            logits = self.model.forward_temporal(h, B, self.mask)
            loss = self.criterion(logits, self.y)
        self.logits = logits.detach()
        if self._one is None:                                  # d(loss)/d(loss): a constant, not a fill launch per step
            self._one = torch.ones_like(loss)
        grads = torch.autograd.grad(loss, [h] + self.t_params, grad_outputs=self._one)
        self._h, self._gh = h, grads[0]
        self._deliver(self.t_params, self.t_views, grads[1:], bind)
        self.loss = loss.detach()

    def _piece_b(self, bind=True):
        grads = torch.autograd.grad(self._h, self.b_params, grad_outputs=self._gh)
        self._deliver(self.b_params, self.b_views, grads, bind)
        self._h = self._gh = None

    def _deliver(self, params, views, grads, bind):
        if views is not None:
            torch._foreach_copy_(views, list(grads))
        elif bind:                                             # one rank: the captured gradient tensors themselves (static addresses)
            for p, g in zip(params, grads):
                p.grad = g

    def _piece_c(self):
        if self.world > 1 and not self._avg:
            self.t_bucket.mul_(1.0 / self.world)
            self.b_bucket.mul_(1.0 / self.world)
        self.optimizer.step()
        if not self._advancing:
            self.counter.add_(1)

    def _reduce(self, bucket):
        if self.world == 1:
            return None
        return dist.all_reduce(bucket, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.group, async_op=True)

    @staticmethod
    def _reduce_wait(work):
        if work is not None:
            work.wait()

    # ---- public ---------------------------------------------------------------------------------------------------------
    def load(self, x, y, mask=None):
        """Copy the next batch into the static input buffers (same shapes as at construction)."""
        self.x.copy_(x, non_blocking=True)
        self.y.copy_(y, non_blocking=True)
        if mask is not None:
            self.mask.copy_(mask, non_blocking=True)

    def step(self):
        """One training step; returns the (device) loss tensor of this step -- reading it synchronises."""
        if self.gs is not None:
            self._bind("step")
            self.gs.replay()
            self.loss = self._loss_step
            return self.loss
        self.ga.replay()
        w1 = self._reduce(self.t_bucket)                       # rides under graph B
        self.gb.replay()
        w2 = self._reduce(self.b_bucket)
        self._reduce_wait(w1)
        self._reduce_wait(w2)
        self.gc_.replay()
        return self.loss

    def fwd_bwd(self):
        """Forward + loss + backward (+ all-reduce) only: no optimizer, the step counter does not advance."""
        self._bind("fb")
        self.ga.replay()
        w1 = self._reduce(self.t_bucket)
        self.gb.replay()
        w2 = self._reduce(self.b_bucket)
        self._reduce_wait(w1)
        self._reduce_wait(w2)
        if self.gs is not None:
            self.loss = self._loss_fb
        return self.loss

    def eager_fwd_bwd(self):
        """The same forward + backward issued launch by launch (for per-kernel event timing, which needs live launches)."""
        self._piece_a(bind=False)
        self._piece_b(bind=False)
        return self.loss

    def steps_done(self):
        """Steps taken through this object, warm-up included (reads the device counter: synchronises)."""
        return int(self.counter.item())

    def sync_optimizer_state(self):
        """Fold the device step counter into the optimizer's Python-side state (before state_dict())."""
        n = self.steps_done()
        for st in self.optimizer.state.values():
            if "step" in st:
                st["step"] = int(st["step"]) + n
        self.counter.zero_()

    def close(self):
        ops.set_step_counter(None)
        self.optimizer.set_step_counter(None)
