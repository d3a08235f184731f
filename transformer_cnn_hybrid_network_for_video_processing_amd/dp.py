"""Batch-of-clips data parallelism: one process per GPU, gradients all-reduced (averaged) over
torch.distributed -- backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.

The reference has no distributed code at all; this is the one exchange step the
sharded hot path needs.  Clips are independent, so there is no data-path collective:
only parameter gradients (6.83 M fp32 = 27.3 MB at BASELINE config 2) are exchanged.  BatchNorm
statistics stay per-rank (the reference has no SyncBN), so W ranks compute "W independent batch-B
BatchNorm forward/backward passes, gradients averaged".

Buckets are filled in backward-readiness order (head -> encoder -> token projection -> conv4 -> ... -> conv1) and are cut
where readiness changes: a bucket is closed when it is full (``bucket_bytes``) and also when the next parameter belongs to
another top-level submodule while the bucket already holds ``min_bucket_bytes``.  At BASELINE config 2 that gives
[head + encoder, 16 MB] [rest of the encoder, 9 MB] [token projection + conv4, 1.7 MB] [conv3..conv1, 0.4 MB]: the first
two are complete as soon as the encoder backward returns and travel under the whole CNN backward (>99 % of the backward
FLOPs, ~0.9 ms), the third under conv3..conv1; only the last 0.4 MB is exposed.  (One 32 MB bucket -- the whole model --
would start its all-reduce only after the last gradient and overlap nothing.)  A bucket's all-reduce is launched
asynchronously from the autograd hook of its last-arriving gradient.  xGMI is point-to-point (7 links x ~153 GB/s per
GPU): a 25 MB ring all-reduce is ~0.3 ms.
"""
import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, module, process_group=None, bucket_bytes=16 << 20, min_bucket_bytes=1 << 20, broadcast=True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        # the named_parameters order, reversed, is ~ the order gradients become ready in backward
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        self.params = [p for _, p in named]
        if broadcast:
            self.broadcast_state()
        # NCCL/RCCL can average inside the collective; gloo cannot (sum, then scale)
        self._avg = dist.get_backend(process_group) == "nccl" and hasattr(dist.ReduceOp, "AVG")
        self.buckets = []            # each: dict(params, flat, views, pending, work)
        cur, cur_bytes, cur_top = [], 0, None
        for name, p in reversed(named):
            nb = p.numel() * 4
            top = name.split(".", 1)[0]
            if cur and (cur_bytes + nb > bucket_bytes or (top != cur_top and cur_bytes >= min_bucket_bytes)):
                self._close_bucket(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
            cur_top = top
        if cur:
            self._close_bucket(cur)
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._bucket_of[p] = bi
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]

    def _close_bucket(self, params):
        total = sum(p.numel() for p in params)
        flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
        views, off = [], 0
        for p in params:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.buckets.append(dict(params=params, flat=flat, views=views, pending=len(params), work=None))

    def broadcast_state(self):
        """Rank 0's parameters and buffers to every rank (one-time, at construction)."""
        with torch.no_grad():
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                dist.broadcast(t.data, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)

    def _hook(self, p):
        b = self.buckets[self._bucket_of[p]]
        b["pending"] -= 1
        if b["pending"] == 0:
            self._fill(b)
            b["work"] = self._launch(b)

    def _fill(self, b):
        dst, src = [], []
        for q, v in zip(b["params"], b["views"]):
            if q.grad is None:
                v.zero_()
            elif q.grad.data_ptr() != v.data_ptr():          # (a gradient that already lives in its view needs no copy)
                dst.append(v); src.append(q.grad)
        if dst:
            torch._foreach_copy_(dst, src)

    def _launch(self, b):
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        return dist.all_reduce(b["flat"], op=op, group=self.group, async_op=True)

    def finalize(self):
        """Wait for the in-flight buckets, average, and write the result back into p.grad."""
        inv = 1.0 / self.world
        for b in self.buckets:
            if b["work"] is None:
                if b["pending"] != len(b["params"]):
                    # some gradients of this bucket never arrived (unused parameters): reduce what is there
                    self._fill(b)
                    b["work"] = self._launch(b)
                else:
                    continue
            b["work"].wait()
            if not self._avg:
                b["flat"].mul_(inv)
            # hand the averaged gradients over WITHOUT a copy: p.grad becomes the bucket view.  The next step's zero_grad
            # (set_to_none, the default) drops the views before the bucket is refilled; if the caller accumulates instead, autograd
            # adds into the view in place and the refill copies it onto itself.
            for q, v in zip(b["params"], b["views"]):
                q.grad = v
            b["work"] = None
            b["pending"] = len(b["params"])

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def shard_batch(global_batch, rank, world):
    """Clips [rank::world] of a global batch -- independent units, no data-path collective."""
    if global_batch % world != 0:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per
