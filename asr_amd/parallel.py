"""Single-node data parallelism: one process per GPU, RCCL all-reduce of the flat gradient buffer in per-layer buckets launched in
backward order.  Two schedules (DS2_DP_MODE):
  * "serial" (default): every bucket's all-reduce is ordered INTO the compute stream right where its gradients become final.  Nothing
    overlaps, but nothing ever runs beside the persistent recurrence kernels either (they need every workgroup resident at once), so
    backward keeps them: 188 MB of all-reduce per step (~1-2.5 ms on 8 GPUs over xGMI) is cheaper than the 7.5 ms the per-step
    backward kernels cost.
  * "overlap": buckets are reduced on a side stream while the rest of backward runs; the trainer then switches the persistent BACKWARD
    recurrence off (ds2_rnn_persistent_enable(1, 0)).

The reference has no live distributed code; semantics defined in SURVEY §8(e):
per-rank BatchNorm statistics (plain DDP), gradients = mean over ranks of each rank's
d(sum_shard CTC / B_local), loss-validity agreed collectively before the optimizer step.
Works on any backend: `nccl` (= RCCL over xGMI on ROCm) on GPUs, `gloo` on CPU tensors for tests.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


class BucketedAllReducer:
    def __init__(self, flat_grad: torch.Tensor, buckets: List[Tuple[str, int, int]], group=None):
        self.flat_grad = flat_grad
        self.buckets: Dict[str, Tuple[int, int]] = {n: (a, b) for n, a, b in buckets}
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # DS2_FORCE_ALLREDUCE=1: run the bucketed all-reduce even with a single rank (exercises the RCCL / side-stream
        # path on a 1-GPU box; a 1-rank SUM is the identity)
        self.force = dist.is_initialized() and os.environ.get("DS2_FORCE_ALLREDUCE") == "1"
        self.mode = os.environ.get("DS2_DP_MODE", "serial")
        self.use_stream = flat_grad.is_cuda and self.mode == "overlap"
        self.comm_stream = torch.cuda.Stream(device=flat_grad.device) if self.use_stream else None
        self._pending = []
        self.launched: List[str] = []

    def on_bucket(self, name: str):
        """Called by engine.backward when bucket `name`'s gradient kernels are enqueued."""
        if self.world == 1 and not self.force:
            return
        a, b = self.buckets[name]
        view = self.flat_grad[a:b]
        self.launched.append(name)
        if self.use_stream:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ready)
                work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending.append(work)
        else:
            work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            if view.is_cuda:
                work.wait()          # stream-level: the compute stream waits for the collective, the host does not
            else:
                self._pending.append(work)

    def finish(self):
        """Block the compute stream until every bucket is reduced.  Gradients hold the SUM over ranks;
        the 1/world factor is folded into the optimizer (FusedAdamW.grad_scale)."""
        for w in self._pending:
            w.wait()
        if self.use_stream:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self._pending = []
        self.launched = []

    def all_valid(self, valid: bool, device) -> bool:
        """Collective agreement on check_loss (every rank must skip the same steps)."""
        if self.world == 1:
            return valid
        flag = torch.tensor([1 if valid else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(flag.item() == 1)
