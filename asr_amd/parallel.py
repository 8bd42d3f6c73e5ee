"""Single-node data parallelism: one process per GPU, RCCL all-reduce of the flat gradient buffer, launched in backward order.
Three schedules (DS2_DP_MODE):
  * "conv" (default) — BASELINE's north_star schedule, "all-reduce overlapped with the backward conv": the fc and RNN buckets are held
    back until the LAST recurrent layer's backward has been enqueued, then reduced as ONE collective over their contiguous span of the flat
    buffer (~99.6 % of the gradient bytes; few large collectives are what the point-to-point xGMI links want) on a communication stream,
    WHILE the conv-stack backward (BN2d/conv2 dgrad + wgrad/conv1 wgrad, ~3 ms at c3) runs on the compute stream; the small conv bucket
    follows.  No collective ever runs beside a persistent recurrence kernel (those need every workgroup resident at once), so backward
    keeps them.
  * "serial": every bucket's all-reduce is ordered INTO the compute stream right where its gradients become final.  Nothing overlaps.
  * "overlap": buckets are reduced on a side stream while the rest of backward runs; the trainer then switches the persistent BACKWARD
    recurrence off (ds2_rnn_persistent_enable(1, 0)), which costs more than the communication it hides.

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
        self.mode = os.environ.get("DS2_DP_MODE", "conv")
        if self.mode not in ("conv", "serial", "overlap"):
            raise ValueError(f"DS2_DP_MODE={self.mode!r}: expected conv, serial or overlap")
        self.use_stream = flat_grad.is_cuda and self.mode in ("overlap", "conv")
        # does a collective ever run while backward's recurrences are still being executed?  (only then must the persistent backward go)
        self.overlaps_recurrence = flat_grad.is_cuda and self.mode == "overlap"
        self.comm_stream = torch.cuda.Stream(device=flat_grad.device) if self.use_stream else None
        self._pending = []
        self.launched: List[str] = []
        self._held: List[str] = []
        # measurement (bench.py sets it to a list): per collective {name, elements, start / end events on the stream it was issued on}, and per
        # step an event on the compute stream where the conv-stack backward starts (= the big bucket's release) and ends (the conv bucket)
        self.timing: Optional[list] = None
        # "conv" schedule: the bucket whose completion releases the held ones = the first recurrent layer (last in backward order)
        rnn_names = [n for n in self.buckets if n.startswith("rnns.")]
        self._release_on = min(rnn_names, key=lambda n: int(n.split(".")[1])) if rnn_names else None

    @property
    def active(self) -> bool:
        """False on a single rank (unless forced): on_bucket / finish do nothing, the gradient buffers are never touched by a collective."""
        return self.world > 1 or bool(self.force)

    def on_bucket(self, name: str):
        """Called by engine.backward when bucket `name`'s gradient kernels are enqueued."""
        if self.world == 1 and not self.force:
            return
        if self.mode == "conv" and name != "conv" and self._release_on is not None:
            # fc / recurrent layers: hold until the last of them is final, then ONE all-reduce over their (contiguous) span
            self._held.append(name)
            if name != self._release_on:
                return
            a = min(self.buckets[n][0] for n in self._held)
            b = max(self.buckets[n][1] for n in self._held)
            assert sum(self.buckets[n][1] - self.buckets[n][0] for n in self._held) == b - a, "held buckets must be contiguous"
            self.launched.append("+".join(self._held))
            self._held = []
        else:
            a, b = self.buckets[name]
            self.launched.append(name)
        view = self.flat_grad[a:b]
        rec = None
        if self.timing is not None and view.is_cuda:
            rec = {"name": self.launched[-1], "elements": b - a, "mark": torch.cuda.Event(enable_timing=True),
                   "start": torch.cuda.Event(enable_timing=True), "end": torch.cuda.Event(enable_timing=True)}
            rec["mark"].record(torch.cuda.current_stream())          # where the caller's stream stands when the bucket is released
            self.timing.append(rec)
        if self.use_stream:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ready)
                if rec:
                    rec["start"].record(self.comm_stream)
                work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if rec:
                    work.wait()                                      # (stream-level for RCCL; gloo stages through the host and blocks it)
                    rec["end"].record(self.comm_stream)
            self._pending.append(work)
        else:
            if rec:
                rec["start"].record(torch.cuda.current_stream())
            work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            if view.is_cuda:
                work.wait()          # stream-level: the compute stream waits for the collective, the host does not
                if rec:
                    rec["end"].record(torch.cuda.current_stream())
            else:
                self._pending.append(work)

    def timing_summary(self) -> dict:
        """Per-step averages (ms) from the event records collected while `timing` was a list: the largest collective's duration on its
        stream, the span of the conv-stack backward on the compute stream (from the release of the fc + RNN bucket to the conv bucket), and
        how long the big collective outlasted the conv-stack backward (0: completely hidden).  Call after a device synchronisation."""
        recs = self.timing or []
        if not recs:
            return {"collectives_per_step": 0}
        big_elems = max(r["elements"] for r in recs)
        bigs = [r for r in recs if r["elements"] == big_elems]
        convs = [r for r in recs if r["name"] == "conv"]
        out = {"collectives_per_step": len(recs) / max(len(bigs), 1), "big_collective_bytes": big_elems * 4,
               "big_collective_ms": sum(r["start"].elapsed_time(r["end"]) for r in bigs) / len(bigs)}
        if self.mode == "conv" and len(convs) == len(bigs):
            span = [b["mark"].elapsed_time(c["mark"]) for b, c in zip(bigs, convs)]
            tail = [max(0.0, b["mark"].elapsed_time(b["end"]) - sp) for b, sp in zip(bigs, span)]
            out["conv_backward_ms"] = sum(span) / len(span)
            out["big_collective_outlasts_conv_backward_ms"] = sum(tail) / len(tail)
        return out

    def finish(self):
        """Block the compute stream until every bucket is reduced.  Gradients hold the SUM over ranks;
        the 1/world factor is folded into the optimizer (FusedAdamW.grad_scale)."""
        for w in self._pending:
            w.wait()
        if self.use_stream:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self._pending = []
        self.launched = []
        assert not self._held, f"buckets never released: {self._held}"

    def all_valid_device(self, flag: torch.Tensor) -> torch.Tensor:
        """Device-side agreement: `flag` (int32 GPU tensor, 1 = this rank's step is valid) becomes the MIN over ranks, in stream order,
        without a host synchronisation (the gated optimizer launch reads it).  "Without a host synchronisation" holds on `nccl` (= RCCL),
        whose `work.wait()` is a stream dependency; `gloo` (CPU tests, or CUDA tensors staged through host memory) blocks the host there —
        correct, but not the overlap the run-ahead train step is built for.  DS2_FORCE_ALLREDUCE=1 runs the reduction with one rank too,
        so that a 1-GPU box exercises the same code as a multi-rank run."""
        if self.world == 1 and not self.force:
            return flag
        work = dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group, async_op=True)
        work.wait()                                  # stream-level for CUDA tensors: the compute stream waits, the host does not
        return flag

    def all_valid(self, valid: bool, device) -> bool:
        """Collective agreement on check_loss (every rank must skip the same steps)."""
        if self.world == 1:
            return valid
        flag = torch.tensor([1 if valid else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(flag.item() == 1)
