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
  * "auto": MEASURE, then choose between "conv" and "serial".  Overlap is not free on this part: the collective's channel kernels take CUs
    and memory bandwidth from the conv-stack backward they hide under (scripts/r6_dp_window.py prices that on one GPU).  Steps 1-2 run
    "conv", steps 3-4 "serial", each with an event at the first bucket of backward and one after the last collective; the MAX over ranks of
    the two spans decides (the same number on every rank, so every rank switches alike), and `auto_report` says what was chosen and why.
    Costs two device synchronisations and one 2-float all-reduce, once, inside the warm-up (>= 5 warm-up steps).

RCCL channels: the schedules assume RCCL's defaults (it sizes its channel count to the topology: on an 8-GPU xGMI mesh up to 32 channels of
one workgroup each for a 260 MB all-reduce).  A collective NEVER runs beside a persistent recurrence in "conv" / "serial" / "auto", so
the channel count only prices the conv-stack window; cap it with NCCL_MAX_NCHANNELS (e.g. 16) if `dist.conv_backward_ms` of a multi-GPU line
grows by more than the collective's own duration.

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
        if self.mode not in ("conv", "serial", "overlap", "auto"):
            raise ValueError(f"DS2_DP_MODE={self.mode!r}: expected conv, serial, overlap or auto")
        # "auto": starts as "conv", measures both, settles on one (see _auto_step); auto_report: None until then
        self.auto = self.mode == "auto" and flat_grad.is_cuda
        self.auto_report: Optional[dict] = None
        self._auto_steps, self._auto_ev, self._auto_spans = 0, None, {}
        if self.mode == "auto":
            self.mode = "conv"
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
        if self.auto and self._auto_ev is None and self._auto_steps in (1, 2, 3, 4):
            self._auto_ev = torch.cuda.Event(enable_timing=True)      # first bucket of this backward (fc): where communication could start
            self._auto_ev.record(torch.cuda.current_stream())
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
            # what the compute stream waits for at the end of backward: the big collective's tail + the (serialised) conv bucket
            out["exposed_comm_ms"] = out["big_collective_outlasts_conv_backward_ms"] + sum(c["start"].elapsed_time(c["end"]) for c in convs) / len(convs)
        elif self.mode == "serial":
            steps = max(len(convs), 1)
            out["exposed_comm_ms"] = sum(r["start"].elapsed_time(r["end"]) for r in recs) / steps      # nothing overlaps: every collective is exposed
        out["schedule_chosen"] = self.mode
        if self.auto_report:
            out["auto"] = self.auto_report
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
        if self.auto:
            self._auto_step()

    def _auto_step(self):
        """DS2_DP_MODE=auto, called at the end of every step until the choice is made: steps 1-2 in "conv", 3-4 in "serial", the span from
        backward's first bucket to the compute stream having all reduced gradients is timed in both, the MAX over ranks decides."""
        k = self._auto_steps
        self._auto_steps += 1
        if self._auto_ev is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record(torch.cuda.current_stream())
            self._auto_spans.setdefault(self.mode, []).append((self._auto_ev, end))
            self._auto_ev = None
        if k == 2:                                       # two "conv" steps timed: now two "serial" ones
            self._set_mode("serial")
        elif k == 4:
            torch.cuda.synchronize(self.flat_grad.device)
            span = {m: sum(a.elapsed_time(b) for a, b in v) / len(v) for m, v in self._auto_spans.items()}
            t = torch.tensor([span.get("conv", 0.0), span.get("serial", 0.0)], dtype=torch.float64, device=self.flat_grad.device)
            if self.world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            conv_ms, serial_ms = float(t[0]), float(t[1])
            chosen = "conv" if conv_ms <= serial_ms else "serial"
            self._set_mode(chosen)
            self.auto_report = {"schedule_chosen": chosen, "backward_with_comm_ms": {"conv": conv_ms, "serial": serial_ms},
                                "measured_over_steps": {m: len(v) for m, v in self._auto_spans.items()}}
            self.auto = False
            self._auto_spans = {}

    def _set_mode(self, mode: str):
        self.mode = mode
        self.use_stream = self.flat_grad.is_cuda and mode in ("overlap", "conv")
        if self.use_stream and self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream(device=self.flat_grad.device)

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
