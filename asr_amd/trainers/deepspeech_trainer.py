"""`DeepSpeechTrainer` — same constructor arguments, `fit`/`train`/`test`/`update`/`load`/`save`
behaviour as asr_deepspeech/trainers/deepspeech_trainer.py:14-188, standalone (the reference derives
from `sakura.ml.SakuraTrainer`, a third-party class whose source is not in the tree), plus:

  * `step(data)`   — the fused MI355X train step: forward, CTC, hand-written backward, bucketed RCCL
                     gradient all-reduce overlapped with backward, fused AdamW; ONE host sync per step
                     (the reference's `loss.item()`), validity of the loss agreed across ranks.
  * data-parallel operation when torch.distributed is initialised (one process per GPU).
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import torch
import torch.distributed as dist

from .. import engine, ops
from ..ctc import CTCLoss, _prep_targets, _prep_targets_host
from ..device import autocast, make_grad_scaler, resolve_device
from ..functional import check_loss
from ..optim import FusedAdamW
from ..parallel import BucketedAllReducer


class Epochs:
    """Iterable epoch counter with the attributes the reference trainer touches
    (`_epochs.current/.total/.best/.start`, deepspeech_trainer.py:50,66,137,166-170)."""

    def __init__(self, total, start=0):
        self.total, self.start, self.current, self.best = int(total), int(start), int(start), int(start)

    def __iter__(self):
        for e in range(self.start, self.total):
            self.current = e
            yield e


def asr_metrics():
    """Minimal stand-in for `sakura.functional.asr_metrics`: metrics.{train,test}.{current,best}."""
    def blank():
        return SimpleNamespace(loss=0.0, wer=None, cer=None)
    return SimpleNamespace(train=SimpleNamespace(current=blank(), best=blank()), test=SimpleNamespace(current=blank(), best=blank()))


class DeepSpeechTrainer:
    starved_steps = 0          # process-wide count of train steps skipped because a persistent recurrence launch starved (bench.py reports it)

    def __init__(self, model, criterion, epochs, metrics, optimizer, model_path, checkpoint_path, device, device_test,
                 mixed_precision, output_file, scheduler=None, overwrite_lr=None):
        self._device = resolve_device(device)
        self._device_test = resolve_device(device_test)
        self._model, self._optimizer, self._scheduler = model, optimizer, scheduler
        self._metrics = metrics() if callable(metrics) else (metrics if metrics is not None else asr_metrics())
        self._epochs = epochs if hasattr(epochs, "__iter__") else Epochs(epochs)
        self._epoch = getattr(self._epochs, "current", 0)
        self._model_path, self._checkpoint_path = model_path, checkpoint_path
        self.criterion = criterion
        # reference: mixed_precision -> torch autocast + GradScaler around fit/backward (deepspeech_trainer.py:72-91).  The MI355X path's
        # reduced-precision mode is bf16 MFMA operands with fp32 accumulation/state, which needs no loss scaling: the flag selects it.
        self.mixed_precision = bool(mixed_precision)
        if self.mixed_precision and hasattr(model, "precision"):
            model.precision = "bf16"
        self.output_file = output_file
        self.overwrite_lr = overwrite_lr
        self._reducer = None
        self._rejected_losses = []          # loss values of steps the device gate rejected after step() had reported them (train() takes them back)
        self._loss_corrections = []         # (reported - recomputed) loss of batches that were re-run after a starved launch
        self._unsettled = None
        self._starved_report = None
        self._last_prep = None
        self.load()

    # -- epoch loop (deepspeech_trainer.py:50-66) -------------------------------------------------
    def run(self, train_loader, test_loader):
        for self._epoch in self._epochs:
            self.train(train_loader)
            self.test(test_loader)

    def checkpoint(self):
        if self._metrics.test.current == self._metrics.test.best:
            self.save()

    def description(self):
        lr = self._optimizer.param_groups[0]["lr"] * pow(10, 5)
        current, best = self._metrics.test.current, self._metrics.test.best
        tcurrent, tbest = self._metrics.train.current, self._metrics.train.best
        cer, bcer = (current.cer or 0.0), (best.cer or 0.0)
        return (f"({self._epochs.best}) {self._model.id} | CER: {cer:.4f} / ({bcer:.4f}) | Loss:{tcurrent.loss:.4f} / "
                f"({tbest.loss:.4f}) | Lr: {lr:.4f}e-5 | Epoch: {self._epochs.current}/{self._epochs.total}")

    # -- train (deepspeech_trainer.py:68-100) -----------------------------------------------------
    def train(self, train_loader):
        self._model.train()
        self._model.to(self._device)
        self.optimizer_to(self._optimizer, self._device)        # deepspeech_trainer.py:71 (a restored optimizer state sits on the CPU)
        current, best = self._metrics.train.current, self._metrics.train.best
        fused = isinstance(self._optimizer, FusedAdamW)
        for data in train_loader:
            if fused:
                valid_loss, loss_value = self.step(data)
            else:
                pct0 = data[2].clone()                            # fit() multiplies the percentages in place (A.5): a re-run needs the originals
                valid_loss, loss, loss_value = self.fit(data)
                for _attempt in range(3):
                    if not valid_loss:
                        break
                    self._optimizer.zero_grad()
                    loss.backward()
                    if getattr(self._model, "_flat", None) is None:
                        break                                     # (not an asr_amd model: nothing to check)
                    torch.cuda.synchronize(self._device)
                    if not self._persistent_starved():            # a starved BACKWARD recurrence left invalid gradients: never apply them
                        break
                    self._restore_bn_stats(self._step_index)      # the re-run's forward repeats this batch's running-statistics update
                    valid_loss, loss, loss_value = self.fit((data[0], data[1], pct0.clone(), data[3]))
                else:
                    valid_loss = False
                if valid_loss:
                    self._optimizer.step()
            if valid_loss:
                current.loss += loss_value
            else:
                print("Loss non valid, skipped")
            current.loss -= self._take_back_rejected()          # a step the DEVICE gate rejected after its loss had been counted (run-ahead)
        if fused:
            self.synchronize()                                  # settle the last step's device verdict before the epoch's bookkeeping
            current.loss -= self._take_back_rejected()
        self.update(current, best, train_loader, update_best=False)
        if self._scheduler is not None:
            self._scheduler.step()

    # -- fit (deepspeech_trainer.py:102-117) ------------------------------------------------------
    def fit(self, data):
        """Reference semantics: returns (valid_loss, loss tensor with graph, loss_value)."""
        inputs, targets, input_percentages, target_sizes = data
        input_sizes = input_percentages.mul_(int(inputs.size(3))).int()      # in place + fp32 truncation (A.5)
        inputs = inputs.to(self._device)
        hip_model = getattr(self._model, "_ensure_flat", None) is not None and getattr(self._model, "bidirectional", True) and inputs.is_cuda
        for _attempt in range(3):
            if hip_model:
                self._model._ensure_flat(inputs.device)
                self._step_index = getattr(self, "_step_index", -1) + 1
                self._snapshot_bn_stats(self._step_index)                    # (put back if this forward has to be repeated)
            out, output_sizes = self._model.forward(inputs, input_sizes)
            out = out.transpose(0, 1)                                        # TxNxC
            float_out = out.float()
            if not isinstance(self.criterion, CTCLoss):
                float_out = float_out.log_softmax(2)    # torch criterion: keep the reference's op sequence
            # (asr_amd.CTCLoss fuses the row log-softmax into the CTC kernels)
            loss = self.criterion(float_out, targets, output_sizes, target_sizes).to(self._device)
            loss = loss / inputs.size(0)
            loss_value = loss.item()
            starved = hip_model and self._persistent_starved()
            if not starved:
                break
            # The reference never skips a VALID batch (deepspeech_trainer.py:86-97: only check_loss decides).  A starved persistent launch is
            # this library's own failure, so the same batch is computed again: the library is in its cooldown now (one-launch-per-step
            # kernels, which cannot starve) and the running statistics the invalid forward wrote are put back first.
            self._restore_bn_stats(self._step_index)
        valid_loss, _ = check_loss(loss, loss_value)
        return valid_loss and not starved, loss, loss_value

    # -- fused step ---------------------------------------------------------------------------------
    def _persistent_starved(self) -> bool:
        """True if a persistent recurrence launch starved during this step (not every workgroup could be resident: something else held CUs).
        The step's results are then invalid: it is reported and counted, the library has switched to the one-launch-per-step kernels for its
        cooldown, and the caller computes the SAME batch again on those (fit / train / _recover) — no batch is dropped, the sequence of
        optimizer updates is the one an un-starved run makes."""
        try:
            ops.rnn_persistent_check(self._device)
            return False
        except Exception as e:                                               # DS2LibraryError
            import sys
            DeepSpeechTrainer.starved_steps += 1
            print(f"[asr_amd] step re-run on the one-launch-per-step kernels: {e}", file=sys.stderr, flush=True)
            return True

    def _get_reducer(self):
        flat, grad = self._model.flat_parameters()
        if self._reducer is None or self._reducer.flat_grad.data_ptr() != grad.data_ptr():
            self._reducer = BucketedAllReducer(grad, self._model._flat.layer_buckets())
            # "overlap" schedule: buckets are all-reduced on a communication stream WHILE backward's recurrences run, so the persistent backward
            # recurrence (which needs every workgroup resident at once) must be off; "conv" (default: collectives only under the conv-stack
            # backward) and "serial" keep it.  The forward recurrence never overlaps a collective in any schedule.
            single = self._reducer.world == 1 and not self._reducer.force          # (a forced 1-rank run behaves like a multi-rank one)
            ops.rnn_persistent_enable(True, single or not self._reducer.overlaps_recurrence, device=self._device)
        return self._reducer

    def step(self, data):
        """One full train step on the HIP kernels; returns (valid_loss, loss_value).

        The host waits for ONE thing: the loss value (the reference's `loss.item()`), which exists as soon as the CTC kernels have run —
        it does not wait for backward or the optimizer.  Whether the update may be applied is decided on the DEVICE (ops.step_gate: loss
        finite and non-negative, no starved persistent recurrence; under data parallelism the MIN over ranks) and the fused AdamW launch
        reads that flag when it runs, so everything of the step is enqueued before the host blocks and the next step's host-side
        preparation overlaps this step's backward.  `valid_loss` is `check_loss` of this rank's loss value; a step that the device gate
        rejected for another reason (a starved launch, another rank's loss) is reported — and the optimizer's step count corrected — when
        the next call (or `synchronize()`) reads the flag back."""
        model = self._model
        if not getattr(model, "bidirectional", True):
            raise NotImplementedError("DeepSpeechTrainer.step is the fused MI355X train step of the bidirectional model; a unidirectional model "
                                      "(torch ops) trains through the reference's own loop: fit(data) -> loss.backward() -> optimizer.step()")
        if not isinstance(self._optimizer, FusedAdamW):
            return self._step_host_gated(data)
        inputs, targets, input_percentages, target_sizes = data
        input_sizes = input_percentages.mul_(int(inputs.size(3))).int()
        inputs = inputs.to(self._device, non_blocking=True)
        B = inputs.size(0)
        output_sizes = model.get_seq_lens(input_sizes.cpu().int())
        model._ensure_flat(inputs.device)
        # every small integer operand of the step (output lengths, flat targets, their offsets and lengths) goes to the GPU in ONE
        # asynchronous copy from pinned memory: a pageable `.to(device)` is stream-ordered AND blocks the host, i.e. it would make the host
        # wait for the previous step's backward after all
        t_h, off_h, tl_h, max_u = _prep_targets_host(targets, target_sizes)
        prep = (inputs, output_sizes.to(torch.int32), t_h, off_h, tl_h, max_u)      # everything a re-run of this batch needs (_recover)
        lens_dev, tg, off, tl = self._stage_ints(inputs.device, *prep[1:5])
        main = torch.cuda.current_stream()
        with torch.no_grad():
            W = model._flat.tensors(model)
            Gr = model._flat.tensors(model, grads=True)
            self._step_index = getattr(self, "_step_index", -1) + 1
            self._snapshot_bn_stats(self._step_index)                        # (restored if the device gate reports this step starved)
            logits, ctx = engine.forward(W, model._cfg, inputs, lens_dev, training=True, save=True)
            nll, dlogits = ops.ctc_loss(logits, tg, off, lens_dev, tl, max_u, 1.0 / B, want_grad=True)
            loss = ops.ctc_batch_mean(nll)                                   # (1,): sum of the per-utterance losses / B
            # the loss value travels to pinned host memory on a copy stream that waits for the CTC kernels only
            pin = self._pinned()
            have_loss = torch.cuda.Event()
            have_loss.record(main)
            with torch.cuda.stream(pin["stream"]):
                pin["stream"].wait_event(have_loss)
                pin["loss"].copy_(loss, non_blocking=True)
                pin["loss_done"].record(pin["stream"])
            loss.record_stream(pin["stream"])
            red = self._get_reducer()
            engine.backward(W, Gr, model._cfg, ctx, dlogits, on_bucket=red.on_bucket if red.active else None)
            red.finish()
            pin["loss_done"].synchronize()                                   # the step's single host wait: CTC done (backward is running)
            loss_value = float(pin["loss"][0])
            prev_starved = self._settle()                                    # the PREVIOUS step's (rank-reduced) verdict is long available
            if prev_starved:
                # The previous step did not update the weights (device gate), and this step's recurrences were launched before that was
                # known (the record they would have left has just been cleared with the other): neither is trusted, NEITHER IS DROPPED —
                # both batches are computed again, in order, synchronously, on the step kernels of the cooldown (every rank takes this
                # branch: the verdict is the reduced one).
                return self._recover(prep)
            valid_loss, _ = check_loss(None, loss_value)
            gate = ops.step_gate(loss)                                       # device verdict of THIS step, behind backward in stream order
            gate = red.all_valid_device(gate)                                # MIN over ranks (no-op for one rank)
            self._optimizer.grad_scale = 1.0 / red.world
            self._optimizer.step(apply_flag=gate)
            gate_ready = torch.cuda.Event()
            gate_ready.record(main)
            with torch.cuda.stream(pin["stream"]):
                pin["stream"].wait_event(gate_ready)
                pin["gate"].copy_(gate, non_blocking=True)
                pin["gate_done"].record(pin["stream"])
            gate.record_stream(pin["stream"])
            self._unsettled, self._unsettled_loss, self._unsettled_index = valid_loss, loss_value, self._step_index
            self._last_prep = prep
        return valid_loss, loss_value

    def _run_sync(self, prep):
        """One train step on a prepared batch, host-gated (one full synchronisation): forward, CTC, backward, all-reduce, starvation check,
        optimizer.  A starved launch (impossible during a cooldown, possible in the host-gated path of a torch optimizer) repeats the batch.
        Returns (valid_loss, loss_value)."""
        model = self._model
        inputs, output_sizes, t_h, off_h, tl_h, max_u = prep
        B = inputs.size(0)
        valid_loss, loss_value = False, float("nan")
        for _attempt in range(3):
            lens_dev, tg, off, tl = self._stage_ints(inputs.device, output_sizes, t_h, off_h, tl_h)
            with torch.no_grad():
                W = model._flat.tensors(model)
                Gr = model._flat.tensors(model, grads=True)
                self._step_index = getattr(self, "_step_index", -1) + 1
                self._snapshot_bn_stats(self._step_index)
                logits, ctx = engine.forward(W, model._cfg, inputs, lens_dev, training=True, save=True)
                nll, dlogits = ops.ctc_loss(logits, tg, off, lens_dev, tl, max_u, 1.0 / B, want_grad=True)
                loss = ops.ctc_batch_mean(nll)[0]
                red = self._get_reducer()
                engine.backward(W, Gr, model._cfg, ctx, dlogits, on_bucket=red.on_bucket if red.active else None)
                red.finish()
                loss_value = loss.item()
                torch.cuda.synchronize(inputs.device)
                starved = self._persistent_starved()                         # the device is idle here
                if not red.all_valid(not starved, inputs.device):            # starved on ANY rank: every rank repeats the batch
                    self._restore_bn_stats(self._step_index)
                    continue
                valid_loss, _ = check_loss(loss, loss_value)
                valid_loss = red.all_valid(valid_loss, inputs.device)
                if valid_loss and isinstance(self._optimizer, FusedAdamW):
                    self._optimizer.grad_scale = 1.0 / red.world
                    self._optimizer.step()
                elif valid_loss:
                    for n, p in model.named_parameters():                    # torch AdamW skips parameters without a gradient: frozen ones get none
                        p.grad = (Gr[n] if red.world == 1 else Gr[n] / red.world) if p.requires_grad else None
                    self._optimizer.step()
                return valid_loss, loss_value
        print("[asr_amd] a persistent recurrence starved three times in a row on the same batch: batch skipped", flush=True)
        return False, loss_value

    def _recover(self, prep_cur):
        """After _settle() reported the previous step starved: compute that batch again (the weights did not move, the BatchNorm statistics
        are back to what they were before it), then the current one.  The epoch-loss bookkeeping is corrected for what step() had
        reported for the previous batch.  prep_cur None: only the previous batch (synchronize() at the end of an epoch)."""
        torch.cuda.synchronize(self._device)
        reported_valid, reported_loss = self._starved_report
        self._starved_report = None
        valid_p, loss_p = self._run_sync(self._last_prep)
        if valid_p:
            self._loss_corrections.append((reported_loss if reported_valid else 0.0) - loss_p)   # train() added `reported_loss` (or nothing)
        elif reported_valid:
            self._rejected_losses.append(reported_loss)
        if prep_cur is None:
            return None
        out = self._run_sync(prep_cur)
        self._last_prep = prep_cur
        return out

    def _stage_ints(self, device, *cpu_int32):
        """One pinned staging buffer, one non-blocking H2D copy; returns device views (16-byte aligned starts).  The buffer is reused
        every step: the host only gets here after it has seen the previous step's loss, i.e. after that step's copy has executed."""
        sizes = [int(t.numel()) for t in cpu_int32]
        starts, total = [], 0
        for n in sizes:
            starts.append(total)
            total += (n + 3) // 4 * 4
        if getattr(self, "_ints_pin", None) is None or self._ints_pin.numel() < total:
            self._ints_pin = torch.empty(max(total, 4096), dtype=torch.int32).pin_memory()
        for t, a, n in zip(cpu_int32, starts, sizes):
            self._ints_pin[a:a + n].copy_(t.reshape(-1))
        dev = torch.empty(total, dtype=torch.int32, device=device)
        dev.copy_(self._ints_pin[:total], non_blocking=True)
        return tuple(dev[a:a + n] for a, n in zip(starts, sizes))

    def _pinned(self):
        if getattr(self, "_pin", None) is None:
            self._pin = {"stream": torch.cuda.Stream(device=self._device), "loss": torch.empty(1, dtype=torch.float32).pin_memory(),
                         "gate": torch.empty(1, dtype=torch.int32).pin_memory(), "loss_done": torch.cuda.Event(), "gate_done": torch.cuda.Event()}
        return self._pin

    def _settle(self):
        """Read back the device verdict of the last gated step (if any): a step this rank's loss check accepted but the device rejected
        (starved persistent launch; another rank's loss) did not update the weights — report it and take the optimizer's count back."""
        pending = getattr(self, "_unsettled", None)
        if pending is None:
            return False
        self._unsettled = None
        self._pin["gate_done"].synchronize()
        verdict = int(self._pin["gate"][0])                                  # MIN over ranks of {1 apply, 0 invalid loss, -1 starved launch}
        applied = verdict > 0
        starved_any = verdict < 0
        if not applied:
            self._optimizer.undo_step_count()
            own = self._persistent_starved() if starved_any else False       # this rank's own record (counted, printed, cleared, cooldown)
            if starved_any and not own:
                print("[asr_amd] step re-run on every rank: a persistent recurrence launch starved on another rank", flush=True)
            if starved_any:
                self._starved_report = (bool(pending), self._unsettled_loss)   # _recover() re-runs the batch and settles the bookkeeping
            elif pending:                                                    # not explained by this rank's own loss
                print("[asr_amd] step skipped on every rank: another rank's loss was not valid", flush=True)
                # the step's loss was reported as valid (and train() has added it to the epoch loss) before this verdict existed:
                # hand it back so that the bookkeeping matches the updates that were really applied
                self._rejected_losses.append(self._unsettled_loss)
            if starved_any:
                # The starved step's forward (and, because starvation is discovered one step late, the forward of the step that is running
                # now) wrote BatchNorm running statistics from invalid activations on the starving rank.  EVERY rank puts back the statistics
                # of BEFORE the starved step — the verdict is the reduced one, so all ranks roll back the same two forwards and their
                # running statistics / counters keep describing the same number of accepted updates, whichever rank checkpoints.
                self._restore_bn_stats(self._unsettled_index)
        return starved_any

    def _take_back_rejected(self) -> float:
        """Sum of the loss values of steps that step() reported as valid but the device gate rejected afterwards (and forget them)."""
        total = sum(self._rejected_losses) + sum(self._loss_corrections)     # (corrections: a re-run batch's loss against what was reported)
        self._loss_corrections = []
        if self._rejected_losses:
            print("Loss non valid, skipped")                                 # the reference's message, for the step it belongs to
            self._rejected_losses = []
        return total

    def _snapshot_bn_stats(self, index: int):
        """One device copy (a few KB): the BatchNorm running statistics and counters as they are BEFORE train step `index` runs."""
        flat = self._model._flat
        if getattr(self, "_bn_snap", None) is None or self._bn_snap[0][0].numel() != flat.stats.numel() or self._bn_snap[0][0].device != flat.stats.device:
            self._bn_snap = [(torch.empty_like(flat.stats), torch.empty_like(flat.counters)) for _ in range(2)]
        st, ct = self._bn_snap[index & 1]
        st.copy_(flat.stats)
        ct.copy_(flat.counters)

    def _restore_bn_stats(self, index: int):
        snap = getattr(self, "_bn_snap", None)
        if snap is None:
            return
        flat = self._model._flat
        st, ct = snap[index & 1]
        flat.stats.copy_(st)
        flat.counters.copy_(ct)

    def synchronize(self):
        """Wait for everything enqueued by step() and settle the last step's device verdict (call before reading weights / counters)."""
        if self._settle():
            self._recover(None)
        torch.cuda.synchronize(self._device)

    def _step_host_gated(self, data):
        """step() for a torch optimizer (it cannot read a device flag): the host decides, with one full synchronisation per step."""
        model = self._model
        inputs, targets, input_percentages, target_sizes = data
        input_sizes = input_percentages.mul_(int(inputs.size(3))).int()
        inputs = inputs.to(self._device, non_blocking=True)
        output_sizes = model.get_seq_lens(input_sizes.cpu().int())
        model._ensure_flat(inputs.device)
        t_h, off_h, tl_h, max_u = _prep_targets_host(targets, target_sizes)
        return self._run_sync((inputs, output_sizes.to(torch.int32), t_h, off_h, tl_h, max_u))

    # -- eval / bookkeeping (deepspeech_trainer.py:119-137) ----------------------------------------
    def test(self, test_loader):
        current, best = self._metrics.test.current, self._metrics.test.best
        # settle the last train step first: the evaluation loop checks the persistent-recurrence record at its own sync points and must not
        # consume the record of an un-settled train step (model(x) in eval mode must not be interleaved with un-settled step() calls)
        if self._unsettled is not None:
            self.synchronize()
        wer, cer, _ = self._model(loader=test_loader, device=self._device_test, output_file=self.output_file)
        current.wer, current.cer = wer, cer
        self.update(current, best, test_loader, update_best=True)
        self.checkpoint()

    def update(self, current, best, loader, update_best=False):
        n = len(loader.dataset) if hasattr(loader, "dataset") else max(len(loader), 1)
        current.loss /= n
        try:
            assert best.cer is not None
            assert best.cer < current.cer
        except (AssertionError, TypeError):
            vars(best).update(vars(current))
            if update_best:
                self._epochs.best = self._epochs.current

    @staticmethod
    def optimizer_to(optim, device):
        for param in getattr(optim, "state", {}).values():
            if isinstance(param, torch.Tensor):
                param.data = param.data.to(device)
            elif isinstance(param, dict):
                for k, sub in param.items():
                    if isinstance(sub, torch.Tensor):
                        sub.data = sub.data.to(device)

    # -- checkpoints (deepspeech_trainer.py:154-188): same dict keys ------------------------------------
    def load(self, all=True):
        model_path = self._model_path
        if model_path and os.path.exists(model_path):
            ckpt = torch.load(model_path, map_location="cpu", weights_only=False)
            self._model.load_state_dict(ckpt["state_dict"])
            if all:
                try:
                    self._optimizer.load_state_dict(ckpt["optimizer"])
                except Exception as e:  # optimizer kind changed (torch AdamW <-> FusedAdamW)
                    print(f"optimizer state not restored: {e}")
                if self.overwrite_lr is not None:
                    self._optimizer.param_groups[0]["lr"] = self.overwrite_lr
                if ckpt.get("scheduler") is not None:
                    self._scheduler = ckpt["scheduler"]
                    self._scheduler.optimizer = self._optimizer
            self._metrics = ckpt["metrics"]
            self._epochs.start = self._epochs.current = self._epochs.best = ckpt["epoch"]
            print(f"restart from {model_path}")

    def save(self):
        os.makedirs(os.path.dirname(os.path.abspath(self._model_path)), exist_ok=True)
        if dist.is_initialized() and dist.get_rank() != 0:
            return  # DDP convention: rank 0 owns the checkpoint (BN running stats are per-rank)
        torch.save({"epoch": self._epochs.best, "metrics": self._metrics, "optimizer": self._optimizer.state_dict(),
                    "scheduler": self._scheduler, "state_dict": self._model.state_dict()}, self._model_path)
        print(f"{self._model_path} saved...")
