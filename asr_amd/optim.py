"""FusedAdamW: torch.optim.AdamW semantics (trainers/__main__.py:41-47) as HIP launches over the
model's flat parameter buffer (asr_amd/params.py), reading gradients from the flat gradient buffer the
backward kernels write.

Like torch.optim.AdamW it leaves parameters without a gradient alone: a parameter with
`requires_grad=False` (DeepSpeech.finetune_from freezes all but the last tensors, deepspeech.py:124-128)
is neither decayed nor updated — the launch covers the contiguous spans of trainable parameters only
(normally one span: the whole buffer)."""
from __future__ import annotations

import torch

from . import ops


class FusedAdamW:
    def __init__(self, model, lr=1.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5):
        self.model = model
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)]
        self.state = {"step": 0, "exp_avg": None, "exp_avg_sq": None}
        self.grad_scale = 1.0   # the DP reducer sets 1/world_size (all-reduce is a SUM)

    def _ensure_state(self):
        flat, _ = self.model.flat_parameters()
        for k in ("exp_avg", "exp_avg_sq"):
            m = self.state[k]
            if m is None or m.shape != flat.shape:
                self.state[k] = torch.zeros_like(flat)        # first step, or a different model
            elif m.device != flat.device or m.dtype != flat.dtype:
                self.state[k] = m.to(device=flat.device, dtype=flat.dtype)   # restored from a checkpoint (map_location="cpu"): keep the moments

    def zero_grad(self, set_to_none: bool = True):
        """The fused schedule overwrites the flat gradient buffer, so there is nothing to clear for `step()`;
        on the autograd path (`fit` -> `loss.backward()`) the parameters' `.grad` views must go, or the next
        backward would ACCUMULATE into them (asr_amd/modules/deepspeech.py:_DS2Function.backward)."""
        for p in self.model.parameters():
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _trainable_spans(self):
        """Contiguous [start, end) element ranges of the flat buffer that hold parameters with requires_grad."""
        fp = self.model._flat
        named = dict(self.model.named_parameters())
        key = tuple(bool(named[n].requires_grad) for n in fp.order)
        if getattr(self, "_span_key", None) != key:
            spans, cur = [], None
            for n, on in zip(fp.order, key):
                o, sz = fp.offsets[n]
                end = o + (sz + 3) // 4 * 4                    # params.ALIGN padding belongs to the tensor in front of it
                if on:
                    if cur is not None and cur[1] == o:
                        cur[1] = end
                    else:
                        cur = [o, end]
                        spans.append(cur)
                else:
                    cur = None
            self._span_key, self._spans = key, [(a, min(b, fp.total)) for a, b in spans]
        return self._spans

    @torch.no_grad()
    def step(self, apply_flag=None):
        """apply_flag: optional int32 GPU tensor (ops.step_gate) the kernels read when they RUN — 0 leaves parameters and moments
        untouched.  The caller that passes one learns the outcome later and must call `undo_step_count()` if the update did not happen
        (the bias corrections are computed on the host from the step count)."""
        self._ensure_state()
        flat, grad = self.model.flat_parameters()
        g = self.param_groups[0]
        self.state["step"] += 1
        for a, b in self._trainable_spans():
            ops.adamw(flat[a:b], grad[a:b], self.state["exp_avg"][a:b], self.state["exp_avg_sq"][a:b], self.state["step"], g["lr"], g["betas"],
                      g["eps"], g["weight_decay"], self.grad_scale, apply_flag=apply_flag)

    def undo_step_count(self):
        """A gated step() turned out to be a no-op on the device: take its count back."""
        self.state["step"] = max(0, self.state["step"] - 1)

    def state_dict(self):
        return {"state": {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.state.items()}, "param_groups": self.param_groups}

    def _from_torch_adamw(self, sd):
        """A torch.optim.AdamW state_dict — what the reference trainer writes (trainers/deepspeech_trainer.py:176-188 with the optimizer
        of trainers/__main__.py:41-47): {'state': {param index: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [{...}]} — converted
        to the flat layout: parameter i of the group is the i-th of model.parameters() (same order as the reference's: the state_dict
        key-order test pins it), its moments go to that tensor's span of the flat buffer."""
        groups = sd.get("param_groups")
        if not isinstance(groups, (list, tuple)) or len(groups) != 1:
            raise ValueError("torch AdamW state_dict: expected exactly one param group")
        g = groups[0]
        if g.get("amsgrad") or g.get("maximize"):
            raise ValueError("torch AdamW state_dict: amsgrad / maximize are not supported by FusedAdamW")
        names = [n for n, _ in self.model.named_parameters()]
        if "params" in g and len(g["params"]) != len(names):
            raise ValueError(f"torch AdamW state_dict covers {len(g['params'])} parameters, the model has {len(names)}")
        fp = self.model._ensure_flat(next(self.model.parameters()).device)
        m = torch.zeros(fp.total, dtype=torch.float32)
        v = torch.zeros(fp.total, dtype=torch.float32)
        steps = set()
        for i, n in enumerate(names):
            st = sd["state"].get(i)
            if st is None:                                   # a parameter that never had a gradient (frozen): no state, like here
                continue
            o, sz = fp.offsets[n]
            if st["exp_avg"].numel() != sz:
                raise ValueError(f"torch AdamW state_dict: parameter {i} ({n}) has {st['exp_avg'].numel()} elements, expected {sz}")
            m[o:o + sz] = st["exp_avg"].detach().reshape(-1).float().cpu()
            v[o:o + sz] = st["exp_avg_sq"].detach().reshape(-1).float().cpu()
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError(f"torch AdamW state_dict: per-parameter step counts differ ({sorted(steps)}); one fused step count cannot represent that")
        self.param_groups = [dict(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g["weight_decay"])]
        self.state = {"step": steps.pop() if steps else 0, "exp_avg": m, "exp_avg_sq": v}   # (moved to the device by _ensure_state)

    def load_state_dict(self, sd):
        """Accepts what `state_dict()` wrote, and a torch.optim.AdamW state_dict as the reference trainer checkpoints it (converted to
        the flat layout, `_from_torch_adamw`).  Anything else raises ValueError, which the trainer reports as "optimizer state not
        restored" instead of failing at the first step."""
        st = sd.get("state") if isinstance(sd, dict) else None
        if isinstance(st, dict) and (not st or all(isinstance(k, int) for k in st)) and "param_groups" in sd:
            return self._from_torch_adamw(sd)
        if not isinstance(st, dict) or set(st.keys()) != {"step", "exp_avg", "exp_avg_sq"}:
            raise ValueError("not a FusedAdamW state_dict (expected state keys step / exp_avg / exp_avg_sq)")
        m, v = st["exp_avg"], st["exp_avg_sq"]
        if (m is None) != (v is None) or (m is not None and (not torch.is_tensor(m) or not torch.is_tensor(v) or m.shape != v.shape or m.dim() != 1)):
            raise ValueError("FusedAdamW state_dict: exp_avg / exp_avg_sq must both be None or flat tensors of one shape")
        if not isinstance(sd.get("param_groups"), (list, tuple)) or len(sd["param_groups"]) != 1:
            raise ValueError("FusedAdamW state_dict: expected exactly one param group")
        missing = {"lr", "betas", "eps", "weight_decay"} - set(sd["param_groups"][0])
        if missing:
            raise ValueError(f"FusedAdamW state_dict: param group lacks {sorted(missing)}")
        self.param_groups = [dict(sd["param_groups"][0])]
        self.state = {"step": int(st["step"]), "exp_avg": m, "exp_avg_sq": v}
