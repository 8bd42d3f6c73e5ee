"""FusedAdamW: torch.optim.AdamW semantics (trainers/__main__.py:41-47) as ONE HIP launch over the
model's flat parameter buffer (asr_amd/params.py), reading gradients from the flat gradient buffer the
backward kernels write.  `zero_grad()` is a no-op: the backward schedule overwrites, never accumulates."""
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
        if self.state["exp_avg"] is None or self.state["exp_avg"].shape != flat.shape or self.state["exp_avg"].device != flat.device:
            self.state["exp_avg"] = torch.zeros_like(flat)
            self.state["exp_avg_sq"] = torch.zeros_like(flat)

    def zero_grad(self, set_to_none: bool = True):
        return None

    @torch.no_grad()
    def step(self):
        self._ensure_state()
        flat, grad = self.model.flat_parameters()
        g = self.param_groups[0]
        self.state["step"] += 1
        ops.adamw(flat, grad, self.state["exp_avg"], self.state["exp_avg_sq"], self.state["step"], g["lr"], g["betas"], g["eps"],
                  g["weight_decay"], self.grad_scale)

    def state_dict(self):
        return {"state": {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.state.items()}, "param_groups": self.param_groups}

    def load_state_dict(self, sd):
        self.param_groups = sd["param_groups"]
        self.state = dict(sd["state"])
