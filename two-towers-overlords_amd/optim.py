"""FusedAdam — torch.optim.Adam with the reference's defaults (backend/training.py:436) as ONE multi-tensor K4 launch."""
from __future__ import annotations

import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr) with the defaults the reference uses
    (backend/training.py:436: betas (0.9, 0.999), eps 1e-8, no weight decay): ONE K4
    launch per step for all parameters that share a step count (tt_adam_multi_f32 —
    the eight trainable tensors of the two towers).  Parameters without
    a gradient are skipped, as torch does for the frozen encoder."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            by_step = {}  # parameters whose step counts differ (one had no gradient for a while) go in separate launches
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                if not p.data.is_contiguous():
                    raise ValueError("FusedAdam: parameters must be contiguous")
                by_step.setdefault(st["step"], []).append((p.data, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"]))
            for step, quads in by_step.items():
                ops.adam_step_multi(quads, step, group["lr"], b1, b2, group["eps"])
        return loss
