"""`lidargen.utils.training` of the reference (lidargen/utils/training.py:7-24): the learning-rate
schedule `tools/train/train_lidm[_cond].py:128-134` builds -- linear warm-up from 0 to 1 over
`num_warmup_steps`, then cos^2-shaped decay: 0.5 (1 + cos(2 pi num_cycles p)), p = progress through
the remaining steps, clamped at 0 (num_cycles = 0.5: one half period, 1 -> 0)."""
from __future__ import annotations

import math

import torch
from torch.optim.lr_scheduler import LambdaLR


class _WarmupCosine:
    """Picklable multiplier (LambdaLR.state_dict() stores callable objects, not lambdas)."""

    def __init__(self, warmup: int, total: int, cycles: float):
        self.warmup, self.total, self.cycles = int(warmup), int(total), float(cycles)

    def __call__(self, step: int) -> float:
        if step < self.warmup:
            return step / max(1, self.warmup)
        p = (step - self.warmup) / max(1, self.total - self.warmup)
        return max(0.0, 0.5 + 0.5 * math.cos(2.0 * math.pi * self.cycles * p))


def get_cosine_schedule_with_warmup(optimizer: torch.optim.Optimizer, num_warmup_steps: int,
                                    num_training_steps: int, num_cycles: float = 0.5,
                                    last_epoch: int = -1) -> LambdaLR:
    return LambdaLR(optimizer, _WarmupCosine(num_warmup_steps, num_training_steps, num_cycles),
                    last_epoch)
