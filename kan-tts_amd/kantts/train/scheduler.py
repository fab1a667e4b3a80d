"""LR schedulers (reference kantts/train/scheduler.py:1-46): NoamLR, FindLR + every torch scheduler
re-exported so that yaml ``scheduler.type`` resolves by name exactly like the reference."""
from torch.optim.lr_scheduler import *  # NOQA
from torch.optim.lr_scheduler import _LRScheduler  # NOQA


class FindLR(_LRScheduler):
    """Exponential LR sweep from base_lr to max_lr over max_steps (reference :7-22)."""

    def __init__(self, optimizer, max_steps, max_lr=10):
        self.max_steps = max_steps
        self.max_lr = max_lr
        super().__init__(optimizer)

    def get_lr(self):
        frac = self.last_epoch / (self.max_steps - 1)
        return [base_lr * ((self.max_lr / base_lr) ** frac) for base_lr in self.base_lrs]


class NoamLR(_LRScheduler):
    """lr = base * sqrt(warmup) * min(step^-0.5, step * warmup^-1.5), step >= 1 (reference :25-46)."""

    def __init__(self, optimizer, warmup_steps):
        self.warmup_steps = warmup_steps
        super().__init__(optimizer)

    def get_lr(self):
        step = max(1, self.last_epoch)
        scale = self.warmup_steps ** 0.5 * min(step ** (-0.5), step * self.warmup_steps ** (-1.5))
        return [base_lr * scale for base_lr in self.base_lrs]
