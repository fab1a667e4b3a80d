"""LR schedulers (reference kantts/train/scheduler.py:1-46): NoamLR, FindLR + every torch scheduler
re-exported so that yaml ``scheduler.type`` resolves by name exactly like the reference.  Both custom schedules are a
scalar factor on the base learning rates; attribute names (``warmup_steps``, ``max_steps``, ``max_lr``) are the ones the
reference's checkpoints carry in ``scheduler.state_dict()``."""
from torch.optim.lr_scheduler import *  # NOQA
from torch.optim.lr_scheduler import _LRScheduler  # NOQA


class _FactorLR(_LRScheduler):
    """lr_i = base_lr_i * factor(step, base_lr_i)."""

    def factor(self, step, base_lr):
        raise NotImplementedError

    def get_lr(self):
        return [base_lr * self.factor(self.last_epoch, base_lr) for base_lr in self.base_lrs]


class FindLR(_FactorLR):
    """Exponential sweep from base_lr to max_lr over max_steps (reference :7-22): base * (max/base)^(step/(N-1))."""

    def __init__(self, optimizer, max_steps, max_lr=10):
        self.max_steps = max_steps
        self.max_lr = max_lr
        super().__init__(optimizer)

    def factor(self, step, base_lr):
        return (self.max_lr / base_lr) ** (step / (self.max_steps - 1))


class NoamLR(_FactorLR):
    """Linear warm-up then inverse-square-root decay, continuous at ``warmup_steps`` (reference :25-46):
    sqrt(warmup) * min(step^-0.5, step * warmup^-1.5) with step >= 1."""

    def __init__(self, optimizer, warmup_steps):
        self.warmup_steps = warmup_steps
        super().__init__(optimizer)

    def factor(self, step, base_lr):
        step = max(1, step)
        return self.warmup_steps ** 0.5 * min(step ** (-0.5), step * self.warmup_steps ** (-1.5))
