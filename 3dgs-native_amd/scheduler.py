"""Learning-rate schedule of the trainer (reference scheduler.py:3-28): geometric interpolation
from lr0 at iteration 0 to lr0 * final_lr_factor at the last iteration,

    lr(i) = lr0 * (final_lr_factor-scaled ratio) ** clamp(i / (T - 1), 0..1)

evaluated in Python floats exactly as the reference evaluates it (ratio = final_lr / initial_lr, then `**`), so the value
that reaches the Adam kernel as float32 is the same."""


def decayed_lr(initial_lr, final_lr, iteration, total_iterations):
    """One evaluation of the schedule; a run of one iteration (or none) stays at the initial rate."""
    if total_iterations <= 1:
        return initial_lr
    fraction = min(iteration / (total_iterations - 1), 1.0)
    return initial_lr * ((final_lr / initial_lr) ** fraction)


class LRScheduler:
    """Same constructor and `get_lr(iteration, total_iterations)` as the reference's class."""

    __slots__ = ("initial_lr", "final_lr")

    def __init__(self, initial_lr, final_lr_factor=0.01):
        self.initial_lr, self.final_lr = initial_lr, initial_lr * final_lr_factor

    def get_lr(self, iteration, total_iterations):
        return decayed_lr(self.initial_lr, self.final_lr, iteration, total_iterations)
