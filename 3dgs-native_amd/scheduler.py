"""Exponential learning-rate decay (reference scheduler.py:15-27): lr(i) = lr0 * final_factor ** min(i/(T-1), 1)."""


class LRScheduler:
    def __init__(self, initial_lr, final_lr_factor=0.01):
        self.initial_lr = initial_lr
        self.final_lr = initial_lr * final_lr_factor

    def get_lr(self, iteration, total_iterations):
        if total_iterations <= 1:
            return self.initial_lr
        progress = min(iteration / (total_iterations - 1), 1.0)
        return self.initial_lr * ((self.final_lr / self.initial_lr) ** progress)
