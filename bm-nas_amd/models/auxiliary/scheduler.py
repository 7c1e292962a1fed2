"""Per-batch learning-rate schedules (reference models/auxiliary/scheduler.py:12-62).

Same arithmetic as the reference's warm-restart cosine rule; `update_optimizer` writes the rate
into the param groups directly instead of round-tripping the whole optimizer state through
state_dict()/load_state_dict() every batch (same effect, no deep copy of the Adam moments).
"""
import numpy as np


class LRCosineAnnealingScheduler():
    def __init__(self, eta_max, eta_min, Ti, Tmultiplier, num_batches_per_epoch):
        self.eta_min, self.eta_max = eta_min, eta_max
        self.Ti, self.Tm = Ti, Tmultiplier
        self.Tcur = 0.0
        self.nbpe = num_batches_per_epoch
        self.iteration_counter = 0.0
        self.eta = eta_max

    def _compute_rule(self):
        self.eta = self.eta_min + 0.5 * (self.eta_max - self.eta_min) * (1 + np.cos(np.pi * self.Tcur / self.Ti))
        return self.eta

    def step(self):
        self.Tcur = self.iteration_counter / self.nbpe
        self.iteration_counter += 1.0
        eta = self._compute_rule()
        if eta <= self.eta_min + 1e-10:          # warm restart with a longer period
            self.Tcur = 0
            self.Ti = self.Ti * self.Tm
            self.iteration_counter = 0
        return eta

    def update_optimizer(self, optimizer):
        for group in optimizer.param_groups:
            group['lr'] = self.eta


class FixedScheduler():
    def __init__(self, lr):
        self.lr = lr

    def step(self):
        return self.lr

    def update_optimizer(self, optimizer):
        for group in optimizer.param_groups:
            group['lr'] = self.lr
