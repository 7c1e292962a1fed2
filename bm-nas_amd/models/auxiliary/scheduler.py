"""Per-batch learning-rate schedules (reference models/auxiliary/scheduler.py:12-62).

Two schedules with the reference's interface — `step()` returns the rate for the coming batch,
`update_optimizer(opt)` applies it.  The rate is written straight into the param groups: the
reference round-trips the whole optimizer state through state_dict()/load_state_dict() every
batch, which has the same effect and deep-copies the Adam moments.  The cosine rule's arithmetic
is evaluated in the reference's order so that the rates are the same doubles
(tests/test_aux_layers.py checks them against values recorded from the reference).
"""
import numpy as np


class _Schedule:
    """What both schedules share: the current rate lives in `self.eta`."""
    eta = 0.0

    def update_optimizer(self, optimizer):
        rate = self.eta
        for group in optimizer.param_groups:
            group['lr'] = rate


class LRCosineAnnealingScheduler(_Schedule):
    """SGDR: cosine decay from eta_max to eta_min over Ti epochs, then a restart with the period
    multiplied by Tmultiplier.  Positions are counted in batches (`num_batches_per_epoch` may be
    fractional, as the trainers pass len(dataset) / batchsize)."""
    _RESTART_EPS = 1e-10

    def __init__(self, eta_max, eta_min, Ti, Tmultiplier, num_batches_per_epoch):
        self.eta_max, self.eta_min = eta_max, eta_min
        self.Ti, self.Tm = Ti, Tmultiplier
        self.nbpe = num_batches_per_epoch
        self.iteration_counter = 0.0       # batches since the last restart
        self.Tcur = 0.0                    # the same, in epochs
        self.eta = eta_max

    def _compute_rule(self):
        span = self.eta_max - self.eta_min
        phase = np.cos(np.pi * self.Tcur / self.Ti)
        self.eta = self.eta_min + 0.5 * span * (1 + phase)
        return self.eta

    def _restart(self):
        self.Ti *= self.Tm
        self.iteration_counter = 0
        self.Tcur = 0

    def step(self):
        self.Tcur = self.iteration_counter / self.nbpe
        self.iteration_counter += 1.0
        rate = self._compute_rule()
        if rate <= self.eta_min + self._RESTART_EPS:
            self._restart()
        return rate


class FixedScheduler(_Schedule):
    def __init__(self, lr):
        self.lr = self.eta = lr

    def step(self):
        return self.lr
