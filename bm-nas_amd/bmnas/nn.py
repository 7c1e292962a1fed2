"""Drop-in nn.Module replacements for the two callers right after the fusion cell: the
central classifier (nn.Linear) and the criterion, both on the gfx950 kernels of
csrc/linear.hip.  Same constructor signatures, parameter names and state_dict keys as the torch
classes they subclass.  Option combinations the kernels do not cover (the reference never uses
them) run the torch parent class after a one-time RuntimeWarning (bmnas.lib.note_off_path):
nothing leaves the HIP path silently."""
import torch
import torch.nn as nn

import contextlib

from . import lib
from .functions import BCEWithLogitsFn, CrossEntropyFn, DeferredLossFn, LinearFn

_FUSED_CRITERION = [False]


@contextlib.contextmanager
def fused_criterion(on=True):
    """Inside this context a criterion applied to the logits of a fused head (FusionNetwork with its
    central classifier, csrc/head.hip) is evaluated by the head's backward launch instead of a
    launch of its own: the returned loss tensor is filled when backward has run.  For code that
    reads the loss only after backward — a captured training step (bmnas.graph.GraphedTrainStep
    turns it on), the benchmark step."""
    prev, _FUSED_CRITERION[0] = _FUSED_CRITERION[0], bool(on)
    try:
        yield
    finally:
        _FUSED_CRITERION[0] = prev


def _deferrable(input):
    head = getattr(input, '_bmnas_head', None)
    if _FUSED_CRITERION[0] and head is not None and torch.is_grad_enabled() and input.requires_grad:
        return head
    return None


class Linear(nn.Linear):
    """nn.Linear whose forward/backward run on the MFMA kernels when out_features <= 128,
    in_features % 16 == 0 and a bias is present (the central_classifier shapes)."""

    def forward(self, x):
        if (x.is_cuda and x.dim() == 2 and self.bias is not None and self.out_features <= 128
                and self.in_features % 16 == 0 and x.dtype == torch.float32):
            return LinearFn.apply(x, self.weight, self.bias)
        lib.note_off_path('bmnas.nn.Linear', f'input {tuple(x.shape)} {x.dtype} on {x.device}, out_features '
                          f'{self.out_features}, in_features {self.in_features}, bias {self.bias is not None}')
        return super().forward(x)


class BCEWithLogitsLoss(nn.BCEWithLogitsLoss):
    def forward(self, input, target):
        if (input.is_cuda and self.weight is None and self.pos_weight is None and self.reduction == 'mean'
                and input.dtype == torch.float32 and target.dtype == torch.float32
                and input.shape == target.shape):
            head = _deferrable(input)
            if head is not None:
                return DeferredLossFn.apply(input, target, head, 'bce')
            return BCEWithLogitsFn.apply(input, target)
        lib.note_off_path('bmnas.nn.BCEWithLogitsLoss', f'input {tuple(input.shape)} {input.dtype} on {input.device}, '
                          f'target {tuple(target.shape)} {target.dtype}, reduction {self.reduction}')
        return super().forward(input, target)


class CrossEntropyLoss(nn.CrossEntropyLoss):
    def forward(self, input, target):
        if (input.is_cuda and input.dim() == 2 and self.weight is None and self.reduction == 'mean'
                and self.label_smoothing == 0.0 and self.ignore_index == -100
                and target.dtype == torch.int64 and target.dim() == 1 and input.dtype == torch.float32):
            head = _deferrable(input)
            if head is not None:
                return DeferredLossFn.apply(input, target, head, 'ce')
            return CrossEntropyFn.apply(input, target)
        lib.note_off_path('bmnas.nn.CrossEntropyLoss', f'input {tuple(input.shape)} {input.dtype} on {input.device}, '
                          f'target {tuple(target.shape)} {target.dtype}, reduction {self.reduction}')
        return super().forward(input, target)
