"""Cell-level edge primitives and the architecture-weighted mixed edge.

Mirror of the reference's models/search/darts/operations.py (OPS :7-12, Zero :14-20,
FC_Relu :22-38, FC_Mish :48-65, Identity :88-93, FusionMixedOp :95-105) on the gfx950
kernels: with the default PRIMITIVES ['none', 'skip'] the mixed edge is the HIP mixsum
kernel (bmnas_mixsum_fwd/bwd).  FC_Relu / FC_Mish are not in the default search space
(SURVEY.md a14); they stay ordinary PyTorch modules so an edited PRIMITIVES list works.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from bmnas.functions import MixSumFn

from .genotypes import *  # noqa: F401,F403
from .genotypes import PRIMITIVES

OPS = {
    'none': lambda C, L, args: Zero(),
    'fc_relu': lambda C, L, args: FC_Relu(C, L, args),
    'fc_mish': lambda C, L, args: FC_Mish(C, L, args),
    'skip': lambda C, L, args: Identity(),
}


class Zero(nn.Module):
    def forward(self, x):
        return x.mul(0.)


class Identity(nn.Module):
    def forward(self, x):
        return x


class Mish(nn.Module):
    def forward(self, x):
        return x * torch.tanh(F.softplus(x))


class _FCBase(nn.Module):
    """Linear over the channel dim -> activation -> BatchNorm1d -> Dropout."""

    def __init__(self, C, L, args):
        super().__init__()
        self.linear = nn.Linear(C, C)
        self.bn = nn.BatchNorm1d(C)
        self.dropout = nn.Dropout(args.drpt)

    def _act(self, x):
        raise NotImplementedError

    def forward(self, x):
        out = self.linear(x.transpose(1, 2)).transpose(1, 2)
        return self.dropout(self.bn(self._act(out)))


class FC_Relu(_FCBase):
    def _act(self, x):
        return F.relu(x)


class FC_Mish(_FCBase):
    def __init__(self, C, L, args):
        super().__init__(C, L, args)
        self.mish = Mish()

    def _act(self, x):
        return self.mish(x)


class FusionMixedOp(nn.Module):
    """sum_p weights[p] * op_p(x) over PRIMITIVES (reference operations.py:95-105)."""

    def __init__(self, C, L, args):
        super().__init__()
        self._ops = nn.ModuleList(OPS[p](C, L, args) for p in PRIMITIVES)
        self._default = list(PRIMITIVES) == ['none', 'skip']

    def forward(self, x, weights):
        if self._default:
            # w_none * (x * 0) + w_skip * x: the 'none' term vanishes for finite x
            w = weights if weights.device == x.device else weights.to(x.device)
            return MixSumFn.apply(w[1:2], x)
        return sum(w * op(x) for w, op in zip(weights, self._ops))


def mixed_edge_sum(states, weights, offset):
    """sum_j FusionMixedOp_j(states[j], weights[offset + j]) as ONE HIP kernel
    (reference model_search.py:58 / node_search.py:54)."""
    n = len(states)
    w = weights if weights.device == states[0].device else weights.to(states[0].device)
    return MixSumFn.apply(w[offset:offset + n, 1], *states)
