"""Discrete (found) fusion network built from a Genotype.

Mirror of the reference's models/search/darts/model.py (Found_FusionCell :16-89,
Found_Random_FusionCell :92-160, Found_FusionNetwork :162-190) on the gfx950 kernels.
"""
import torch.nn as nn

from bmnas.functions import CatLnFn

from .genotypes import *  # noqa: F401,F403
from .node import Found_FusionNode
from .operations import OPS


class Found_FusionCell(nn.Module):
    def __init__(self, steps, args, genotype):
        super().__init__()
        self.C, self.L = args.C, args.L
        self.args = args
        op_names, indices = zip(*genotype.edges)
        self._compile(self.C, self.L, op_names, indices, genotype.concat, genotype.steps, args)
        self._steps = steps
        self.ln = nn.LayerNorm([self.C * self._multiplier, self.L])

    def _compile(self, C, L, op_names, indices, concat, gene_step_nodes, args):
        assert len(op_names) == len(indices)
        self._steps = len(op_names) // 2
        self._concat = concat
        self._multiplier = len(concat)
        self._ops = nn.ModuleList(OPS[name](C, L, args) for name in op_names)
        self._indices = indices
        self._step_nodes = nn.ModuleList(
            Found_FusionNode(args.node_steps, args.node_multiplier, args, g) for g in gene_step_nodes)

    def forward(self, input_features):
        states = list(input_features)
        for i in range(self._steps):
            h1 = self._ops[2 * i](states[self._indices[2 * i]])
            h2 = self._ops[2 * i + 1](states[self._indices[2 * i + 1]])
            states.append(self._step_nodes[i](h1, h2))
        out = CatLnFn.apply(True, self.ln.weight, self.ln.bias, None, *states[-self._multiplier:])
        return out.view(out.size(0), -1)


class Found_Random_FusionCell(Found_FusionCell):
    """Same computation as Found_FusionCell (the reference keeps two identical classes)."""


class Found_FusionNetwork(nn.Module):
    def __init__(self, steps, multiplier, num_input_nodes, num_keep_edges, args, criterion, genotype):
        super().__init__()
        self._steps = steps
        self._multiplier = multiplier
        self._criterion = criterion
        self._genotype = genotype
        self._num_input_nodes = num_input_nodes
        self._num_keep_edges = num_keep_edges
        self.cell = Found_Random_FusionCell(steps, args, genotype)

    def forward(self, input_features):
        assert self._num_input_nodes == len(input_features)
        return self.cell(input_features)

    def _loss(self, input_features, labels):
        return self._criterion(self(input_features), labels)

    def get_genotype(self):
        return self._genotype
