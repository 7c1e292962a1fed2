"""Discrete (found) fusion network built from a Genotype.

Mirror of the reference's models/search/darts/model.py (Found_FusionCell :16-89,
Found_Random_FusionCell :92-160, Found_FusionNetwork :162-190) on the gfx950 kernels.
"""
import torch.nn as nn

from bmnas.functions import CatLnFn, FoundHeadFn

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
        # the tail concatenates the LAST len(concat) states (model.py:157): where those are all step-node outputs the
        # nodes also hand out their per-sample sums, for the fused head (forward with a classifier)
        if self._multiplier <= len(self._step_nodes):
            for node in list(self._step_nodes)[-self._multiplier:]:
                node.node_cell.want_sums = True

    def head_fusable(self, classifier, states):
        """K7 + `classifier` (+ criterion) as the two launches of csrc/head.hip: a bmnas.nn.Linear with bias, <= 128
        classes, every concatenated state a step-node output that carries its per-sample sums."""
        from bmnas import cell as K
        M = self._multiplier
        return (K.FUSE_HEAD and isinstance(classifier, nn.Linear) and type(classifier).__module__.startswith('bmnas')
                and classifier.bias is not None and classifier.out_features <= 128 and M <= min(self._steps, 4)
                and (self.C * self.L) % 16 == 0 and classifier.in_features == M * self.C * self.L
                and states[0].is_cuda and states[0].dtype == classifier.weight.dtype
                and all(getattr(s, '_bmnas_sums', None) is not None for s in states[-M:]))

    def forward(self, input_features, classifier=None):
        """classifier (Found_FusionNetwork.forward_classified): the bmnas.nn.Linear the cell's output feeds — the call
        then returns ITS output."""
        states = list(input_features)
        for i in range(self._steps):
            h1 = self._ops[2 * i](states[self._indices[2 * i]])
            h2 = self._ops[2 * i + 1](states[self._indices[2 * i + 1]])
            states.append(self._step_nodes[i](h1, h2))
        M = self._multiplier
        if classifier is not None and self.head_fusable(classifier, states):
            from bmnas import cell as K
            tail = states[-M:]
            out = FoundHeadFn.apply(self.ln.weight, self.ln.bias, classifier.weight, classifier.bias, M, *tail,
                                    *[s._bmnas_sums for s in tail])
            out._bmnas_head = K.LAST_HEAD.pop()       # lets a fused criterion find its head (bmnas.nn)
            return out
        out = CatLnFn.apply(True, self.ln.weight, self.ln.bias, None, *states[-M:])
        out = out.view(out.size(0), -1)
        return out if classifier is None else classifier(out)


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

    def forward_classified(self, input_features, classifier):
        """classifier(self(input_features)) — what Found_*_Net.forward does next (mmimdb_darts_searchable.py:185-188) —
        with the cell's LayerNorm tail and the classifier as one launch where the shapes allow
        (Found_FusionCell.head_fusable), else exactly that composition."""
        assert self._num_input_nodes == len(input_features)
        return self.cell(input_features, classifier)

    def _loss(self, input_features, labels):
        return self._criterion(self(input_features), labels)

    def get_genotype(self):
        return self._genotype
