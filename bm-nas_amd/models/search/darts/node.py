"""Discrete (found) step node built from a StepGenotype.

Mirror of the reference's models/search/darts/node.py (Found_NodeCell :8-76,
Found_FusionNode :78-91).  The reference's ablation nodes (:94-184) look up primitive
names that are not in STEP_STEP_OPS and cannot be constructed; they are not mirrored.
"""
import os

import torch
import torch.nn as nn

from bmnas.functions import CatLnFn, CatLnSumsFn, ConvBnActFn, ConvBnReluLnFn

from .node_operations import STEP_STEP_OPS
from .operations import OPS

# BMNAS_FOUND_FUSE_TAIL=0: the node tail as ConvBnActFn + CatLnFn (A/B runs)
FOUND_FUSE_TAIL = os.environ.get('BMNAS_FOUND_FUSE_TAIL', '1') != '0'
# BMNAS_FOUND_THRU=0: every reader of an inner state as an autograd node of its own (gradients summed by the engine)
FOUND_THRU = os.environ.get('BMNAS_FOUND_THRU', '1') != '0'


class Found_NodeCell(nn.Module):
    def __init__(self, node_steps, node_multiplier, args, step_genotype):
        super().__init__()
        self.args = args
        self.node_steps = node_steps
        self.node_multiplier = node_multiplier
        self.C, self.L = args.C, args.L
        self.num_input_nodes = 2

        self.edge_ops = nn.ModuleList()
        self.node_ops = nn.ModuleList()
        op_names, indices = zip(*step_genotype.inner_edges)
        self.compile(op_names, indices, step_genotype.inner_steps)

        if node_multiplier != 1:
            self.out_conv = nn.Conv1d(self.C * node_multiplier, self.C, 1, 1)
            self.bn = nn.BatchNorm1d(self.C)
            self.out_dropout = nn.Dropout(args.drpt)
        self.ln = nn.LayerNorm([self.C, self.L])
        self.dropout = nn.Dropout(args.drpt)
        # set by Found_FusionCell on the nodes whose output its tail concatenates: the node then also hands out each
        # sample's (sum, sum of squares) of its output, from which the fused head takes the K7 LayerNorm statistics
        self.want_sums = False

    def compile(self, edge_op_names, edge_indices, inner_steps):
        for name in edge_op_names:
            self.edge_ops.append(OPS[name](self.C, self.L, self.args))
        self.edge_indices = edge_indices
        for name in inner_steps:
            self.node_ops.append(STEP_STEP_OPS[name](self.C, self.L, self.args))

    def forward(self, x, y):
        states = [x, y]
        thru = FOUND_THRU and x.is_cuda and torch.is_grad_enabled()
        for i in range(self.node_steps):
            ix, iy = self.edge_indices[2 * i], self.edge_indices[2 * i + 1]
            in_x = self.edge_ops[2 * i](states[ix])
            in_y = self.edge_ops[2 * i + 1](states[iy])
            op = self.node_ops[i]
            if thru and hasattr(op, 'forward_thru') and in_x is not in_y:
                # a state that is read again later (another inner step, the out_conv tail, the residual) travels THROUGH
                # this op: the later readers take the alias, and their gradients are accumulated by this op's
                # data-gradient launch instead of by autograd `add` launches (bmnas.functions.ConvBnActThruFn)
                s, ax, ay = op.forward_thru(in_x, in_y)
                if in_x is states[ix]:
                    states[ix] = ax
                if in_y is states[iy]:
                    states[iy] = ay
                states.append(s)
            else:
                states.append(op(in_x, in_y))
        x = states[0]                       # (the residual reads the end of x's chain)
        tail = states[-self.node_multiplier:]
        if (self.node_multiplier != 1 and x.is_cuda and FOUND_FUSE_TAIL
                and ConvBnReluLnFn.usable(x.shape[0], self.C)):
            # out_conv's BatchNorm / ReLU / dropout tail, the residual and the node's LayerNorm as ONE launch each way
            bn = self.bn
            o, sums = ConvBnReluLnFn.apply(self.out_dropout.p, self.training, self.want_sums, bn.running_mean,
                                           bn.running_var, bn.num_batches_tracked, self.out_conv.weight,
                                           self.out_conv.bias, bn.weight, bn.bias, self.ln.weight, self.ln.bias, x,
                                           *tail)
            if self.want_sums:
                o._bmnas_sums = sums
            return o
        if self.node_multiplier != 1:
            bn = self.bn
            out = ConvBnActFn.apply('relu', self.out_dropout.p, self.training, bn.running_mean,
                                    bn.running_var, bn.num_batches_tracked, self.out_conv.weight,
                                    self.out_conv.bias, bn.weight, bn.bias, *tail)
        else:
            out = tail[0]
        if self.want_sums and out.is_cuda:
            o, sums = CatLnSumsFn.apply(self.ln.weight, self.ln.bias, x, out)
            o._bmnas_sums = sums
            return o
        return CatLnFn.apply(False, self.ln.weight, self.ln.bias, x, out)


class Found_FusionNode(nn.Module):
    def __init__(self, node_steps, node_multiplier, args, step_genotype):
        super().__init__()
        self.node_steps = node_steps
        self.node_multiplier = node_multiplier
        self.node_cell = Found_NodeCell(node_steps, node_multiplier, args, step_genotype)
        self.num_input_nodes = 2
        self.num_keep_edges = 2

    def forward(self, x, y):
        return self.node_cell(x, y)
