"""Searchable step node: inner DAG of mixed edges and NodeMixedOps.

Mirror of the reference's models/search/darts/node_search.py (NodeCell :12-70,
FusionNode :72-163): same names, signatures, arch-parameter layout and genotype rules.
The forward runs on the gfx950 kernels; ``node_genotype`` stays host-side Python.
"""
import torch
import torch.nn as nn

from bmnas.cell import Pack
from bmnas.functions import CatLnFn, ConvBnActFn, arch_softmax

from .genotypes import PRIMITIVES, STEP_EDGE_PRIMITIVES, STEP_STEP_PRIMITIVES, StepGenotype
from .node_operations import NodeMixedOp
from .operations import FusionMixedOp, mixed_edge_sum


class NodeCell(nn.Module):
    def __init__(self, node_steps, node_multiplier, args):
        super().__init__()
        self.args = args
        self.node_steps = node_steps
        self.node_multiplier = node_multiplier
        self.C, self.L = args.C, args.L
        self.num_input_nodes = 2

        self.edge_ops = nn.ModuleList()
        self.node_ops = nn.ModuleList()
        for i in range(node_steps):
            for _ in range(self.num_input_nodes + i):
                self.edge_ops.append(FusionMixedOp(self.C, self.L, args))
        for _ in range(node_steps):
            self.node_ops.append(NodeMixedOp(self.C, self.L, args))

        if node_multiplier != 1:
            self.out_conv = nn.Conv1d(self.C * node_multiplier, self.C, 1, 1)
            self.bn = nn.BatchNorm1d(self.C)
            self.out_dropout = nn.Dropout(args.drpt)

        self.ln = nn.LayerNorm([self.C, self.L])
        self.dropout = nn.Dropout(args.drpt)      # constructed but never applied (as in the reference)

    # -- packs for the fused FusionCell path -----------------------------------------------
    def pack(self):
        p = Pack(mixed=[op.pack() for op in self.node_ops], ln_w=self.ln.weight.detach(),
                 ln_b=self.ln.bias.detach())
        if self.node_multiplier != 1:
            p.out_conv_w = self.out_conv.weight.detach()
            p.out_conv_b = self.out_conv.bias.detach()
            p.bn_w, p.bn_b = self.bn.weight.detach(), self.bn.bias.detach()
            p.bn_rm, p.bn_rv, p.bn_nbt = self.bn.running_mean, self.bn.running_var, self.bn.num_batches_tracked
            p.out_p = self.out_dropout.p
        return p

    def param_list(self):
        ps = []
        for op in self.node_ops:
            ps += op.param_list()
        if self.node_multiplier != 1:
            ps += [self.out_conv.weight, self.out_conv.bias, self.bn.weight, self.bn.bias]
        return ps + [self.ln.weight, self.ln.bias]

    def plan_grads(self, arena):
        C, L, nm = self.C, self.L, self.node_multiplier
        h = Pack(mixed=[op.plan_grads(arena) for op in self.node_ops])
        if nm != 1:
            h.oc = (arena.ask(C, nm * C, 1), arena.ask(C), arena.ask(2 * C))
        h.ln = (arena.ask(C, L), arena.ask(C, L))
        return h

    def bind_grads(self, arena, h):
        g = Pack(mixed=[op.bind_grads(arena, hm) for op, hm in zip(self.node_ops, h.mixed)],
                 dln_w=arena.view(h.ln[0]), dln_b=arena.view(h.ln[1]))
        if self.node_multiplier != 1:
            g.out_conv_dW, g.out_conv_db, g.bn_grad = (arena.view(i) for i in h.oc)
        return g

    def grads_in_param_order(self, G):
        C = self.C
        gs = []
        for op, gm in zip(self.node_ops, G.mixed):
            gs += op.grads_in_param_order(gm)
        if self.node_multiplier != 1:
            gs += [G.out_conv_dW, G.out_conv_db, G.bn_grad[:C], G.bn_grad[C:]]
        return gs + [G.dln_w, G.dln_b]

    # -- standalone forward (same call signature as the reference) --------------------------
    def forward(self, x, y, edge_weights, node_weights):
        states = [x, y]
        offset = 0
        default_edges = all(op._default for op in self.edge_ops)
        for i in range(self.node_steps):
            if default_edges:
                z = mixed_edge_sum(states, edge_weights, offset)
            else:
                # an edited PRIMITIVES list reaches the inner edges too (they are FusionMixedOps,
                # reference node_search.py:31): composed op by op; zip() inside FusionMixedOp stops
                # at the len(STEP_EDGE_PRIMITIVES) weights of the row, like the reference's
                z = sum(self.edge_ops[offset + j](h, edge_weights[offset + j]) for j, h in enumerate(states))
            s = self.node_ops[i](z, z, node_weights[i])
            offset += len(states)
            states.append(s)
        tail = states[-self.node_multiplier:]
        if self.node_multiplier != 1:
            bn = self.bn
            out = ConvBnActFn.apply('relu', self.out_dropout.p, self.training, bn.running_mean,
                                    bn.running_var, bn.num_batches_tracked, self.out_conv.weight,
                                    self.out_conv.bias, bn.weight, bn.bias, *tail)
        else:
            out = tail[0]
        # residual with the first input, then LayerNorm([C, L])
        return CatLnFn.apply(False, self.ln.weight, self.ln.bias, x, out)


class FusionNode(nn.Module):
    def __init__(self, node_steps, node_multiplier, args):
        super().__init__()
        self.node_steps = node_steps
        self.node_multiplier = node_multiplier
        self.node_cell = NodeCell(node_steps, node_multiplier, args)
        self.num_input_nodes = 2
        self.num_keep_edges = 2
        self._initialize_betas()
        self._initialize_gammas()
        self._arch_parameters = [self.betas, self.gammas]

    def _initialize_betas(self):
        k = sum(self.num_input_nodes + i for i in range(self.node_steps))
        # betas weigh the inner edges; unregistered leaf tensor, like the reference
        self.betas = (1e-3 * torch.randn(k, len(STEP_EDGE_PRIMITIVES))).requires_grad_(True)

    def _initialize_gammas(self):
        # gammas weigh the fusion primitive of each inner step
        self.gammas = (1e-3 * torch.randn(self.node_steps, len(STEP_STEP_PRIMITIVES))).requires_grad_(True)

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        # keep tensor identity (the arch optimizer holds references) while following .to()/.cuda()
        for t in (self.betas, self.gammas):
            t.data = fn(t.data)
            if t.grad is not None:
                t.grad.data = fn(t.grad.data)
        return self

    def forward(self, x, y):
        edge_weights = arch_softmax(self.betas, x.device)
        node_weights = arch_softmax(self.gammas, x.device)
        return self.node_cell(x, y, edge_weights, node_weights)

    def arch_parameters(self):
        return self._arch_parameters

    def node_genotype(self):
        none_idx = STEP_EDGE_PRIMITIVES.index('none')
        ew = torch.softmax(self.betas.detach().float().cpu(), dim=-1).numpy()
        nw = torch.softmax(self.gammas.detach().float().cpu(), dim=-1).numpy()

        def best_op(row):
            best = None
            for k in range(len(row)):
                if k != none_idx and (best is None or row[k] > row[best]):
                    best = k
            return best

        edge_gene = []
        start = 0
        for i in range(self.node_steps):
            n = self.num_input_nodes + i
            W = ew[start:start + n]
            # the edge RANKING skips PRIMITIVES.index('none') — the cell-level list — as the reference does
            # (node_search.py:121); the op choice below skips STEP_EDGE_PRIMITIVES.index('none') (:126).  The two
            # coincide unless someone moves 'none' in an edited PRIMITIVES
            rank_none = PRIMITIVES.index('none')
            strength = [max(W[j][k] for k in range(W.shape[1]) if k != rank_none) for j in range(n)]
            # stable descending sort: ties keep the lower edge index first
            keep = sorted(range(n), key=lambda j: -strength[j])[:self.num_keep_edges]
            edge_gene += [(STEP_EDGE_PRIMITIVES[best_op(W[j])], j) for j in keep]
            start += n

        node_gene = []
        for i in range(self.node_steps):
            row = nw[i]
            best = 0
            for k in range(1, len(row)):
                if row[k] > row[best]:
                    best = k
            node_gene.append(STEP_STEP_PRIMITIVES[best])

        lo = self.num_input_nodes + self.node_steps - self.node_multiplier
        concat_gene = list(range(lo, self.node_steps + self.num_input_nodes))
        return StepGenotype(inner_edges=edge_gene, inner_steps=node_gene, inner_concat=concat_gene)


if __name__ == '__main__':
    # the reference's smoke block (node_search.py:165-183), on the HIP device
    class _Args:
        def __init__(self, C, L):
            self.C, self.L, self.drpt = C, L, 0.1

    _a = _Args(16, 8)
    _node = FusionNode(2, 1, _a).cuda()
    _x = torch.randn(4, 16, 8, device='cuda')
    print(_node(_x, _x).shape, _node.node_genotype())
