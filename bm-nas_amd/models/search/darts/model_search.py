"""Searchable fusion cell and network (the DARTS-style hypernet of BM-NAS).

Mirror of the reference's models/search/darts/model_search.py (FusionCell :13-68,
FusionNetwork :70-182): same names, constructor / forward signatures, arch_parameters()
layout, genotype() rules and state_dict keys.  FusionCell.forward runs the whole cell as
ONE autograd node over the gfx950 kernels (bmnas.functions.FusedCellFn).
"""
import torch
import torch.nn as nn

from bmnas.cell import Arena, Pack
from bmnas.functions import CatLnFn, FusedCellFn, arch_softmax

from .genotypes import PRIMITIVES, Genotype
from .node_search import FusionNode
from .operations import FusionMixedOp, mixed_edge_sum


class FusionCell(nn.Module):
    def __init__(self, steps, multiplier, args):
        super().__init__()
        self._steps = steps
        self._multiplier = multiplier
        self.args = args
        self.num_input_nodes = args.num_input_nodes
        self.C, self.L = args.C, args.L

        self._ops = nn.ModuleList()
        self._step_nodes = nn.ModuleList()
        self.ln = nn.LayerNorm([self.C * multiplier, self.L])
        for i in range(steps):
            for _ in range(self.num_input_nodes + i):
                self._ops.append(FusionMixedOp(self.C, self.L, args))
        self._initialize_step_nodes(args)
        self._fusable = all(op._default for op in self._ops)

    def _initialize_step_nodes(self, args):
        for _ in range(self._steps):
            self._step_nodes.append(FusionNode(args.node_steps, args.node_multiplier, args))

    def arch_parameters(self):
        self._arch_parameters = []
        for node in self._step_nodes:
            self._arch_parameters += node.arch_parameters()
        return self._arch_parameters

    # -- packs for the fused path ----------------------------------------------------------------
    def pack(self):
        return Pack(nodes=[n.node_cell.pack() for n in self._step_nodes], ln_w=self.ln.weight.detach(),
                    ln_b=self.ln.bias.detach())

    def param_list(self):
        ps = []
        for n in self._step_nodes:
            ps += n.node_cell.param_list()
        return ps + [self.ln.weight, self.ln.bias]

    def grad_pack(self, device, alpha_w, beta_ws, gamma_ws, shards=None):
        """One zero-filled arena holding every atomically accumulated gradient of the cell
        plus the gradients w.r.t. the softmaxed arch weights.  shards: copies of the arch section
        (bmnas.cell.arch_shards; default ARCH_SHARDS)."""
        from bmnas.cell import ARCH_SHARDS
        if shards is not None:
            ARCH_SHARDS = int(shards)
        arena = Arena()
        hn = [n.node_cell.plan_grads(arena) for n in self._step_nodes]
        hl = (arena.ask(self.C * self._multiplier, self.L), arena.ask(self.C * self._multiplier, self.L))
        # arch-weight gradients: ARCH_SHARDS copies of one section (shard 0 is handed out; the
        # kernels spread their atomics over the copies, the softmax backward sums them)
        base = arena.total
        ha = arena.ask(*alpha_w.shape)
        hb = [arena.ask(*t.shape) for t in beta_ws]
        hg = [arena.ask(*t.shape) for t in gamma_ws]
        stride = arena.total - base
        arena.total += stride * (ARCH_SHARDS - 1)
        arena.alloc(device, zero=False)      # cleared by the first kernel of the backward (K7's)
        CG = Pack(scrub=arena.buf, nodes=[n.node_cell.bind_grads(arena, h) for n, h in zip(self._step_nodes, hn)],
                  dln_w=arena.view(hl[0]), dln_b=arena.view(hl[1]), shards=ARCH_SHARDS, shard_stride=stride)
        for g in CG.nodes:
            g.shards, g.shard_stride = ARCH_SHARDS, stride
        return CG, arena.view(ha), [arena.view(i) for i in hb], [arena.view(i) for i in hg]

    def grads_in_param_order(self, CG):
        gs = []
        for n, g in zip(self._step_nodes, CG.nodes):
            gs += n.node_cell.grads_in_param_order(g)
        return gs + [CG.dln_w, CG.dln_b]

    def head_fusable(self, classifier):
        """Whether the cell's LayerNorm tail can continue into `classifier` in one launch
        (csrc/head.hip): the concatenated states are all step-node outputs, <= 128 classes."""
        from bmnas import cell as K
        return (K.FUSE_HEAD and self._fusable and isinstance(classifier, nn.Linear)
                and type(classifier).__module__.startswith('bmnas') and classifier.bias is not None
                and classifier.out_features <= 128 and self._multiplier <= self._steps
                and self._multiplier <= 4 and (self.C * self.L) % 16 == 0
                and classifier.in_features == self._multiplier * self.C * self.L
                and all(op._default for n in self._step_nodes for op in n.node_cell.node_ops))

    def forward(self, input_features, weights, weights_are_logits=False, classifier=None):
        """weights: the softmaxed alphas (k, 2), as in the reference; FusionNetwork passes the
        raw alphas with weights_are_logits=True so that every arch softmax of the cell runs
        in one kernel launch.  classifier (FusionNetwork.forward_classified): a bmnas.nn.Linear
        the cell's output feeds — the call then returns ITS output, with K7 + Linear as one launch."""
        states = list(input_features)
        dev = states[0].device
        w = weights if weights.device == dev else weights.to(dev)
        if self._fusable and all(op._default for n in self._step_nodes for op in n.node_cell.node_ops):
            arch = []
            for n in self._step_nodes:
                arch += [t if t.device == dev else t.to(dev) for t in (n.betas, n.gammas)]
            if classifier is not None and states[0].is_cuda and self.head_fusable(classifier):
                from bmnas import cell as K
                out = FusedCellFn.apply(self, self.training, weights_are_logits, w, 2, *states, *arch,
                                        *self.param_list(), classifier.weight, classifier.bias)
                out._bmnas_head = K.LAST_HEAD.pop()      # lets a fused criterion find its head
                return out
            out = FusedCellFn.apply(self, self.training, weights_are_logits, w, 0, *states, *arch,
                                    *self.param_list())
            return out if classifier is None else classifier(out)
        if weights_are_logits:
            w = arch_softmax(w, dev)
        # edited primitive lists: same dataflow, composed op by op
        offset = 0
        for i in range(self._steps):
            if self._fusable:
                sif = mixed_edge_sum(states, w, offset)
            else:
                sif = sum(self._ops[offset + j](h, w[offset + j]) for j, h in enumerate(states))
            s = self._step_nodes[i](sif, sif)
            offset += len(states)
            states.append(s)
        out = CatLnFn.apply(True, self.ln.weight, self.ln.bias, None, *states[-self._multiplier:])
        out = out.view(out.size(0), -1)
        return out if classifier is None else classifier(out)


class FusionNetwork(nn.Module):
    def __init__(self, steps, multiplier, num_input_nodes, num_keep_edges, args, criterion=None,
                 logger=None):
        super().__init__()
        self.logger = logger
        self._steps = steps
        self._multiplier = multiplier
        self._criterion = criterion
        self._num_input_nodes = num_input_nodes
        self._num_keep_edges = num_keep_edges

        self.cell = FusionCell(steps, multiplier, args)
        self.cell_arch_parameters = self.cell.arch_parameters()
        self._initialize_alphas()
        self._arch_parameters = [self.alphas_edges] + self.cell_arch_parameters

    def _initialize_alphas(self):
        k = sum(self._num_input_nodes + i for i in range(self._steps))
        # unregistered leaf tensor (not in state_dict / parameters()), like the reference
        self.alphas_edges = (1e-3 * torch.randn(k, len(PRIMITIVES))).requires_grad_(True)

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        t = self.alphas_edges
        t.data = fn(t.data)
        if t.grad is not None:
            t.grad.data = fn(t.grad.data)
        return self

    def forward(self, input_features):
        assert self._num_input_nodes == len(input_features)
        # softmax(alphas_edges) (reference :95) is folded into the fused cell call
        return self.cell(input_features, self.alphas_edges, weights_are_logits=True)

    def forward_classified(self, input_features, classifier):
        """classifier(self(input_features)) — what Searchable_*.forward does next
        (mmimdb_darts_searchable.py:113-114) — with the cell's LayerNorm tail and the classifier as
        one launch where the shapes allow (FusionCell.head_fusable), else exactly that composition."""
        assert self._num_input_nodes == len(input_features)
        return self.cell(input_features, self.alphas_edges, weights_are_logits=True, classifier=classifier)

    def _loss(self, input_features, labels):
        return self._criterion(self(input_features), labels)

    def arch_parameters(self):
        return self._arch_parameters

    def genotype(self):
        none_idx = PRIMITIVES.index('none')
        W_all = torch.softmax(self.alphas_edges.detach().float().cpu(), dim=-1).numpy()
        N = self._num_input_nodes

        def strongest(row):
            return max(row[t] for t in range(len(row)) if t != none_idx)

        def best_op(row):
            best = None
            for k in range(len(row)):
                if k != none_idx and (best is None or row[k] > row[best]):
                    best = k
            return best

        gene_edges = []
        used = set()
        start = 0
        for i in range(self._steps):
            W = W_all[start:start + N + i]
            # candidate pairs of ORIGINAL input nodes with at least one node not used yet,
            # scored by the product of their strongest non-'none' weights (fp32, like numpy)
            cands = [(j, k, strongest(W[j]) * strongest(W[k]))
                     for j in range(N) for k in range(j + 1, N) if not (j in used and k in used)]
            # stable sort + [0] == first maximal pair; an empty list raises IndexError as the
            # reference does when every input node has been used
            j, k, _ = sorted(cands, key=lambda c: -c[2])[:1][0]
            used.update((j, k))
            for e in (j, k):
                gene_edges.append((PRIMITIVES[best_op(W[e])], e))
            start += N + i

        gene_steps = [node.node_genotype() for node in self.cell._step_nodes]
        gene_concat = list(range(N + self._steps - self._multiplier, self._steps + N))
        return Genotype(edges=gene_edges, concat=gene_concat, steps=gene_steps)
