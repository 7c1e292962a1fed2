"""CPU oracle for the BM-NAS fusion-search hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional restatement (plain PyTorch CPU ops, no
nn.Module state) of the reference's fusion hypernet forward pass; gradients come
from torch.autograd over the restated forward.  It is *not* part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.  The product
path (``bm-nas_amd/``) never imports anything from ``oracle/`` and fails loudly
when the HIP extension is missing.

Pinning: the reference ships no tests or golden vectors for this path (SURVEY.md
section 4), so the oracle is pinned against outputs of the reference itself, run in
the build container by ``tests/golden/make_golden.py`` (imports /root/reference
read-only) and committed as ``tests/golden/*.npz|json``.
``tests/test_oracle_golden.py`` checks this file against every one of them.

Every function cites the reference file:line (relative to /root/reference) whose
behaviour it restates.  The op *sequence* deliberately mirrors the reference's
(e.g. the ``w0*(x*0.) + w1*x`` form of the mixed edge), so that timing this oracle
on CPU is a fair stand-in for timing the reference, and so that non-finite inputs
propagate the same way.
"""
from __future__ import annotations

import math
from collections import namedtuple
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

EPS = 1e-5            # nn.LayerNorm / nn.BatchNorm1d default eps
BN_MOMENTUM = 0.1     # nn.BatchNorm1d default momentum
ATTN_DROP = 0.1       # hard-coded in ScaledDotAttn (node_operations.py:89)

# Same field names as models/search/darts/genotypes.py:3-4 (separate namedtuple
# types: the oracle compares field-by-field, never by pickled identity).
Genotype = namedtuple('Genotype', 'edges steps concat')
StepGenotype = namedtuple('StepGenotype', 'inner_edges inner_steps inner_concat')

PRIMITIVES = ['none', 'skip']                                   # genotypes.py:6-9
STEP_EDGE_PRIMITIVES = ['none', 'skip']                         # genotypes.py:11-14
STEP_STEP_PRIMITIVES = ['Sum', 'ScaleDotAttn', 'LinearGLU', 'ConcatFC']  # genotypes.py:16-21


class Cfg(dict):
    """Hypernet configuration: N (num_input_nodes), C, L, S (steps), M (multiplier),
    ns (node_steps), nm (node_multiplier), drpt."""
    __getattr__ = dict.__getitem__


def make_cfg(N, C, L, S=2, M=2, ns=1, nm=1, drpt=0.1) -> Cfg:
    return Cfg(N=N, C=C, L=L, S=S, M=M, ns=ns, nm=nm, drpt=drpt)


CONFIGS = {
    # main_darts_searchable_mmimdb.py:17-58 / _ntu.py:17-63 / _ego.py:17-67 defaults
    'mmimdb': make_cfg(N=6, C=192, L=16, S=2, M=2, ns=1, nm=1, drpt=0.1),
    'ntu': make_cfg(N=8, C=128, L=8, S=2, M=2, ns=2, nm=2, drpt=0.2),
    'ego': make_cfg(N=8, C=128, L=8, S=2, M=2, ns=3, nm=3, drpt=0.0),
}


# --------------------------------------------------------------------------- shapes
def num_cell_edges(cfg) -> int:
    """k of alphas_edges (model_search.py:100)."""
    return sum(cfg.N + i for i in range(cfg.S))


def num_node_edges(cfg) -> int:
    """k of betas (node_search.py:90)."""
    return sum(2 + t for t in range(cfg.ns))


def arch_shapes(cfg, primitives=None) -> List[tuple]:
    """Shapes of arch_parameters(): [alphas_edges, betas_0, gammas_0, betas_1, ...]
    (model_search.py:91, model_search.py:44-48, node_search.py:87).  ``primitives``: an edited
    cell-level PRIMITIVES list (genotypes.py:6-9), default ['none', 'skip']."""
    shapes = [(num_cell_edges(cfg), len(primitives or PRIMITIVES))]
    for _ in range(cfg.S):
        shapes.append((num_node_edges(cfg), len(STEP_EDGE_PRIMITIVES)))
        shapes.append((cfg.ns, len(STEP_STEP_PRIMITIVES)))
    return shapes


def param_shapes(cfg, primitives=None) -> "Dict[str, tuple]":
    """state_dict() key -> shape of the search FusionNetwork, in registration order
    (module construction order of model_search.py:20-35, node_search.py:20-46,
    node_operations.py:88-90,25-27,44-46).  ``num_batches_tracked`` entries are
    int64 scalars (shape ()).  With an edited cell-level ``primitives`` list every mixed edge
    ``cell._ops.{e}`` also owns the parameters of its fc_relu / fc_mish primitives
    (operations.py:22-27, 48-54, 97-102), appended after the default entries."""
    C, L = cfg.C, cfg.L
    out: Dict[str, tuple] = {}
    out['cell.ln.weight'] = (cfg.M * C, L)
    out['cell.ln.bias'] = (cfg.M * C, L)

    def bn(prefix, ch):
        out[prefix + '.weight'] = (ch,)
        out[prefix + '.bias'] = (ch,)
        out[prefix + '.running_mean'] = (ch,)
        out[prefix + '.running_var'] = (ch,)
        out[prefix + '.num_batches_tracked'] = ()

    for i in range(cfg.S):
        nc = f'cell._step_nodes.{i}.node_cell'
        for t in range(cfg.ns):
            ops = f'{nc}.node_ops.{t}._ops'
            out[f'{ops}.1.ln.weight'] = (C, L)
            out[f'{ops}.1.ln.bias'] = (C, L)
            out[f'{ops}.2.conv.weight'] = (2 * C, 2 * C, 1)
            out[f'{ops}.2.conv.bias'] = (2 * C,)
            bn(f'{ops}.2.bn', 2 * C)
            out[f'{ops}.3.conv.weight'] = (C, 2 * C, 1)
            out[f'{ops}.3.conv.bias'] = (C,)
            bn(f'{ops}.3.bn', C)
        if cfg.nm != 1:
            out[f'{nc}.out_conv.weight'] = (C, cfg.nm * C, 1)
            out[f'{nc}.out_conv.bias'] = (C,)
            bn(f'{nc}.bn', C)
        out[f'{nc}.ln.weight'] = (C, L)
        out[f'{nc}.ln.bias'] = (C, L)
    # FusionMixedOp is ALSO the class of NodeCell.edge_ops (node_search.py:10, 31), so an edited
    # PRIMITIVES list puts the fc modules on the inner edges too
    owners = [f'cell._ops.{e}' for e in range(num_cell_edges(cfg))]
    owners += [f'cell._step_nodes.{i}.node_cell.edge_ops.{e}' for i in range(cfg.S)
               for e in range(num_node_edges(cfg))]
    for owner in owners:
        for pi, prim in enumerate(primitives or PRIMITIVES):
            if prim in ('fc_relu', 'fc_mish'):
                out[f'{owner}._ops.{pi}.linear.weight'] = (C, C)
                out[f'{owner}._ops.{pi}.linear.bias'] = (C,)
                bn(f'{owner}._ops.{pi}.bn', C)
    return out


def is_buffer(key: str) -> bool:
    return key.endswith(('running_mean', 'running_var', 'num_batches_tracked'))


# ------------------------------------------------------------------ primitive ops
_INJECTED = None      # iterator over multiplier tensors while `injected_masks` is active


class injected_masks:
    """with injected_masks(masks): every LIVE dropout site (train mode, p > 0) multiplies its input by the next
    tensor of `masks` instead of drawing from torch's generator.  A mask holds the site's multipliers — 0 for a
    dropped element, 1/(1-p) for a kept one — which is all nn.Dropout does (x * Bernoulli(1-p) / (1-p),
    node_operations.py:38, :55, :105; node_search.py:64).  Sites are visited in the reference's execution
    order: per inner step ScaledDotAttn, LinearGLU, ConcatFC (node_operations.py:119 evaluates `_ops` in list
    order), then the node's out_conv dropout, cell step by cell step.  `.used` counts the masks consumed."""

    def __init__(self, masks):
        self.masks = list(masks)
        self.used = 0

    def __enter__(self):
        global _INJECTED
        assert _INJECTED is None, 'injected_masks does not nest'
        _INJECTED = self
        return self

    def __exit__(self, *exc):
        global _INJECTED
        _INJECTED = None
        return False

    def next(self, x):
        if self.used >= len(self.masks):
            raise IndexError(f'dropout site {self.used}: no mask left ({len(self.masks)} injected)')
        m = self.masks[self.used]
        self.used += 1
        if m.numel() != x.numel():
            raise ValueError(f'dropout site {self.used - 1}: mask of {m.numel()} elements for a tensor {tuple(x.shape)}')
        return m.reshape(x.shape).to(x.dtype)


def _dropout(x, p, training):
    # nn.Dropout: identity in eval; in train mode Bernoulli(1-p) mask scaled by 1/(1-p).
    if _INJECTED is not None and training and p > 0.0:
        return x * _INJECTED.next(x)
    return F.dropout(x, p=p, training=training)


_RELU_POLICY = None     # a relu_decisions instance while one is active


class relu_decisions:
    """with relu_decisions(near=d, flips=F) as rd: ... — bookkeeping of the ReLU branch decisions of one oracle
    evaluation (F.relu at node_operations.py:54, node_search.py:63, model_search.py:65, aux_models.py:73, :112,
    operations.py:34).  A ReLU input within floating-point round-off of zero may legitimately fall on either
    side: two correct fp32 evaluations of the same math (different summation orders) can disagree on it, and
    ONE such element moves whole gradient tensors by ~1e-2 of their scale (it removes or adds one sample's term
    of a batch reduction).  `near`: every element with |input| < near is recorded as (site, flat index, value)
    in `.ambiguous` (sites numbered in execution order).  `flips`: a set of (site, flat index) that take the
    OPPOSITE branch in this evaluation (output x * [x <= 0] there), so a checker can ask whether a result
    equals the reference math under SOME assignment of the ambiguous decisions."""

    def __init__(self, near=0.0, flips=()):
        self.near = near
        self.flips = {}
        for site, idx in flips:
            self.flips.setdefault(site, []).append(idx)
        self.ambiguous = []
        self.site = 0

    def __enter__(self):
        global _RELU_POLICY
        assert _RELU_POLICY is None, 'relu_decisions does not nest'
        _RELU_POLICY = self
        return self

    def __exit__(self, *exc):
        global _RELU_POLICY
        _RELU_POLICY = None
        return False

    def apply(self, x):
        site = self.site
        self.site += 1
        flat = x.detach().reshape(-1)
        if self.near > 0:
            for i in torch.nonzero(flat.abs() < self.near).reshape(-1).tolist():
                self.ambiguous.append((site, i, float(flat[i])))
        if site not in self.flips:
            return F.relu(x)
        keep = (flat > 0)
        for i in self.flips[site]:
            keep[i] = not bool(keep[i])
        return x * keep.reshape(x.shape).to(x.dtype)


def _relu(x):
    return F.relu(x) if _RELU_POLICY is None else _RELU_POLICY.apply(x)


def mixed_edge(x, w):
    """FusionMixedOp.forward (operations.py:104-105) with PRIMITIVES ['none','skip']:
    sum(w_p * op_p(x)) = 0 + w[0]*Zero(x) + w[1]*Identity(x); Zero is x.mul(0.)
    (operations.py:18-20), Identity returns x (operations.py:92-93)."""
    return 0 + w[0] * x.mul(0.) + w[1] * x


def mixed_edge_sum(states: Sequence[torch.Tensor], W, offset: int):
    """sum over incoming edges (model_search.py:58, node_search.py:54)."""
    acc = 0
    for j, h in enumerate(states):
        acc = acc + mixed_edge(h, W[offset + j])
    return acc


def op_fc(x, p, prefix, kind, training, drpt):
    """FC_Relu.forward (operations.py:30-38) / FC_Mish.forward (operations.py:56-65):
    Linear(C, C) over the channel dim (transpose, linear, transpose) -> ReLU | Mish
    (x * tanh(softplus(x)), operations.py:44-46) -> BatchNorm1d(C) -> Dropout(drpt)."""
    out = F.linear(x.transpose(1, 2), p[prefix + '.linear.weight'], p[prefix + '.linear.bias']).transpose(1, 2)
    out = _relu(out) if kind == 'fc_relu' else out * torch.tanh(F.softplus(out))
    out = F.batch_norm(out, p[prefix + '.bn.running_mean'], p[prefix + '.bn.running_var'],
                       p[prefix + '.bn.weight'], p[prefix + '.bn.bias'], training, BN_MOMENTUM, EPS)
    _bump_nbt(p, prefix + '.bn.num_batches_tracked', training)
    return _dropout(out, drpt, training)


def mixed_edge_general(x, w, p, prefix, primitives, training, drpt):
    """FusionMixedOp.forward (operations.py:104-105) for an edited PRIMITIVES list:
    sum(w_p * OPS[p](x)) in list order (Python's sum starts from 0).  ``zip(weights, self._ops)``
    stops at the shorter sequence: the inner edges of a NodeCell hand a row of
    len(STEP_EDGE_PRIMITIVES) = 2 weights to a module with len(PRIMITIVES) ops, so only the
    first two primitives take part there (node_search.py:54, 92)."""
    acc = 0
    for pi, prim in enumerate(primitives[:len(w)]):
        if prim == 'none':
            o = x.mul(0.)
        elif prim == 'skip':
            o = x
        else:
            o = op_fc(x, p, f'{prefix}._ops.{pi}', prim, training, drpt)
        acc = acc + w[pi] * o
    return acc


def op_sum(x, y):
    """Sum.forward (node_operations.py:19-20)."""
    return x + y


def op_scaled_dot_attn(x, y, ln_w, ln_b, training, attn_drop=ATTN_DROP):
    """ScaledDotAttn.forward (node_operations.py:92-108): q = x^T, k = y, v = y^T;
    scores = q@k / sqrt(d_k) with d_k = q.size(-1) = C; softmax(-1); out = (attn@v)^T;
    dropout(0.1); LayerNorm([C, L])."""
    q = x.transpose(1, 2)
    k = y
    v = y.transpose(1, 2)
    d_k = q.size(-1)
    scores = torch.matmul(q, k) / math.sqrt(d_k)
    attn = F.softmax(scores, dim=-1)
    out = torch.matmul(attn, v).transpose(1, 2)
    out = _dropout(out, attn_drop, training)
    return F.layer_norm(out, tuple(ln_w.shape), ln_w, ln_b, EPS)


def _conv_bn(cat, conv_w, conv_b, bn_w, bn_b, rm, rv, training):
    out = F.conv1d(cat, conv_w, conv_b)
    return F.batch_norm(out, rm, rv, bn_w, bn_b, training, BN_MOMENTUM, EPS)


def op_linear_glu(x, y, p, prefix, training, drpt):
    """LinearGLU.forward (node_operations.py:30-39): cat -> Conv1d(2C,2C,1) ->
    BatchNorm1d(2C) -> glu(dim=1) -> Dropout(drpt)."""
    cat = torch.cat([x, y], dim=1)
    out = _conv_bn(cat, p[prefix + '.conv.weight'], p[prefix + '.conv.bias'],
                   p[prefix + '.bn.weight'], p[prefix + '.bn.bias'],
                   p[prefix + '.bn.running_mean'], p[prefix + '.bn.running_var'], training)
    out = F.glu(out, dim=1)
    return _dropout(out, drpt, training)


def op_concat_fc(x, y, p, prefix, training, drpt):
    """ConcatFC.forward (node_operations.py:49-56): cat -> Conv1d(2C,C,1) ->
    BatchNorm1d(C) -> ReLU -> Dropout(drpt)."""
    cat = torch.cat([x, y], dim=1)
    out = _conv_bn(cat, p[prefix + '.conv.weight'], p[prefix + '.conv.bias'],
                   p[prefix + '.bn.weight'], p[prefix + '.bn.bias'],
                   p[prefix + '.bn.running_mean'], p[prefix + '.bn.running_var'], training)
    out = _relu(out)
    return _dropout(out, drpt, training)


def node_mixed_op(x, y, gamma_row, p, prefix, training, drpt, attn_drop=ATTN_DROP):
    """NodeMixedOp.forward (node_operations.py:118-120):
    sum(w * op(x, y)) over STEP_STEP_PRIMITIVES order [Sum, ScaleDotAttn, LinearGLU, ConcatFC]."""
    outs = [
        op_sum(x, y),
        op_scaled_dot_attn(x, y, p[prefix + '.1.ln.weight'], p[prefix + '.1.ln.bias'],
                           training, attn_drop),
        op_linear_glu(x, y, p, prefix + '.2', training, drpt),
        op_concat_fc(x, y, p, prefix + '.3', training, drpt),
    ]
    acc = 0
    for w, o in zip(gamma_row, outs):
        acc = acc + w * o
    return acc


def _bump_nbt(p, key, training):
    # nn.BatchNorm1d increments num_batches_tracked once per training forward.
    if training and key in p and p[key] is not None:
        p[key] += 1


def node_cell(x, y, beta_w, gamma_w, p, prefix, cfg, training, attn_drop=ATTN_DROP, primitives=None):
    """NodeCell.forward (node_search.py:48-70)."""
    states = [x, y]
    offset = 0
    for t in range(cfg.ns):
        if primitives is None or list(primitives) == PRIMITIVES:
            z = mixed_edge_sum(states, beta_w, offset)
        else:
            z = 0
            for j, h in enumerate(states):
                z = z + mixed_edge_general(h, beta_w[offset + j], p, f'{prefix}.edge_ops.{offset + j}',
                                           primitives, training, cfg.drpt)
        s = node_mixed_op(z, z, gamma_w[t], p, f'{prefix}.node_ops.{t}._ops', training,
                          cfg.drpt, attn_drop)
        _bump_nbt(p, f'{prefix}.node_ops.{t}._ops.2.bn.num_batches_tracked', training)
        _bump_nbt(p, f'{prefix}.node_ops.{t}._ops.3.bn.num_batches_tracked', training)
        offset += len(states)
        states.append(s)
    out = torch.cat(states[-cfg.nm:], dim=1)
    if cfg.nm != 1:
        out = _conv_bn(out, p[prefix + '.out_conv.weight'], p[prefix + '.out_conv.bias'],
                       p[prefix + '.bn.weight'], p[prefix + '.bn.bias'],
                       p[prefix + '.bn.running_mean'], p[prefix + '.bn.running_var'], training)
        _bump_nbt(p, prefix + '.bn.num_batches_tracked', training)
        out = _relu(out)
        out = _dropout(out, cfg.drpt, training)
    out = out + x      # reference does the in-place ``out += x`` (node_search.py:67)
    ln_w, ln_b = p[prefix + '.ln.weight'], p[prefix + '.ln.bias']
    return F.layer_norm(out, tuple(ln_w.shape), ln_w, ln_b, EPS)


def fusion_cell(inputs: Sequence[torch.Tensor], arch: Sequence[torch.Tensor], p, cfg,
                training: bool, attn_drop: float = ATTN_DROP, primitives=None):
    """FusionNetwork.forward + FusionCell.forward (model_search.py:93-97, 50-68) and
    FusionNode.forward (node_search.py:101-105).  ``arch`` is the arch_parameters()
    list (raw alphas/betas/gammas; the softmaxes are applied here)."""
    assert len(inputs) == cfg.N
    W = F.softmax(arch[0], dim=-1)
    states = list(inputs)
    offset = 0
    for i in range(cfg.S):
        if primitives is None or list(primitives) == PRIMITIVES:
            sif = mixed_edge_sum(states, W, offset)
        else:
            sif = 0
            for j, h in enumerate(states):
                sif = sif + mixed_edge_general(h, W[offset + j], p, f'cell._ops.{offset + j}', primitives,
                                               training, cfg.drpt)
        beta_w = F.softmax(arch[1 + 2 * i], dim=-1)
        gamma_w = F.softmax(arch[2 + 2 * i], dim=-1)
        s = node_cell(sif, sif, beta_w, gamma_w, p, f'cell._step_nodes.{i}.node_cell', cfg,
                      training, attn_drop, primitives)
        offset += len(states)
        states.append(s)
    out = torch.cat(states[-cfg.M:], dim=1)
    ln_w, ln_b = p['cell.ln.weight'], p['cell.ln.bias']
    out = F.layer_norm(out, tuple(ln_w.shape), ln_w, ln_b, EPS)
    out = _relu(out)
    return out.view(out.size(0), -1)


def hypernet_logits(inputs, arch, p, cls_w, cls_b, cfg, training, attn_drop=ATTN_DROP, primitives=None):
    """fusion_net + central_classifier (mmimdb_darts_searchable.py:113-114)."""
    return F.linear(fusion_cell(inputs, arch, p, cfg, training, attn_drop, primitives), cls_w, cls_b)


# ---------------------------------------------------------------- reshape layers (row f1)
def reshape_pool(x, L, kind):
    """The pooling in front of the reshape conv.  kind 'mmimdb': ReshapeInputLayer_MMIMDB.forward
    (aux_models.py:101-108): (b, C_in[, H, W]) -> unsqueeze twice -> view(b, C_in, d2, -1) ->
    AdaptiveMaxPool2d((sqrt L, sqrt L)) -> (b, C_in, L).  kind 'video': ReshapeInputLayer.forward
    (aux_models.py:62-70): view(b, C_in, T, -1) -> AdaptiveMaxPool2d((L, 1)) -> (b, C_in, L) ->
    F.interpolate(out, L) (nearest; an identity once the pool already returns L positions)."""
    if kind == 'mmimdb':
        side = int(math.sqrt(L * 1.0))
        assert side * side == L
        out = x.unsqueeze(-1).unsqueeze(-1)
        out = out.view(out.size(0), out.size(1), out.size(2), -1)
        out = F.adaptive_max_pool2d(out, (side, side))
        return out.view(out.size(0), out.size(1), -1)
    out = x.unsqueeze(-1)
    out = out.view(out.size(0), out.size(1), out.size(2), -1)
    out = F.adaptive_max_pool2d(out, (L, 1))
    out = out.view(out.size(0), out.size(1), -1)
    return F.interpolate(out, L)


def reshape_layer(x, p, prefix, L, kind, training, drpt):
    """ReshapeInputLayer{,_MMIMDB}.forward (aux_models.py:62-76, 101-115): pool -> Conv1d(C_in, C, 1) ->
    BatchNorm1d(C) -> ReLU -> Dropout(drpt)."""
    out = reshape_pool(x, L, kind)
    out = _conv_bn(out, p[prefix + '.conv.weight'], p[prefix + '.conv.bias'], p[prefix + '.bn.weight'],
                   p[prefix + '.bn.bias'], p[prefix + '.bn.running_mean'], p[prefix + '.bn.running_var'], training)
    _bump_nbt(p, prefix + '.bn.num_batches_tracked', training)
    return _dropout(_relu(out), drpt, training)


def loss_fn(kind: str):
    """'bce' = BCEWithLogitsLoss (mmimdb_darts_searchable.py:22); 'ce' = CrossEntropyLoss
    (ntu_darts_searchable.py:25, ego_darts_searchable.py:24)."""
    if kind == 'bce':
        return F.binary_cross_entropy_with_logits
    if kind == 'ce':
        return F.cross_entropy
    raise ValueError(kind)


def search_step(inputs, labels, arch, p, cls_w, cls_b, cfg, loss_kind, training=True,
                attn_drop=ATTN_DROP, primitives=None):
    """One forward + backward of the hypernet (the benchmarked 'search step'):
    returns (logits, loss, grads) with grads for every float param in ``p``, the
    classifier, the arch list and the N inputs.  Buffers in ``p`` are updated in
    place in training mode, like nn.BatchNorm1d does."""
    leaves = {}
    pp = {}
    for k, v in p.items():
        if is_buffer(k):
            pp[k] = v
        else:
            pp[k] = v.detach().requires_grad_(True)
            leaves['p:' + k] = pp[k]
    a = [t.detach().requires_grad_(True) for t in arch]
    xs = [t.detach().requires_grad_(True) for t in inputs]
    cw = cls_w.detach().requires_grad_(True)
    cb = cls_b.detach().requires_grad_(True)
    logits = hypernet_logits(xs, a, pp, cw, cb, cfg, training, attn_drop, primitives)
    loss = loss_fn(loss_kind)(logits, labels)
    loss.backward()
    grads = {k[2:]: t.grad for k, t in leaves.items()}
    grads['central_classifier.weight'] = cw.grad
    grads['central_classifier.bias'] = cb.grad
    for i, t in enumerate(a):
        grads[f'arch.{i}'] = t.grad
    for i, t in enumerate(xs):
        grads[f'input.{i}'] = t.grad
    return logits.detach(), loss.detach(), grads


# ---------------------------------------------------------------- found (discrete)
def found_param_shapes(cfg, genotype) -> "Dict[str, tuple]":
    """state_dict() key -> shape of Found_FusionNetwork (model.py:92-131, node.py:8-42)."""
    C, L = cfg.C, cfg.L
    out: Dict[str, tuple] = {}

    def bn(prefix, ch):
        out[prefix + '.weight'] = (ch,)
        out[prefix + '.bias'] = (ch,)
        out[prefix + '.running_mean'] = (ch,)
        out[prefix + '.running_var'] = (ch,)
        out[prefix + '.num_batches_tracked'] = ()

    for i, sg in enumerate(genotype.steps):
        nc = f'cell._step_nodes.{i}.node_cell'
        for t, name in enumerate(sg.inner_steps):
            op = f'{nc}.node_ops.{t}'
            if name == 'ScaleDotAttn':
                out[op + '.ln.weight'] = (C, L)
                out[op + '.ln.bias'] = (C, L)
            elif name == 'LinearGLU':
                out[op + '.conv.weight'] = (2 * C, 2 * C, 1)
                out[op + '.conv.bias'] = (2 * C,)
                bn(op + '.bn', 2 * C)
            elif name == 'ConcatFC':
                out[op + '.conv.weight'] = (C, 2 * C, 1)
                out[op + '.conv.bias'] = (C,)
                bn(op + '.bn', C)
        if cfg.nm != 1:
            out[f'{nc}.out_conv.weight'] = (C, cfg.nm * C, 1)
            out[f'{nc}.out_conv.bias'] = (C,)
            bn(f'{nc}.bn', C)
        out[f'{nc}.ln.weight'] = (C, L)
        out[f'{nc}.ln.bias'] = (C, L)
    out['cell.ln.weight'] = (cfg.M * C, L)
    out['cell.ln.bias'] = (cfg.M * C, L)
    return out


def _edge_op(name, x):
    # OPS registry (operations.py:7-12); only the default primitives are on the path.
    if name == 'skip':
        return x
    if name == 'none':
        return x.mul(0.)
    raise ValueError(name)


def _found_node_op(name, x, y, p, prefix, training, drpt, attn_drop):
    if name == 'Sum':
        return op_sum(x, y)
    if name == 'ScaleDotAttn':
        return op_scaled_dot_attn(x, y, p[prefix + '.ln.weight'], p[prefix + '.ln.bias'],
                                  training, attn_drop)
    if name == 'LinearGLU':
        r = op_linear_glu(x, y, p, prefix, training, drpt)
        _bump_nbt(p, prefix + '.bn.num_batches_tracked', training)
        return r
    if name == 'ConcatFC':
        r = op_concat_fc(x, y, p, prefix, training, drpt)
        _bump_nbt(p, prefix + '.bn.num_batches_tracked', training)
        return r
    raise ValueError(name)


def found_node_cell(x, y, sg, p, prefix, cfg, training, attn_drop=ATTN_DROP):
    """Found_NodeCell.forward (node.py:45-76)."""
    names, idx = zip(*sg.inner_edges)
    states = [x, y]
    for t in range(cfg.ns):
        ix = _edge_op(names[2 * t], states[idx[2 * t]])
        iy = _edge_op(names[2 * t + 1], states[idx[2 * t + 1]])
        states.append(_found_node_op(sg.inner_steps[t], ix, iy, p, f'{prefix}.node_ops.{t}',
                                     training, cfg.drpt, attn_drop))
    out = torch.cat(states[-cfg.nm:], dim=1)
    if cfg.nm != 1:
        out = _conv_bn(out, p[prefix + '.out_conv.weight'], p[prefix + '.out_conv.bias'],
                       p[prefix + '.bn.weight'], p[prefix + '.bn.bias'],
                       p[prefix + '.bn.running_mean'], p[prefix + '.bn.running_var'], training)
        _bump_nbt(p, prefix + '.bn.num_batches_tracked', training)
        out = _relu(out)
        out = _dropout(out, cfg.drpt, training)
    out = out + x
    ln_w, ln_b = p[prefix + '.ln.weight'], p[prefix + '.ln.bias']
    return F.layer_norm(out, tuple(ln_w.shape), ln_w, ln_b, EPS)


def found_cell(inputs, genotype, p, cfg, training, attn_drop=ATTN_DROP):
    """Found_Random_FusionCell.forward (model.py:133-160)."""
    names, idx = zip(*genotype.edges)
    states = list(inputs)
    for i in range(cfg.S):
        h1 = _edge_op(names[2 * i], states[idx[2 * i]])
        h2 = _edge_op(names[2 * i + 1], states[idx[2 * i + 1]])
        states.append(found_node_cell(h1, h2, genotype.steps[i], p,
                                      f'cell._step_nodes.{i}.node_cell', cfg, training, attn_drop))
    M = len(genotype.concat)
    out = torch.cat(states[-M:], dim=1)
    ln_w, ln_b = p['cell.ln.weight'], p['cell.ln.bias']
    out = _relu(F.layer_norm(out, tuple(ln_w.shape), ln_w, ln_b, EPS))
    return out.view(out.size(0), -1)


# ------------------------------------------------------------------------ genotype
def _best_non_none(row, names):
    """argmax over non-'none' primitives, first max wins (strict '>', model_search.py:150-154)."""
    none = names.index('none')
    best = None
    for k in range(len(row)):
        if k == none:
            continue
        if best is None or row[k] > row[best]:
            best = k
    return best


def node_genotype(betas, gammas, cfg) -> StepGenotype:
    """FusionNode.node_genotype (node_search.py:110-163)."""
    ew = F.softmax(betas.detach().float().cpu(), dim=-1).numpy()
    nw = F.softmax(gammas.detach().float().cpu(), dim=-1).numpy()
    none = STEP_EDGE_PRIMITIVES.index('none')
    edge_gene, node_gene = [], []
    start, n = 0, 2
    for t in range(cfg.ns):
        W = ew[start:start + n]
        strength = [max(W[j][k] for k in range(W.shape[1]) if k != none) for j in range(n)]
        # python's sorted() is stable: descending strength, ties keep index order
        order = sorted(range(n), key=lambda j: -strength[j])[:2]
        for j in order:
            edge_gene.append((STEP_EDGE_PRIMITIVES[_best_non_none(W[j], STEP_EDGE_PRIMITIVES)], j))
        start += n
        n += 1
    for t in range(cfg.ns):
        row = nw[t]
        best = 0
        for k in range(1, len(row)):
            if row[k] > row[best]:
                best = k
        node_gene.append(STEP_STEP_PRIMITIVES[best])
    concat = list(range(2 + cfg.ns - cfg.nm, 2 + cfg.ns))
    return StepGenotype(inner_edges=edge_gene, inner_steps=node_gene, inner_concat=concat)


def network_genotype(arch, cfg, primitives=None) -> Genotype:
    """FusionNetwork.genotype (model_search.py:111-182): per step pick the pair (j<k) of
    ORIGINAL input nodes, at least one not selected before, maximising the product of
    their best non-none weights (first maximum wins)."""
    W_all = F.softmax(arch[0].detach().float().cpu(), dim=-1).numpy()
    prims = list(primitives or PRIMITIVES)
    none = prims.index('none')
    gene = []
    selected = set()
    start, n = 0, cfg.N
    for i in range(cfg.S):
        W = W_all[start:start + n]
        best_pair, best_val = None, None
        for j in range(cfg.N):
            for k in range(j + 1, cfg.N):
                if j in selected and k in selected:
                    continue
                wj = max(W[j][t] for t in range(W.shape[1]) if t != none)
                wk = max(W[k][t] for t in range(W.shape[1]) if t != none)
                val = wj * wk
                if best_val is None or val > best_val:      # stable sort on -val: first max
                    best_pair, best_val = (j, k), val
        if best_pair is None:
            # every input node already selected: the reference indexes an empty list
            # (model_search.py:143) -> IndexError; kept as the error behaviour.
            raise IndexError('list index out of range')
        selected.update(best_pair)
        for j in best_pair:
            gene.append((prims[_best_non_none(W[j], prims)], j))
        start += n
        n += 1
    steps = [node_genotype(arch[1 + 2 * i], arch[2 + 2 * i], cfg) for i in range(cfg.S)]
    concat = list(range(cfg.N + cfg.S - cfg.M, cfg.S + cfg.N))
    return Genotype(edges=gene, steps=steps, concat=concat)


def genotype_to_jsonable(g):
    return {
        'edges': [[n, int(j)] for n, j in g.edges],
        'concat': [int(c) for c in g.concat],
        'steps': [{'inner_edges': [[n, int(j)] for n, j in s.inner_edges],
                   'inner_steps': list(s.inner_steps),
                   'inner_concat': [int(c) for c in s.inner_concat]} for s in g.steps],
    }


def genotype_from_jsonable(d) -> Genotype:
    return Genotype(
        edges=[(n, j) for n, j in d['edges']],
        steps=[StepGenotype(inner_edges=[(n, j) for n, j in s['inner_edges']],
                            inner_steps=list(s['inner_steps']),
                            inner_concat=list(s['inner_concat'])) for s in d['steps']],
        concat=list(d['concat']))
