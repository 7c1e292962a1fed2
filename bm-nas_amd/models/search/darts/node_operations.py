"""Fusion primitives of a step node and the gamma-weighted NodeMixedOp.

Mirror of the reference's models/search/darts/node_operations.py (STEP_STEP_OPS :9-14,
Sum :16-20, LinearGLU :22-39, ConcatFC :41-56, ScaledDotAttn :84-108, NodeMixedOp :110-120):
same class names, constructor/forward signatures and state_dict keys, but every forward
runs on the gfx950 kernels of libbmnas_hip.so (no eager-PyTorch math on the hot path).
"""
import torch
import torch.nn as nn

from bmnas.cell import Arena, Pack
from bmnas.functions import ConvBnActFn, ConvBnActThruFn, MixSumFn, NodeMixedFn, SdpaLnFn

from .genotypes import *  # noqa: F401,F403
from .genotypes import STEP_STEP_PRIMITIVES

# every node operation takes two (b, C, L) inputs and returns one (b, C, L) output
STEP_STEP_OPS = {
    'Sum': lambda C, L, args: Sum(),
    'ScaleDotAttn': lambda C, L, args: ScaledDotAttn(C, L),
    'LinearGLU': lambda C, L, args: LinearGLU(C, args),
    'ConcatFC': lambda C, L, args: ConcatFC(C, args),
}

_ONES = {}


def _ones2(device):
    t = _ONES.get(device)
    if t is None:
        t = torch.ones(2, device=device, dtype=torch.float32)
        _ONES[device] = t
    return t


class Sum(nn.Module):
    """x + y (reference :16-20), as the two-input mixed-sum kernel with unit weights."""

    def forward(self, x, y):
        return MixSumFn.apply(_ones2(x.device), x, y)


class _CatConvBn(nn.Module):
    """cat([x, y], 1) -> Conv1d(2C, M, 1) -> BatchNorm1d(M) -> act -> Dropout(args.drpt)."""
    _act = None

    def __init__(self, C, M, args):
        super().__init__()
        self.conv = nn.Conv1d(2 * C, M, 1, 1)
        self.bn = nn.BatchNorm1d(M)
        self.dropout = nn.Dropout(args.drpt)

    def forward(self, x, y):
        bn = self.bn
        return ConvBnActFn.apply(self._act, self.dropout.p, self.training, bn.running_mean,
                                 bn.running_var, bn.num_batches_tracked, self.conv.weight,
                                 self.conv.bias, bn.weight, bn.bias, x, y)

    def forward_thru(self, x, y):
        """-> (out, x', y'): the same, with the two inputs handed back for their LATER readers (bmnas.functions
        ConvBnActThruFn: those readers' gradients are then accumulated by this op's data-gradient launch, not by
        autograd `add` launches)."""
        bn = self.bn
        return ConvBnActThruFn.apply(self._act, self.dropout.p, self.training, bn.running_mean,
                                     bn.running_var, bn.num_batches_tracked, self.conv.weight,
                                     self.conv.bias, bn.weight, bn.bias, x, y)


class LinearGLU(_CatConvBn):
    """reference :22-39 (glu over the channel dim halves 2C -> C)."""
    _act = 'glu'

    def __init__(self, C, args):
        super().__init__(C, 2 * C, args)


class ConcatFC(_CatConvBn):
    """reference :41-56."""
    _act = 'relu'

    def __init__(self, C, args):
        super().__init__(C, C, args)


class ScaledDotAttn(nn.Module):
    """Scaled dot-product attention without projections (reference :84-108):
    q = x^T, k = y, v = y^T; softmax(q k / sqrt(C)) v, Dropout(0.1), LayerNorm([C, L])."""

    def __init__(self, C, L):
        super().__init__()
        self.dropout = nn.Dropout(0.1)
        self.ln = nn.LayerNorm([C, L])

    def forward(self, x, y):
        return SdpaLnFn.apply(x, y, self.ln.weight, self.ln.bias, self.dropout.p, self.training)


class NodeMixedOp(nn.Module):
    """sum_p weights[p] * op_p(x, y) over STEP_STEP_PRIMITIVES (reference :110-120).

    With the default primitive list the whole mixed op is one fused kernel sequence
    (bmnas.functions.NodeMixedFn).  To feed ONE stacked GEMM, the LinearGLU and ConcatFC
    conv / BatchNorm parameters and buffers are kept as views into stacked tensors
    (rows [0, 2C) = LinearGLU, rows [2C, 3C) = ConcatFC); names, shapes and state_dict
    keys are exactly the reference's."""

    def __init__(self, C, L, args):
        super().__init__()
        self._ops = nn.ModuleList(STEP_STEP_OPS[p](C, L, args) for p in STEP_STEP_PRIMITIVES)
        self.C, self.L = C, L
        self._default = list(STEP_STEP_PRIMITIVES) == ['Sum', 'ScaleDotAttn', 'LinearGLU', 'ConcatFC']
        self._stack = None

    # -- stacked storage ---------------------------------------------------------------
    def _stack_ok(self):
        st = self._stack
        if st is None:
            return False
        glu, cfc = self._ops[2], self._ops[3]
        C = self.C
        return (glu.conv.weight.data_ptr() == st.W.data_ptr()
                and cfc.conv.weight.data_ptr() == st.W[2 * C:].data_ptr()
                and glu.bn.running_mean.data_ptr() == st.rm.data_ptr()
                and cfc.bn.running_var.data_ptr() == st.rv[2 * C:].data_ptr()
                and glu.bn.weight.data_ptr() == st.bn_w.data_ptr()
                and cfc.conv.bias.data_ptr() == st.bias[2 * C:].data_ptr())

    @torch.no_grad()
    def _restack(self):
        glu, cfc = self._ops[2], self._ops[3]
        C = self.C
        dev = glu.conv.weight.device

        def stack(a, b, shape_a, shape_b, rows, dtype=torch.float32):
            buf = torch.empty((3 * C,) + rows, device=dev, dtype=dtype)
            buf[:2 * C].copy_(a.detach().reshape((2 * C,) + rows))
            buf[2 * C:].copy_(b.detach().reshape((C,) + rows))
            a.data = buf[:2 * C].view(shape_a)
            b.data = buf[2 * C:].view(shape_b)
            return buf

        W = stack(glu.conv.weight, cfc.conv.weight, (2 * C, 2 * C, 1), (C, 2 * C, 1), (2 * C,))
        bias = stack(glu.conv.bias, cfc.conv.bias, (2 * C,), (C,), ())
        bn_w = stack(glu.bn.weight, cfc.bn.weight, (2 * C,), (C,), ())
        bn_b = stack(glu.bn.bias, cfc.bn.bias, (2 * C,), (C,), ())
        rm = stack(glu.bn.running_mean, cfc.bn.running_mean, (2 * C,), (C,), ())
        rv = stack(glu.bn.running_var, cfc.bn.running_var, (2 * C,), (C,), ())
        nbt = torch.stack([glu.bn.num_batches_tracked.detach().to(dev),
                           cfc.bn.num_batches_tracked.detach().to(dev)])
        glu.bn.num_batches_tracked.data = nbt[0]
        cfc.bn.num_batches_tracked.data = nbt[1]
        self._stack = Pack(W=W, bias=bias, bn_w=bn_w, bn_b=bn_b, rm=rm, rv=rv, nbt=nbt)

    def pack(self):
        """Parameter pack consumed by bmnas.cell.node_mixed_fwd."""
        if not self._stack_ok():
            self._restack()
        st, attn = self._stack, self._ops[1]
        return Pack(ln_w=attn.ln.weight.detach(), ln_b=attn.ln.bias.detach(), attn_p=attn.dropout.p,
                    glu_p=self._ops[2].dropout.p, fc_p=self._ops[3].dropout.p,
                    stack_W=st.W, stack_bias=st.bias, stack_bn_w=st.bn_w, stack_bn_b=st.bn_b,
                    stack_rm=st.rm, stack_rv=st.rv, stack_nbt=st.nbt)

    def param_list(self):
        attn, glu, cfc = self._ops[1], self._ops[2], self._ops[3]
        return [attn.ln.weight, attn.ln.bias, glu.conv.weight, glu.conv.bias, glu.bn.weight, glu.bn.bias,
                cfc.conv.weight, cfc.conv.bias, cfc.bn.weight, cfc.bn.bias]

    def plan_grads(self, arena):
        C, L = self.C, self.L
        return (arena.ask(3 * C, 2 * C), arena.ask(3 * C), arena.ask(6 * C), arena.ask(C, L), arena.ask(C, L))

    def bind_grads(self, arena, h):
        return Pack(stack_dW=arena.view(h[0]), stack_dbias=arena.view(h[1]), stack_bn_grad=arena.view(h[2]),
                    dln_w=arena.view(h[3]), dln_b=arena.view(h[4]))

    def grad_pack(self, device):
        arena = Arena()
        h = self.plan_grads(arena)
        arena.alloc(device)
        return self.bind_grads(arena, h)

    def grads_in_param_order(self, G):
        C = self.C
        dW, db, bn = G.stack_dW, G.stack_dbias, G.stack_bn_grad
        return [G.dln_w, G.dln_b,
                dW[:2 * C].view(2 * C, 2 * C, 1), db[:2 * C], bn[0:2 * C], bn[3 * C:5 * C],
                dW[2 * C:].view(C, 2 * C, 1), db[2 * C:], bn[2 * C:3 * C], bn[5 * C:6 * C]]

    def forward(self, x, y, weights):
        if not self._default:
            return sum(w * op(x, y) for w, op in zip(weights, self._ops))
        w = weights if weights.device == x.device else weights.to(x.device)
        return NodeMixedFn.apply(self, self.training, x, y, w, *self.param_list())
