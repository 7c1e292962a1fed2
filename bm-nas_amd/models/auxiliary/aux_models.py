"""Input reshape layers that produce the fusion cell's (b, C, L) inputs, plus the small pooling
helpers the reference's backbones import from this module.

Mirror of the reference's models/auxiliary/aux_models.py: ReshapeInputLayer (:51-76),
ReshapeInputLayer_MMIMDB (:87-115), Identity (:8-10), GlobalPooling2D (:39-48),
GlobalPooling1D (:117-124).  The pooling is ordinary PyTorch; the Conv1d(k=1) -> BatchNorm1d ->
ReLU -> Dropout tail runs on the same gfx950 GEMM + BN kernels as ConcatFC (K = C_in up to 2048).
The MFAS-legacy cells further down the reference file are unused by BM-NAS and not mirrored.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from bmnas.functions import ConvBnActFn, PoolGroupFn, ReshapeGroupFn


class Identity(nn.Module):
    def forward(self, inputs):
        return inputs


class GlobalPooling2D(nn.Module):
    def forward(self, x):
        return x.view(x.size(0), x.size(1), -1).mean(2)


class GlobalPooling1D(nn.Module):
    def forward(self, x):
        return torch.mean(x, 2)


class _ReshapeBase(nn.Module):
    def __init__(self, C_in, C, L, args):
        super().__init__()
        self.C, self.L = C, L
        self.conv = nn.Conv1d(C_in, C, 1, 1)
        self.bn = nn.BatchNorm1d(C)
        self.dropout = nn.Dropout(args.drpt)

    def _tail(self, pooled):
        """conv -> bn -> relu -> dropout on a (b, C_in, L) tensor."""
        bn = self.bn
        if pooled.is_cuda and self.conv.in_channels % 16 == 0 and self.C % 16 == 0 and self.L in (4, 8, 16):
            return ConvBnActFn.apply('relu', self.dropout.p, self.training, bn.running_mean, bn.running_var,
                                     bn.num_batches_tracked, self.conv.weight, self.conv.bias, bn.weight,
                                     bn.bias, pooled.contiguous())
        from bmnas import lib
        lib.note_off_path(type(self).__name__, f'pooled input {tuple(pooled.shape)} on {pooled.device} '
                          f'(needs HIP device, C_in % 16 == 0, C % 16 == 0, L in 4/8/16)')
        return self.dropout(F.relu(bn(self.conv(pooled))))


class ReshapeInputLayer(_ReshapeBase):
    """(b, C_in, T, ...) -> adaptive max pool to (L, 1) over (T, rest) -> (b, C, L)."""

    def __init__(self, C_in, C, L, args):
        super().__init__(C_in, C, L, args)
        self.pool = nn.AdaptiveMaxPool2d((L, 1))

    def pooled(self, x):
        out = x.unsqueeze(-1)
        out = out.view(out.size(0), out.size(1), out.size(2), -1)
        out = self.pool(out).view(out.size(0), out.size(1), -1)
        return F.interpolate(out, self.L)

    def pool_dims(self, x):
        """(C, H, W, oh, ow) of the AdaptiveMaxPool2d that `pooled` applies to x viewed as (b, C, H, W); the
        F.interpolate(., L) after it is an identity at that size."""
        H = x.size(2) if x.dim() > 2 else 1
        return (x.size(1), H, max(1, x[0, 0].numel() // H), self.L, 1)

    def forward(self, x):
        return self._tail(self.pooled(x))


class ReshapeInputLayer_MMIMDB(_ReshapeBase):
    """(b, C_in[, H, W]) -> adaptive max pool to (sqrt L, sqrt L) -> (b, C, L)."""

    def __init__(self, C_in, C, L, args):
        super().__init__(C_in, C, L, args)
        side = int(math.sqrt(L * 1.0))
        assert side * side == L
        self.pool = nn.AdaptiveMaxPool2d((side, side))

    def pooled(self, x):
        out = x.unsqueeze(-1).unsqueeze(-1)
        out = out.view(out.size(0), out.size(1), out.size(2), -1)
        return self.pool(out).view(out.size(0), out.size(1), -1)

    def pool_dims(self, x):
        H = x.size(2) if x.dim() > 2 else 1
        side = int(math.sqrt(self.L * 1.0))
        return (x.size(1), H, max(1, x[0, 0].numel() // H), side, side)

    def forward(self, x):
        return self._tail(self.pooled(x))


def reshape_tails(layers, pooled):
    """[layer._tail(f) for layer, f in zip(layers, pooled)] — the conv -> bn -> relu -> dropout stacks of the
    reshape layers on their pooled inputs (b, C_in_i, L) — with every eligible layer of the list in ONE grouped
    set of launches (bmnas.functions.ReshapeGroupFn: 4 launches for the group instead of 5 per layer).  Layers the
    group cannot take (placeholders of a found net, CPU tensors, channel counts off the kernels' grid) run on their
    own, exactly as before."""
    from bmnas import lib
    outs = [None] * len(layers)
    idx = [i for i, (layer, f) in enumerate(zip(layers, pooled))
           if isinstance(layer, _ReshapeBase) and torch.is_tensor(f) and f.is_cuda and f.dim() == 3
           and f.dtype == torch.float32]
    grp = []
    if len(idx) >= 2:
        l0 = layers[idx[0]]
        same = [i for i in idx if layers[i].C == l0.C and layers[i].L == l0.L and layers[i].training == l0.training
                and layers[i].dropout.p == l0.dropout.p and pooled[i].shape[0] == pooled[idx[0]].shape[0]
                and pooled[i].shape[2] == l0.L and layers[i].conv.in_channels == pooled[i].shape[1]]
        same = same[:lib.MAX_GROUP]
        if len(same) >= 2 and lib.conv1x1_group_ok([layers[i].conv.in_channels for i in same],
                                                   pooled[same[0]].shape[0], l0.L, l0.C):
            grp = same
    if grp:
        ls = [layers[i] for i in grp]
        buffers = [(m.bn.running_mean, m.bn.running_var, m.bn.num_batches_tracked) for m in ls]
        params = []
        for m in ls:
            params += [m.conv.weight, m.conv.bias, m.bn.weight, m.bn.bias]
        res = ReshapeGroupFn.apply(len(grp), ls[0].dropout.p, ls[0].training, buffers, *[pooled[i] for i in grp],
                                   *params)
        for i, r in zip(grp, res):
            outs[i] = r
    for i, (layer, f) in enumerate(zip(layers, pooled)):
        if outs[i] is None:
            outs[i] = layer._tail(f) if isinstance(layer, _ReshapeBase) else layer(f)
    return outs


def reshape_all(layers, features):
    """[layer(f) for layer, f in zip(layers, features)] with the conv stacks grouped (reshape_tails)."""
    idx = [i for i, (layer, f) in enumerate(zip(layers, features))
           if isinstance(layer, _ReshapeBase) and torch.is_tensor(f) and f.is_cuda and f.dtype == torch.float32
           and f.dim() >= 2 and f.numel() > 0 and f[0, 0].numel() < (1 << 31)]
    pooled = list(features)
    if len(idx) >= 2 and len({features[i].shape[0] for i in idx}) == 1:
        # the adaptive max pools of all modalities in one launch (PoolGroupFn), at most lib.MAX_GROUP at a time
        from bmnas import lib
        for k in range(0, len(idx), lib.MAX_GROUP):
            part = idx[k:k + lib.MAX_GROUP]
            dims = [layers[i].pool_dims(features[i]) for i in part]
            res = PoolGroupFn.apply(dims, *[features[i] for i in part])
            for i, r in zip(part, res):
                pooled[i] = r
        idx = set(idx)
    else:
        idx = set()
    pooled = [f if (i in idx or not isinstance(layer, _ReshapeBase)) else layer.pooled(f)
              for i, (layer, f) in enumerate(zip(layers, pooled))]
    return reshape_tails(layers, pooled)
