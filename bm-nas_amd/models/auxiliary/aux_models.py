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

from bmnas.functions import ConvBnActFn


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

    def forward(self, x):
        out = x.unsqueeze(-1)
        out = out.view(out.size(0), out.size(1), out.size(2), -1)
        out = self.pool(out).view(out.size(0), out.size(1), -1)
        out = F.interpolate(out, self.L)
        return self._tail(out)


class ReshapeInputLayer_MMIMDB(_ReshapeBase):
    """(b, C_in[, H, W]) -> adaptive max pool to (sqrt L, sqrt L) -> (b, C, L)."""

    def __init__(self, C_in, C, L, args):
        super().__init__(C_in, C, L, args)
        side = int(math.sqrt(L * 1.0))
        assert side * side == L
        self.pool = nn.AdaptiveMaxPool2d((side, side))

    def forward(self, x):
        out = x.unsqueeze(-1).unsqueeze(-1)
        out = out.view(out.size(0), out.size(1), out.size(2), -1)
        out = self.pool(out).view(out.size(0), out.size(1), -1)
        return self._tail(out)
