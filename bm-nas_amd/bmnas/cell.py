"""Forward / backward procedures of the BM-NAS fusion cell on the HIP kernels.

No autograd here: every ``*_fwd`` returns its outputs plus a ``saved`` object, every
``*_bwd`` consumes it.  ``bmnas.functions`` wraps them into torch.autograd.Function so the
nn.Module mirror (models/search/darts/*.py) behaves like the reference's modules.

Math and reference citations: SURVEY.md Appendix A; kernel contracts: include/bmnas_hip.h.
"""
from __future__ import annotations

import os

import torch

from . import lib

ATTN_DROP = 0.1     # ScaledDotAttn's hard-coded nn.Dropout(0.1) (node_operations.py:89)

# Independent kernels of a NodeMixedOp CAN run on a second HIP stream (attention next to the conv
# GEMM in the forward, the weight-gradient GEMM next to the attention backward); the fork / join
# are stream waits, which a hipGraph capture turns into parallel branches.  Measured on MI355X
# (MM-IMDB b = 128): with the four fork/join pairs per step the replayed graph got SLOWER
# (0.364 vs 0.308 ms/step; eager 1.69 vs 1.38 ms) — the cross-stream dependencies cost more than
# the ~10 us kernels they overlap.  Off by default.
OVERLAP = False
_SIDE = {}


class _Fork:
    """with _Fork(device) as f: ... f.side(lambda: kernel launches) ...; joins on exit."""

    def __init__(self, device):
        self.main = torch.cuda.current_stream(device)
        if OVERLAP:
            key = (device.index, self.main.cuda_stream)
            s = _SIDE.get(key)
            if s is None:
                s = torch.cuda.Stream(device)
                _SIDE[key] = s
            self.stream = s
        else:
            self.stream = None
        self.used = False

    def __enter__(self):
        return self

    def side(self, fn):
        if self.stream is None:
            fn()
            return
        if not self.used:
            self.stream.wait_stream(self.main)
            self.used = True
        with torch.cuda.stream(self.stream):
            fn()

    def __exit__(self, *exc):
        if self.used:
            self.main.wait_stream(self.stream)
        return False


# ------------------------------------------------------------------------ dropout state
class _DropState:
    """Counter-based dropout bookkeeping: seed = torch.initial_seed(), offset advances with
    every dropout site so masks never repeat; the backward reuses the saved descriptor.
    Under hipGraph capture (bmnas.graph) the offsets restart at 0 inside the captured
    region and a DEVICE counter, advanced by the graph itself, is added at run time."""

    def __init__(self):
        self.offset = 0
        self.device_counter = None       # int64 cuda tensor while capturing / replaying
        self.pending_advance = None      # (counter, span) to hand to the first cell prologue of a capture
        self.record = None               # a list: every live site is appended as (descriptor, numel), in call order

    def make(self, p, numel, training):
        if not training or p <= 0.0:
            return lib.NO_DROP
        ctr = self.device_counter
        d = lib.make_dropout(p, torch.initial_seed(), self.offset,
                             None if ctr is None else ctr.data_ptr())
        self.offset += (numel + 3) // 4
        if self.record is not None and d.thr != 0:
            self.record.append((d, numel))
        return d

    def take_advance(self):
        """The captured step's counter advance, for the cell prologue launch to carry — but only while NO
        dropout site has been issued yet: a site in front of the prologue (the reshape layers' dropout) reads
        the counter before the advance in its forward and after it in its backward, i.e. would regenerate a
        mask the forward never applied.  Such a step keeps the add at the end of the graph instead."""
        if self.pending_advance is None or self.offset != 0:
            return None
        adv, self.pending_advance = self.pending_advance, None
        return adv


DROP = _DropState()


# ---- debug: |mean| / std of every BatchNorm input (BMNAS_BN_RATIO_CHECK=1) ---------------------------------------
# The batch statistics are one-pass sums of d = u - bias and d^2: the forward error follows
# ~2 eps r^2 with r = |E[d]| / std(d), pinned to r ~ 27 for the 1e-4 parity bound (tests/test_numerics_gpu.py).  Real
# pre-BatchNorm activations sit at r ~ 1-3, but nothing at run time noticed a layer far outside — the reshape layers
# see post-ReLU backbone features with up to 2048 input channels.  With the switch on, every training-mode BatchNorm of
# the path is noted when launched and evaluated at the next note / bn_ratio_report(): one extra reduction and a host
# read per layer (a debug mode: never inside a hipGraph capture, where nothing can be read back).
BN_RATIO_CHECK = os.environ.get('BMNAS_BN_RATIO_CHECK', '0') not in ('0', '', 'false', 'False')
BN_RATIO_WARN = 20.0
BN_RATIOS = {}                  # where -> the largest r seen
_BN_PENDING = []


def bn_ratio_note(chan, bias, M, where, training=True):
    if not BN_RATIO_CHECK or not training or torch.cuda.is_current_stream_capturing():
        return
    bn_ratio_report()
    _BN_PENDING.append((chan, bias, M, where))


def bn_ratio_report():
    """Evaluate the noted layers (their `chan` = mean | rstd | scale | shift is complete once the kernel that applies the
    BatchNorm has run — it has, by the time the next layer is noted) -> {where: r}; warns above BN_RATIO_WARN."""
    import warnings
    while _BN_PENDING:
        chan, bias, M, where = _BN_PENDING.pop(0)
        mean, rstd = chan[:M], chan[M:2 * M]
        d = mean if bias is None else mean - bias.reshape(-1)[:M]
        r = float((d.abs() * rstd).max())
        BN_RATIOS[where] = max(BN_RATIOS.get(where, 0.0), r)
        if r > BN_RATIO_WARN:
            warnings.warn(f'bmnas: BatchNorm input of {where}: |mean| / std = {r:.1f} > {BN_RATIO_WARN:g} — the one-pass '
                          f'batch variance loses ~{2.5e-7 * r * r:.1e} of relative precision here (DESIGN.md section 1)',
                          RuntimeWarning)
    return dict(BN_RATIOS)


def _empty(like, *shape):
    return torch.empty(shape, device=like.device, dtype=torch.float32)


def _zeros(like, *shape):
    return torch.zeros(shape, device=like.device, dtype=torch.float32)


class Arena:
    """One zero-filled fp32 buffer carved into the gradient tensors that the kernels
    accumulate into with atomics (a single memset instead of one per tensor)."""

    def __init__(self):
        self.req = []
        self.total = 0
        self.buf = None

    def ask(self, *shape):
        n = 1
        for d in shape:
            n *= int(d)
        self.req.append((self.total, n, tuple(shape)))
        self.total += (n + 3) // 4 * 4
        return len(self.req) - 1

    def alloc(self, device, zero=True):
        """zero=False: the caller has the first kernel of its backward clear the buffer
        (the `scrub` side job of bmnas_cat_ln_bwd) instead of paying a memset launch."""
        self.buf = (torch.zeros if zero else torch.empty)(self.total, device=device, dtype=torch.float32)
        return self

    def view(self, idx):
        off, n, shape = self.req[idx]
        return self.buf[off:off + n].view(shape)


class Pack:
    """Plain attribute bag for parameter / gradient packs."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class HeadState(Pack):
    """What the fused head (csrc/head.hip) shares between the cell's forward, the criterion and the
    backward: classifier weights, the zero-filled hb = logits | A | B buffer, the loss scalar, and —
    when the criterion was deferred into the backward launch (bmnas.nn.fused_criterion) — its kind
    and labels.  `marker` stands in for dlogits in that case: the backward recognises it by address."""

    deferred = None      # ('bce' | 'ce', labels)
    gscale = None        # 0-dim device tensor multiplying dlogits (None: 1)

    def resolve(self, g):
        """-> (mode, dlogits, gscale, labels) for bmnas_head_bwd given the incoming gradient g."""
        if self.deferred is not None:
            if g.data_ptr() != self.marker.data_ptr():
                # autograd summed the (uninitialised) marker with another gradient of the logits: a second
                # loss term on them inside bmnas.nn.fused_criterion().  There is no dlogits tensor to add to.
                raise lib.BmnasError('fused criterion: the logits received a gradient besides the deferred '
                                     'criterion\'s (an auxiliary loss on the logits?); evaluate the losses outside '
                                     'bmnas.nn.fused_criterion() or combine them into one criterion call')
            kind, labels = self.deferred
            return (1 if kind == 'bce' else 2), None, self.gscale, labels
        return 0, (g if g.is_contiguous() else g.contiguous()), None, None


LAST_HEAD = []           # FusedCellFn.forward leaves its HeadState here for the caller to pick up


class GradSlot:
    """A lazily allocated gradient buffer that remembers whether it has been written
    (first writer overwrites, later writers accumulate)."""
    __slots__ = ('t', 'like', 'written', 'extra')

    def __init__(self, like):
        self.like = like
        self.t = None
        self.written = False
        self.extra = None     # a second addend of this gradient, produced concurrently (merged launches)

    def buf(self):
        if self.t is None:
            self.t = torch.empty_like(self.like)
        return self.t

    def acc_bit(self):
        """Returns 1 if the kernel must accumulate, then marks the slot written."""
        a = 1 if self.written else 0
        self.written = True
        return a

    def get(self):
        if self.t is None or not self.written:
            return None
        return self.t


def _write_group(slots):
    """(buffers, accumulate_mask) for a kernel writing slots[j] (None = skip).  The same slot
    may appear several times only for kernels that document in-order read-modify-write
    (mixsum_bwd); it is then overwritten by its first occurrence and accumulated after."""
    bufs, mask = [], 0
    for j, s in enumerate(slots):
        if s is None:
            bufs.append(None)
            continue
        bufs.append(s.buf())
        if s.acc_bit():
            mask |= (1 << j)
    return bufs, mask


# ---------------------------------------------------------------------------- K1 mixsum
def mixsum_fwd(xs, w_row0, w_stride=2):
    """out = sum_j w_j xs[j]; w_row0 = view whose data_ptr is edge 0's weight."""
    out = torch.empty_like(xs[0])
    lib.mixsum_fwd(xs, w_row0, w_stride, out)
    return out


ARCH_SHARDS = 16    # copies of the arch-weight gradient buffers (atomic contention spreading)


def mixsum_bwd(xs, slots, w_row0, g, dw_row0, w_stride=2, shards=1, shard_stride=0, g2=None):
    bufs, mask = _write_group(slots)
    lib.mixsum_bwd(xs, bufs, w_row0, w_stride, g, dw_row0, mask, shards, shard_stride, g2)


# -------------------------------------------------------------------- conv + BatchNorm
class ConvBnSaved:
    __slots__ = ('srcs', 'C_src', 'W', 'ldw', 'U', 'chan', 'M', 'training', 'dup', 'fold', 'fin')


STAT_SHARDS = 4     # copies of the BatchNorm batch-sum buffers (same-address atomics serialise; 1 / 2 / 4 / 8 measured in round 2)


class StatArena:
    """The BatchNorm batch-sum buffers of one cell forward, carved out of ONE fp32 buffer that the
    cell prologue launch zero-fills (bmnas_cell_prologue's scrub): the forward GEMMs add their
    per-channel sums into them with atomics and the kernels that apply the BatchNorm finalise them
    in place of a bmnas_bn_finalize launch per conv."""

    def __init__(self, like, channel_counts, buf=None):
        self.sizes = [STAT_SHARDS * M * 2 for M in channel_counts]
        self.buf = torch.empty(sum(self.sizes), device=like.device, dtype=torch.float32) if buf is None else buf
        assert self.buf.numel() >= sum(self.sizes)
        self.off = 0

    @staticmethod
    def numel_for(channel_counts):
        return sum(STAT_SHARDS * M * 2 for M in channel_counts)

    def take(self, M):
        n = STAT_SHARDS * M * 2
        if self.off + n > self.buf.numel():
            raise lib.BmnasError('StatArena: more BatchNorms in the forward than were planned')
        v = self.buf[self.off:self.off + n]
        self.off += n
        return v


def conv_bn_fwd(srcs, C_src, W, ldw, bias, bn_w, bn_b, rm, rv, nbt, training, dup=0, fold=0, attn=None,
                stats=None):
    """U = conv1x1(cat(srcs)) (+ batch statistics) and the fused BN affine `chan`.
    W is (M, ldw) row-major with the first len(srcs)*C_src columns used.
    stats: a StatArena -> the statistics are finalised by the CONSUMER kernel (sv.fin is the
    descriptor to hand to it; `chan` is written by that kernel); None -> bmnas_bn_finalize here."""
    x0 = srcs[0]
    b, L = x0.shape[0], x0.shape[2]
    M = bn_w.numel()
    U = _empty(x0, b, M, L)
    part, n_part, shards = None, 0, 0
    if training:
        if b * L < 2:
            raise ValueError('Expected more than 1 value per channel when training, got input size '
                             f'{[b, M, L]}')            # same refusal as nn.BatchNorm1d
        if stats is not None:
            part, shards = stats.take(M), STAT_SHARDS
        else:
            n_part = lib.conv1x1_num_partials(b, L)
            part = _empty(x0, n_part * M * 2)
    if attn is None:
        lib.conv1x1_fwd(srcs, C_src, W, ldw, bias, U, part, b, L, M, fold, shards)
    else:                                    # the attention branch rides in the same launch
        lib.conv1x1_fwd_sdpa(srcs, C_src, W, ldw, bias, U, part, b, L, M, fold, *attn, shards)
    chan = _empty(x0, 4 * M)
    sv = ConvBnSaved()
    if stats is not None:
        sv.fin = lib.make_bn_fin(part, shards, bias, bn_w, bn_b, rm, rv, nbt, training)
    else:
        sv.fin = lib.NO_FIN
        lib.bn_finalize(part, n_part, b, L, M, bn_w, bn_b, rm, rv, nbt, training, chan)
    sv.srcs, sv.C_src, sv.W, sv.ldw, sv.U, sv.chan, sv.M = list(srcs), C_src, W, ldw, U, chan, M
    sv.training, sv.dup, sv.fold = training, dup, fold
    bn_ratio_note(chan, bias, M, f'conv {len(srcs)}x{C_src}->{M}', training)
    return U, chan, sv


def conv_bn_bwd(sv, dV, bn_grad, src_slots, dW, dbias, fork=None, attn=None, mix=None):
    """dV (gradient w.r.t. the BN output, with bn_grad already reduced) -> in place dU;
    then data gradient into src_slots and weight/bias gradient (+=) into dW / dbias."""
    b, L = sv.U.shape[0], sv.U.shape[2]
    if not WANT_PARAM_GRADS:
        dW = dbias = None                     # (the merged launch then carries no weight-gradient tiles)
    # search mode: the merged backward launch applies the BatchNorm input gradient while it stages its
    # operands (bmnas_conv1x1_bwd_all_sdpa, bn_U) — no launch in between
    fold_bn = attn is not None and FUSE_BN_APPLY
    # a conv with no attention beside it (out_conv): BatchNorm apply + both gradients behind one C-ABI
    # call, which is one launch at small grids
    live = [s for s in src_slots if s is not None]
    pair = (attn is None and dW is not None and fork is None and FUSE_BWD_PAIR
            and len({id(s) for s in live}) == len(live))
    if not fold_bn and not pair:
        lib.bn_bwd_apply(dV, sv.U, sv.chan, bn_grad, b, sv.M, L, sv.training)
    # destinations that alias each other inside ONE data-gradient launch would race:
    # give later duplicates a scratch buffer and add it afterwards (rare: node_multiplier
    # reaching back to the duplicated x/y inputs).
    seen, extra, slots = {}, [], []
    for q, s in enumerate(src_slots):
        if s is not None and id(s) in seen:
            tmp = GradSlot(s.like)
            extra.append((s, tmp))
            slots.append(tmp)
        else:
            if s is not None:
                seen[id(s)] = q
            slots.append(s)
    bufs, mask = _write_group(slots)
    if pair:
        lib.conv1x1_bwd_all(dV, sv.W, sv.ldw, bufs, sv.C_src, mask, b, L, sv.M, sv.fold, sv.srcs, dW,
                            dW.shape[1], dbias, sv.dup, (sv.U, sv.chan, bn_grad, sv.training), mix)
        return
    assert mix is None, 'mix epilogue: only with the one-launch out_conv backward'
    if attn is not None:
        # data gradient, weight gradient (if wanted) and the attention backward share one grid
        lib.conv1x1_bwd_all_sdpa(dV, sv.W, sv.ldw, bufs, sv.C_src, mask, b, L, sv.M, sv.fold, sv.srcs, dW,
                                 0 if dW is None else dW.shape[1], dbias, sv.dup, *attn,
                                 (sv.U, sv.chan, bn_grad, sv.training) if fold_bn else None)
        for s, tmp in extra:
            s.buf().add_(tmp.buf())
        return
    assert attn is None, 'an attention branch beside the conv needs the weight-gradient buffers (one merged launch)'
    if any(x is not None for x in bufs):
        lib.conv1x1_bwd_data(dV, sv.W, sv.ldw, bufs, sv.C_src, mask, b, L, sv.M, sv.fold)
    for s, tmp in extra:
        s.buf().add_(tmp.buf())
    if dW is not None:
        def weight_grad():
            lib.conv1x1_bwd_weight(dV, sv.srcs, sv.C_src, dW, dW.shape[1], dbias, sv.dup, b, L, sv.M)
        if fork is not None:
            fork.side(weight_grad)           # independent of everything that follows on the main stream
        else:
            weight_grad()


# -------------------------------------------------------------- search-mode NodeMixedOp
class MixedSaved:
    pass


def node_mixed_fwd(x, y, gamma_row, P, training, ln=None, Weff=None, stats=None, nxt=None, launch_mix=True):
    """NodeMixedOp.forward (node_operations.py:118-120).  P: parameter pack of one NodeMixedOp
    (see models.search.darts.node_operations.NodeMixedOp.pack()).  x may be y (search).
    ln = (resid, ln_w, ln_b, stats): fuse the NodeCell tail `out += x; ln(out)` (node_search.py:67-68)
    into the mix kernel; returns LN(mix + resid) and keeps the pre-norm sum in sv.pre."""
    b, C, L = x.shape
    same = x is y or x.data_ptr() == y.data_ptr()
    sv = MixedSaved()
    sv.x, sv.y, sv.same, sv.gamma, sv.P, sv.training = x, y, same, gamma_row, P, training
    # attention branch
    sv.d_attn = DROP.make(P.attn_p, x.numel(), training)
    p1 = torch.empty_like(x)
    sv.xhat1 = torch.empty_like(x)
    sv.stats1 = _empty(x, b * 2)
    sv.p1 = p1
    sv.d_glu = DROP.make(P.glu_p, x.numel(), training)
    sv.d_fc = DROP.make(P.fc_p, x.numel(), training)
    sv.merged = same and FUSE_ATTN_GEMM
    if sv.merged:
        U, chan = _mixed_conv_fwd(sv, x, y, same, P, training, C,
                                  attn=(x, y, P.ln_w, P.ln_b, p1, sv.xhat1, sv.stats1, C, sv.d_attn), Weff=Weff,
                                  stats=stats)
    else:
        with _Fork(x.device) as fork:
            fork.side(lambda: lib.sdpa_ln_fwd(x, y, P.ln_w, P.ln_b, p1, sv.xhat1, sv.stats1, b, C, L, sv.d_attn))
            U, chan = _mixed_conv_fwd(sv, x, y, same, P, training, C, Weff=Weff, stats=stats)
    out = torch.empty_like(x)
    fin = sv.conv.fin                        # BatchNorm finalised inside the mix kernel (or NO_FIN)
    if not launch_mix:                       # the caller's out_conv launch forms `out` (bmnas_node_mix_conv_fwd)
        sv.pending_mix = (x, y, p1, U, chan, gamma_row, out, sv.d_glu, sv.d_fc, fin)
        return out, sv
    if ln is None:
        lib.node_mix_fwd(x, y, p1, U, chan, gamma_row, out, b, C, L, sv.d_glu, sv.d_fc, fin, nxt)
    else:
        resid, ln_w, ln_b, ln_stats, out_sums, lazy = ln
        sv.pre = torch.empty_like(x)
        if lazy:
            # streaming producer: pre + moment records; `out` is formed by its first consumer (if any)
            P = lib.lazy_ln_parts(C, L)
            rec, prm = _empty(x, b * P * 8), _empty(x, P * 8)
            lib.node_mix_pre_fwd(x, y, p1, U, chan, gamma_row, resid, ln_w, ln_b, sv.pre, rec, prm, b, C, L,
                                 sv.d_glu, sv.d_fc, fin)
            sv.lazy = LazyNode(desc=lib.make_lazy(sv.pre, rec, prm, ln_w, ln_b, ln_stats), out=out, P=P,
                               rec=rec, prm=prm)
        else:
            lib.node_mix_ln_fwd(x, y, p1, U, chan, gamma_row, resid, ln_w, ln_b, sv.pre, out, ln_stats, b, C, L,
                                sv.d_glu, sv.d_fc, fin, out_sums)
    return out, sv


# search mode: attention branch and conv GEMM share a launch (fwd and bwd); BMNAS_FUSE_ATTN_GEMM=0 for A/B runs
FUSE_ATTN_GEMM = os.environ.get('BMNAS_FUSE_ATTN_GEMM', '1') != '0'
# arch softmaxes + folded conv weights of a cell in one launch
FUSE_PROLOGUE = os.environ.get('BMNAS_FUSE_PROLOGUE', '1') != '0'
# LayerNorm affine gradients + arch-softmax backward in one launch at the end of a cell's backward
FUSE_EPILOGUE = os.environ.get('BMNAS_FUSE_EPILOGUE', '1') != '0'
# BatchNorm statistics accumulated by the GEMM epilogues (atomics) and finalised inside the kernel
# that applies the BatchNorm, instead of one bn_finalize launch per conv (needs FUSE_PROLOGUE)
FUSE_BN_FINALIZE = os.environ.get('BMNAS_FUSE_BN_FINALIZE', '1') != '0'
# BatchNorm input gradient applied inside the merged backward GEMM launch (no bn_bwd_apply launch)
FUSE_BN_APPLY = os.environ.get('BMNAS_FUSE_BN_APPLY', '1') != '0'
# small grids, node_multiplier != 1: the last inner step's mix as the producer of out_conv's last operand
FUSE_MIX_GEMM = os.environ.get('BMNAS_FUSE_MIX_GEMM', '1') != '0'
# out_conv backward (no attention beside it): BatchNorm apply + data + weight gradient as one launch at small grids
FUSE_BWD_PAIR = os.environ.get('BMNAS_FUSE_BWD_PAIR', '1') != '0'
# NodeCell tail with node_multiplier != 1: BatchNorm + ReLU + dropout + residual + LayerNorm as one launch per
# direction (one workgroup per sample; above BN_TAIL_MAX_B samples the per-sample BatchNorm atomics of its
# backward would serialise, and the streaming kernels take over)
FUSE_BN_TAIL = os.environ.get('BMNAS_FUSE_BN_TAIL', '1') != '0'
BN_TAIL_MAX_B = 128
# the next inner step's mixed sum (and its backward) inside the previous step's mix launch
FUSE_INNER_SUM = os.environ.get('BMNAS_FUSE_INNER_SUM', '1') != '0'
MIX_PREV_MAX = 5
# the cell prologue inside the launch of the first step's pair sum (needs FUSE_PROLOGUE and FUSE_PAIR)
FUSE_PROLOGUE_PAIR = os.environ.get('BMNAS_FUSE_PROLOGUE_PAIR', '1') != '0'
# node_multiplier == 1, <= 128 samples: the node's LayerNorm backward inside the mix-backward launch
FUSE_LN_BWD = os.environ.get('BMNAS_FUSE_LN_BWD', '1') != '0'
# small batches, node_multiplier != 1: the next cell step's K1 pair sum inside the node's tail launch
FUSE_NEXT_PAIR = os.environ.get('BMNAS_FUSE_NEXT_PAIR', '1') != '0'
# small grids, node_multiplier != 1: the last inner step's mix backward as the epilogue of the out_conv
# data-gradient tiles (needs the one-launch out_conv backward, FUSE_BWD_PAIR)
FUSE_MIX_EPILOGUE = os.environ.get('BMNAS_FUSE_MIX_EPILOGUE', '1') != '0'
# the cell's K7 tail + central classifier (+ criterion) as two launches (csrc/head.hip)
FUSE_HEAD = os.environ.get('BMNAS_FUSE_HEAD', '1') != '0'
# node_multiplier == 1 under the fused head: the step node's LayerNorm is applied by its consumers (the next step's K1
# pair sum, the head) and differentiated from per-workgroup partial sums — streaming grids instead of one workgroup
# per sample in both directions (csrc/lazyln.hip)
LAZY_LN = os.environ.get('BMNAS_LAZY_LN', '1') != '0'
# with it: the cell-input gradients written once by the first step's K1 backward from every step's stored G
WRITE_ONCE = os.environ.get('BMNAS_WRITE_ONCE', '1') != '0'


# BMNAS_STRICT_ZERO=1 (debug): the reference's 'none' primitive is Zero(x) = x.mul(0.) (operations.py:18-20), so a NaN
# or Inf in a cell input reaches EVERY step's mixed sum as NaN through w_none * (x * 0.) (model_search.py:58) — the
# kernels drop that term (finite inputs: exactly 0).  With the switch the cell adds sum_j (x_j * 0.) of its inputs to every
# step's sum (two extra torch launches per step) and restores the per-sample NaN that the kernels' ReLU (`v > 0 ? v : 0`)
# would clear at the cell's output: the non-finite pattern of the cell output / logits then equals the reference's.
# Forward only — for finite values the terms and their gradients are zero.  (L = 16: exact per sample.  L = 8 / 4: samples
# that share a 16-column attention tile with a non-finite one turn NaN as well — the tile's block-diagonal mask multiplies.)
STRICT_ZERO = os.environ.get('BMNAS_STRICT_ZERO', '0') not in ('0', '', 'false', 'False')


# Run-to-run bit-identical results (the reference's CPU path is deterministic; the default kernels sum batch reductions
# with fp32 atomics in whatever order workgroups finish).  Covers the search step with node_multiplier == 1 under the
# fused head (the lazy-LayerNorm path): BatchNorm statistics as per-n-group partials + bn_finalize, head logits and
# criterion / BatchNorm affine gradients as partials summed in order, one arch-gradient shard per workgroup, weight
# gradients without batch splits, LayerNorm affine reductions in one chunk (include/bmnas_hip.h, "deterministic mode").
DETERMINISTIC = os.environ.get('BMNAS_DETERMINISTIC', '0') not in ('0', '', 'false', 'False')
_DET_APPLIED = [None]


def apply_deterministic():
    """Hand the current DETERMINISTIC setting to the library (its host-side launch choices); cheap when unchanged."""
    if _DET_APPLIED[0] != DETERMINISTIC:
        lib.set_deterministic(DETERMINISTIC)
        _DET_APPLIED[0] = DETERMINISTIC


def arch_shards(b, C, L):
    """Copies of the arch-gradient buffers: ARCH_SHARDS to spread same-address atomics — or, deterministic mode, one per
    workgroup of the largest launch that adds into them (a single add per address; the epilogue sums in shard order)."""
    if not DETERMINISTIC:
        return ARCH_SHARDS
    cl4 = C * L // 4
    parts = (cl4 + 255) // 256
    streaming = min(2048, (b * cl4 + 255) // 256)
    mix = ((cl4 + 63) // 64) * lib.node_mix_lnp_bwd_rows(b)
    return max(ARCH_SHARDS, b * parts, streaming, mix)


class LazyNode(Pack):
    """A step-node output whose LayerNorm is pending: desc (lib.LazyLn over pre / rec / prm / affine / stats),
    out (the (b, C, L) tensor the first K1 consumer fills; never written for the last node), P parts per sample;
    backward: lnp_head (b, C L / 64, 2) and lnp_k1 (b, k1_n P, 2) partial sums of the LayerNorm backward."""
    materialised = False
    lnp_head = None
    lnp_k1 = None
    k1_n = 0


def _mixed_conv_fwd(sv, x, y, same, P, training, C, attn=None, Weff=None, stats=None):
    # stacked [LinearGLU | ConcatFC] conv + BN
    if same:
        # conv(cat[z, z]) = (W[:, :C] + W[:, C:]) z: K is C instead of 2C.  The halves are added once
        # into a folded copy (2.3 us); letting the GEMMs add them while fetching their operand
        # (fold_cols of the C ABI) was measured slower: +4 us per GEMM for the doubled weight loads.
        if Weff is None:                     # (the fused cell folds every node's weights up front)
            Weff = _empty(x, 3 * C, C)
            lib.fold_weight(P.stack_W, Weff, 3 * C, C)
        U, chan, sv.conv = conv_bn_fwd([x], C, Weff, C, P.stack_bias, P.stack_bn_w, P.stack_bn_b,
                                       P.stack_rm, P.stack_rv, P.stack_nbt, training, dup=C, attn=attn,
                                       stats=stats)
    else:
        U, chan, sv.conv = conv_bn_fwd([x, y], C, P.stack_W, 2 * C, P.stack_bias, P.stack_bn_w,
                                       P.stack_bn_b, P.stack_rm, P.stack_rv, P.stack_nbt, training,
                                       stats=stats)
    return U, chan


class Deferred:
    """LayerNorm affine-gradient problems collected during a backward pass and flushed in
    ONE launch at its end (they feed nothing downstream)."""

    def __init__(self):
        self.probs = []

    def add(self, **p):
        self.probs.append(p)

    def flush(self, b, L):
        for i in range(0, len(self.probs), 8):
            lib.ln_affine_bwd_multi(self.probs[i:i + 8], b, L)
        self.probs = []


def _ln_affine(deferred, g, gscale, srcs, resid, ln_w, ln_b, stats, dln_w, dln_b, b, C, L, relu, prenorm):
    if not WANT_PARAM_GRADS:
        return
    if deferred is None:
        lib.ln_affine_bwd(g, gscale, srcs, resid, ln_w, ln_b, stats, dln_w, dln_b, b, C, L, relu, prenorm)
    else:
        deferred.add(g=g, gscale=gscale, srcs=list(srcs), resid=resid, ln_w=ln_w, ln_b=ln_b, stats=stats,
                     dln_w=dln_w, dln_b=dln_b, C=C, relu=relu, prenorm=prenorm)


def _attn_affine_bwd(sv, g, G, deferred=None):
    b, C, L = sv.x.shape
    _ln_affine(deferred, g, sv.gamma[1:2], [sv.xhat1], None, None, None, None, G.dln_w, G.dln_b, b, C, L,
               False, True)


def node_mixed_bwd(sv, g, dgamma_row, x_slot, y_slot, G, shards=1, shard_stride=0, deferred=None, nxt=None,
                   ln=None, pre_done=None):
    """g: grad of the mixed output (None with nxt: nothing accumulated yet).  nxt: see lib.node_mix_bwd.
    dgamma_row (4 floats, +=), x_slot / y_slot: GradSlots (y_slot None when x is y).  G: gradient pack (stack_dW, stack_dbias, stack_bn_grad,
    dln_w, dln_b), all += .
    ln = (gy, pre, ln_w, stats, resid_slot, g_slot): the mixed output went through `+ resid -> LayerNorm`
    (node_multiplier == 1) and gy is the gradient of THAT output; the launch does the LayerNorm backward
    first (bmnas_node_mix_ln_bwd), writes its input gradient to g_slot and resid_slot.  x is y only."""
    x, y = sv.x, sv.y
    b, C, L = x.shape
    M = 3 * C
    dV = _empty(x, b, M, L) if pre_done is None else pre_done[0]
    bn_grad = G.stack_bn_grad            # [dW_bn (3C) | dB_bn (3C)], zero-initialised by caller
    with _Fork(x.device) as fork:
        if sv.same and pre_done is not None:
            # the mix backward already ran as the epilogue of the out_conv data-gradient tiles
            # (bmnas_conv1x1_bwd_all_mix): dV, bn_grad, dgamma and x_slot are done; the contractions remain
            dxb = pre_done[1]
            if sv.merged and x_slot.extra is None:
                x_slot.extra = torch.empty_like(x)
                conv_bn_bwd(sv.conv, dV, bn_grad, [x_slot], G.stack_dW, G.stack_dbias, fork,
                            attn=(g, sv.gamma[1:2], x, y, sv.P.ln_w, sv.xhat1, sv.stats1, x_slot.extra, None,
                                  0, C, sv.d_attn))
            else:
                conv_bn_bwd(sv.conv, dV, bn_grad, [x_slot], G.stack_dW, G.stack_dbias, fork)
                lib.sdpa_ln_bwd(g, sv.gamma[1:2], x, y, sv.P.ln_w, sv.xhat1, sv.stats1, dxb, None, 1, b, C, L,
                                sv.d_attn)
        elif sv.same:
            if ln is not None:
                gy, pre, ln_w, stats, r_slot, g_slot, lazy = ln
                g = g_slot.buf()
                g_slot.written = True
                rbuf, racc = r_slot.buf(), r_slot.acc_bit()
            dxb, acc = x_slot.buf(), x_slot.acc_bit()
            if ln is not None and lazy is not None:
                bn_part = _empty(x, lib.node_mix_lnp_bwd_rows(b) * 6 * C) if DETERMINISTIC else None
                lib.node_mix_lnp_bwd(gy, pre, ln_w, stats, lazy.lnp_head, lazy.lnp_k1, g, rbuf, racc, x, y, sv.p1,
                                     sv.conv.U, sv.conv.chan, sv.gamma, dgamma_row, dxb, None, acc, dV, bn_grad,
                                     b, C, L, sv.d_glu, sv.d_fc, shards, shard_stride, bn_part)
            elif ln is not None:
                lib.node_mix_ln_bwd(gy, pre, ln_w, stats, g, rbuf, racc, x, y, sv.p1, sv.conv.U, sv.conv.chan,
                                    sv.gamma, dgamma_row, dxb, None, acc, dV, bn_grad, b, C, L, sv.d_glu,
                                    sv.d_fc, shards, shard_stride)
            else:
                lib.node_mix_bwd(g, x, y, sv.p1, sv.conv.U, sv.conv.chan, sv.gamma, dgamma_row, dxb, None, acc,
                                 dV, bn_grad, b, C, L, sv.d_glu, sv.d_fc, shards, shard_stride, nxt)
            if nxt is not None:
                g = nxt[-1]                  # the launch completed this step's gradient there
            if sv.merged and x_slot.extra is None:
                # the attention gradient is produced next to the data-gradient GEMM, into its own
                # buffer; whoever consumes x_slot adds the two parts (g2 / gz2 of the K1 backward)
                x_slot.extra = torch.empty_like(x)
                conv_bn_bwd(sv.conv, dV, bn_grad, [x_slot], G.stack_dW, G.stack_dbias, fork,
                            attn=(g, sv.gamma[1:2], x, y, sv.P.ln_w, sv.xhat1, sv.stats1, x_slot.extra, None,
                                  0, C, sv.d_attn))
            else:
                conv_bn_bwd(sv.conv, dV, bn_grad, [x_slot], G.stack_dW, G.stack_dbias, fork)
                lib.sdpa_ln_bwd(g, sv.gamma[1:2], x, y, sv.P.ln_w, sv.xhat1, sv.stats1, dxb, None, 1, b, C, L,
                                sv.d_attn)
        else:
            dxb, dyb = x_slot.buf(), y_slot.buf()
            acc = x_slot.acc_bit() | (y_slot.acc_bit() << 1)
            lib.node_mix_bwd(g, x, y, sv.p1, sv.conv.U, sv.conv.chan, sv.gamma, dgamma_row, dxb, dyb, acc,
                             dV, bn_grad, b, C, L, sv.d_glu, sv.d_fc, shards, shard_stride, nxt)
            if nxt is not None:
                g = nxt[-1]
            conv_bn_bwd(sv.conv, dV, bn_grad, [x_slot, y_slot], G.stack_dW, G.stack_dbias, fork)
            lib.sdpa_ln_bwd(g, sv.gamma[1:2], x, y, sv.P.ln_w, sv.xhat1, sv.stats1, dxb, dyb, 3, b, C, L,
                            sv.d_attn)
    _attn_affine_bwd(sv, g, G, deferred)


# ------------------------------------------------------------------- search-mode NodeCell
class NodeCellSaved:
    pass


FUSE_TAIL = True   # node_multiplier == 1: NodeMixedOp + residual + LayerNorm in one launch


FUSE_PAIR = True   # search mode: cell-level mixed sum + the node's first inner sum in one launch


def _mix_conv_fwd(pending, tail, C, Wo, ldw, NP, training, stats):
    """out_conv over cat(tail) whose LAST tensor is the mix output that `pending` describes and nobody has formed
    yet: mix + GEMM in one launch (bmnas_node_mix_conv_fwd).  Returns what conv_bn_fwd would (the BatchNorm is
    finalised by the consumer, like every conv of the fused cell)."""
    x, y, p1, U, chan_mix, gamma_row, out, d_glu, d_fc, fin = pending
    assert out is tail[-1]
    b, L = x.shape[0], x.shape[2]
    V = _empty(x, b, C, L)
    part, shards = None, 0
    if training:
        if b * L < 2:
            raise ValueError('Expected more than 1 value per channel when training, got input size '
                             f'{[b, C, L]}')
        part, shards = stats.take(C), STAT_SHARDS
    lib.node_mix_conv_fwd(x, y, p1, U, chan_mix, gamma_row, out, d_glu, d_fc, fin, tail[:-1], Wo, ldw,
                          NP.out_conv_b, V, part, shards, b, C, L)
    sv = ConvBnSaved()
    sv.fin = lib.make_bn_fin(part, shards, NP.out_conv_b, NP.bn_w, NP.bn_b, NP.bn_rm, NP.bn_rv, NP.bn_nbt, training)
    sv.srcs, sv.C_src, sv.W, sv.ldw, sv.U, sv.chan, sv.M = list(tail), C, Wo, ldw, V, _empty(x, 4 * C), C
    sv.training, sv.dup, sv.fold = training, 0, 0
    bn_ratio_note(sv.chan, NP.out_conv_b, C, f'out_conv {len(tail)}x{C}->{C}', training)
    return V, sv.chan, sv


def node_cell_fwd(x, y, beta_w, gamma_w, NP, training, ns, nm, z0=None, weffs=None, stats=None,
                  want_sums=False, next_pair=None, lazy=False):
    """NodeCell.forward (node_search.py:48-70).  beta_w (k_in, 2), gamma_w (ns, 4): softmaxed
    device tensors.  NP: parameter pack of the NodeCell.  z0: the first inner mixed sum when the
    caller already formed it (bmnas_mixsum_pair_fwd)."""
    b, C, L = x.shape
    sv = NodeCellSaved()
    sv.x, sv.ns, sv.nm, sv.NP, sv.training = x, ns, nm, NP, training
    sv.beta_w, sv.gamma_w = beta_w, gamma_w
    states = [x, y]
    sv.zs, sv.mixed, sv.offsets = [], [], []
    offset = 0
    sv.fused_tail = nm == 1 and FUSE_TAIL
    sv.next_pair_done = False
    sv.stats = _empty(x, b * 2)
    # per-sample (sum, sum of squares) of the node output, for the head's K7 LayerNorm (head.hip)
    sv.osum = _empty(x, b * 2) if want_sums else None
    sv.next_fused = []
    z_next = None
    for t in range(ns):
        if z_next is not None:
            z = z_next                                   # formed by the previous step's mix launch
        else:
            z = z0 if (t == 0 and z0 is not None) else mixsum_fwd(states, beta_w[offset:, 1])
        last = sv.fused_tail and t == ns - 1
        nxt, z_next = None, None
        if FUSE_INNER_SUM and t + 1 < ns and len(states) <= MIX_PREV_MAX:
            z_next = torch.empty_like(x)
            nxt = (list(states), beta_w[offset + len(states):, 1], 2, z_next)
        sv.next_fused.append(nxt is not None)
        # the last inner step's mix rides in the out_conv launch below (small grids)
        defer = (FUSE_MIX_GEMM and nm != 1 and t == ns - 1 and stats is not None and x.is_cuda
                 and lib.node_mix_conv_fwd_ok(b, C, L, nm - 1))
        s, msv = node_mixed_fwd(z, z, gamma_w[t], NP.mixed[t], training,
                                (x, NP.ln_w, NP.ln_b, sv.stats, sv.osum, lazy) if last else None,
                                None if weffs is None else weffs[t], stats, nxt, launch_mix=not defer)
        sv.zs.append(z)
        sv.mixed.append(msv)
        sv.offsets.append(offset)
        offset += len(states)
        states.append(s)
    sv.states = states
    sv.lazy = None
    if sv.fused_tail:
        sv.o = sv.mixed[-1].pre                          # pre-norm sum o + x; states[-1] is LN(o + x)
        sv.lazy = getattr(sv.mixed[-1], 'lazy', None)    # ... once a consumer has applied the LayerNorm
        return states[-1], sv
    tail = states[-nm:]
    if nm != 1:
        Wo = NP.out_conv_w.view(C, nm * C)
        pending = getattr(sv.mixed[-1], 'pending_mix', None)
        if pending is not None:
            V, chan, sv.oconv = _mix_conv_fwd(pending, tail, C, Wo, nm * C, NP, training, stats)
        else:
            V, chan, sv.oconv = conv_bn_fwd(tail, C, Wo, nm * C, NP.out_conv_b, NP.bn_w, NP.bn_b,
                                            NP.bn_rm, NP.bn_rv, NP.bn_nbt, training, stats=stats)
        sv.d_out = DROP.make(NP.out_p, x.numel(), training)
        o = torch.empty_like(x)
        sv.fused_bn_tail = FUSE_BN_TAIL and b <= BN_TAIL_MAX_B and C <= 1024
        if sv.fused_bn_tail:             # BatchNorm + ReLU + dropout + residual + LayerNorm: one launch
            out = torch.empty_like(x)
            sv.next_pair_done = next_pair is not None
            lib.bn_relu_ln_fwd(V, chan, x, NP.ln_w, NP.ln_b, o, out, sv.stats, b, C, L, sv.d_out,
                               sv.oconv.fin, sv.osum, next_pair)
            sv.o = o
            return out, sv
        lib.bn_relu_fwd(V, chan, o, b, C, L, sv.d_out, sv.oconv.fin)
    else:
        o = tail[0]
    sv.o = o
    out = torch.empty_like(x)
    lib.cat_ln_fwd([o], x, NP.ln_w, NP.ln_b, out, sv.stats, b, C, L, False, sv.osum)
    return out, sv


def node_cell_bwd(sv, g, x_slot, y_slot, dbeta_w, dgamma_w, NG, deferred=None, defer_first=False, pre_pair=None):
    """g: grad of the node output.  x_slot / y_slot: GradSlots of the two inputs (the same
    object in search mode).  dbeta_w / dgamma_w: zero-initialised (k_in,2)/(ns,4) buffers
    receiving the gradients w.r.t. the SOFTMAXED weights.  NG: gradient pack.
    defer_first: do not run the backward of the first inner mixed sum; return the gradient of
    z0 instead (the caller folds it into bmnas_mixsum_pair_bwd)."""
    x = sv.x
    b, C, L = x.shape
    ns, nm, NP = sv.ns, sv.nm, sv.NP
    slots = [x_slot, y_slot] + [GradSlot(x) for _ in range(ns)]
    tail = list(range(2 + ns - nm, 2 + ns))
    resid = None if sv.fused_tail else x                 # fused: sv.o already holds o + x
    ln_job = None
    mix_done = None
    if nm != 1:
        dV = _empty(x, b, C, L)
        if sv.fused_bn_tail:
            racc = x_slot.acc_bit()
            pre = None
            if pre_pair is not None:
                # the next cell step's K1 pair backward runs first in this launch and completes g
                pbufs, pmask = _write_group(pre_pair['slots'])
                g_slot = pre_pair['g_slot']
                g_full = g_slot.buf()
                g_slot.written = True
                pre = (pre_pair['xs'], pbufs, pmask, pre_pair['out'], pre_pair['w'], 2, pre_pair['w2'], 2,
                       pre_pair['h'], pre_pair['gh'], pre_pair['gz'], pre_pair['gz2'], pre_pair['dw'],
                       pre_pair['dw2'], pre_pair['shards'], pre_pair['stride'], g_full)
            lib.bn_relu_ln_bwd(g, sv.o, x, NP.ln_w, sv.stats, sv.oconv.U, sv.oconv.chan, dV, NG.bn_grad,
                               x_slot.buf(), racc, b, C, L, sv.d_out, pre)
            if pre is not None:
                g = pre[-1]
        else:
            d_o = GradSlot(x)
            bufs, mask = _write_group([d_o])
            racc = x_slot.acc_bit()
            lib.cat_ln_bwd(g, [sv.o], x, NP.ln_w, NP.ln_b, sv.stats, bufs, x_slot.buf(),
                           mask | (racc << 31), None, None, b, C, L, False)
            lib.bn_relu_bwd(d_o.buf(), sv.oconv.U, sv.oconv.chan, dV, NG.bn_grad, b, C, L, sv.d_out)
        mix = None
        mlast = sv.mixed[ns - 1]
        last_slot = slots[2 + ns - 1]
        if (FUSE_MIX_EPILOGUE and FUSE_BWD_PAIR and WANT_PARAM_GRADS and mlast.same and last_slot.get() is None
                and len({id(slots[j]) for j in tail}) == nm and sv.oconv.fold == 0
                and lib.conv1x1_bwd_all_mix_ok(b, L, C, nm, C)):
            # the last inner step's mix backward rides in the out_conv data-gradient tiles of its channels
            mz_slot = GradSlot(x)
            mdV = _empty(x, b, 3 * C, L)
            mG = NG.mixed[ns - 1]
            mix = (nm - 1, mlast.conv.U, mlast.conv.chan, mlast.x, mlast.p1, mlast.gamma, dgamma_w[ns - 1],
                   NG.shards, NG.shard_stride, mz_slot.buf(), mz_slot.acc_bit(), mdV, mG.stack_bn_grad,
                   mlast.d_glu, mlast.d_fc)
            mix_done = (mdV, mz_slot)
        conv_bn_bwd(sv.oconv, dV, NG.bn_grad, [slots[j] for j in tail],
                    NG.out_conv_dW.view(C, nm * C), NG.out_conv_db, mix=mix)
    elif sv.lazy is not None:
        # streaming LayerNorm + mix backward from the partial sums its gradient's producers left (lazyln.hip)
        assert sv.mixed[ns - 1].same and slots[tail[0]].get() is None
        ln_job = (g, sv.o, NP.ln_w, sv.stats, x_slot, slots[tail[0]], sv.lazy)
    elif (FUSE_LN_BWD and sv.fused_tail and sv.mixed[ns - 1].same and slots[tail[0]].get() is None
          and lib.node_mix_ln_bwd_ok(b, C, L)):
        # the LayerNorm backward rides in the last inner step's mix-backward launch
        ln_job = (g, sv.o, NP.ln_w, sv.stats, x_slot, slots[tail[0]], None)
    else:
        bufs, mask = _write_group([slots[tail[0]]])
        racc = x_slot.acc_bit()
        lib.cat_ln_bwd(g, [sv.o], resid, NP.ln_w, NP.ln_b, sv.stats, bufs, x_slot.buf(),
                       mask | (racc << 31), None, None, b, C, L, False)
    _ln_affine(deferred, g, None, [sv.o], resid, NP.ln_w, NP.ln_b, sv.stats, NG.dln_w, NG.dln_b, b, C, L,
               False, False)
    pending = None               # (gz, gz2, offset, n_in) of step t + 1's mixed sum, folded into step t's mix backward
    for t in reversed(range(ns)):
        ln = ln_job if t == ns - 1 else None
        gs = None if ln is not None else slots[2 + t].get()
        if gs is None and pending is None and ln is None:
            continue                                    # this inner state feeds nothing
        pre_done = None
        if mix_done is not None and t == ns - 1:
            z_slot = mix_done[1]
            pre_done = (mix_done[0], z_slot.buf())
        else:
            z_slot = GradSlot(x)
        nxt = None
        if pending is not None:
            gz, gz2, off_n, n_in = pending              # states[:n_in - 1] + this step's output states[n_in - 1]
            bufs, mask = _write_group(slots[:n_in - 1])
            g_out = slots[2 + t].buf()                  # in place when something was accumulated already
            slots[2 + t].written = True
            nxt = (sv.states[:n_in - 1], bufs, mask, sv.beta_w[off_n:, 1], 2, dbeta_w[off_n:, 1], NG.shards,
                   NG.shard_stride, sv.states[n_in - 1], gz, gz2, g_out)
            pending = None
        node_mixed_bwd(sv.mixed[t], gs, dgamma_w[t], z_slot, None, NG.mixed[t], NG.shards, NG.shard_stride,
                       deferred, nxt, ln, pre_done)
        if t == 0 and defer_first:
            return z_slot.buf(), z_slot.extra
        off = sv.offsets[t]
        n_in = 2 + t
        if t >= 1 and sv.next_fused[t - 1]:
            pending = (z_slot.buf(), z_slot.extra, off, n_in)
            continue
        mixsum_bwd(sv.states[:n_in], slots[:n_in], sv.beta_w[off:, 1], z_slot.buf(),
                   dbeta_w[off:, 1], 2, NG.shards, NG.shard_stride, g2=z_slot.extra)
    return None


# ------------------------------------------------------------------- search-mode FusionCell
class CellSaved:
    pass


def fusion_cell_fwd(xs, alpha_w, beta_ws, gamma_ws, CP, training, S, M, ns, nm, weffs=None, stats=None,
                    head=None, prologue=None):
    """FusionCell.forward (model_search.py:50-68) with the step nodes in search mode
    (FusionNode(x, x), model_search.py:59).  alpha_w (k, 2) softmaxed device tensor.
    head: Pack(W, bias, hb) -> the cell's LayerNorm tail continues into the central classifier in
    ONE launch (bmnas_head_fwd) and the function returns the logits (b, O) = head.hb[0]."""
    N = len(xs)
    b, C, L = xs[0].shape
    sv = CellSaved()
    sv.N, sv.S, sv.M, sv.CP, sv.alpha_w = N, S, M, CP, alpha_w
    states = list(xs)
    sv.sifs, sv.nodes, sv.offsets = [], [], []
    offset = 0
    pending_pair = None                          # (sif, z0) of step i formed by step i - 1's tail launch
    # the step nodes' LayerNorm applied by their consumers (streaming grids, csrc/lazyln.hip)
    sv.lazy_on = bool(LAZY_LN and head is not None and nm == 1 and FUSE_TAIL and FUSE_PAIR and N + S <= 15
                      and 1 <= S <= 3 and M <= S and (C * L) % 64 == 0 and xs[0].is_cuda and lib.lazy_ln_ok(C, L))
    for i in range(S):
        if pending_pair is not None:
            sif, z0 = pending_pair
            pending_pair = None
        elif FUSE_PAIR and len(states) <= 15:
            sif, z0 = torch.empty_like(xs[0]), torch.empty_like(xs[0])
            lz = sv.nodes[-1].lazy if (sv.lazy_on and i >= 1) else None
            if i == 0 and prologue is not None:
                prologue(states, sif, z0)        # the cell prologue rides in this launch (bmnas_cell_prologue_pair)
            elif lz is not None and not lz.materialised:
                # the previous node's output is still un-normalised: this launch applies its LayerNorm and writes it
                lib.mixsum_pair_fwd_lazy(states[:-1], alpha_w[offset:, 1], 2, beta_ws[i][:, 1], 2, lz.desc,
                                         states[-1], sv.nodes[-1].osum, sif, z0, b, C, L)
                lz.materialised = True
            else:
                lib.mixsum_pair_fwd(states, alpha_w[offset:, 1], 2, beta_ws[i][:, 1], 2, sif, z0)
        else:
            sif, z0 = mixsum_fwd(states, alpha_w[offset:, 1]), None
        if STRICT_ZERO:
            if i == 0:
                poison = xs[0] * 0.
                for x_ in xs[1:]:
                    poison = poison + x_ * 0.
            sif.add_(poison)
            if z0 is not None:
                z0.add_(poison)
        # small batches: step i + 1's pair sum (its inputs = today's states + this node's output) rides in the
        # launch that ends this node (bmnas_bn_relu_ln_fwd_pair)
        next_pair = None
        if (FUSE_NEXT_PAIR and FUSE_PAIR and FUSE_BN_TAIL and i + 1 < S and nm != 1 and len(states) + 1 <= 15
                and lib.bn_relu_ln_fwd_pair_ok(b, C, L, len(states))):
            nsif, nz0 = torch.empty_like(xs[0]), torch.empty_like(xs[0])
            next_pair = (list(states), alpha_w[offset + len(states):, 1], 2, beta_ws[i + 1][:, 1], 2, nsif, nz0)
        out, nsv = node_cell_fwd(sif, sif, beta_ws[i], gamma_ws[i], CP.nodes[i], training, ns, nm, z0,
                                 None if weffs is None else weffs[i * ns:(i + 1) * ns], stats,
                                 want_sums=head is not None, next_pair=next_pair, lazy=sv.lazy_on)
        if next_pair is not None and nsv.next_pair_done:
            pending_pair = (next_pair[5], next_pair[6])
        nsv.paired = z0 is not None
        sv.sifs.append(sif)
        sv.nodes.append(nsv)
        sv.offsets.append(offset)
        offset += len(states)
        states.append(out)
    sv.states = states
    sv.stats = _empty(xs[0], b * 2)
    sv.head = head
    if head is not None:
        if M > S:
            raise lib.BmnasError('fused head: the cell concatenates input states (multiplier > steps)')
        head.sums = [nsv.osum for nsv in sv.nodes[S - M:]]
        if DETERMINISTIC and (not sv.lazy_on or ns != 1):
            # (node_steps >= 2: the inner steps t < ns - 1 run bmnas_node_mix_bwd, whose BatchNorm-affine / dgamma
            # reductions are atomics from many workgroups — refused rather than silently non-reproducible)
            raise lib.BmnasError('BMNAS_DETERMINISTIC covers the search cell with node_steps == 1 and node_multiplier '
                                 '== 1 under the fused head (the lazy-LayerNorm path); this configuration is outside it')
        last = sv.nodes[-1].lazy if sv.lazy_on else None
        if last is not None and not last.materialised:
            # the last node's output exists only inside the classifier GEMM's operand fetch
            O_ = head.W.shape[0]
            hb_part = _empty(xs[0], lib.head_fwd_part_floats(b, C, L, M, O_)) if DETERMINISTIC else None
            lib.head_fwd_lazy(states[-M:-1] + [sv.nodes[-1].o], head.sums[:-1] + [None], M - 1, last.desc, CP.ln_w,
                              CP.ln_b, head.W, head.bias, head.hb, sv.stats, b, C, L, O_, hb_part)
        else:
            lib.head_fwd(states[-M:], head.sums, CP.ln_w, CP.ln_b, head.W, head.bias, head.hb, sv.stats, b, C, L,
                         head.W.shape[0])
        if STRICT_ZERO and S > 0:
            head.hb[0].add_(poison.sum(dim=(1, 2))[:, None])    # (see below)
        return head.hb[0], sv
    out = _empty(xs[0], b, M * C * L)
    lib.cat_ln_fwd(states[-M:], None, CP.ln_w, CP.ln_b, out, sv.stats, b, C, L, True)
    if STRICT_ZERO and S > 0:
        # The kernels' ReLU is `v > 0 ? v : 0`, which maps NaN to 0 where torch's relu keeps it: a sample with a non-finite
        # cell input is NaN from its first step node on (the node's LayerNorm spreads it over the sample), and the
        # reference's K7 LayerNorm + ReLU return NaN for that whole sample — restored here per sample.
        out.add_(poison.sum(dim=(1, 2))[:, None])
    return out, sv


# False while a backward runs in which NO parameter of the cell (nor of the fused classifier) needs a gradient — the
# architecture step of the search loop (architect.py:21-29) differentiates alpha / beta / gamma only.  The reference
# computes every weight gradient there anyway (loss.backward()) and throws them away at the next zero_grad(); a
# captured architecture step (bmnas.graph.GraphedTrainStep over the arch optimizer) asks for the arch gradients only
# and says so with `arch_grads_only()`: the weight-gradient GEMM tiles, the LayerNorm-affine reductions and the
# classifier's weight-gradient product are then not launched / carried.  (autograd cannot tell a custom Function which
# of its inputs a torch.autograd.grad call is after: ctx.needs_input_grad is fixed at forward time.)
WANT_PARAM_GRADS = True
_ARCH_ONLY = [False]


class arch_grads_only:
    """with arch_grads_only(on): a backward that runs inside differentiates architecture tensors only — no module
    parameter and no input of the fused cell gets a gradient (so nothing upstream of the cell runs its backward)."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev, _ARCH_ONLY[0] = _ARCH_ONLY[0], self.on
        return self

    def __exit__(self, *exc):
        _ARCH_ONLY[0] = self.prev
        return False


# The mirror image: the WEIGHT step of the search loop (train_searchable/*.py: its optimizer holds model.parameters(),
# none of alpha / beta / gamma).  The reference's loss.backward() forms the architecture gradients there as well and
# zeroes them before the next architecture step; a captured weight step says `weight_grads_only()` and the fused cell
# then runs no arch-softmax backward and its cell-level K1 pair backward launches neither form the edge-weight dot
# products nor load the operands read only for them (WANT_ARCH_GRADS below).
_NO_ARCH = [False]


class weight_grads_only:
    """with weight_grads_only(on): a backward that runs inside asks for no gradient of an architecture tensor."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev, _NO_ARCH[0] = _NO_ARCH[0], self.on
        return self

    def __exit__(self, *exc):
        _NO_ARCH[0] = self.prev
        return False

# Data-parallel overlap (bench.py's 'overlap' shape): a callable (i, NG) invoked right after the launches of
# step node i's backward have been issued.  From that point the node's conv / BatchNorm gradients in NG
# (stack_dW, stack_dbias, stack_bn_grad, out_conv_dW / _db, bn_grad) are final — its LayerNorm-affine gradients
# are not (the epilogue launch sums them) — so an all-reduce of them can run on a forked stream while the
# remaining nodes' backward continues.
NODE_DONE_HOOK = None

# Does anybody differentiate alpha / beta / gamma in the backward that is running?  (set by FusedCellFn.backward: no
# when none of them requires a gradient or inside weight_grads_only().)  False: the cell-level K1 pair backward launches skip
# their dot products and the N + 1 operand streams read only for them (dw = dw2 = NULL at the C ABI).
WANT_ARCH_GRADS = True


def fusion_cell_bwd(sv, g, need_input_grads, dalpha_w, dbeta_ws, dgamma_ws, CG, epilogue=None):
    """Returns the list of input gradients (None where not needed).  d*_w: zero-initialised
    buffers for the gradients w.r.t. the softmaxed arch weights.  CG: gradient pack."""
    N, S, M, CP = sv.N, sv.S, sv.M, sv.CP
    x0 = sv.states[0]
    b, C, L = x0.shape
    slots = [GradSlot(x0) if (j >= N or need_input_grads[j]) else None for j in range(N + S)]
    tail = slots[-M:]
    bufs, mask = _write_group(tail)
    deferred = Deferred()
    head, sums = sv.head, ()
    if head is not None:
        # classifier backward + (deferred) criterion + K7 backward in one launch; its batch reductions
        # (dWcls, dbcls, the K7 affine gradients) leave as per-chunk partials summed by the epilogue
        O, D = head.W.shape[0], M * C * L
        n_chunk = lib.head_chunks(b)
        want = WANT_PARAM_GRADS
        part = _empty(x0, n_chunk * (O + 3) * D) if want else None
        mode, gten, gscale, labels = head.resolve(g)
        if sv.lazy_on:
            # every source as (pre, node LayerNorm): the launch also leaves the partial sums of the nodes' LayerNorm
            # backward; later cell steps' K1 backward launches add theirs (k1 buffers allocated here)
            P = sv.nodes[0].lazy.P
            for t, nsv in enumerate(sv.nodes):
                lzn = nsv.lazy
                lzn.lnp_head = _empty(x0, b * (C * L // 64) * 2) if t >= S - M else None
                lzn.k1_n = S - 1 - t
                lzn.lnp_k1 = _empty(x0, b * lzn.k1_n * P * 2) if lzn.k1_n else None
            tailn = sv.nodes[S - M:]
            loss_part = _empty(x0, n_chunk) if (DETERMINISTIC and mode != 0) else None
            lib.head_bwd_lazy([n.lazy.desc for n in tailn], [n.lazy.lnp_head for n in tailn], bufs, mask, CP.ln_w,
                              CP.ln_b, head.W, head.hb, sv.stats, mode, gten, gscale, labels, head.loss, part, b, C,
                              L, O, getattr(CG, 'scrub', None), loss_part)
            if loss_part is not None:            # the chunks' shares in a fixed order (a four-element sum)
                head.loss.copy_(loss_part.sum(0, keepdim=True))
        else:
            lib.head_bwd(sv.states[-M:], head.sums, bufs, mask, CP.ln_w, CP.ln_b, head.W, head.hb, sv.stats, mode,
                         gten, gscale, labels, head.loss, part, b, C, L, O, getattr(CG, 'scrub', None))
        head.dW = head.dbias = None
        if want:
            hsum = _empty(x0, (O + 3) * D)
            sums = ((part, hsum, n_chunk),)
            head.dW, head.dbias = hsum[:O * D].view(O, D), hsum[(O + 2) * D:(O + 2) * D + O]
            CG.dln_w = hsum[O * D:(O + 1) * D].view(M * C, L)
            CG.dln_b = hsum[(O + 1) * D:(O + 2) * D].view(M * C, L)
    else:
        lib.cat_ln_bwd(g, sv.states[-M:], None, CP.ln_w, CP.ln_b, sv.stats, bufs, None, mask, None, None,
                       b, C, L, True, getattr(CG, 'scrub', None))
        _ln_affine(deferred, g, None, sv.states[-M:], None, CP.ln_w, CP.ln_b, sv.stats, CG.dln_w, CG.dln_b,
                   b, C, L, True, False)
    pending = None                 # the K1 pair backward of step i + 1, to run inside node i's first launch
    g_fulls = {}                   # step -> its stored G (write-once input gradients)
    for i in reversed(range(S)):
        gn = slots[N + i].get()
        if gn is None and pending is None:
            continue
        sif_slot = GradSlot(x0)
        nsv = sv.nodes[i]
        pre_pair, pending = pending, None
        gz = node_cell_bwd(nsv, gn, sif_slot, sif_slot, dbeta_ws[i], dgamma_ws[i], CG.nodes[i], deferred,
                           defer_first=nsv.paired, pre_pair=pre_pair)
        if NODE_DONE_HOOK is not None:
            NODE_DONE_HOOK(i, CG.nodes[i])
        off = sv.offsets[i]
        n_in = N + i
        if gz is not None:
            gz, gz2 = gz
            prev = sv.nodes[i - 1] if i >= 1 else None
            if (FUSE_NEXT_PAIR and prev is not None and prev.nm != 1 and getattr(prev, 'fused_bn_tail', False)
                    and lib.bn_relu_ln_fwd_pair_ok(b, C, L, n_in - 1)):
                # small batches: this backward rides at the start of the previous node's tail launch
                # (bmnas_bn_relu_ln_bwd_pair), whose input gradient it completes
                pending = dict(xs=sv.states[:n_in - 1], slots=slots[:n_in - 1], g_slot=slots[n_in - 1],
                               out=sv.states[n_in - 1], w=sv.alpha_w[off:, 1], w2=nsv.beta_w[:, 1],
                               h=sv.sifs[i], gh=sif_slot.get(), gz=gz, gz2=gz2, dw=dalpha_w[off:, 1],
                               dw2=dbeta_ws[i][:, 1], shards=CG.shards, stride=CG.shard_stride)
                continue
            # write-once input gradients: steps i >= 1 store their G and leave dx_j (j < N) to step 0's launch
            wonce = sv.lazy_on and WRITE_ONCE and any(s_ is not None for s_ in slots[:N])
            if wonce and i >= 1:
                g_fulls[i] = torch.empty_like(x0)
                bufs, mask = _write_group([None] * N + slots[N:n_in])
            else:
                bufs, mask = _write_group(slots[:n_in])
            da_w, db_w = (dalpha_w[off:, 1], dbeta_ws[i][:, 1]) if WANT_ARCH_GRADS else (None, None)
            if sv.lazy_on and i >= 1:
                # its last i inputs are step-node outputs with a streaming LayerNorm backward: leave their partials
                P = sv.nodes[0].lazy.P
                lzs = [sv.nodes[t].lazy for t in range(i)]
                views = [lz.lnp_k1[(i - 1 - t) * P * 2:] for t, lz in enumerate(lzs)]
                lib.mixsum_pair_bwd_lazy(sv.states[:n_in], bufs, sv.alpha_w[off:, 1], 2, nsv.beta_w[:, 1], 2,
                                         sv.sifs[i], sif_slot.get(), gz, da_w, db_w, mask,
                                         [lz.desc for lz in lzs], views, [lz.k1_n * P for lz in lzs], b, C, L,
                                         CG.shards, CG.shard_stride, gz2, g_fulls.get(i))
            elif g_fulls:
                ts = sorted(g_fulls)
                lib.mixsum_pair_bwd_x(sv.states[:n_in], bufs, sv.alpha_w[off:, 1], 2, nsv.beta_w[:, 1], 2,
                                      sv.sifs[i], sif_slot.get(), gz, da_w, db_w, mask,
                                      [g_fulls[t] for t in ts], [sv.alpha_w[sv.offsets[t]:, 1] for t in ts],
                                      CG.shards, CG.shard_stride, gz2)
            else:
                lib.mixsum_pair_bwd(sv.states[:n_in], bufs, sv.alpha_w[off:, 1], 2, nsv.beta_w[:, 1], 2,
                                    sv.sifs[i], sif_slot.get(), gz, da_w, db_w, mask, CG.shards, CG.shard_stride, gz2)
        else:
            mixsum_bwd(sv.states[:n_in], slots[:n_in], sv.alpha_w[off:, 1], sif_slot.buf(),
                       dalpha_w[off:, 1], 2, CG.shards, CG.shard_stride)
    if epilogue is not None and FUSE_EPILOGUE and 0 < len(deferred.probs) <= 8:
        # the LayerNorm affine reductions and the arch-softmax backward end the pass in ONE launch
        ws, dws, outs = epilogue
        lib.backward_epilogue(deferred.probs, b, L, ws, dws, outs, CG.shards, CG.shard_stride, sums)
        deferred.probs = []
        sv.epilogue_done = True
    else:
        deferred.flush(b, L)
        for part, out, n_chunk in sums:
            lib.sum_chunks(part, out, n_chunk)
        sv.epilogue_done = False
    return [s.get() if s is not None else None for s in slots[:N]]
