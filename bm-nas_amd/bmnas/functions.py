"""torch.autograd.Function wrappers over bmnas.cell / bmnas.lib.

Every Function requires CUDA(HIP) fp32 tensors and the built shared library; there is no
CPU or eager-PyTorch fallback on the product path (bmnas.lib raises if the .so is absent).
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import cell as K
from . import lib


def _require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f'bmnas: {what} needs tensors on a HIP device (got {t.device}); the fusion-cell '
                           'hot path runs only on the gfx950 kernels — there is no CPU fallback')


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _f32(t):
    if t.dtype != torch.float32:
        raise TypeError(f'bmnas kernels are fp32 only (got {t.dtype})')
    return t


# ------------------------------------------------------------------- arch softmax
class ArchSoftmaxFn(Function):
    """softmax(logits, dim=-1) for the (rows, 2|4) architecture parameters
    (model_search.py:95, node_search.py:102-103)."""

    @staticmethod
    def forward(ctx, logits):
        _require_gpu(logits, 'arch softmax')
        a = _c(_f32(logits))
        w = torch.empty_like(a)
        lib.arch_softmax_fwd(a, w, a.shape[0], a.shape[1])
        ctx.save_for_backward(w)
        return w

    @staticmethod
    def backward(ctx, dw):
        (w,) = ctx.saved_tensors
        da = torch.empty_like(w)
        lib.arch_softmax_bwd(w, _c(dw), da, w.shape[0], w.shape[1])
        return da


def arch_softmax(logits, device):
    if logits.device != device:
        logits = logits.to(device)        # reference keeps alphas on the CPU; tracked copy
    return ArchSoftmaxFn.apply(logits)


# --------------------------------------------------------------------------- K1
class MixSumFn(Function):
    """out = sum_j w[j] * xs[j]  (w: (n_in,) device tensor)."""

    @staticmethod
    def forward(ctx, w, *xs):
        _require_gpu(xs[0], 'mixed-edge sum')
        xs = [_c(_f32(x)) for x in xs]
        w = _c(_f32(w))
        out = torch.empty_like(xs[0])
        lib.mixsum_fwd(xs, w, 1, out)
        ctx.xs, ctx.w = xs, w
        return out

    @staticmethod
    def backward(ctx, g):
        xs, w = ctx.xs, ctx.w
        g = _c(g)
        need_w = ctx.needs_input_grad[0]
        dw = torch.zeros_like(w) if need_w else None
        dxs = [torch.empty_like(x) if ctx.needs_input_grad[1 + j] else None for j, x in enumerate(xs)]
        lib.mixsum_bwd(xs, dxs, w, 1, g, dw, 0)
        return (dw, *dxs)


# ---- LayerNorm-affine reductions of the per-op path: ONE launch per backward pass ----------------------------------
# The fused search cell ends its backward with one batched launch for every LayerNorm's (dweight, dbias) (reductions
# over the batch: bmnas_ln_affine_bwd_multi); the per-op Functions below (Found_* nets, standalone modules) paid one
# launch per LayerNorm — 4 of the 22 launches of an MM-IMDB found-stage step.  Inside `deferred_affine()` (entered by
# bmnas.graph.GraphedTrainStep around its torch.autograd.grad call, whose results nobody reads before the context
# ends) they only note the job; the context's exit launches them together.  Plain `loss.backward()` keeps the
# immediate launches: an AccumulateGrad node may ADD a parameter's incoming gradient to an existing .grad right away.
# Constraint: a deferred job's (dweight, dbias) are handed to autograd BEFORE the launch that fills them has run,
# so nothing may read them inside the torch.autograd.grad call.  That holds when the LayerNorm's weight / bias are leaf
# tensors used by ONE Function of the pass (the engine passes a single incoming gradient through untouched).  A parameter
# that reaches the Function through an op (a cast, a view, a parametrization: not a leaf) or that is used twice (the engine
# adds the two gradients out of place as soon as the second arrives) gets its launches immediately instead — `_ln_affine`
# checks both.
AFFINE_DEFER = None
_DEFER_KEYS = {}          # id(ln_w storage) -> its pending job, for the pass in progress


class deferred_affine:
    def __enter__(self):
        global AFFINE_DEFER, _DEFER_KEYS
        self.prev, AFFINE_DEFER = AFFINE_DEFER, []
        self.prev_keys, _DEFER_KEYS = _DEFER_KEYS, {}
        return self

    def __exit__(self, *exc):
        global AFFINE_DEFER, _DEFER_KEYS
        jobs, AFFINE_DEFER = AFFINE_DEFER, self.prev
        _DEFER_KEYS = self.prev_keys
        groups, sums = {}, []
        for b, L, prob in jobs:
            if prob is None:
                continue                          # launched early (its parameter came up a second time)
            if b is None:
                sums.append(prob)                 # (part, out, n_chunk): the fused found head's per-chunk partials
            else:
                groups.setdefault((b, L), []).append(prob)
        for (b, L), probs in groups.items():
            for i in range(0, len(probs), 8):
                if sums:                          # ride in the same launch (bmnas_backward_epilogue: affine + sums)
                    lib.backward_epilogue(probs[i:i + 8], b, L, [], [], [], 1, 0, sums=tuple(sums[:2]))
                    sums = sums[2:]
                else:
                    lib.ln_affine_bwd_multi(probs[i:i + 8], b, L)
        for part, out, n_chunk in sums:
            lib.sum_chunks(part, out, n_chunk)
        return False


def _ln_affine(g, srcs, resid, ln_w, ln_b, stats, dw, db, b, C, L, relu, prenorm, key=None, leaf=True):
    """key: identifies the LayerNorm's weight (its data pointer) — a second job for the same key inside one deferral, or a
    weight that is not a leaf (`leaf=False`), is launched at once, and so is the earlier job of that key."""
    if AFFINE_DEFER is not None and leaf and (key is None or key not in _DEFER_KEYS):
        AFFINE_DEFER.append((b, L, dict(g=g, gscale=None, srcs=list(srcs), resid=resid, ln_w=ln_w, ln_b=ln_b, stats=stats,
                                        dln_w=dw, dln_b=db, C=C, relu=relu, prenorm=prenorm)))
        if key is not None:
            _DEFER_KEYS[key] = len(AFFINE_DEFER) - 1
        return
    if AFFINE_DEFER is not None and key is not None and _DEFER_KEYS.get(key) is not None:
        i = _DEFER_KEYS[key]                      # the first use's job: its gradient is about to be added to this one
        b0, L0, p0 = AFFINE_DEFER[i]
        AFFINE_DEFER[i] = (b0, L0, None)
        _DEFER_KEYS[key] = None
        lib.ln_affine_bwd(p0['g'], None, p0['srcs'], p0['resid'], p0['ln_w'], p0['ln_b'], p0['stats'], p0['dln_w'],
                          p0['dln_b'], b0, p0['C'], L0, p0['relu'], p0['prenorm'])
    lib.ln_affine_bwd(g, None, srcs, resid, ln_w, ln_b, stats, dw, db, b, C, L, relu, prenorm)


# ---- the zero-filled accumulators of a CAPTURED per-op step: one persistent arena, cleared in front of the replay --------
# The per-op Functions (Found_* nets, standalone modules, the torch-side classifier) add into zero-filled buffers with
# atomics: BatchNorm batch sums (_FwdStatPool), bn_grad | dW | dbias and the LayerNorm affine gradients (_ZeroPool), the
# classifier's split-K output.  Each pool cost a torch.zeros fill launch inside the step (and LinearFn a zero_fill of its
# own).  A captured step (bmnas.graph) now measures what a pass needs during its dress rehearsal (`arena_measure()`), owns
# ONE persistent fp32 arena of that size, and the launch that copies the batch in front of every replay also clears it
# (bmnas_copy_batch's zero-fill jobs): inside the capture (`arena_use(arena)`) the pools carve from it without any fill.
# Anything beyond the measured size falls back to a fill of its own, as before.
class _StepArena:
    def __init__(self, buf):
        self.buf, self.off = buf, 0

    def take(self, n):
        """n floats (exactly n: callers view the slice by shape); the cursor advances by n rounded up to a multiple of
        four so that every slice stays 16-byte aligned."""
        step = (n + 3) // 4 * 4
        if self.off + step > self.buf.numel():
            return None
        v = self.buf[self.off:self.off + n]
        self.off += step
        return v


_ARENA = None            # the arena of the capture in progress
_ARENA_NEED = None       # [floats] while a rehearsal pass is being measured


class arena_measure:
    """with arena_measure() as m: one pass of the step -> m.need = floats of zero-filled scratch it took."""

    def __enter__(self):
        global _ARENA_NEED
        self.prev, _ARENA_NEED = _ARENA_NEED, [0]
        self.cell = _ARENA_NEED
        return self

    def __exit__(self, *exc):
        global _ARENA_NEED
        self.need, _ARENA_NEED = self.cell[0], self.prev
        return False


class arena_use:
    def __init__(self, buf):
        self.arena = _StepArena(buf) if buf is not None and buf.numel() else None

    def __enter__(self):
        global _ARENA
        self.prev, _ARENA = _ARENA, self.arena
        reset_pools()                     # nothing of an earlier (eager) pass may be handed out inside the capture
        return self

    def __exit__(self, *exc):
        global _ARENA
        _ARENA = self.prev
        reset_pools()                     # ... nor arena slices to the eager passes that follow
        return False


def zeros_for_step(n, device):
    """n zero-filled floats: a slice of the captured step's arena (cleared in front of every replay by the batch-copy
    launch), else torch.zeros — one fill launch."""
    if _ARENA_NEED is not None:
        _ARENA_NEED[0] += (n + 3) // 4 * 4
    if _ARENA is not None and _ARENA.buf.device == device:
        v = _ARENA.take(n)
        if v is not None:
            return v
    return torch.zeros(n, device=device, dtype=torch.float32)


def _zero_pair(like):
    """Two zero-filled tensors shaped like `like` (a LayerNorm's dln_w, dln_b), slices of the backward pass's one
    zero-filled chunk (ZERO_POOL; the forward announced them)."""
    n = like.numel()
    buf = ZERO_POOL.take(2 * n, like.device)
    return buf[:n].view_as(like), buf[n:2 * n].view_as(like)


# ------------------------------------------------------------------------ K6 / K7
class CatLnFn(Function):
    """LayerNorm([n_src*C, L]) of cat(srcs, 1) (+ resid), optional ReLU; output (b, n_src*C, L)."""

    @staticmethod
    def forward(ctx, relu, ln_w, ln_b, resid, *srcs):
        _require_gpu(srcs[0], 'concat + LayerNorm')
        srcs = [_c(_f32(s)) for s in srcs]
        resid = None if resid is None else _c(resid)
        b, C, L = srcs[0].shape
        n = len(srcs)
        out = torch.empty((b, n * C, L), device=srcs[0].device, dtype=torch.float32)
        stats = torch.empty(b * 2, device=srcs[0].device, dtype=torch.float32)
        lw, lb = _c(ln_w), _c(ln_b)
        lib.cat_ln_fwd(srcs, resid, lw, lb, out, stats, b, C, L, relu)
        ctx.relu, ctx.srcs, ctx.resid, ctx.lw, ctx.lb, ctx.stats = relu, srcs, resid, lw, lb, stats
        ctx.leaf = ln_w.is_leaf and ln_b.is_leaf
        if any(ctx.needs_input_grad):
            ZERO_POOL.announce(2 * lw.numel())      # dln_w | dln_b of the backward: one fill per pass for all modules
        return out

    @staticmethod
    def backward(ctx, g):
        srcs, resid = ctx.srcs, ctx.resid
        b, C, L = srcs[0].shape
        g = _c(g)
        dsrcs = [torch.empty_like(s) if ctx.needs_input_grad[4 + q] else None for q, s in enumerate(srcs)]
        dres = torch.empty_like(resid) if (resid is not None and ctx.needs_input_grad[3]) else None
        dw, db = _zero_pair(ctx.lw)
        lib.cat_ln_bwd(g, srcs, resid, ctx.lw, ctx.lb, ctx.stats, dsrcs, dres, 0, None, None, b, C, L,
                       ctx.relu)
        _ln_affine(g, srcs, resid, ctx.lw, ctx.lb, ctx.stats, dw, db, b, C, L, ctx.relu, False,
                   key=ctx.lw.data_ptr(), leaf=ctx.leaf)
        return (None, dw, db, dres, *dsrcs)


class CatLnSumsFn(Function):
    """CatLnFn (no ReLU) that also hands out each sample's (sum, sum of squares) of its output — what the fused head
    (FoundHeadFn, csrc/head.hip) takes the K7 LayerNorm statistics from.  -> (out, sums (b, 2))."""

    @staticmethod
    def forward(ctx, ln_w, ln_b, resid, *srcs):
        _require_gpu(srcs[0], 'concat + LayerNorm')
        srcs = [_c(_f32(s)) for s in srcs]
        resid = None if resid is None else _c(resid)
        b, C, L = srcs[0].shape
        n = len(srcs)
        out = torch.empty((b, n * C, L), device=srcs[0].device, dtype=torch.float32)
        stats = torch.empty(b * 2, device=srcs[0].device, dtype=torch.float32)
        sums = torch.empty((b, 2), device=srcs[0].device, dtype=torch.float32)
        lw, lb = _c(ln_w), _c(ln_b)
        lib.cat_ln_fwd(srcs, resid, lw, lb, out, stats, b, C, L, False, sums)
        ctx.relu, ctx.srcs, ctx.resid, ctx.lw, ctx.lb, ctx.stats = False, srcs, resid, lw, lb, stats
        ctx.leaf = ln_w.is_leaf and ln_b.is_leaf
        if any(ctx.needs_input_grad):
            ZERO_POOL.announce(2 * lw.numel())
        ctx.mark_non_differentiable(sums)
        ctx.set_materialize_grads(False)          # (else autograd fills a zero tensor for `sums`' absent gradient: a launch)
        return out, sums

    @staticmethod
    def backward(ctx, g, _g_sums):
        srcs, resid = ctx.srcs, ctx.resid
        b, C, L = srcs[0].shape
        if g is None:                             # nobody differentiated the output (materialisation is off)
            return (None,) * (3 + len(srcs))
        g = _c(g)
        dsrcs = [torch.empty_like(s) if ctx.needs_input_grad[3 + q] else None for q, s in enumerate(srcs)]
        dres = torch.empty_like(resid) if (resid is not None and ctx.needs_input_grad[2]) else None
        dw, db = _zero_pair(ctx.lw)
        lib.cat_ln_bwd(g, srcs, resid, ctx.lw, ctx.lb, ctx.stats, dsrcs, dres, 0, None, None, b, C, L, False)
        _ln_affine(g, srcs, resid, ctx.lw, ctx.lb, ctx.stats, dw, db, b, C, L, False, False,
                   key=ctx.lw.data_ptr(), leaf=ctx.leaf)
        return (dw, db, dres, *dsrcs)


class FoundHeadFn(Function):
    """The found cell's tail + central classifier (+ criterion) as the two launches of csrc/head.hip — K7
    `relu(LayerNorm(cat(states[-M:])))` (model.py:157-160), `central_classifier` (mmimdb_darts_searchable.py:185-188) and,
    inside bmnas.nn.fused_criterion(), the criterion evaluated by the backward launch — instead of cat_ln + linear +
    criterion and their three backward launches.  states: the M concatenated step-node outputs with their per-sample
    sums (CatLnSumsFn).  Returns the logits; the caller attaches K.LAST_HEAD.pop() to them as `_bmnas_head`."""

    @staticmethod
    def forward(ctx, ln_w, ln_b, W, bias, M, *tensors):
        states = [_c(_f32(t)) for t in tensors[:M]]
        sums = [_c(t) for t in tensors[M:2 * M]]
        _require_gpu(states[0], 'found cell head')
        dev = states[0].device
        b, C, L = states[0].shape
        Wc, bc = _c(_f32(W)), _c(_f32(bias))
        O = Wc.shape[0]
        hb_n = (3 * b * O + 3) // 4 * 4
        buf = zeros_for_step(hb_n + 4, dev)                   # logits | A | B accumulate with atomics; + the loss scalar
        head = K.HeadState(W=Wc, bias=bc, hb=buf[:3 * b * O].view(3, b, O), loss=buf[hb_n:hb_n + 1],
                           marker=torch.empty((b, O), device=dev, dtype=torch.float32))
        stats = torch.empty(b * 2, device=dev, dtype=torch.float32)
        lw, lb = _c(ln_w), _c(ln_b)
        lib.head_fwd(states, sums, lw, lb, Wc, bc, head.hb, stats, b, C, L, O)
        ctx.states, ctx.sums, ctx.lw, ctx.lb, ctx.stats, ctx.head, ctx.M = states, sums, lw, lb, stats, head, M
        ctx.leaf = all(t.is_leaf for t in (ln_w, ln_b, W, bias))      # (see deferred_affine: only leaves may be deferred)
        K.LAST_HEAD.append(head)
        return head.hb[0]

    @staticmethod
    def backward(ctx, g):
        states, sums, head, M = ctx.states, ctx.sums, ctx.head, ctx.M
        b, C, L = states[0].shape
        O, D = head.W.shape[0], M * C * L
        mode, gten, gscale, labels = head.resolve(g)
        want = any(ctx.needs_input_grad[:4])
        n_chunk = lib.head_chunks(b)
        part = torch.empty(n_chunk * (O + 3) * D, device=states[0].device, dtype=torch.float32) if want else None
        dstates = [torch.empty_like(s) if ctx.needs_input_grad[5 + i] else None for i, s in enumerate(states)]
        lib.head_bwd(states, sums, dstates, 0, ctx.lw, ctx.lb, head.W, head.hb, ctx.stats, mode, gten, gscale, labels,
                     head.loss, part, b, C, L, O, None)
        dlw = dlb = dW = dbias = None
        if want:
            hsum = torch.empty((O + 3) * D, device=states[0].device, dtype=torch.float32)
            if AFFINE_DEFER is not None and ctx.leaf:
                AFFINE_DEFER.append((None, None, (part, hsum, n_chunk)))      # summed with the pass's affine reductions
            else:
                lib.sum_chunks(part, hsum, n_chunk)
            dW = hsum[:O * D].view(O, D)
            dlw = hsum[O * D:(O + 1) * D].view_as(ctx.lw)
            dlb = hsum[(O + 1) * D:(O + 2) * D].view_as(ctx.lb)
            dbias = hsum[(O + 2) * D:(O + 2) * D + O]
        return (dlw, dlb, dW, dbias, None, *dstates, *([None] * M))


# ----------------------------------------------------------------------------- K3
class SdpaLnFn(Function):
    """ScaledDotAttn.forward (node_operations.py:92-108)."""

    @staticmethod
    def forward(ctx, x, y, ln_w, ln_b, p, training):
        _require_gpu(x, 'scaled-dot attention')
        x, y = _c(_f32(x)), _c(_f32(y))
        b, C, L = x.shape
        out = torch.empty_like(x)
        stats = torch.empty(b * 2, device=x.device, dtype=torch.float32)
        drop = K.DROP.make(p, x.numel(), training)
        lw, lb = _c(ln_w), _c(ln_b)
        xhat = torch.empty_like(x)
        lib.sdpa_ln_fwd(x, y, lw, lb, out, xhat, stats, b, C, L, drop)
        ctx.x, ctx.y, ctx.lw, ctx.stats, ctx.drop, ctx.xhat = x, y, lw, stats, drop, xhat
        ctx.leaf = ln_w.is_leaf and ln_b.is_leaf
        if any(ctx.needs_input_grad):
            ZERO_POOL.announce(2 * lw.numel())
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.x, ctx.y
        b, C, L = x.shape
        dx, dy = torch.empty_like(x), torch.empty_like(y)
        dw, db = _zero_pair(ctx.lw)
        g = _c(g)
        lib.sdpa_ln_bwd(g, None, x, y, ctx.lw, ctx.xhat, ctx.stats, dx, dy, 0, b, C, L, ctx.drop)
        _ln_affine(g, [ctx.xhat], None, None, None, None, dw, db, b, C, L, False, True,
                   key=ctx.lw.data_ptr(), leaf=ctx.leaf)
        return dx, dy, dw, db, None, None


# ------------------------------------------------- conv1x1 + BN + {GLU | ReLU} + dropout
class _ZeroPool:
    """Zero-filled scratch for the gradients that the standalone conv + BatchNorm modules accumulate with
    atomics (bn_grad | dW | dbias per module): ONE torch.zeros per backward pass instead of one per module
    (the six reshape layers of a step paid six ~5 us fills).  Forward calls announce their need; the first
    backward call of a pass allocates for all of them — also inside a hipGraph capture, where the fill
    becomes one memset node.  Anything unexpected (a second backward over the same graph, a module that
    never announced) gets its own torch.zeros, as before."""

    def __init__(self):
        self.pending = 0          # floats announced by forwards since the last backward pass began
        self.chunk = None
        self.off = 0
        self.in_backward = False

    @staticmethod
    def _round(n):
        return (n + 3) // 4 * 4   # slices stay 16-byte aligned

    def announce(self, n):
        if self.in_backward:      # a new step's forward: the previous pass is over
            self.in_backward, self.pending, self.chunk = False, 0, None
        self.pending += self._round(n)

    def take(self, n, device):
        n = self._round(n)
        cap = device.type == 'cuda' and torch.cuda.is_current_stream_capturing()
        if self.chunk is not None and getattr(self, 'captured', cap) != cap:
            self.chunk, self.in_backward = None, False      # allocated inside / outside a capture that is over
        self.captured = cap
        if not self.in_backward:
            self.in_backward = True
            FWD_STAT_POOL.close()
            # (capped: forwards that were never followed by a backward must not inflate the fill)
            self.chunk = zeros_for_step(max(min(self.pending, 1 << 24), n), device)
            self.off = 0
        if self.chunk is None or self.chunk.device != device or self.off + n > self.chunk.numel():
            return zeros_for_step(n, device)
        v = self.chunk[self.off:self.off + n]
        self.off += n
        return v


ZERO_POOL = _ZeroPool()


class _FwdStatPool:
    """Zero-filled BatchNorm batch-sum buffers (StatArena layout) for the standalone conv + BatchNorm + ReLU
    modules, so that their statistics are finalised inside bn_relu_fwd like on the search path instead of by
    a bn_finalize launch per module.  A forward pass cannot announce its needs, so the chunk is sized by the
    previous pass (a pass ends when a backward pass begins); the first pass, and anything beyond the chunk,
    gets a fill of its own."""
    CAP = 1 << 24                 # floats: never carry more than 64 MB from a run of forwards without backward

    def __init__(self):
        self.open = False
        self.chunk = None
        self.off = 0
        self.total = 0
        self.last_total = 0

    def close(self):              # a backward pass began (called by the first ZERO_POOL.take of the pass)
        if self.open:
            self.open = False
            self.last_total = min(self.total, self.CAP)

    def take(self, M):
        """-> zero-filled view of STAT_SHARDS * M * 2 floats (the interface conv_bn_fwd expects of `stats`)."""
        n = (K.STAT_SHARDS * M * 2 + 3) // 4 * 4
        dev = self.device
        cap = dev.type == 'cuda' and torch.cuda.is_current_stream_capturing()
        if self.open and getattr(self, 'captured', cap) != cap:
            self.open, self.chunk = False, None             # the pass that opened the chunk belonged to a capture
        self.captured = cap
        if not self.open:
            self.open, self.off, self.total = True, 0, 0
            self.chunk = zeros_for_step(self.last_total, dev) if self.last_total else None
        self.total += n
        if self.chunk is None or self.chunk.device != dev or self.off + n > self.chunk.numel():
            return zeros_for_step(n, dev)
        v = self.chunk[self.off:self.off + n]
        self.off += n
        return v


FWD_STAT_POOL = _FwdStatPool()


def reset_pools():
    """Forget every buffer the two pools hold.  Called when a hipGraph capture aborts: slices handed out inside
    the dead capture point at memory of a released capture pool whose zero-fill node never ran — the next
    eager forward must not be given the rest of that chunk as 'zero-filled' BatchNorm sums."""
    ZERO_POOL.pending, ZERO_POOL.chunk, ZERO_POOL.off, ZERO_POOL.in_backward = 0, None, 0, False
    FWD_STAT_POOL.open, FWD_STAT_POOL.chunk, FWD_STAT_POOL.off, FWD_STAT_POOL.total = False, None, 0, 0
# BMNAS_FUSE_BN_FINALIZE=0: bn_finalize launches, as in round 1
FUSE_STANDALONE_BN = K.FUSE_BN_FINALIZE


class ConvBnActFn(Function):
    """cat(srcs) -> Conv1d(k=1) -> BatchNorm1d -> glu(dim=1) | relu -> Dropout(p).
    LinearGLU (node_operations.py:30-39), ConcatFC (:49-56), NodeCell out_conv
    (node_search.py:59-64).  bn buffers are updated in place in training mode."""

    @staticmethod
    def forward(ctx, act, p, training, rm, rv, nbt, conv_w, conv_b, bn_w, bn_b, *srcs):
        _require_gpu(srcs[0], 'conv1x1 + BatchNorm')
        srcs = [_c(_f32(s)) for s in srcs]
        b, C_src, L = srcs[0].shape
        M = conv_w.shape[0]
        K_in = len(srcs) * C_src
        W = _c(conv_w).view(M, K_in)
        pool = None
        if training and FUSE_STANDALONE_BN:
            pool = FWD_STAT_POOL             # statistics finalised inside bn_relu_fwd / bn_glu_fwd (no bn_finalize launch)
            pool.device = srcs[0].device
        U, chan, sv = K.conv_bn_fwd(srcs, C_src, W, K_in, _c(conv_b), _c(bn_w), _c(bn_b), rm, rv, nbt,
                                    training, stats=pool)
        if act == 'glu':
            Cout = M // 2
            out = torch.empty((b, Cout, L), device=U.device, dtype=torch.float32)
            drop = K.DROP.make(p, out.numel(), training)
            lib.bn_glu_fwd(U, chan, out, b, Cout, L, drop, sv.fin)
        else:
            out = torch.empty((b, M, L), device=U.device, dtype=torch.float32)
            drop = K.DROP.make(p, out.numel(), training)
            lib.bn_relu_fwd(U, chan, out, b, M, L, drop, sv.fin)
        ctx.act, ctx.sv, ctx.drop, ctx.wshape = act, sv, drop, tuple(conv_w.shape)
        if any(ctx.needs_input_grad):
            ZERO_POOL.announce(2 * M + M * sv.ldw + M)
        return out

    @staticmethod
    def backward(ctx, g):
        sv, act = ctx.sv, ctx.act
        U = sv.U
        b, M, L = U.shape
        g = _c(g)
        dV = torch.empty_like(U)
        # ONE zero-filled buffer for the three gradients that are accumulated with atomics, carved out of
        # one fill per backward pass (_ZeroPool)
        zero = ZERO_POOL.take(2 * M + M * sv.ldw + M, U.device)
        bn_grad = zero[:2 * M]
        dW = zero[2 * M:2 * M + M * sv.ldw].view(M, sv.ldw)
        dbias = zero[2 * M + M * sv.ldw:]
        if act == 'glu':
            lib.bn_glu_bwd(g, U, sv.chan, dV, bn_grad, b, M // 2, L, ctx.drop)
        else:
            lib.bn_relu_bwd(g, U, sv.chan, dV, bn_grad, b, M, L, ctx.drop)
        slots = [K.GradSlot(s) if ctx.needs_input_grad[10 + q] else None for q, s in enumerate(sv.srcs)]
        K.conv_bn_bwd(sv, dV, bn_grad, slots, dW, dbias)
        dsrcs = [s.get() if s is not None else None for s in slots]
        return (None, None, None, None, None, None, dW.view(ctx.wshape), dbias, bn_grad[:M], bn_grad[M:],
                *dsrcs)


class ConvBnActThruFn(Function):
    """ConvBnActFn whose sources are ALSO handed back as outputs: -> (out, *srcs).  A discrete step node reads an inner
    state more than once (Found_NodeCell.forward, node.py:45-76: a later inner step, the out_conv tail, the residual); as
    separate autograd nodes every extra reader costs an `at::add` launch when the engine sums the gradients.  Here the
    later readers take the handed-back alias instead, so their gradient arrives as this node's grad_output for it — and
    the data-gradient launch ACCUMULATES onto that tensor in place (the kernels' accumulate mask) instead of writing a
    fresh one for the engine to add.  (The arriving gradient is written in place: it is the fresh result of the later
    reader's backward, which nothing else holds — no retain_grad / hooks on these inner tensors.)"""

    @staticmethod
    def forward(ctx, act, p, training, rm, rv, nbt, conv_w, conv_b, bn_w, bn_b, *srcs):
        out = ConvBnActFn.forward(ctx, act, p, training, rm, rv, nbt, conv_w, conv_b, bn_w, bn_b, *srcs)
        ctx.set_materialize_grads(False)
        return (out, *[s.view_as(s) for s in srcs])

    @staticmethod
    def backward(ctx, g, *g_thru):
        sv, act = ctx.sv, ctx.act
        n = len(sv.srcs)
        if g is None:                       # only the handed-back sources were differentiated: identity
            return (None,) * 10 + tuple(g_thru)
        U = sv.U
        b, M, L = U.shape
        g = _c(g)
        dV = torch.empty_like(U)
        zero = ZERO_POOL.take(2 * M + M * sv.ldw + M, U.device)
        bn_grad = zero[:2 * M]
        dW = zero[2 * M:2 * M + M * sv.ldw].view(M, sv.ldw)
        dbias = zero[2 * M + M * sv.ldw:]
        if act == 'glu':
            lib.bn_glu_bwd(g, U, sv.chan, dV, bn_grad, b, M // 2, L, ctx.drop)
        else:
            lib.bn_relu_bwd(g, U, sv.chan, dV, bn_grad, b, M, L, ctx.drop)
        slots = []
        for q, s in enumerate(sv.srcs):
            gt = g_thru[q] if q < len(g_thru) else None
            if not ctx.needs_input_grad[10 + q]:
                slots.append(None)
                continue
            slot = K.GradSlot(s)
            if gt is not None and gt.is_contiguous() and gt.dtype == torch.float32:
                slot.t, slot.written = gt, True          # the later readers' gradient: accumulated onto, in place
            slots.append(slot)
        K.conv_bn_bwd(sv, dV, bn_grad, slots, dW, dbias)
        dsrcs = []
        for q, slot in enumerate(slots):
            gt = g_thru[q] if q < len(g_thru) else None
            d = slot.get() if slot is not None else None
            if slot is not None and gt is not None and d is not gt:
                d = gt if d is None else d + gt          # (a non-contiguous arrival: the engine's way)
            dsrcs.append(d)
        return (None, None, None, None, None, None, dW.view(ctx.wshape), dbias, bn_grad[:M], bn_grad[M:], *dsrcs)


class ConvBnReluLnFn(Function):
    """The tail of a discrete step node with node_multiplier != 1 (Found_NodeCell.forward, node.py:62-76):
        out = LayerNorm_[C, L]( dropout(relu(bn(out_conv(cat(tail))))) + x )
    as the conv GEMM + ONE tail launch per direction (bmnas_bn_relu_ln_fwd / _bwd: the kernels the search path's
    NodeCell ends in) instead of ConvBnActFn + CatLnFn (conv, BatchNorm tail, LayerNorm: one launch more each way).
    -> (out, sums): sums (b, 2) = each sample's (sum, sum of squares) of `out` when want_sums (the fused head's K7
    statistics), else an empty tensor.  Callers check `usable(b, C)` first."""

    @staticmethod
    def usable(b, C):
        return K.FUSE_BN_TAIL and b <= K.BN_TAIL_MAX_B and C <= 1024

    @staticmethod
    def forward(ctx, p, training, want_sums, rm, rv, nbt, conv_w, conv_b, bn_w, bn_b, ln_w, ln_b, x, *srcs):
        _require_gpu(srcs[0], 'conv1x1 + BatchNorm + LayerNorm tail')
        srcs = [_c(_f32(s)) for s in srcs]
        x = _c(_f32(x))
        b, C_src, L = srcs[0].shape
        M = conv_w.shape[0]
        K_in = len(srcs) * C_src
        W = _c(conv_w).view(M, K_in)
        pool = None
        if training and FUSE_STANDALONE_BN:
            pool = FWD_STAT_POOL
            pool.device = srcs[0].device
        U, chan, sv = K.conv_bn_fwd(srcs, C_src, W, K_in, _c(conv_b), _c(bn_w), _c(bn_b), rm, rv, nbt,
                                    training, stats=pool)
        o = torch.empty_like(x)                  # dropout(relu(bn(U))), saved for the backward
        out = torch.empty_like(x)
        stats = torch.empty(b * 2, device=x.device, dtype=torch.float32)
        sums = torch.empty((b, 2) if want_sums else (0,), device=x.device, dtype=torch.float32)
        drop = K.DROP.make(p, out.numel(), training)
        lw, lb = _c(ln_w), _c(ln_b)
        lib.bn_relu_ln_fwd(U, chan, x, lw, lb, o, out, stats, b, M, L, drop, sv.fin, sums if want_sums else None)
        ctx.sv, ctx.drop, ctx.wshape, ctx.x, ctx.o, ctx.stats, ctx.lw, ctx.lb = sv, drop, tuple(conv_w.shape), x, o, stats, lw, lb
        ctx.leaf = ln_w.is_leaf and ln_b.is_leaf
        if any(ctx.needs_input_grad):
            ZERO_POOL.announce(2 * M + M * sv.ldw + M)
            ZERO_POOL.announce(2 * lw.numel())
        ctx.mark_non_differentiable(sums)
        ctx.set_materialize_grads(False)
        return out, sums

    @staticmethod
    def backward(ctx, g, _g_sums):
        sv = ctx.sv
        n_in = 13 + len(sv.srcs)
        if g is None:
            return (None,) * n_in
        U, x = sv.U, ctx.x
        b, M, L = U.shape
        g = _c(g)
        dV = torch.empty_like(U)
        zero = ZERO_POOL.take(2 * M + M * sv.ldw + M, U.device)
        bn_grad = zero[:2 * M]
        dW = zero[2 * M:2 * M + M * sv.ldw].view(M, sv.ldw)
        dbias = zero[2 * M + M * sv.ldw:]
        dx = torch.empty_like(x) if ctx.needs_input_grad[12] else None
        dlw, dlb = _zero_pair(ctx.lw)
        # LayerNorm input gradient, ReLU / dropout mask, BatchNorm reductions and the residual's gradient: one launch
        lib.bn_relu_ln_bwd(g, ctx.o, x, ctx.lw, ctx.stats, U, sv.chan, dV, bn_grad, dx, 0, b, M, L, ctx.drop)
        slots = [K.GradSlot(s) if ctx.needs_input_grad[13 + q] else None for q, s in enumerate(sv.srcs)]
        K.conv_bn_bwd(sv, dV, bn_grad, slots, dW, dbias)
        _ln_affine(g, [ctx.o], x, ctx.lw, ctx.lb, ctx.stats, dlw, dlb, b, M, L, False, False,
                   key=ctx.lw.data_ptr(), leaf=ctx.leaf)
        dsrcs = [s.get() if s is not None else None for s in slots]
        return (None, None, None, None, None, None, dW.view(ctx.wshape), dbias, bn_grad[:M], bn_grad[M:], dlw, dlb, dx,
                *dsrcs)


class PoolGroupFn(Function):
    """The AdaptiveMaxPool2d in front of every reshape conv (aux_models.py:62-70, 101-108), all modalities in ONE
    launch per direction (csrc/pool.hip), output in the (b, C_in, L) layout the grouped GEMM reads.
    dims[i] = (C, H, W, oh, ow) of input i viewed as (b, C, H, W)."""

    @staticmethod
    def forward(ctx, dims, *xs):
        xs = [_c(_f32(x)) for x in xs]
        _require_gpu(xs[0], 'reshape-layer pooling')
        b, dev = xs[0].shape[0], xs[0].device
        outs = [torch.empty((b, d[0], d[3] * d[4]), device=dev, dtype=torch.float32) for d in dims]
        need = any(ctx.needs_input_grad)
        idxs = [torch.empty((b, d[0], d[3] * d[4]), device=dev, dtype=torch.int32) if need else None for d in dims]
        lib.adaptive_maxpool_group(xs, dims, outs, idxs, b)
        ctx.dims, ctx.idxs, ctx.shapes, ctx.b = dims, idxs, [tuple(x.shape) for x in xs], b
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        sel = [i for i in range(len(gs)) if ctx.needs_input_grad[1 + i]]
        dxs = [None] * len(gs)
        if sel:
            dev = ctx.idxs[sel[0]].device
            g = [torch.zeros_like(ctx.idxs[i], dtype=torch.float32) if gs[i] is None else _c(gs[i]) for i in sel]
            dx = [torch.empty(ctx.shapes[i], device=dev, dtype=torch.float32) for i in sel]
            lib.adaptive_maxpool_group_bwd(g, [ctx.idxs[i] for i in sel], dx, [ctx.dims[i] for i in sel], ctx.b)
            for i, t in zip(sel, dx):
                dxs[i] = t
        return (None, *dxs)


class ReshapeGroupFn(Function):
    """The N reshape layers in front of the fusion cell (ReshapeInputLayer{,_MMIMDB}.forward after the pooling,
    aux_models.py:71-74 / 111-114: Conv1d(C_in_i, C, 1) -> BatchNorm1d(C) -> ReLU -> Dropout(drpt) on each of the
    N pooled modality features) as FOUR launches for the whole group instead of five per layer: grouped GEMM
    (+ BatchNorm batch sums), grouped BN-finalise + ReLU + dropout; grouped activation / BatchNorm-reduction
    backward, grouped weight- and data-gradient GEMMs with the BatchNorm input gradient folded in.
    Same kernels bodies, same arithmetic and same dropout sites (layer 0 first) as N ConvBnActFn calls."""

    @staticmethod
    def forward(ctx, n, p, training, buffers, *tensors):
        xs = [_c(_f32(t)) for t in tensors[:n]]
        _require_gpu(xs[0], 'reshape layers')
        prm = [tensors[n + 4 * i:n + 4 * i + 4] for i in range(n)]          # conv.weight, conv.bias, bn.weight, bn.bias
        b, L = xs[0].shape[0], xs[0].shape[2]
        M = prm[0][0].shape[0]
        dev = xs[0].device
        Ws = [_c(w).view(M, -1) for w, _, _, _ in prm]
        cbs = [_c(cb) for _, cb, _, _ in prm]
        Us = [torch.empty((b, M, L), device=dev, dtype=torch.float32) for _ in range(n)]
        chans = [torch.empty(4 * M, device=dev, dtype=torch.float32) for _ in range(n)]
        stats, shards = None, 0
        if training and b * L < 2:
            raise ValueError(f'Expected more than 1 value per channel when training, got input size {[b, M, L]}')
        # ONE zero-filled buffer for everything the group accumulates with atomics — the forward's BatchNorm
        # batch sums and (when a backward will follow) bn_grad | dW | dbias of every layer — cleared by ONE
        # launch, which under hipGraph capture also advances the dropout step counter: it is the first launch
        # of the step, in front of every dropout site (bmnas.cell._DropState.take_advance)
        per = K.STAT_SHARDS * M * 2 if training else 0
        want_bwd = any(ctx.needs_input_grad)              # (all False when autograd is not recording)
        sizes = [2 * M + M * W.shape[1] + M for W in Ws] if want_bwd else []
        offs = [n * per]
        for sz in sizes:
            offs.append(offs[-1] + (sz + 3) // 4 * 4)
        adv = K.DROP.take_advance() if (training and p > 0.0) else None
        pool = torch.empty(offs[-1], device=dev, dtype=torch.float32) if offs[-1] else None
        if pool is not None or adv is not None:
            lib.cell_prologue([], [], [], [], 4, 4, adv, pool)
        if training:
            shards = K.STAT_SHARDS
            stats = [pool[i * per:(i + 1) * per] for i in range(n)]
        ctx.zero = (pool, offs, sizes) if want_bwd else None
        lib.conv1x1_fwd_group(xs, Ws, cbs, Us, stats, shards, b, L, M)
        outs = [torch.empty((b, M, L), device=dev, dtype=torch.float32) for _ in range(n)]
        drops = [K.DROP.make(p, outs[i].numel(), training) for i in range(n)]
        fins = [lib.make_bn_fin(None if stats is None else stats[i], shards, cbs[i], _c(prm[i][2]), _c(prm[i][3]),
                                buffers[i][0], buffers[i][1], buffers[i][2], training) for i in range(n)]
        lib.bn_relu_fwd_group(Us, chans, outs, fins, drops, b, M, L)
        ctx.n, ctx.xs, ctx.Ws, ctx.Us, ctx.chans, ctx.drops, ctx.training = n, xs, Ws, Us, chans, drops, training
        for i in range(n):
            K.bn_ratio_note(chans[i], cbs[i], M, f'reshape layer {i} ({Ws[i].shape[1]}->{M})', training)
        ctx.wshapes = [tuple(w.shape) for w, _, _, _ in prm]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        n, xs, Ws, Us = ctx.n, ctx.xs, ctx.Ws, ctx.Us
        b, M, L = Us[0].shape
        dev = Us[0].device
        gs = [torch.zeros_like(Us[i]) if g is None else _c(g) for i, g in enumerate(gs)]
        # bn_grad | dW | dbias of every layer: the buffer the forward's first launch cleared (a second backward
        # over the same graph gets a fresh one)
        if ctx.zero is not None:
            zero, offs, sizes = ctx.zero
            ctx.zero = None
        else:
            sizes = [2 * M + M * W.shape[1] + M for W in Ws]
            offs = [0]
            for sz in sizes:
                offs.append(offs[-1] + (sz + 3) // 4 * 4)
            zero = torch.zeros(offs[-1], device=dev, dtype=torch.float32)
        bn_grads = [zero[offs[i]:offs[i] + 2 * M] for i in range(n)]
        dWs = [zero[offs[i] + 2 * M:offs[i] + 2 * M + M * Ws[i].shape[1]].view(M, Ws[i].shape[1]) for i in range(n)]
        dbs = [zero[offs[i] + 2 * M + M * Ws[i].shape[1]:offs[i] + sizes[i]] for i in range(n)]
        dVs = [torch.empty_like(U) for U in Us]
        lib.bn_relu_bwd_group(gs, Us, ctx.chans, dVs, bn_grads, ctx.drops, b, M, L)
        dxs = [torch.empty_like(x) if ctx.needs_input_grad[4 + i] else None for i, x in enumerate(xs)]
        lib.conv1x1_bwd_group(dVs, Ws, xs, dxs, dWs, dbs, Us, ctx.chans, bn_grads, ctx.training, b, L, M)
        grads = []
        for i in range(n):
            grads += [dWs[i].view(ctx.wshapes[i]), dbs[i], bn_grads[i][:M], bn_grads[i][M:]]
        return (None, None, None, None, *dxs, *grads)


# ---------------------------------------------------------------- search NodeMixedOp
class NodeMixedFn(Function):
    """NodeMixedOp.forward(x, y, weights) (node_operations.py:118-120) as one fused
    sequence (attention+LN, stacked conv GEMM + BN statistics, gamma-mix)."""

    @staticmethod
    def forward(ctx, op, training, x, y, gamma_row, *params):
        _require_gpu(x, 'NodeMixedOp')
        same = x is y
        x = _c(_f32(x))
        y = x if same else _c(_f32(y))
        P = op.pack()
        out, sv = K.node_mixed_fwd(x, y, _c(gamma_row), P, training)
        ctx.op, ctx.sv = op, sv
        return out

    @staticmethod
    def backward(ctx, g):
        sv, op = ctx.sv, ctx.op
        x = sv.x
        G = op.grad_pack(x.device)
        dgamma = torch.zeros(4, device=x.device, dtype=torch.float32)
        xs = K.GradSlot(x)
        ys = None if sv.same else K.GradSlot(sv.y)
        K.node_mixed_bwd(sv, _c(g), dgamma, xs, ys, G)
        dx = xs.get()
        if xs.extra is not None:                 # attention part produced next to the GEMM
            dx = dx.add_(xs.extra)
        if sv.same:
            # x and y are the same tensor object: autograd adds both returned halves
            dx_half, dy = dx, torch.zeros_like(dx)
        else:
            dx_half, dy = dx, ys.get()
        return (None, None, dx_half, dy, dgamma, *op.grads_in_param_order(G))


# -------------------------------------------------------------------- fused FusionCell
class FusedCellFn(Function):
    """FusionCell.forward in search mode (model_search.py:50-68 with FusionNode(x, x)):
    the whole cell — mixed edges, step nodes, LayerNorm tail — as one autograd node."""

    @staticmethod
    def forward(ctx, cell, training, alpha_is_logits, alpha, n_head, *tensors):
        """tensors = N inputs, 2*S arch tensors, the cell's parameters and — n_head = 2 — the central
        classifier's weight and bias: the cell then ends in the classifier (bmnas_head_fwd) and the
        function returns the logits."""
        N, S = cell.num_input_nodes, cell._steps
        xs = [_c(_f32(t)) for t in tensors[:N]]
        _require_gpu(xs[0], 'FusionCell')
        dev = xs[0].device
        logits = [_c(t) for t in tensors[N:N + 2 * S]]
        if alpha_is_logits:
            logits = [_c(alpha)] + logits
        ws = [torch.empty_like(t) for t in logits]
        CP = cell.pack()
        # every arch softmax and (search mode: conv applied to cat[z, z]) every NodeMixedOp's folded
        # conv weight in ONE launch
        mixed = [m for n in CP.nodes for m in n.mixed]
        b, C_ = xs[0].shape[0], xs[0].shape[1]
        # forward accumulation arena, zero-filled by the prologue launch: the BatchNorm batch sums that
        # the GEMM epilogues add into (finalised by the mix / out_conv kernels: no bn_finalize
        # launches) and the head's logits | A | B | loss
        counts = []
        K.apply_deterministic()
        # (deterministic mode: the BatchNorm statistics go through per-n-group partials + bn_finalize launches — plain
        # stores combined in a fixed order — instead of atomically accumulated sums)
        if K.FUSE_BN_FINALIZE and not K.DETERMINISTIC:
            for n in CP.nodes:
                counts += [3 * C_] * len(n.mixed) + ([C_] if cell.args.node_multiplier != 1 else [])
        stat_n = K.StatArena.numel_for(counts)
        head, head_n = None, 0
        if n_head:
            Wc, bc = _c(_f32(tensors[-2])), _c(_f32(tensors[-1]))
            O = Wc.shape[0]
            hb_n = (3 * b * O + 3) // 4 * 4
            head_n = hb_n + 4
        arena = torch.empty(stat_n + head_n, device=dev, dtype=torch.float32) if stat_n + head_n else None
        stats = K.StatArena(xs[0], counts, arena[:stat_n]) if stat_n else None
        if n_head:
            head = K.HeadState(W=Wc, bias=bc, hb=arena[stat_n:stat_n + 3 * b * O].view(3, b, O),
                               loss=arena[stat_n + hb_n:stat_n + hb_n + 1],
                               marker=torch.empty((b, O), device=dev, dtype=torch.float32))
        weffs, prologue = None, None
        if K.FUSE_PROLOGUE and 0 < len(mixed) <= 8:
            weffs = [torch.empty((3 * C_, C_), device=dev, dtype=torch.float32) for _ in mixed]
            # under capture the first prologue of the step also advances the dropout step counter
            # (bmnas.graph.GraphedStep), saving the separate add launch at the end of every replay
            adv = K.DROP.take_advance()
            stackW = [m.stack_W for m in mixed]
            if (K.FUSE_PROLOGUE_PAIR and K.FUSE_PAIR and alpha_is_logits and N <= 15
                    and logits[0].shape[1] == 2 and logits[1].shape[1] == 2 and logits[1].shape[0] >= 2):
                # the prologue's jobs ride in the launch of the first step's pair sum, which takes
                # its edge weights straight from the alpha / beta logits
                ws_all = ws                          # (`ws` is re-bound below: do not capture the name)

                def prologue(states, sif, z0):
                    lib.cell_prologue_pair(logits, ws_all, stackW, weffs, 3 * C_, C_, adv, arena, states,
                                           logits[0], logits[1], sif, z0)
            else:
                lib.cell_prologue(logits, ws, stackW, weffs, 3 * C_, C_, adv, arena)
        else:
            stats = None                                     # bn_finalize launches (no zero-filled sums)
            if head is not None:
                arena[stat_n:].zero_()
            lib.arch_softmax_multi(logits, None, ws, False)      # every arch tensor, one launch
        if alpha_is_logits:
            alpha_w, ws = ws[0], ws[1:]
        else:
            alpha_w = _c(alpha)
        beta_ws, gamma_ws = ws[0::2], ws[1::2]
        ctx.alpha_is_logits = alpha_is_logits
        out, sv = K.fusion_cell_fwd(xs, alpha_w, beta_ws, gamma_ws, CP, training, S,
                                    cell._multiplier, cell.args.node_steps, cell.args.node_multiplier, weffs,
                                    stats, head, prologue)
        ctx.cell, ctx.sv, ctx.beta_ws, ctx.gamma_ws, ctx.N, ctx.S = cell, sv, beta_ws, gamma_ws, N, S
        ctx.dev, ctx.n_head = dev, n_head
        if head is not None:
            K.LAST_HEAD.append(head)
        return out

    @staticmethod
    def backward(ctx, g):
        cell, sv, N, S = ctx.cell, ctx.sv, ctx.N, ctx.S
        dev = ctx.dev
        need_in = [ctx.needs_input_grad[5 + j] and not K._ARCH_ONLY[0] for j in range(N)]
        # one zero-filled arena for every gradient that is accumulated with atomics
        x0 = sv.states[0]
        CG, dalpha_w, dbeta_ws, dgamma_ws = cell.grad_pack(dev, sv.alpha_w, ctx.beta_ws, ctx.gamma_ws,
                                                           K.arch_shards(x0.shape[0], x0.shape[1], x0.shape[2]))
        ws, dws = [], []
        for i in range(S):
            ws += [ctx.beta_ws[i], ctx.gamma_ws[i]]
            dws += [dbeta_ws[i], dgamma_ws[i]]
        if ctx.alpha_is_logits:
            ws, dws = [sv.alpha_w] + ws, [dalpha_w] + dws
        # does anybody differentiate alpha / beta / gamma?  (a captured weight step does not — its optimizer holds the
        # network weights only, K.weight_grads_only — and the pass then ends without the arch-softmax backward)
        need_arch = ((bool(ctx.needs_input_grad[3]) or any(ctx.needs_input_grad[5 + N:5 + N + 2 * S]))
                     and not K._NO_ARCH[0])
        if not need_arch:
            ws, dws = [], []
        darch = [torch.empty_like(w) for w in ws]
        # does ANY parameter of the cell / the fused classifier need its gradient?  (a captured architecture step
        # says no: K.arch_grads_only)
        first_param = 5 + N + 2 * S
        want = any(ctx.needs_input_grad[first_param:]) and not K._ARCH_ONLY[0]
        saved, K.WANT_PARAM_GRADS = K.WANT_PARAM_GRADS, want
        saved_arch, K.WANT_ARCH_GRADS = K.WANT_ARCH_GRADS, need_arch
        try:
            dxs = K.fusion_cell_bwd(sv, g if sv.head is not None else _c(g), need_in, dalpha_w, dbeta_ws, dgamma_ws,
                                    CG, (ws, dws, darch))
        finally:
            K.WANT_PARAM_GRADS = saved
            K.WANT_ARCH_GRADS = saved_arch
        if not sv.epilogue_done and need_arch:
            lib.arch_softmax_multi(ws, dws, darch, True, CG.shards, CG.shard_stride)
        if not need_arch:
            dalpha, darch = None, [None] * (2 * S)
        elif ctx.alpha_is_logits:
            dalpha, darch = darch[0], darch[1:]
        else:
            # the caller owns the alpha softmax: hand back the sum of the atomic shards
            dalpha = torch.as_strided(dalpha_w, (CG.shards, *dalpha_w.shape),
                                      (CG.shard_stride, *dalpha_w.stride()),
                                      dalpha_w.storage_offset()).sum(0)
        head_grads = (sv.head.dW, sv.head.dbias) if ctx.n_head else ()
        pgrads = cell.grads_in_param_order(CG) if want else [None] * len(cell.param_list())
        return (None, None, None, dalpha, None, *dxs, *darch, *pgrads, *head_grads)


# ------------------------------------------------------- classifier + criterion epilogue
class LinearFn(Function):
    """F.linear(feat, W, bias) for the skinny central classifier (classes <= 128)."""

    @staticmethod
    def forward(ctx, feat, W, bias):
        _require_gpu(feat, 'central classifier')
        feat, W, bias = _c(_f32(feat)), _c(_f32(W)), _c(bias)
        b, Kd = feat.shape
        O = W.shape[0]
        # the k-slices add into `out`: zero-filled by the captured step's arena where there is one (no fill launch)
        if _ARENA_NEED is not None:
            _ARENA_NEED[0] += (b * O + 3) // 4 * 4
        pre = _ARENA.take(b * O) if (_ARENA is not None and _ARENA.buf.device == feat.device) else None
        out = pre.view(b, O) if pre is not None else torch.empty((b, O), device=feat.device, dtype=torch.float32)
        lib.linear_fwd(feat, W, bias, out, b, O, Kd, out_is_zero=pre is not None)
        ctx.save_for_backward(feat, W)
        return out

    @staticmethod
    def backward(ctx, g):
        feat, W = ctx.saved_tensors
        b, Kd = feat.shape
        O = W.shape[0]
        need = ctx.needs_input_grad
        dfeat = torch.empty_like(feat) if need[0] else None
        dW = torch.empty_like(W) if need[1] else None
        db = torch.empty(O, device=feat.device, dtype=torch.float32) if need[2] else None
        lib.linear_bwd(_c(g), None, feat, W, dfeat, dW, db, b, O, Kd)
        return dfeat, dW, db


_UNIT = {}


def unit_grad(device):
    """A cached 0-dim tensor holding 1.0 to pass as `grad_outputs` / `loss.backward(gradient=...)`.
    autograd otherwise launches a fill kernel for ones_like(loss) and our loss functions a multiply
    by it; when the incoming gradient IS this constant the criterion hands its stored dloss/dlogits
    back untouched (two ~4 us launches less per step; same numbers)."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    t = _UNIT.get(key)
    if t is None:
        t = _UNIT[key] = torch.ones((), device=f'{key[0]}:{key[1]}', dtype=torch.float32)
    return t


class _LossFn(Function):
    """mean loss with dloss/dlogits produced in the forward pass."""

    @staticmethod
    def _finish(ctx, loss, dz):
        ctx.save_for_backward(dz)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        (dz,) = ctx.saved_tensors
        u = _UNIT.get((gl.device.type, gl.device.index))
        if u is not None and gl.data_ptr() == u.data_ptr():
            return dz, None                      # gradient of the loss is the constant 1
        return dz * gl, None


class DeferredLossFn(Function):
    """The criterion of a fused head, evaluated by the head's BACKWARD launch (bmnas_head_bwd modes
    1 / 2): forward only records kind and labels and hands out the loss scalar that the backward
    will fill — valid once backward has run, which is all a captured step needs (nothing can read a
    value between two nodes of a hipGraph).  Enabled by bmnas.nn.fused_criterion()."""

    @staticmethod
    def forward(ctx, z, target, head, kind):
        if head.deferred is not None:
            raise lib.BmnasError('fused criterion: a second criterion on the same logits (the head evaluates ONE '
                                 'criterion in its backward launch); call it outside bmnas.nn.fused_criterion()')
        head.deferred = (kind, target if target.is_contiguous() else target.contiguous())
        head.gscale = None
        ctx.head = head
        return head.loss.view(())

    @staticmethod
    def backward(ctx, gl):
        head = ctx.head
        u = _UNIT.get((gl.device.type, gl.device.index))
        if not (u is not None and gl.data_ptr() == u.data_ptr()):
            head.gscale = gl.contiguous().float()          # d(loss)/d(loss) other than the constant 1
        return head.marker, None, None, None


class BCEWithLogitsFn(_LossFn):
    @staticmethod
    def forward(ctx, z, y):
        _require_gpu(z, 'BCEWithLogits loss')
        z, y = _c(_f32(z)), _c(_f32(y))
        loss = torch.empty(1, device=z.device, dtype=torch.float32)
        dz = torch.empty_like(z)
        lib.bce_logits(z, y, loss, dz)
        return _LossFn._finish(ctx, loss, dz)


class CrossEntropyFn(_LossFn):
    @staticmethod
    def forward(ctx, z, label):
        _require_gpu(z, 'CrossEntropy loss')
        z = _c(_f32(z))
        b, O = z.shape
        loss = torch.empty(1, device=z.device, dtype=torch.float32)
        dz = torch.empty_like(z)
        rows = torch.empty(b, device=z.device, dtype=torch.float32)
        lib.cross_entropy(z, label.contiguous(), loss, dz, rows, b, O)
        return _LossFn._finish(ctx, loss, dz)
