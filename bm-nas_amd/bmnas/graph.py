"""hipGraph capture of a launch-bound search step.

One fwd+bwd of the fusion hypernet is ~60 short kernels; eager Python issues them at
~2 ms/step while the GPU needs a fraction of that.  GraphedStep captures the whole step
(our HIP kernels are launched on torch's current stream, so they land in the capture like
any aten op) and replays it with one hipGraphLaunch.  Dropout stays fresh across replays:
the kernels add a DEVICE counter to their Philox offsets and the graph advances it.
"""
import torch

from . import cell as K
from .functions import arena_measure, deferred_affine, unit_grad


def _step_arena_on():
    """BMNAS_STEP_ARENA=0 (A/B runs): captured per-op steps keep their in-graph fill launches and counter add."""
    import os
    return os.environ.get('BMNAS_STEP_ARENA', '1') != '0'


class GraphedStep:
    """fn() must read its inputs from static tensors, set ``.grad = None`` on everything it
    differentiates (so the backward writes instead of accumulating) and return tensors that
    stay referenced (they become static graph outputs).

    Dropout under replay: the Philox offsets of the captured sites restart at 0 and the kernels add
    a device counter that the graph advances by the step's span on every replay.  Each instance
    starts its counter at `2^60 + (instance << 40)` (eager offsets count up from 0), so two graphs over the same sites (the weight step
    and the Architect step) never draw the same masks.  The seed is torch.initial_seed() AT
    CAPTURE: a later torch.manual_seed() does not reach replays (re-capture to reseed)."""

    _instances = 0

    def __init__(self, fn, warmup=3, external_advance=False, arena=None):
        """external_advance: the caller advances the dropout step counter itself in front of every replay (the launch
        that copies the batch in: `self.external = (counter, span)`), so a step without a fused cell prologue needs no
        `counter.add_` node at its end.  arena: the zero-filled fp32 scratch the captured pass carves its accumulators
        from (bmnas.functions.arena_use); the caller clears it in front of every replay."""
        quiet = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
        if quiet is not None:
            quiet(False)         # leaves outside the differentiated set keep nodes from the warm-up stream
        dev = torch.cuda.current_device()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        GraphedStep._instances += 1
        self.counter_base = (1 << 60) + (GraphedStep._instances << 40)
        self.counter = torch.full((1,), self.counter_base, dtype=torch.int64, device=f'cuda:{dev}')
        self.span_dev = torch.zeros(1, dtype=torch.int64, device=f'cuda:{dev}')
        saved = (K.DROP.offset, K.DROP.device_counter)
        K.DROP.offset, K.DROP.device_counter = 0, self.counter
        # the first fused-cell prologue of the step advances the counter (no launch of its own);
        # if the step has none, an add at the end of the graph does
        K.DROP.pending_advance = (self.counter, self.span_dev)
        self.graph = torch.cuda.CUDAGraph()
        self.external = None
        from .functions import arena_use
        try:
            with torch.cuda.graph(self.graph), arena_use(arena):
                out = fn()
                self.span = K.DROP.offset
                # True: a cell prologue advanced the counter in front of every dropout site of the step (they
                # all read the advanced value); False: the sites read the old value and this add ends the graph
                self.advanced_first = K.DROP.pending_advance is None
                if self.span > 0 and not self.advanced_first:
                    if external_advance:
                        # the caller's pre-replay launch adds the span: the sites then read the advanced value too
                        self.external = (self.counter, self.span)
                        self.advanced_first = True
                    else:
                        self.counter.add_(self.span)      # next replay draws new dropout masks
            self.span_dev.fill_(self.span)
            # keep the static storage, not the Python autograd graph that produced it (a retained
            # graph would get in the way of later captures: GraphedTrainStep._live_graph_tensors)
            det = lambda o: o.detach() if torch.is_tensor(o) else o
            self.outputs = type(out)(det(o) for o in out) if isinstance(out, (tuple, list)) else det(out)
            del out
        except BaseException:
            from . import functions
            functions.reset_pools()                   # see reset_pools: nothing of a dead capture may be handed out
            raise
        finally:
            K.DROP.offset, K.DROP.device_counter = saved
            K.DROP.pending_advance = None

    def replay(self):
        self.graph.replay()
        return self.outputs

    def site_step_value(self):
        """The value of the step counter that the dropout sites of the LAST replay read (synchronises)."""
        now = int(self.counter.item())
        return now if self.advanced_first else now - self.span


def classify_targets(model, targets):
    """(arch_only, no_arch) of an optimizer's tensors against model.arch_parameters(): every target IS an architecture
    tensor of the model / none of them is.  Both False when the model has no architecture tensors or no targets (a
    found-stage net, an empty optimizer): nothing is identified, the backward stays the full one."""
    arch_fn = getattr(model, 'arch_parameters', None)
    aids = {id(t) for t in arch_fn()} if callable(arch_fn) else set()
    targets = list(targets)
    if not targets or not aids:
        return False, False
    hits = sum(id(t) in aids for t in targets)
    return hits == len(targets), hits == 0


class GraphedTrainStep:
    """One optimisation step of the search loop — forward, criterion, backward and the Adam
    update — as ONE hipGraph replay ("HIP-graph capture of the whole
    fwd+bwd+allreduce+Adam step").  The reference's loops issue the same work eagerly
    (train_searchable/mmimdb.py:85-101 for the weights, architect.py:21-29 for alpha).

        step = GraphedTrainStep(model, criterion, optimizer, inputs, labels)
        loss, logits = step(inputs, labels)          # copies the batch in, replays, returns statics

    * The tensors differentiated are the optimizer's own (central_params for the weight phase,
      arch_parameters for the alpha phase); their gradients come from torch.autograd.grad, so no
      other leaf's .grad is touched — the stepped tensors end up exactly as after
      zero_grad + backward + step.
    * `optimizer`: a bmnas.optim.Adam (its step is capturable; per-batch learning rates and bias
      corrections stay exact: the scalars of a replay, 32 bytes per row, travel BY VALUE in the
      kernel arguments of the launch that copies the batch into the step's static tensors —
      `Adam.replay_blob()`, bmnas_copy_batch — so the captured step holds no H2D copy node and the
      host never waits for the previous replay.  Optimizers with more scalar rows than that blob
      holds (> 8) fall back to a copy node fed from a pinned staging buffer, guarded by
      `wait_staging()`).
    * Data parallel (optimizer passed through bmnas.dist.attach, world size > 1): the captured step
      writes the gradients straight into the reducer's flat bucket.  Under an RCCL process group the
      all-reduce(avg) of the bucket is a launch INSIDE the captured step (`bmnas_allreduce_f32`
      through the C-ABI communicator, `FlatGradAllReducer.plan() == 'native'`), followed by the
      one-launch Adam step in the same replay; under gloo, or when the communicator could not be
      created on every rank, the replay ends after the backward and the host issues the same
      collective (`reduce_bucket()`) and the Adam step.
    * The batch shape is fixed at capture: check `matches(inputs, labels)` and run a ragged last
      batch through the eager path.
    Returned `loss` / `logits` are static tensors that the next call overwrites."""

    @staticmethod
    def _live_graph_tensors(device):
        """Non-leaf tensors on `device` that are still referenced somewhere: evidence of an autograd
        graph from an earlier eager step being kept alive (a held `loss` / `logits`).  Such a graph
        pins the AccumulateGrad nodes of the parameters to the stream they were created on; the
        engine then makes that stream wait on the capture stream, the capture never joins, and
        HIP's capture_end segfaults instead of reporting it."""
        import gc
        import warnings
        n = 0
        with warnings.catch_warnings():
            # (the walk touches every live object, among them torch.distributed's deprecated `reduce_op` shim, whose
            # attribute access warns: one FutureWarning per capture in every log otherwise)
            warnings.simplefilter('ignore')
            for o in gc.get_objects():
                try:
                    if type(o) is torch.Tensor or type(o) is torch.nn.Parameter or isinstance(o, torch.Tensor):
                        if o.grad_fn is not None and o.device == device:
                            n += 1
                except Exception:                # noqa: BLE001 — objects in odd states during gc walk
                    pass
        return n

    def __init__(self, model, criterion, optimizer, inputs, labels, warmup=2, metric_forward=False, k=1):
        """k: optimisation steps per replay.  k > 1 captures k CONSECUTIVE steps — each over a static batch of its own,
        each with its own Adam scalars (learning rate, bias corrections: `stage(i, inputs, labels)` per batch, then
        `replay_staged()`) — into one graph: what separates two replays (~4 us of launch gap and ~30 us of host work)
        is then paid once per k batches.  The reference's loop takes its batches one by one
        (train_searchable/mmimdb.py:73-113); the k batches of a replay are processed in the same order with the same
        arithmetic.  Needs the update inside the graph.
        metric_forward: the replay ENDS with a gradient-free `criterion(model(inputs), labels)` over the same static
        batch, evaluated after the update — the dev phase of a search does exactly this after every `architect.step`
        (train_searchable/mmimdb.py:66-84: Architect.step, then the metric forward on the same batch, which sees the
        updated alphas).  One batch copy and one replay instead of two of each; `__call__` then returns
        (loss, logits, metric_loss, metric_logits).  Needs the update inside the graph (`in_graph_step`)."""
        import gc
        gc.collect()
        live = self._live_graph_tensors(labels.device)
        if live:
            raise RuntimeError(f'{live} tensor(s) of an earlier autograd graph are still referenced (a held '
                               'loss / output?): capturing now would tie the capture to their streams; '
                               'drop them (or call this before the first eager backward)')
        self.optimizer = optimizer
        self.targets = [p for g in optimizer.param_groups for p in g['params']]
        # the architecture step differentiates alpha / beta / gamma only -> the fused cell skips every weight-gradient
        # product of its backward (bmnas.cell.arch_grads_only).  Only on POSITIVE identification: every target IS one
        # of the model's architecture tensors.  (An optimizer over anything else that is not a module parameter —
        # learnable inputs, criterion parameters, a wrapper's own tensors — keeps the full backward.)
        # ... and the weight step none of them: no arch-softmax backward, no edge-weight dot products in the cell-level
        # K1 backward launches (bmnas.cell.weight_grads_only)
        self.arch_only, self.no_arch = classify_targets(model, self.targets)
        reducer = getattr(optimizer, '_bmnas_reducer', None)
        if reducer is not None and reducer.world <= 1 and not getattr(reducer, 'selftest', False):
            reducer = None
        self.reducer = reducer
        self.k = int(k)
        if self.k < 1:
            raise ValueError('k >= 1')
        # one static batch per captured step; slot 0 doubles as `static_batch()`
        self.slots = [([x.detach().clone().requires_grad_(x.requires_grad) for x in inputs], labels.detach().clone())
                      for _ in range(self.k)]
        self.inputs, self.labels = self.slots[0]
        self._staged = []
        self._batch_in = _BatchIn(self.inputs, self.labels)
        # RCCL through the C ABI (bmnas.dist.NativeComm; BMNAS_NATIVE_RCCL=0 turns it off) is a plain launch
        # on the capture stream: the all-reduce and the Adam step then live INSIDE the graph.
        # reducer.plan() is decided once, collectively, and is the same for captured and eager steps.
        self.native = reducer is not None and reducer.plan() == 'native'
        self.in_graph_step = reducer is None or self.native
        self.metric_forward = bool(metric_forward)
        if self.metric_forward and not self.in_graph_step:
            raise RuntimeError('metric_forward needs the optimizer step inside the captured graph')
        if self.k > 1 and not self.in_graph_step:
            raise RuntimeError('k steps per replay need the optimizer step inside the captured graph')
        views = reducer.ensure_bucket() if reducer is not None else None
        # with an averaging collective (RCCL) the captured step is the single-GPU one: unscaled loss, constant
        # unit gradient; otherwise (gloo) the loss is pre-scaled by 1/world and the bucket is summed
        scale = reducer.loss_scale if reducer is not None else 1.0
        # (the scale is a constant of the capture: `matches` refuses a batch whose shard weight differs — an uneven
        # scatter of another global batch size that happens to give this rank the captured shape)
        self.loss_scale = scale
        armed = [False]

        from . import nn as bnn

        def fn():
            if self.k == 1:
                return one(self.inputs, self.labels)
            out = ()
            for xs_i, y_i in self.slots:             # (loss_0, logits_0[, metric_loss_0, metric_logits_0], loss_1, ...)
                out += one(xs_i, y_i)
            return out

        def one(xs_, y_):
            # the criterion of a fused head is evaluated by the head's backward launch: the static
            # `loss` output is complete when the replay is (bmnas.nn.fused_criterion)
            with bnn.fused_criterion():
                logits = model(xs_)
                if isinstance(logits, tuple):
                    logits = logits[-1]
                loss = criterion(logits, y_)
            # (deferred_affine: the per-op path's LayerNorm-affine reductions of this pass as ONE launch at its end)
            with K.arch_grads_only(self.arch_only), K.weight_grads_only(self.no_arch), deferred_affine():
                if scale != 1.0:
                    grads = torch.autograd.grad(loss * scale, self.targets, allow_unused=True)
                else:
                    grads = torch.autograd.grad(loss, self.targets, grad_outputs=unit_grad(loss.device),
                                                allow_unused=True)
            if reducer is not None:
                have = [(v, g) for v, g in zip(views, grads) if g is not None]
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
                for t, v, g in zip(self.targets, views, grads):
                    t.grad = v if g is not None else None
            else:
                for t, g in zip(self.targets, grads):
                    t.grad = g
            if self.native:
                # only in the capture: a warm-up pass that raised on one rank must not leave the others waiting
                # inside a real collective (the communicator's channels were set up at its creation)
                if torch.cuda.is_current_stream_capturing():
                    reducer.reduce_bucket()                     # captured: a launch on this stream
                reducer.reduced = True                          # the step pre-hook must not reduce again
            if self.in_graph_step and armed[0]:
                optimizer.step()
            if self.native:
                reducer.reduced = False
            if self.metric_forward:
                # behind the Adam launch on the same stream: this forward's prologue reads the UPDATED architecture
                # tensors; its criterion is the eager kernel (nothing follows that could evaluate a deferred one)
                with torch.no_grad():
                    mout = model(xs_)
                    if isinstance(mout, tuple):
                        mout = mout[-1]
                    mloss = criterion(mout, y_)
                return loss, logits, mloss, mout
            return loss, logits

        # The warm-up passes run WITHOUT the update (they settle allocations and lazy
        # initialisation; extra optimizer steps on the example batch would change training) and
        # their BatchNorm running-statistics updates are undone afterwards.
        state = {k: v.clone() for k, v in model.state_dict().items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):
                    fn()
                # dress rehearsal: anything that synchronises with the host (.item(), .cpu(), nonzero,
                # ...) cannot be captured.  Find out HERE, with torch's sync detector, not by letting
                # a real capture fail: a capture that dies half-way was seen to leave the HIP runtime
                # in a state where a later, unrelated capture segfaults.
                mode = torch.cuda.get_sync_debug_mode()
                torch.cuda.set_sync_debug_mode('error')
                try:
                    with arena_measure() as need:        # ... and what zero-filled scratch a pass of the step takes
                        fn()
                finally:
                    torch.cuda.set_sync_debug_mode(mode)
        except BaseException:
            torch.cuda.synchronize()
            model.load_state_dict(state)
            raise
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        try:
            if self.in_graph_step:
                # no H2D node in the graph: the scalars (of every one of the k steps) ride with the batch copy
                optimizer.capture_safe(poke=True, slots=self.k)
            armed[0] = True
            # the step's accumulation arena: cleared, like the dropout step counter advanced, by the launch that copies
            # the batch in front of every replay — no fill / add launches inside the step (bmnas.functions._StepArena)
            self._arena = (torch.zeros(need.need, device=self.labels.device, dtype=torch.float32)
                           if need.need and _step_arena_on() else None)
            self._g = GraphedStep(fn, warmup=0, external_advance=_step_arena_on(), arena=self._arena)
            flat = [t for xs_i, y_i in self.slots for t in list(xs_i) + [y_i]]
            self._batch_in = _BatchIn(flat[:-1], flat[-1], zero=[self._arena], advance=self._g.external)
            # the graph reads THIS plan's staging buffers and writes THESE gradient tensors for good
            self.plan = optimizer.captured_plan() if self.in_graph_step else None
            self.static_grads = [t.grad for t in self.targets]
        finally:
            model.load_state_dict(state)        # also when the capture fails and the caller stays eager

    @staticmethod
    def try_build(model, criterion, optimizer, inputs, labels, logger=None, metric_forward=False, k=1):
        """-> a GraphedTrainStep, or False when this step cannot be captured (inputs that are not a
        flat list of tensors, a module that synchronises with the host, ...); callers then keep
        the eager path."""
        if not (isinstance(inputs, (list, tuple)) and all(torch.is_tensor(x) for x in inputs)
                and torch.is_tensor(labels)):
            return False
        from .dist import all_ranks_agree
        try:
            step = GraphedTrainStep(model, criterion, optimizer, inputs, labels, metric_forward=metric_forward, k=k)
        except Exception as e:                       # noqa: BLE001 — capture errors are of many types
            torch.cuda.synchronize()
            from . import functions
            functions.reset_pools()                  # buffers handed out inside the dead capture are gone
            if logger is not None:
                logger.info('hipGraph capture of the step failed ({}: {}); staying eager'.format(
                    type(e).__name__, e))
            step = False
        # data parallel: every rank replays, or none does (capture success is decided per rank; the eager
        # reducer issues the same collective as a replaying rank, but the step's structure — where Adam
        # runs, which buffers hold the gradients — should not differ between replicas either)
        if not all_ranks_agree(bool(step), labels.device):
            if step and logger is not None:
                logger.info('another rank could not capture its step: staying eager on every rank')
            step = False
        return step

    @staticmethod
    def enabled(args):
        """Whether the trainer loops replay their steps as hipGraphs: `args.hip_graph` if the caller
        set it, else the BMNAS_HIP_GRAPH environment variable, else ON (a step that cannot be
        captured falls back to eager by itself).  Under data parallelism the captured step writes
        its gradients into the reducer's flat bucket; the replay is followed by one all-reduce and
        the one-launch Adam step."""
        import os
        v = getattr(args, 'hip_graph', None)
        if v is None and os.environ.get('BMNAS_HIP_GRAPH') is not None:
            v = os.environ['BMNAS_HIP_GRAPH'] not in ('0', '', 'false', 'False')
        if v is None:
            v = True
        return bool(v) and torch.cuda.is_available()

    def matches(self, inputs, labels):
        if self.reducer is not None and abs(self.reducer.loss_scale - self.loss_scale) > 1e-12:
            return False
        return (len(inputs) == len(self.inputs) and labels.shape == self.labels.shape and
                all(a.shape == b.shape for a, b in zip(inputs, self.inputs)))

    def static_batch(self):
        """(inputs, labels) the captured step reads: a producer that writes the batch INTO these tensors (and then
        passes them to `__call__`) saves the copy itself — 5.9 us of kernel for the 9.4 MB of an MM-IMDB b128 batch, ~1 us net:
        it overlaps the gap between two replays (bench.py `input_copy_us`)."""
        return self.inputs, self.labels

    def stage(self, i, inputs, labels):
        """k > 1: hand over batch i (0 ... k - 1, in order) of the next replay.  Call it AFTER the scheduler has set the
        learning rates this batch's step runs with: the step's Adam scalars are fixed here.  -> those learning rates, as
        staged (one per scalar row)."""
        opt = self.optimizer
        if i == 0:
            opt.activate(self.plan)          # an eager step in between must not leak into the replay
            self._staged = []
        if i != len(self._staged) // (len(self.inputs) + 1):
            raise RuntimeError('GraphedTrainStep.stage: batches must be staged in order, one call per slot')
        opt.prepare_replay(slot=i)
        self._staged += list(inputs) + [labels]
        # (decoded here, right after this slot's rows were written: `count` still is this step's)
        return opt.staged_lrs(slot=i)

    def replay_staged(self):
        """-> [(loss, logits[, metric_loss, metric_logits])] * k of the k staged batches (static tensors, overwritten by
        the next replay)."""
        opt = self.optimizer
        if len(self._staged) != self.k * (len(self.inputs) + 1):
            raise RuntimeError(f'GraphedTrainStep.replay_staged: {self.k} batches must be staged first')
        # ONE launch in front of the replay: the k batches into the static tensors + the k steps' Adam scalars
        self._batch_in(self._staged[:-1], self._staged[-1], opt.replay_blob())
        self._staged = []
        out = self._g.replay()
        opt.mark_launched()
        n = 4 if self.metric_forward else 2
        return [tuple(out[n * i:n * (i + 1)]) for i in range(self.k)]

    def __call__(self, inputs, labels):
        opt = self.optimizer
        if self.k != 1:
            raise RuntimeError('a k-step GraphedTrainStep is driven by stage() / replay_staged()')
        if self.in_graph_step:
            opt.activate(self.plan)          # an eager step in between must not leak into the replay
            opt.prepare_replay()
            # ONE launch in front of the replay: the batch into the static tensors + this step's Adam scalars (by value)
            self._batch_in(inputs, labels, opt.replay_blob())
            out = self._g.replay()
            opt.mark_launched()
            return out                       # (loss, logits[, metric_loss, metric_logits])
        else:
            self._batch_in(inputs, labels)
            loss, logits = self._g.replay()
            # an eager step in between (ragged last batch) re-pointed .grad at its own tensors:
            # the optimizer must read the bucket views the graph has just written
            for t, g in zip(self.targets, self.static_grads):
                t.grad = g
            self.reducer.all_reduce_bucket()
            opt.step()
        return loss, logits


class _BatchIn:
    """The batch into a graph's static tensors: ONE launch (bmnas_copy_batch: features and labels of any dtypes together,
    plus — `blob` — a captured optimizer step's per-replay scalars by value) for everything that is already on the
    device, plain copies for anything else (host tensors: the copy IS the H2D transfer; a dtype / layout change:
    torch's converting copy).  Tensors that already ARE the static ones (GraphedTrainStep.static_batch()) cost nothing."""

    def __init__(self, static_inputs, static_labels, zero=(), advance=None):
        from . import lib
        self.copier = lib.BatchCopier(list(static_inputs) + [static_labels], zero=zero, advance=advance)

    def __call__(self, inputs, labels, blob=None):
        with torch.no_grad():
            for dst, src in self.copier(list(inputs) + [labels], blob):
                dst.copy_(src, non_blocking=True)


class GraphedForward:
    """`output = model(inputs); loss = criterion(output, labels)` WITHOUT gradients as one hipGraph replay: the
    metric pass of the search loop's dev phase (after `architect.step`, reference train_searchable/mmimdb.py:66-84
    with `phase == 'dev'`: model.train(), so dropout is live and BatchNorm keeps updating — both stay so under
    replay) and the eval / test passes (`test_*_track_acc`).  Issued eagerly that forward is host-bound: 0.55 ms for
    reshape layers + hypernet + classifier + criterion at MM-IMDB batch 128, 0.86 ms at NTU batch 8, against 0.11 ms
    as a replay.  The module's mode (train / eval) and the batch shape are fixed at capture: `matches()`.

        fwd = GraphedForward.try_build(model, criterion, inputs, labels)
        loss, output = fwd(inputs, labels)            # static tensors, overwritten by the next call"""

    def __init__(self, model, criterion, inputs, labels, warmup=2):
        self.training = model.training
        self.inputs = [x.detach().clone() for x in inputs]
        self.labels = labels.detach().clone()
        self._batch_in = _BatchIn(self.inputs, self.labels)

        def fn():
            with torch.no_grad():
                out = model(self.inputs)
                if isinstance(out, tuple):
                    out = out[-1]
                return criterion(out, self.labels), out

        # warm-up passes must not change training: their BatchNorm updates are undone (eval mode updates nothing:
        # no snapshot, no restore)
        state = {k: v.clone() for k, v in model.state_dict().items()} if model.training else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):
                    fn()
                mode = torch.cuda.get_sync_debug_mode()      # dress rehearsal (see GraphedTrainStep)
                torch.cuda.set_sync_debug_mode('error')
                try:
                    with arena_measure() as need:
                        fn()
                finally:
                    torch.cuda.set_sync_debug_mode(mode)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._arena = (torch.zeros(need.need, device=self.labels.device, dtype=torch.float32)
                           if need.need and _step_arena_on() else None)
            self._g = GraphedStep(fn, warmup=0, external_advance=_step_arena_on(), arena=self._arena)
            self._batch_in = _BatchIn(self.inputs, self.labels, zero=[self._arena], advance=self._g.external)
        finally:
            torch.cuda.synchronize()
            if state is not None:
                model.load_state_dict(state)

    @staticmethod
    def try_build(model, criterion, inputs, labels, logger=None):
        """-> a GraphedForward, or False (inputs that are not a flat list of device tensors, a module that
        synchronises with the host, a capture error): the caller keeps the eager forward.  No collective is
        involved, so under data parallelism every rank decides for itself."""
        if not (isinstance(inputs, (list, tuple)) and all(torch.is_tensor(x) and x.is_cuda for x in inputs)
                and torch.is_tensor(labels) and labels.is_cuda):
            return False
        try:
            return GraphedForward(model, criterion, inputs, labels)
        except Exception as e:                       # noqa: BLE001 — capture errors are of many types
            torch.cuda.synchronize()
            from . import functions
            functions.reset_pools()
            if logger is not None:
                logger.info('hipGraph capture of the forward pass failed ({}: {}); staying eager'.format(
                    type(e).__name__, e))
            return False

    def matches(self, model, inputs, labels):
        return (model.training == self.training and isinstance(inputs, (list, tuple))
                and len(inputs) == len(self.inputs) and labels.shape == self.labels.shape
                and labels.dtype == self.labels.dtype
                and all(torch.is_tensor(a) and a.shape == b.shape and a.dtype == b.dtype
                        for a, b in zip(inputs, self.inputs)))

    def __call__(self, inputs, labels):
        self._batch_in(inputs, labels)
        return self._g.replay()
