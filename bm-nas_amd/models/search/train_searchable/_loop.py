"""The epoch / phase loop shared by the three dataset trainers.

Behaviour follows the reference's train_searchable/{mmimdb,ntu,ego}.py: phase order, model.train()
in BOTH the train and dev phases of a search, Architect.step on every dev batch followed by a
no-grad forward for the metric, per-batch cosine schedule, best-dev checkpoint + genotype pickle.
Differences, all on the host side of the hot path:
  * the per-batch .item() / .cpu() syncs are gone: loss and metric statistics accumulate on the
    device and are read once per phase (same numbers, no stall per batch);
  * `parallel` means one process per GPU (bmnas.dist); statistics are summed over ranks and only
    rank 0 writes checkpoints — there is no `.module` indirection;
  * the weight step (forward + criterion + backward + Adam) and the Architect step are captured
    once and replayed as one hipGraph launch per batch (bmnas.graph.GraphedTrainStep); a ragged
    last batch, or a model that cannot be captured, runs the eager path.  On by default for
    single-process runs; `args.hip_graph` / BMNAS_HIP_GRAPH switch it (GraphedTrainStep.enabled);
  * the gradient-free passes — the metric forward of every dev batch after `architect.step`, the
    eval / test passes — are hipGraph replays too (bmnas.graph.GraphedForward, one graph per module
    mode and batch shape): issued eagerly that forward is host-bound, 0.55-0.86 ms against 0.11 ms.
"""
import copy
import os

import torch
import torch.distributed as dist

import models.auxiliary.scheduler as sc
from models.search.darts.utils import count_parameters, save, save_pickle


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _all_sum(t):
    if _world() > 1:
        from bmnas import dist as bdist
        bdist.all_reduce(t, dist.ReduceOp.SUM)
    return t


def _model_device(model, device):
    """Under data parallelism every rank's replica sits on its own GPU (search_setup places it by
    LOCAL_RANK) while the unchanged mains hand `cuda:0` to every rank: follow the model."""
    if _world() > 1:
        for p in model.parameters():
            return p.device
    return device


def _is_sharded(loader):
    from torch.utils.data.distributed import DistributedSampler
    return isinstance(getattr(loader, 'sampler', None), DistributedSampler)


def _shard_batch(inputs, labels):
    """A loader without a DistributedSampler yields the GLOBAL batch on every rank: keep this rank's contiguous slice,
    cut the way nn.DataParallel's scatter cuts it (mmimdb_darts_searchable.py:36-37; `bmnas.dist.uneven_bounds`:
    Tensor.chunk — ceil(n / world) samples per replica until the batch is used up, so the last replicas may get fewer
    or none).  The reference's loss is ONE mean over the n gathered outputs; a rank's mean over its own n_rank samples
    enters the gradient average with weight n_rank * world / n (`bmnas.dist.set_shard_weight`, applied by the reducer:
    eager steps scale the bucket, captured steps their loss — and are replayed only for the weight they were captured
    with).  A rank left without samples idles through the batch (`_idle_step`): zero gradients into the same
    collective, the same averaged update."""
    from bmnas import dist as bdist
    world, rank = _world(), _rank()
    n = labels.shape[0]
    start, length = bdist.uneven_bounds(n, rank, world)
    bdist.set_shard_weight(length * world / n if n else 1.0)
    cut = lambda t: t.narrow(0, start, length) if torch.is_tensor(t) and t.dim() > 0 else t
    if isinstance(inputs, (tuple, list)):
        inputs = type(inputs)(cut(t) for t in inputs)
    else:
        inputs = cut(inputs)
    return inputs, cut(labels)


def _weight():
    from bmnas import dist as bdist
    return bdist.shard_weight()


def _set_weight(w):
    from bmnas import dist as bdist
    bdist.set_shard_weight(w)


def _idle_step(optimizer):
    """This rank got no sample of the batch (uneven scatter): it still joins the step's ONE collective — with zero
    gradients, its shard weight is 0 — and applies the averaged gradients like every other rank."""
    for group in optimizer.param_groups:
        for p in group['params']:
            p.grad = torch.zeros_like(p)
    optimizer.step()


class AccuracyMeter:
    name = 'Acc'

    def reset(self, device):
        self.correct = torch.zeros((), device=device, dtype=torch.float64)

    def update(self, output, labels):
        self.correct += (output.argmax(1) == labels).sum()

    def compute(self, n):
        return float(_all_sum(self.correct.clone())) / n


class F1Meter:
    """weighted / macro / ... F1 of sigmoid(output) > threshold, like the reference's sklearn call."""

    def __init__(self, f1_type='weighted', th=0.3):
        self.f1_type, self.th = f1_type, th
        self.name = f'{f1_type} F1'

    def reset(self, device):
        self.preds, self.labels = [], []

    def update(self, output, labels):
        self.preds.append(torch.sigmoid(output) > self.th)
        self.labels.append(labels)

    def compute(self, n):
        from sklearn.metrics import f1_score
        y_pred = torch.cat(self.preds).cpu().numpy()
        y_true = torch.cat(self.labels).cpu().numpy()
        if _world() > 1:                       # gather every rank's shard on every rank
            objs = [None] * _world()
            dist.all_gather_object(objs, (y_pred, y_true))
            import numpy as np
            y_pred = np.concatenate([o[0] for o in objs])
            y_true = np.concatenate([o[1] for o in objs])
        return float(f1_score(y_true, y_pred, average=self.f1_type, zero_division=1))


def fusion_params(model):
    n = sum(count_parameters(layer) for layer in model.reshape_layers)
    return n + count_parameters(model.fusion_net)


class _ForwardGraphs:
    """The gradient-free passes of the loops (metric pass of the dev phase, eval / test) as hipGraph replays
    (bmnas.graph.GraphedForward): one graph per (module mode, batch shape), at most three capture attempts; anything
    that does not fit — a ragged last batch, host tensors, a module that cannot be captured — stays eager."""

    def __init__(self, args, logger=None):
        from bmnas.graph import GraphedTrainStep
        self.on = GraphedTrainStep.enabled(args)
        self.graphs, self.attempts, self.logger = [], 0, logger
        self.replays = 0

    @staticmethod
    def of(model, criterion, args, logger=None):
        """The graphs of (model, criterion), kept ON the model across run() / evaluate() calls: a capture costs three
        eager warm-ups, a rehearsal and a private pool — more than a short eval loader saves if every epoch's eval
        pass captured again.  (The graphs read the parameters in place: optimizer steps and load_state_dict are seen.)"""
        cache = model.__dict__.setdefault('_bmnas_forward_graphs', {})
        # keyed by id(), but the entry HOLDS its criterion: an id can be re-used by a new object once the old one is
        # freed (a criterion rebuilt per stage with another pos_weight), and a captured graph has its criterion baked
        # in — an entry whose criterion is not this very object is dropped together with its graphs' pools
        entry = cache.get(id(criterion))
        if entry is not None and entry[0] is not criterion:
            del cache[id(criterion)]
            entry = None
        if entry is None:
            entry = cache[id(criterion)] = (criterion, _ForwardGraphs(args, logger))
        fg = entry[1]
        from bmnas.graph import GraphedTrainStep
        fg.on = GraphedTrainStep.enabled(args)
        fg.replays = 0
        return fg

    def __call__(self, model, criterion, inputs, labels):
        """-> (loss, output) from a replay, or None: run the pass eagerly."""
        if not self.on:
            return None
        for g in self.graphs:
            if g.matches(model, inputs, labels):
                self.replays += 1
                return g(inputs, labels)
        if self.attempts >= 3:
            return None
        self.attempts += 1
        from bmnas.graph import GraphedForward
        g = GraphedForward.try_build(model, criterion, inputs, labels, self.logger)
        if not g:
            return None
        self.graphs.append(g)
        self.replays += 1
        return g(inputs, labels)


def steps_per_replay(args):
    """How many weight steps the trainer captures into ONE hipGraph replay (`args.steps_per_replay`, else the environment
    variable BMNAS_STEPS_PER_REPLAY, else 1; at most 8).  With k > 1 the loop keeps k batches resident and replays a
    k-step graph — same batches, same order, same per-batch learning rates as the reference's one-by-one loop
    (train_searchable/mmimdb.py:73-113) —; a ragged tail of fewer than k batches runs as single steps."""
    v = getattr(args, 'steps_per_replay', None)
    if v is None:
        v = os.environ.get('BMNAS_STEPS_PER_REPLAY')
    try:
        k = int(v) if v is not None else 1
    except (TypeError, ValueError):
        k = 1
    return max(1, min(k, 8))


def _observe(event, **info):
    """tests/test_outer_loop_golden_gpu.py pins the loop against the reference's trainers through this hook
    (run.observer = callable(event, **info)); None in production: no host synchronisation is added."""
    obs = getattr(run, 'observer', None)
    if obs is not None:
        obs(event, **info)


def run(model, architect, criterion, optimizer, scheduler, dataloaders, dataset_sizes, device,
        num_epochs, logger, plotter, args, status, unpack, meter, eval_phases, better, task=None,
        nan_escape=False, init=None):
    """-> dict(best_dev, best_dev_genotype, best_test, best_test_genotype, last_genotype, nan_abort).
    init: what the first epoch's metric is compared with (the reference starts from best_f1 = init_f1 with `>`,
    train_searchable/mmimdb.py:19,162, and from best_acc = 0 with `>=`, ntu.py:18,135); None: always accepted."""
    cosine = isinstance(scheduler, sc.LRCosineAnnealingScheduler)
    device = _model_device(model, device)
    from bmnas.graph import GraphedTrainStep
    use_graph = GraphedTrainStep.enabled(args)
    w_graph, w_attempts = None, 0
    k_steps = steps_per_replay(args) if use_graph else 1
    wk_graph, wk_attempts = None, 0
    f_graphs = _ForwardGraphs.of(model, criterion, args, logger)
    stats = run.stats = dict(graph_replays=0, eager_steps=0, forward_replays=0, k_step_replays=0)
    best = dict(best_dev=None, best_dev_genotype=None, best_dev_epoch=0, best_test=None,
                best_test_genotype=None, best_test_epoch=0, last_genotype=None, nan_abort=False)
    for epoch in range(num_epochs):
        logger.info('Epoch: {}'.format(epoch))
        logger.info('EXP: {}'.format(args.save))
        phases = ['train', 'dev'] if status == 'search' else eval_phases
        for phase in phases:
            if phase == 'train':
                if not cosine:
                    scheduler.step()
                if architect is not None:
                    architect.log_learning_rate(logger)
                model.train()
            elif phase == 'dev':
                if status == 'eval' and not cosine:
                    scheduler.step()
                model.train()                  # the reference keeps BN/dropout in train mode here
            else:
                model.eval()
            meter.reset(device)
            loss_sum = torch.zeros((), device=device, dtype=torch.float64)
            seen = 0
            learn = phase == 'train' or (phase == 'dev' and status == 'eval')
            loader = dataloaders[phase]
            split = _world() > 1 and not _is_sharded(loader)
            if _world() > 1 and _is_sharded(loader):
                loader.sampler.set_epoch(epoch)          # a new shuffle per epoch, the same on every rank
            _set_weight(1.0)                             # (a shard weight belongs to ONE scattered batch: none carries over)
            def one_batch(inputs, labels):
                nonlocal w_graph, w_attempts, loss_sum
                # nothing of the previous batch's autograd graph may stay referenced while a step is
                # being captured (see GraphedTrainStep._live_graph_tensors)
                output = loss = None
                got = None
                if labels.size(0) == 0:
                    # an idle replica of an uneven scatter: no forward, the collectives of the batch only
                    if status == 'search' and phase in ('dev', 'test') and architect is not None:
                        _idle_step(architect.optimizer)
                    if learn:
                        if cosine:
                            scheduler.step()
                            scheduler.update_optimizer(optimizer)
                        _idle_step(optimizer)
                    stats['idle_steps'] = stats.get('idle_steps', 0) + 1
                    return
                if status == 'search' and phase in ('dev', 'test') and architect is not None:
                    # (captured: the metric forward below rides at the end of the architecture step's replay — one
                    # batch copy and one hipGraph launch for both; None: it did not, evaluate it here)
                    got = architect.step(inputs, labels, logger, metric=not learn and use_graph and f_graphs.on)
                if learn and use_graph:
                    if w_graph is None and w_attempts < 3:
                        w_attempts += 1
                        w_graph = GraphedTrainStep.try_build(model, criterion, optimizer, inputs, labels,
                                                             logger) or None
                    if w_graph and w_graph.matches(inputs, labels):
                        if cosine:
                            scheduler.step()
                            scheduler.update_optimizer(optimizer)
                        loss, output = w_graph(inputs, labels)
                        loss_sum += loss.detach().double() * labels.size(0)
                        meter.update(output.detach(), labels)
                        stats['graph_replays'] += 1
                        _observe('batch', epoch=epoch, phase=phase, loss=loss, output=output, optimizer=optimizer,
                                 learn=True, how='graph')
                        return
                if not learn:
                    # the metric pass: no gradients, one replay (the step above — architect.step in the dev phase —
                    # has already happened: this forward sees the updated alphas, like the reference's)
                    if got is None:
                        got = f_graphs(model, criterion, inputs, labels)
                    else:
                        stats['merged_metric_replays'] = stats.get('merged_metric_replays', 0) + 1
                    if got is not None:
                        loss, output = got
                        loss_sum += loss.detach().double() * labels.size(0)
                        meter.update(output.detach(), labels)
                        stats['forward_replays'] += 1
                        _observe('batch', epoch=epoch, phase=phase, loss=loss, output=output, optimizer=optimizer,
                                 learn=False, how='graph')
                        return
                stats['eager_steps'] += 1
                optimizer.zero_grad()
                with torch.set_grad_enabled(learn):
                    output = model(inputs)
                    if isinstance(output, tuple):
                        output = output[-1]
                    loss = criterion(output, labels)
                    if learn:
                        if cosine:
                            scheduler.step()
                            scheduler.update_optimizer(optimizer)
                        loss.backward()
                        optimizer.step()
                loss_sum += loss.detach().double() * labels.size(0)
                meter.update(output.detach(), labels)
                _observe('batch', epoch=epoch, phase=phase, loss=loss, output=output, optimizer=optimizer,
                         learn=learn, how='eager')

            def k_batches(batches):
                """k resident batches as ONE replay of a k-step graph (each step with the learning rate the scheduler gives
                its batch); anything that does not fit — no capture, another batch shape — runs batch by batch."""
                nonlocal wk_graph, wk_attempts, loss_sum
                x0, y0 = batches[0]
                if wk_graph is None and wk_attempts < 3:
                    wk_attempts += 1
                    wk_graph = GraphedTrainStep.try_build(model, criterion, optimizer, x0, y0, logger, k=k_steps) or None
                if not (wk_graph and all(wk_graph.matches(x_, y_) for x_, y_ in batches)):
                    for x_, y_ in batches:
                        one_batch(x_, y_)
                    return
                lrs = []
                for j, (x_, y_) in enumerate(batches):
                    if cosine:
                        scheduler.step()
                        scheduler.update_optimizer(optimizer)
                    lrs.append(wk_graph.stage(j, x_, y_))
                stats['k_step_replays'] += 1
                for (loss, output), (x_, y_), lr in zip(wk_graph.replay_staged(), batches, lrs):
                    loss_sum += loss.detach().double() * y_.size(0)
                    meter.update(output.detach(), y_)
                    stats['graph_replays'] += 1
                    # (lr: the rates THIS batch's step was staged with — the optimizer's own have moved on to the last
                    # batch of the group by now)
                    _observe('batch', epoch=epoch, phase=phase, loss=loss, output=output, optimizer=optimizer,
                             learn=True, how='graph', lr=lr)

            def k_arch_batches(batches):
                """The dev phase of a search: k (architecture step + metric forward) pairs as ONE replay."""
                nonlocal loss_sum
                outs = architect.step_k(batches, logger)
                if outs is None:
                    for x_, y_ in batches:
                        one_batch(x_, y_)
                    return
                stats['k_step_replays'] += 1
                for (loss, output), (x_, y_) in zip(outs, batches):
                    loss_sum += loss.detach().double() * y_.size(0)
                    meter.update(output.detach(), y_)
                    stats['forward_replays'] += 1
                    stats['merged_metric_replays'] = stats.get('merged_metric_replays', 0) + 1
                    _observe('batch', epoch=epoch, phase=phase, loss=loss, output=output, optimizer=optimizer,
                             learn=False, how='graph')

            # which batches wait for a k-step replay: the weight steps of a learning phase; the (architecture step +
            # metric forward) pairs of a search's dev phase
            arch_phase = (status == 'search' and phase in ('dev', 'test') and architect is not None and not learn
                          and f_graphs.on and hasattr(architect, 'step_k'))
            group = k_batches if learn else (k_arch_batches if arch_phase else None)
            held, held_w = [], []                       # batches waiting for their k-step replay (+ their shard weights)

            def flush():
                # (a shard weight belongs to ITS batch: uneven scatters of different global sizes must not share a replay)
                nonlocal held, held_w
                if len(held) == k_steps and len(set(held_w)) == 1:
                    _set_weight(held_w[0])
                    group(held)
                else:                                   # the ragged tail of a phase: single steps
                    for (x_, y_), w_ in zip(held, held_w):
                        _set_weight(w_)
                        one_batch(x_, y_)
                held, held_w = [], []

            for data in loader:
                inputs, labels = unpack(data, device)
                if split:
                    inputs, labels = _shard_batch(inputs, labels)
                seen += labels.size(0)
                if group is not None and k_steps > 1 and isinstance(inputs, (list, tuple)) and labels.size(0):
                    held.append((inputs, labels))
                    held_w.append(_weight())
                    if len(held) == k_steps:
                        flush()
                    continue
                w_now = _weight()
                flush()                                 # keep the batch order
                _set_weight(w_now)
                one_batch(inputs, labels)
            flush()
            _set_weight(1.0)
            n = dataset_sizes[phase]
            if _world() > 1:
                # what the ranks processed together (a DistributedSampler pads; a split covers every sample)
                n = int(_all_sum(torch.tensor(float(seen), device=device, dtype=torch.float64)))
            epoch_loss = float(_all_sum(loss_sum)) / n
            epoch_metric = meter.compute(n)
            logger.info('{} Loss: {:.4f}, {}: {:.4f}'.format(phase, epoch_loss, meter.name, epoch_metric))
            logger.info('Fusion Model Params: {}'.format(fusion_params(model)))
            genotype = model.genotype()
            best['last_genotype'] = genotype
            logger.info(str(genotype))
            _observe('phase', epoch=epoch, phase=phase, loss=epoch_loss, metric=epoch_metric, genotype=genotype)
            if nan_escape and phase == 'train' and epoch_loss != epoch_loss:
                logger.info('Nan loss during training, escaping')
                model.eval()
                best['nan_abort'] = True
                return best
            for which, fname in (('dev', 'best_model.pt'), ('test', 'best_test_model.pt')):
                if phase != which or (which == 'dev' and status != 'search' and eval_phases[-1] != 'dev'):
                    continue
                key = 'best_' + which
                ref = best[key] if best[key] is not None else init
                if ref is None or better(epoch_metric, ref):
                    best[key] = epoch_metric
                    best[key + '_genotype'] = copy.deepcopy(genotype)
                    best[key + '_epoch'] = epoch
                    if _rank() == 0:
                        save(model, os.path.join(args.save, 'best', fname))
                        save_pickle(best[key + '_genotype'],
                                    os.path.join(args.save, 'best', fname.replace('model.pt', 'genotype.pkl')))
        plotter.plot(best['last_genotype'], os.path.join(args.save, 'architectures', 'epoch_{}'.format(epoch)),
                     task=task)
        logger.info('Current best dev {}: {}, at training epoch: {}'.format(meter.name, best['best_dev'],
                                                                          best['best_dev_epoch']))
        logger.info('Current best test {}: {}, at training epoch: {}'.format(meter.name, best['best_test'],
                                                                           best['best_test_epoch']))
    return best


@torch.no_grad()
def evaluate(model, criterion, loader, n, device, logger, args, unpack, meter, phase='test'):
    model.eval()
    logger.info('EXP: {}'.format(args.save))
    device = _model_device(model, device)
    meter.reset(device)
    loss_sum = torch.zeros((), device=device, dtype=torch.float64)
    split = _world() > 1 and not _is_sharded(loader)
    seen = 0
    f_graphs = _ForwardGraphs.of(model, criterion, args, logger)
    for data in loader:
        inputs, labels = unpack(data, device)
        if split:
            inputs, labels = _shard_batch(inputs, labels)
        seen += labels.size(0)
        got = f_graphs(model, criterion, inputs, labels)
        if got is not None:
            loss, output = got
        else:
            output = model(inputs)
            if isinstance(output, tuple):
                output = output[-1]
            loss = criterion(output, labels)
        loss_sum += loss.double() * labels.size(0)
        meter.update(output, labels)
    evaluate.forward_replays = f_graphs.replays
    if _world() > 1:
        n = int(_all_sum(torch.tensor(float(seen), device=device, dtype=torch.float64)))
    epoch_loss = float(_all_sum(loss_sum)) / n
    metric = meter.compute(n)
    logger.info('{} Loss: {:.4f}, {}: {:.4f}'.format(phase, epoch_loss, meter.name, metric))
    logger.info('Fusion Model Params: {}'.format(fusion_params(model)))
    return metric
