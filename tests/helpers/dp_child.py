"""Child process of tests/test_dp_gpu.py: ONE rank of a 2-rank data-parallel run through the
reference-shaped call chain, both ranks on the same GPU (BMNAS_FORCE_DEVICE) over gloo
(BMNAS_DIST_BACKEND) — RCCL refuses two ranks on one device.  Started fresh (never a re-exec of a
GPU-initialised process); reads RANK / WORLD_SIZE / MASTER_* from the environment.

    python dp_child.py grads  <outdir>     all-reduced w- and arch-gradients of one captured step
    python dp_child.py driver <outdir>     train_darts_model end to end (synthetic loaders)
    python dp_child.py uneven <outdir>     an uneven global batch (7 = 4 + 3) through the loop's scatter
"""
import logging
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, 'bm-nas_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.nn as nn

GLOBAL_BATCH, NOUT, SEED = 8, 23, 6
UNEVEN_BATCH = 7


def cfg_small():
    from oracle import fusion_oracle as fo
    return fo.Cfg({**fo.CONFIGS['mmimdb'], 'C': 32, 'drpt': 0.0})


class Args:
    pass


def make_args(cfg):
    a = Args()
    a.C, a.L, a.drpt = cfg.C, cfg.L, cfg.drpt
    a.num_input_nodes, a.num_keep_edges, a.steps, a.multiplier = cfg.N, 2, cfg.S, cfg.M
    a.node_steps, a.node_multiplier, a.num_outputs = cfg.ns, cfg.nm, NOUT
    a.batchsize, a.epochs = GLOBAL_BATCH, 1
    a.eta_max, a.eta_min, a.Ti, a.Tm = 1e-3, 1e-6, 1, 2
    a.arch_learning_rate, a.arch_weight_decay, a.weight_decay = 3e-4, 1e-3, 1e-4
    a.f1_type = 'weighted'
    a.use_dataparallel = True             # what the MM-IMDB / NTU mains define
    return a


def run_grads(out):
    """search_setup (the body of train_darts_model) on the real HIP hypernet, one captured weight
    step and one captured Architect step with lr = 0 (so both gradients are taken at the initial
    point), then two real steps to compare the replicas."""
    from gpu_util import set_mode
    from oracle import synth
    from bmnas import dist as bdist
    from bmnas import nn as bnn
    from bmnas.graph import GraphedTrainStep
    from models.search._common import HyperNetBase, search_setup
    cfg = cfg_small()
    args = make_args(cfg)

    class Net(HyperNetBase):
        def __init__(self, criterion):
            super().__init__()
            self._build_head(args, criterion, nn.ModuleList([nn.Identity() for _ in range(cfg.N)]), cfg.N, 2)

        def forward(self, feats):
            return self.fuse(feats)

    crit = bnn.BCEWithLogitsLoss()
    model = Net(crit)
    model.fusion_net.load_state_dict(synth.make_params(cfg, SEED))
    for dst, src in zip(model.arch_parameters(), synth.make_arch(cfg, SEED, 0.5)):
        dst.data.copy_(src)
    cw, cb = synth.make_classifier(cfg, NOUT, SEED)
    model.central_classifier.weight.data.copy_(cw)
    model.central_classifier.bias.data.copy_(cb)
    if int(os.environ['RANK']) == 1:          # broadcast_state must repair this
        with torch.no_grad():
            model.central_classifier.bias.add_(1.0)
    # the unchanged mains hand cuda:0 to every rank (main_darts_searchable_mmimdb.py:86)
    optimizer, scheduler, architect, _ = search_setup(model, args, crit, torch.device('cuda:0'),
                                                      1.0, args.weight_decay)
    rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
    device = next(model.parameters()).device
    set_mode(model, 'train_nodrop')
    X = [x.to(device) for x in synth.make_inputs(cfg, GLOBAL_BATCH, SEED)]
    Y = synth.make_labels('bce', GLOBAL_BATCH, NOUT, SEED).to(device)
    xs = [bdist.shard(x, rank, world).contiguous() for x in X]
    y = bdist.shard(Y, rank, world).contiguous()
    for o in (optimizer, architect.optimizer):
        for g in o.param_groups:
            g['lr'] = 0.0
    assert GraphedTrainStep.enabled(args), 'graph + flat bucket must be the data-parallel default'
    wg = GraphedTrainStep(model, crit, optimizer, xs, y)
    assert wg.reducer is not None and wg.reducer.world == 2
    wg(xs, y)
    names = [n for grp in ('reshape_layers', 'fusion_net', 'central_classifier')
             for n, _ in getattr(model, grp).named_parameters(prefix=grp)]
    dump = {'wgrad:' + n: v.detach().cpu().clone() for n, v in zip(names, wg.reducer.views)}
    architect.step(xs, y, None)
    assert architect.graph_replays == 1
    for i, v in enumerate(architect.optimizer._bmnas_reducer.views):
        dump[f'agrad:{i}'] = v.detach().cpu().clone()
    # an eager (ragged) step between replays, then replays again — with real learning rates
    for o, lr in ((optimizer, 1e-3), (architect.optimizer, 3e-3)):
        for g in o.param_groups:
            g['lr'] = lr
    xr, yr = [x[:3].contiguous() for x in xs], y[:3].contiguous()
    for it in range(3):
        if it == 1:
            optimizer.zero_grad()
            crit(model(xr), yr).backward()
            optimizer.step()
        else:
            wg(xs, y)
        architect.step(xs, y, None)
    torch.cuda.synchronize()
    for n, v in model.state_dict().items():
        dump['state:' + n] = v.detach().cpu().clone()
    for i, a in enumerate(model.arch_parameters()):
        dump[f'arch:{i}'] = a.detach().cpu().clone()
    dump['device'] = str(device)
    torch.save(dump, os.path.join(out, f'grads_rank{rank}.pt'))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def run_uneven(out):
    """An UNEVEN global batch (7 samples over 2 ranks = 4 + 3, nn.DataParallel's Tensor.chunk scatter) through the trainer
    loop's own `_shard_batch`: captured weight step and Architect step with lr = 0; the all-reduced buckets must hold the
    gradient of ONE mean over all 7 samples (per-replica BatchNorm statistics) — each shard's mean weighted n_rank * 2 / 7."""
    from gpu_util import set_mode
    from oracle import synth
    from bmnas import dist as bdist
    from bmnas import nn as bnn
    from bmnas.graph import GraphedTrainStep
    from models.search._common import HyperNetBase, search_setup
    import models.search.train_searchable._loop as loop
    cfg = cfg_small()
    args = make_args(cfg)

    class Net(HyperNetBase):
        def __init__(self, criterion):
            super().__init__()
            self._build_head(args, criterion, nn.ModuleList([nn.Identity() for _ in range(cfg.N)]), cfg.N, 2)

        def forward(self, feats):
            return self.fuse(feats)

    crit = bnn.BCEWithLogitsLoss()
    model = Net(crit)
    model.fusion_net.load_state_dict(synth.make_params(cfg, SEED))
    for dst, src in zip(model.arch_parameters(), synth.make_arch(cfg, SEED, 0.5)):
        dst.data.copy_(src)
    cw, cb = synth.make_classifier(cfg, NOUT, SEED)
    model.central_classifier.weight.data.copy_(cw)
    model.central_classifier.bias.data.copy_(cb)
    optimizer, scheduler, architect, _ = search_setup(model, args, crit, torch.device('cuda:0'),
                                                      1.0, args.weight_decay)
    rank = torch.distributed.get_rank()
    device = next(model.parameters()).device
    set_mode(model, 'train_nodrop')
    X = [x[:UNEVEN_BATCH].to(device) for x in synth.make_inputs(cfg, GLOBAL_BATCH, SEED)]
    Y = synth.make_labels('bce', GLOBAL_BATCH, NOUT, SEED)[:UNEVEN_BATCH].to(device)
    xs, y = loop._shard_batch(X, Y)
    xs, y = [x.contiguous() for x in xs], y.contiguous()
    assert y.shape[0] == (4 if rank == 0 else 3) and abs(bdist.shard_weight() - y.shape[0] * 2 / 7) < 1e-12
    for o in (optimizer, architect.optimizer):
        for g in o.param_groups:
            g['lr'] = 0.0
    wg = GraphedTrainStep(model, crit, optimizer, xs, y)
    assert wg.matches(xs, y)
    wg(xs, y)
    names = [n for grp in ('reshape_layers', 'fusion_net', 'central_classifier')
             for n, _ in getattr(model, grp).named_parameters(prefix=grp)]
    dump = {'wgrad:' + n: v.detach().cpu().clone() for n, v in zip(names, wg.reducer.views)}
    architect.step(xs, y, None)
    assert architect.graph_replays == 1
    for i, v in enumerate(architect.optimizer._bmnas_reducer.views):
        dump[f'agrad:{i}'] = v.detach().cpu().clone()
    # another global batch size that gives this rank the SAME shard shape must not replay the captured step
    bdist.set_shard_weight(1.0)
    assert not wg.matches(xs, y)
    # ... and the eager path weights its bucket the same way: same reduced gradients
    bdist.set_shard_weight(y.shape[0] * 2 / 7)
    optimizer.zero_grad()
    crit(model(xs), y).backward()
    optimizer.step()                       # lr = 0: the pre-hook reduces, nothing moves
    for n, p_ in zip(names, [p for grp in ('reshape_layers', 'fusion_net', 'central_classifier')
                             for p in getattr(model, grp).parameters()]):
        if p_.grad is not None:
            dump['egrad:' + n] = p_.grad.detach().cpu().clone()
    torch.cuda.synchronize()
    torch.save(dump, os.path.join(out, f'uneven_rank{rank}.pt'))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


class _VGG(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.proj = nn.Linear(3 * 16 * 16, 512)

    def forward(self, image):
        f = self.proj(image.flatten(1)).relu()
        mk = lambda h, w: f[:, :, None, None].expand(-1, -1, h, w) * torch.linspace(
            0.5, 1.5, h * w, device=f.device).view(1, 1, h, w)
        return [mk(20, 32), mk(20, 32), mk(10, 16), mk(5, 8), f[:, :23]]


class _MLP(nn.Module):
    def __init__(self, args):
        super().__init__()

    def forward(self, text):
        return [text[:, :64].relu(), text[:, :128].relu(), text[:, :23]]


class _DS(torch.utils.data.Dataset):
    def __init__(self, n, seed):
        g = torch.Generator().manual_seed(seed)
        self.img = torch.randn(n, 3, 16, 16, generator=g)
        self.txt = torch.randn(n, 300, generator=g)
        self.lab = (torch.rand(n, 23, generator=g) < 0.2).float()

    def __len__(self):
        return len(self.lab)

    def __getitem__(self, i):
        return {'image': self.img[i], 'text': self.txt[i], 'label': self.lab[i]}


def run_driver(out):
    """The reference's own chain: train_darts_model -> train_mmimdb_track_f1 -> Architect.step, with
    the caller's UNSHARDED loaders and device cuda:0 on every rank, exactly what an unchanged
    main_darts_searchable_mmimdb.py does under torch.distributed.run."""
    from torch.utils.data import DataLoader
    central = types.ModuleType('models.central')
    fake = types.ModuleType('models.central.mmimdb')
    fake.GP_VGG, fake.MaxOut_MLP = _VGG, _MLP
    central.mmimdb = fake
    sys.modules['models.central'] = central
    sys.modules['models.central.mmimdb'] = fake
    import models.search.mmimdb_darts_searchable as drv
    import models.search.train_searchable._loop as loop
    from models.search.darts.utils import create_exp_dir
    cfg = cfg_small()
    a = make_args(cfg)
    a.drpt = 0.1
    a.epochs = 2
    rank = int(os.environ['RANK'])
    a.save = os.path.join(out, f'exp_rank{rank}')
    create_exp_dir(a.save)
    torch.manual_seed(2)                  # same initial weights and the same shuffles on every rank
    loaders = {k: DataLoader(_DS(n, s), batch_size=a.batchsize, shuffle=True, drop_last=False,
                             generator=torch.Generator().manual_seed(10 + s))
               for k, n, s in (('train', 20, 1), ('dev', 12, 2), ('test', 8, 3))}
    best_f1, genotype = drv.train_darts_model(loaders, a, torch.device('cuda:0'), logging.getLogger('dp'))
    torch.save({'best_f1': best_f1, 'genotype': repr(genotype), 'stats': dict(loop.run.stats),
                'files': sorted(os.listdir(os.path.join(a.save, 'best')))},
               os.path.join(out, f'driver_rank{rank}.pt'))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    {'grads': run_grads, 'driver': run_driver, 'uneven': run_uneven}[sys.argv[1]](sys.argv[2])
