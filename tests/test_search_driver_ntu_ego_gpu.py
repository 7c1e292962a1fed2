"""-m gpu: the NTU RGB+D and EgoGesture call chains end to end — train_darts_model -> train_{ntu,ego}_track_acc ->
Architect.step (reference ntu_darts_searchable.py:21-72, ego_darts_searchable.py:20-69, train_searchable/ntu.py:12-227,
ego.py:13-223) — search stage, found stage (status='eval') and the testers, on in-memory synthetic loaders.  The
unimodal backbones are out of scope: stand-ins for `models.central.{ntu,ego}` produce feature maps with the real
channel widths (C_in 512 ... 2048), small spatial extents."""
import logging
import os
import pickle
import sys
import types

import pytest
import torch
from torch.utils.data import DataLoader, Dataset

pytestmark = pytest.mark.gpu


def _maps(f, shapes):
    """feature maps (b, C_i, *spatial_i) that depend on the input through f (b, 64) — differentiable, no host sync"""
    outs = []
    for C, sp in shapes:
        base = f.repeat(1, (C + f.shape[1] - 1) // f.shape[1])[:, :C]
        n = 1
        for s in sp:
            n *= s
        ramp = torch.linspace(0.5, 1.5, n, device=f.device).view(1, 1, *sp) if sp else None
        outs.append(base.view(base.shape[0], C, *([1] * len(sp))) * ramp if sp else base)
    return outs


class _Visual(torch.nn.Module):          # models.central.ntu.Visual: rgbnet(image)[-5:-1] are the four visual features
    def __init__(self, args):
        super().__init__()
        self.proj = torch.nn.Linear(3 * 4 * 8 * 8, 64)

    def forward(self, image):
        f = self.proj(image.flatten(1)).relu()
        return [f] + _maps(f, [(512, (4, 6, 6)), (1024, (4, 3, 3)), (2048, (2, 2, 2)), (2048, ())]) + [f[:, :60]]


class _Skeleton(torch.nn.Module):        # models.central.ntu.Skeleton: (features, logits); features[-4:] are used
    def __init__(self, args):
        super().__init__()
        self.proj = torch.nn.Linear(3 * 8 * 5, 64)

    def forward(self, ske):
        f = self.proj(ske.flatten(1)).relu()
        return [f] + _maps(f, [(128, (4, 4)), (256, (2, 2)), (1024, ()), (512, ())]), f[:, :60]


class _NtuDS(Dataset):
    def __init__(self, n, seed):
        g = torch.Generator().manual_seed(seed)
        self.rgb = torch.randn(n, 3, 4, 8, 8, generator=g)
        self.ske = torch.randn(n, 3, 8, 5, generator=g)
        self.lab = torch.randint(0, 60, (n,), generator=g)

    def __len__(self):
        return len(self.lab)

    def __getitem__(self, i):
        return {'rgb': self.rgb[i], 'ske': self.ske[i], 'label': self.lab[i]}


class _EgoNet(torch.nn.Module):          # models.central.ego.get_{rgb,depth}_model: net(x)[0:-1] are the four features
    def __init__(self, cin):
        super().__init__()
        self.proj = torch.nn.Linear(cin * 4 * 8 * 8, 64)

    def forward(self, x):
        f = self.proj(x.flatten(1)).relu()
        return _maps(f, [(512, (4, 5, 5)), (1024, (2, 3, 3)), (2048, (2, 2, 2)), (2048, (1, 1, 1))]) + [f[:, :83]]


class _EgoDS(Dataset):
    def __init__(self, n, seed):
        g = torch.Generator().manual_seed(seed)
        self.clip = torch.randn(n, 4, 4, 8, 8, generator=g)        # RGB in channels 0:3, depth in 3:
        self.lab = torch.randint(0, 83, (n,), generator=g)

    def __len__(self):
        return len(self.lab)

    def __getitem__(self, i):
        return self.clip[i], self.lab[i]


class _A:
    pass


def _args(tmp_path, ns, nm, nout, drpt):
    from models.search.darts.utils import create_exp_dir
    a = _A()
    a.C, a.L, a.drpt = 32, 8, drpt
    a.num_input_nodes, a.num_keep_edges, a.steps, a.multiplier = 8, 2, 2, 2
    a.node_steps, a.node_multiplier, a.num_outputs = ns, nm, nout
    a.batchsize, a.epochs = 8, 2
    a.eta_max, a.eta_min, a.Ti, a.Tm = 1e-3, 1e-6, 1, 2
    a.arch_learning_rate, a.arch_weight_decay, a.weight_decay = 3e-4, 1e-3, 1e-4
    a.parallel = False
    a.checkpointdir = str(tmp_path)
    a.save = str(tmp_path / 'exp')
    create_exp_dir(a.save)
    return a


def _fake_central(monkeypatch, name, **attrs):
    central = types.ModuleType('models.central')
    fake = types.ModuleType('models.central.' + name)
    for k, v in attrs.items():
        setattr(fake, k, v)
    setattr(central, name, fake)
    monkeypatch.setitem(sys.modules, 'models.central', central)
    monkeypatch.setitem(sys.modules, 'models.central.' + name, fake)


def test_ntu_search_found_and_test_stages(tmp_path, monkeypatch):
    _fake_central(monkeypatch, 'ntu', Visual=_Visual, Skeleton=_Skeleton)
    import models.auxiliary.scheduler as sc
    import models.search.ntu_darts_searchable as drv
    import models.search.train_searchable._loop as loop
    import models.search.train_searchable.ntu as tr
    from bmnas import nn as bnn
    from bmnas.optim import Adam
    from models.search.plot_genotype import Plotter
    a = _args(tmp_path, ns=2, nm=2, nout=60, drpt=0.2)
    a.ske_cp, a.rgb_cp = 'ske.pt', 'rgb.pt'
    torch.manual_seed(3)
    torch.save(_Skeleton(a).state_dict(), tmp_path / a.ske_cp)
    torch.save(_Visual(a).state_dict(), tmp_path / a.rgb_cp)
    loaders = {k: DataLoader(_NtuDS(n, s), batch_size=a.batchsize, shuffle=(k == 'train'), drop_last=False)
               for k, n, s in (('train', 20, 1), ('dev', 12, 2), ('test', 11, 3))}        # ragged last batches
    logger = logging.getLogger('bmnas-test')
    device = torch.device('cuda:0')
    best_acc, genotype = drv.train_darts_model(loaders, a, device, logger)
    assert 0.0 <= best_acc <= 1.0
    assert len(genotype.edges) == 4 and len(genotype.steps) == 2 and len(genotype.steps[0].inner_steps) == 2
    # full batches of both phases were graph replays, the ragged ones eager
    from bmnas.graph import GraphedTrainStep
    if GraphedTrainStep.enabled(a):                      # (BMNAS_HIP_GRAPH=0 runs of the switch matrix: all eager)
        assert loop.run.stats['graph_replays'] >= 2 * 2 and loop.run.stats['eager_steps'] >= 2
    with open(os.path.join(a.save, 'best', 'best_genotype.pkl'), 'rb') as f:
        assert pickle.load(f) == genotype
    sd = torch.load(os.path.join(a.save, 'best', 'best_model.pt'))
    assert 'fusion_net.cell._step_nodes.1.node_cell.out_conv.weight' in sd and 'reshape_layers.7.conv.weight' in sd
    # the reference optimises fusion_net + classifier only on NTU (ntu_darts_searchable.py:157-162)
    m = drv.Searchable_Skeleton_Image_Net(a, bnn.CrossEntropyLoss(), logger)
    groups = m.central_params()
    assert len(groups) == 2
    ids = {id(p) for g in groups for p in g['params']}
    assert not any(id(p) in ids for p in m.reshape_layers.parameters())
    # ---- found stage (main_darts_found_ntu.py): retrain the discrete net, then the tester
    criterion = bnn.CrossEntropyLoss()
    found = drv.Found_Skeleton_Image_Net(a, criterion, genotype).to(device)
    sizes = {k: len(v.dataset) for k, v in loaders.items()}
    opt = Adam(found.central_params(), lr=a.eta_max, weight_decay=1e-4)
    sched = sc.LRCosineAnnealingScheduler(a.eta_max, a.eta_min, a.Ti, a.Tm, sizes['train'] / a.batchsize)
    w0 = found.central_classifier.weight.detach().clone()
    test_acc, g2 = tr.train_ntu_track_acc(found, None, criterion, opt, sched, loaders, sizes, device=device,
                                          num_epochs=2, parallel=False, logger=logger, plotter=Plotter(a), args=a,
                                          status='eval')
    # reference quirk kept: with status != 'search' the NTU / Ego trainers return the best DEV genotype
    # (train_searchable/ntu.py:182, ego.py:177), which their found-stage phases ['train', 'test'] never set
    assert 0.0 <= test_acc <= 1.0 and g2 is None
    assert float((found.central_classifier.weight.detach() - w0).abs().max()) > 0
    acc = tr.test_ntu_track_acc(found, loaders, criterion, genotype, sizes, device, logger, a)
    found.eval()
    hit = 0
    with torch.no_grad():
        for d in loaders['test']:
            out = found((d['rgb'].to(device), d['ske'].to(device)))
            hit += int((out.argmax(1).cpu() == d['label']).sum())
    assert abs(acc - hit / sizes['test']) < 1e-9


def test_ego_search_found_and_test_stages(tmp_path, monkeypatch):
    _fake_central(monkeypatch, 'ego', get_rgb_model=lambda opt: _EgoNet(3), get_depth_model=lambda opt: _EgoNet(1))
    import models.auxiliary.scheduler as sc
    import models.search.ego_darts_searchable as drv
    import models.search.train_searchable._loop as loop
    import models.search.train_searchable.ego as tr
    from bmnas import nn as bnn
    from bmnas.optim import Adam
    from models.search.plot_genotype import Plotter
    a = _args(tmp_path, ns=3, nm=3, nout=83, drpt=0.0)        # the Ego main's defaults: drpt 0, 3 inner steps
    a.rgb_cp, a.depth_cp = 'rgb.pt', 'depth.pt'
    torch.manual_seed(4)
    torch.save(_EgoNet(3).state_dict(), tmp_path / a.rgb_cp)
    torch.save(_EgoNet(1).state_dict(), tmp_path / a.depth_cp)
    loaders = {k: DataLoader(_EgoDS(n, s), batch_size=a.batchsize, shuffle=(k == 'train'), drop_last=False)
               for k, n, s in (('train', 20, 1), ('dev', 12, 2), ('test', 11, 3))}
    logger = logging.getLogger('bmnas-test')
    device = torch.device('cuda:0')
    best_acc, genotype = drv.train_darts_model(loaders, a, object(), device, logger)
    assert 0.0 <= best_acc <= 1.0
    assert len(genotype.steps) == 2 and len(genotype.steps[0].inner_steps) == 3
    assert genotype.steps[0].inner_concat == [2, 3, 4]
    from bmnas.graph import GraphedTrainStep
    if GraphedTrainStep.enabled(a):
        assert loop.run.stats['graph_replays'] >= 2 * 2
    sd = torch.load(os.path.join(a.save, 'best', 'best_model.pt'))
    assert 'fusion_net.cell._step_nodes.0.node_cell.node_ops.2._ops.2.conv.weight' in sd
    # Ego optimises fusion_net, classifier AND the reshape layers (ego_darts_searchable.py:160-166)
    m = drv.Searchable_RGB_Depth_Net(a, object(), bnn.CrossEntropyLoss())
    assert len(m.central_params()) == 3
    criterion = bnn.CrossEntropyLoss()
    found = drv.Found_RGB_Depth_Net(a, object(), criterion, genotype).to(device)
    sizes = {k: len(v.dataset) for k, v in loaders.items()}
    opt = Adam(found.central_params(), lr=a.eta_max, weight_decay=1e-4)
    sched = sc.LRCosineAnnealingScheduler(a.eta_max, a.eta_min, a.Ti, a.Tm, sizes['train'] / a.batchsize)
    test_acc, g2 = tr.train_ego_track_acc(found, None, criterion, opt, sched, loaders, sizes, device=device,
                                          num_epochs=2, parallel=False, logger=logger, plotter=Plotter(a), args=a,
                                          status='eval')
    assert 0.0 <= test_acc <= 1.0 and g2 is None          # (the same reference quirk as NTU)
    acc = tr.test_ego_track_acc(found, loaders, criterion, genotype, sizes, device, logger, a)
    found.eval()
    hit = 0
    with torch.no_grad():
        for clip, lab in loaders['test']:
            out = found((clip[:, 0:3].to(device), clip[:, 3:].to(device)))
            hit += int((out.argmax(1).cpu() == lab).sum())
    assert abs(acc - hit / sizes['test']) < 1e-9
