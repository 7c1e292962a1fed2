"""-m gpu: the reference-shaped call chain train_darts_model -> train_mmimdb_track_f1 ->
Architect.step end to end on an in-memory synthetic DataLoader (the unimodal backbones are out of
scope: a stand-in `models.central.mmimdb` producing feature maps of the right shapes is injected)."""
import logging
import os
import pickle
import sys
import types

import pytest
import torch
from torch.utils.data import DataLoader, Dataset

pytestmark = pytest.mark.gpu


class _FakeVGG(torch.nn.Module):
    def __init__(self, args):
        super().__init__()
        self.p = torch.nn.Parameter(torch.ones(1))

    def forward(self, image):
        b = image.shape[0]
        g = torch.Generator(device='cpu').manual_seed(int(image.sum().item() * 1000) % 1000)
        mk = lambda *s: torch.randn(b, *s, generator=g).to(image.device).relu()
        return [mk(512, 20, 32), mk(512, 20, 32), mk(512, 10, 16), mk(512, 5, 8), mk(23)]


class _FakeMLP(torch.nn.Module):
    def __init__(self, args):
        super().__init__()

    def forward(self, text):
        b = text.shape[0]
        return [text[:, :64].relu(), text[:, :128].relu(), text[:, :23]]


class _DS(Dataset):
    def __init__(self, n, seed):
        g = torch.Generator().manual_seed(seed)
        self.img = torch.randn(n, 3, 16, 16, generator=g)
        self.txt = torch.randn(n, 300, generator=g)
        self.lab = (torch.rand(n, 23, generator=g) < 0.2).float()

    def __len__(self):
        return len(self.lab)

    def __getitem__(self, i):
        return {'image': self.img[i], 'text': self.txt[i], 'label': self.lab[i]}


def test_mmimdb_search_driver_runs_end_to_end(tmp_path, monkeypatch):
    central = types.ModuleType('models.central')
    fake = types.ModuleType('models.central.mmimdb')
    fake.GP_VGG, fake.MaxOut_MLP = _FakeVGG, _FakeMLP
    central.mmimdb = fake
    monkeypatch.setitem(sys.modules, 'models.central', central)
    monkeypatch.setitem(sys.modules, 'models.central.mmimdb', fake)
    import models.search.mmimdb_darts_searchable as drv
    from models.search.darts.utils import create_exp_dir

    class Args:
        pass

    a = Args()
    a.C, a.L, a.drpt = 32, 16, 0.1
    a.num_input_nodes, a.num_keep_edges, a.steps, a.multiplier = 6, 2, 2, 2
    a.node_steps, a.node_multiplier, a.num_outputs = 1, 1, 23
    a.batchsize, a.epochs = 8, 2
    a.eta_max, a.eta_min, a.Ti, a.Tm = 1e-3, 1e-6, 1, 2
    a.arch_learning_rate, a.arch_weight_decay, a.weight_decay = 3e-4, 1e-3, 1e-4
    a.f1_type = 'weighted'
    a.use_dataparallel = False            # the mains define this; the library must not need .parallel
    a.save = str(tmp_path / 'exp')
    create_exp_dir(a.save)
    loaders = {k: DataLoader(_DS(n, s), batch_size=a.batchsize, shuffle=True, drop_last=False)
               for k, n, s in (('train', 20, 1), ('dev', 12, 2), ('test', 8, 3))}   # ragged last batches
    logger = logging.getLogger('bmnas-test')
    best_f1, genotype = drv.train_darts_model(loaders, a, torch.device('cuda:0'), logger)
    assert 0.0 <= best_f1 <= 1.0
    assert len(genotype.edges) == 4 and len(genotype.steps) == 2
    with open(os.path.join(a.save, 'best', 'best_genotype.pkl'), 'rb') as f:
        assert pickle.load(f) == genotype
    sd = torch.load(os.path.join(a.save, 'best', 'best_model.pt'))
    assert 'fusion_net.cell.ln.weight' in sd and 'reshape_layers.0.conv.weight' in sd
    assert not any('alphas' in k or 'betas' in k or 'gammas' in k for k in sd)


class _CapturableVGG(torch.nn.Module):
    """Stand-in backbone without host synchronisation (the real GP_VGG is a plain conv stack)."""

    def __init__(self, args):
        super().__init__()
        self.proj = torch.nn.Linear(3 * 16 * 16, 512)

    def forward(self, image):
        f = self.proj(image.flatten(1)).relu()                       # (b, 512)
        mk = lambda h, w: f[:, :, None, None].expand(-1, -1, h, w) * torch.linspace(
            0.5, 1.5, h * w, device=f.device).view(1, 1, h, w)
        return [mk(20, 32), mk(20, 32), mk(10, 16), mk(5, 8), f[:, :23]]


def test_mmimdb_search_driver_with_hip_graph_steps(tmp_path, monkeypatch):
    """args.hip_graph: every full batch of both phases is one graph replay; the ragged last
    batches take the eager path; results are written as usual."""
    central = types.ModuleType('models.central')
    fake = types.ModuleType('models.central.mmimdb')
    fake.GP_VGG, fake.MaxOut_MLP = _CapturableVGG, _FakeMLP
    central.mmimdb = fake
    monkeypatch.setitem(sys.modules, 'models.central', central)
    monkeypatch.setitem(sys.modules, 'models.central.mmimdb', fake)
    import models.search.mmimdb_darts_searchable as drv
    import models.search.train_searchable._loop as loop
    from models.search.darts.utils import create_exp_dir

    class Args:
        pass

    a = Args()
    a.C, a.L, a.drpt = 32, 16, 0.1
    a.num_input_nodes, a.num_keep_edges, a.steps, a.multiplier = 6, 2, 2, 2
    a.node_steps, a.node_multiplier, a.num_outputs = 1, 1, 23
    a.batchsize, a.epochs = 8, 2
    a.eta_max, a.eta_min, a.Ti, a.Tm = 1e-3, 1e-6, 1, 2
    a.arch_learning_rate, a.arch_weight_decay, a.weight_decay = 3e-4, 1e-3, 1e-4
    a.f1_type = 'weighted'
    a.use_dataparallel = False
    a.hip_graph = True
    a.save = str(tmp_path / 'exp')
    create_exp_dir(a.save)
    loaders = {k: DataLoader(_DS(n, s), batch_size=a.batchsize, shuffle=True, drop_last=False)
               for k, n, s in (('train', 20, 1), ('dev', 12, 2), ('test', 8, 3))}
    logger = logging.getLogger('bmnas-test')
    best_f1, genotype = drv.train_darts_model(loaders, a, torch.device('cuda:0'), logger)
    assert 0.0 <= best_f1 <= 1.0
    assert len(genotype.edges) == 4 and len(genotype.steps) == 2
    # 2 epochs x (2 full + 1 ragged) train batches
    assert loop.run.stats['graph_replays'] == 4, loop.run.stats
    sd = torch.load(os.path.join(a.save, 'best', 'best_model.pt'))
    assert all(torch.isfinite(v.float()).all() for v in sd.values())


def test_hip_graph_falls_back_when_the_model_cannot_be_captured(tmp_path, monkeypatch):
    """A backbone that synchronises with the host (here: .item()) cannot be captured: the
    trainer must notice, stay on the eager path and still finish the search."""
    central = types.ModuleType('models.central')
    fake = types.ModuleType('models.central.mmimdb')
    fake.GP_VGG, fake.MaxOut_MLP = _FakeVGG, _FakeMLP
    central.mmimdb = fake
    monkeypatch.setitem(sys.modules, 'models.central', central)
    monkeypatch.setitem(sys.modules, 'models.central.mmimdb', fake)
    import models.search.mmimdb_darts_searchable as drv
    import models.search.train_searchable._loop as loop
    from models.search.darts.utils import create_exp_dir

    class Args:
        pass

    a = Args()
    a.C, a.L, a.drpt = 32, 16, 0.1
    a.num_input_nodes, a.num_keep_edges, a.steps, a.multiplier = 6, 2, 2, 2
    a.node_steps, a.node_multiplier, a.num_outputs = 1, 1, 23
    a.batchsize, a.epochs = 8, 1
    a.eta_max, a.eta_min, a.Ti, a.Tm = 1e-3, 1e-6, 1, 2
    a.arch_learning_rate, a.arch_weight_decay, a.weight_decay = 3e-4, 1e-3, 1e-4
    a.f1_type = 'weighted'
    a.use_dataparallel = False
    a.hip_graph = True
    a.save = str(tmp_path / 'exp')
    create_exp_dir(a.save)
    loaders = {k: DataLoader(_DS(n, s), batch_size=a.batchsize, shuffle=True, drop_last=False)
               for k, n, s in (('train', 16, 1), ('dev', 8, 2), ('test', 8, 3))}
    logger = logging.getLogger('bmnas-test')
    best_f1, genotype = drv.train_darts_model(loaders, a, torch.device('cuda:0'), logger)
    assert 0.0 <= best_f1 <= 1.0
    assert loop.run.stats['graph_replays'] == 0 and loop.run.stats['eager_steps'] > 0
    # the GPU is still usable afterwards
    x = torch.ones(4, device='cuda')
    assert float((x * 2).sum()) == 8.0
