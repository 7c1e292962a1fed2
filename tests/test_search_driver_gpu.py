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
    # the metric pass of every dev batch (2 epochs x (1 full + 1 ragged): the ragged shape gets a graph of its own)
    assert loop.run.stats['forward_replays'] == 4, loop.run.stats
    # round 5: the full dev batches' metric forward rides at the end of the architecture step's replay (one batch copy,
    # one launch); the ragged ones run architect.step eagerly and replay a forward graph of their own
    assert loop.run.stats.get('merged_metric_replays', 0) == 2, loop.run.stats
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


def test_found_stage_trainer_and_tester_end_to_end(tmp_path, monkeypatch):
    """the found-stage loop of main_darts_found_mmimdb.py:95-145 — a
    Found_Image_Text_Net built from a genotype, trained with status='eval' (phases train, dev,
    test; the dev phase also learns, train_searchable/mmimdb.py:38, 92), best-test checkpoint,
    then test_mmimdb_track_f1 on the reloaded weights.  The tester's F1 must equal the F1
    computed here directly from the model's outputs (sklearn, weighted, threshold 0.3)."""
    from sklearn.metrics import f1_score
    central = types.ModuleType('models.central')
    fake = types.ModuleType('models.central.mmimdb')
    fake.GP_VGG, fake.MaxOut_MLP = _CapturableVGG, _FakeMLP
    central.mmimdb = fake
    monkeypatch.setitem(sys.modules, 'models.central', central)
    monkeypatch.setitem(sys.modules, 'models.central.mmimdb', fake)
    import models.auxiliary.scheduler as sc
    import models.search.mmimdb_darts_searchable as drv
    import models.search.train_searchable._loop as loop
    import models.search.train_searchable.mmimdb as tr
    from bmnas import nn as bnn
    from bmnas.optim import Adam
    from models.search.darts.genotypes import Genotype, StepGenotype
    from models.search.darts.utils import create_exp_dir
    from models.search.plot_genotype import Plotter

    class Args:
        pass

    a = Args()
    a.C, a.L, a.drpt = 32, 16, 0.1
    a.num_input_nodes, a.num_keep_edges, a.steps, a.multiplier = 6, 2, 2, 2
    a.node_steps, a.node_multiplier, a.num_outputs = 2, 2, 23
    a.batchsize, a.epochs = 8, 2
    a.eta_max, a.eta_min, a.Ti, a.Tm = 1e-3, 1e-6, 1, 2
    a.f1_type = 'weighted'
    a.use_dataparallel = False
    a.save = str(tmp_path / 'exp')
    create_exp_dir(a.save)
    genotype = Genotype(
        edges=[('skip', 1), ('skip', 4), ('skip', 0), ('skip', 5)],
        steps=[StepGenotype(inner_edges=[('skip', 0), ('skip', 1), ('skip', 2), ('skip', 0)],
                            inner_steps=['ScaleDotAttn', 'LinearGLU'], inner_concat=[2, 3]),
               StepGenotype(inner_edges=[('skip', 1), ('skip', 0), ('skip', 1), ('skip', 2)],
                            inner_steps=['ConcatFC', 'Sum'], inner_concat=[2, 3])],
        concat=[6, 7])
    device = torch.device('cuda:0')
    torch.manual_seed(5)
    criterion = bnn.BCEWithLogitsLoss()
    model = drv.Found_Image_Text_Net(a, criterion, genotype)
    loaders = {k: DataLoader(_DS(n, s), batch_size=a.batchsize, shuffle=(k == 'train'), drop_last=False)
               for k, n, s in (('train', 20, 1), ('dev', 12, 2), ('test', 11, 3))}
    sizes = {k: len(v.dataset) for k, v in loaders.items()}
    model.to(device)
    optimizer = Adam(model.parameters(), lr=a.eta_max, weight_decay=1e-4)
    scheduler = sc.LRCosineAnnealingScheduler(a.eta_max, a.eta_min, a.Ti, a.Tm, sizes['train'] / a.batchsize)
    logger = logging.getLogger('bmnas-test')
    w0 = model.central_classifier.weight.detach().clone()
    test_f1, test_genotype = tr.train_mmimdb_track_f1(model, None, criterion, optimizer, scheduler, loaders, sizes,
                                                      device, a.epochs, False, logger, Plotter(a), a, a.f1_type,
                                                      0.0, 0.3, 'eval')
    assert 0.0 <= test_f1 <= 1.0 and test_genotype == genotype
    assert float((model.central_classifier.weight.detach() - w0).abs().max()) > 0      # it did train
    # train AND dev phases learn in the found stage: 2 epochs x (3 + 2) batches
    assert loop.run.stats['graph_replays'] + loop.run.stats['eager_steps'] >= 2 * (3 + 2)
    ckpt = os.path.join(a.save, 'best', 'best_test_model.pt')
    assert os.path.exists(ckpt)
    # the tester on the reloaded best-test weights (main_darts_found_mmimdb.py:126-139)
    model2 = drv.Found_Image_Text_Net(a, criterion, genotype)
    model2.load_state_dict(torch.load(ckpt))
    model2.to(device)
    f1 = tr.test_mmimdb_track_f1(model2, criterion, loaders, sizes, device, False, logger, a, a.f1_type,
                                 init_f1=0.0, th_fscore=0.3)
    assert isinstance(f1, float)
    model2.eval()
    preds, labels = [], []
    with torch.no_grad():
        for d in loaders['test']:
            out = model2((d['text'].to(device), d['image'].to(device)))
            preds.append((torch.sigmoid(out) > 0.3).cpu())
            labels.append(d['label'])
    want = f1_score(torch.cat(labels).numpy(), torch.cat(preds).numpy(), average='weighted', zero_division=1)
    assert abs(f1 - want) < 1e-12
    assert abs(f1 - test_f1) < 1e-6        # the checkpoint holds the weights that scored the best test F1
