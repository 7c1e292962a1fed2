"""-m gpu: the OUTER LOOP against the reference's own trainers.

tests/golden/loop_mmimdb_{search,found}.json were recorded by running the unmodified reference
(train_darts_model -> train_mmimdb_track_f1 -> Architect.step, models/search/mmimdb_darts_searchable.py:18-55,
train_searchable/mmimdb.py:10-285, darts/architect.py:21-29; the found stage as main_darts_found_mmimdb.py:95-139
drives it) on the deterministic in-memory data of tests/helpers/loop_stubs.py (tests/golden/make_golden_r04.py).
Here this repo's loop (models/search/train_searchable/_loop.py) runs on the same data, eager and with hipGraph
steps, and must reproduce: every batch's loss and logits, the learning rate every weight step ran with (the
per-batch cosine schedule), every phase's epoch loss / F1, the genotype after every phase, the best F1 / genotype /
checkpoint, the final architecture parameters and weights, and the tester's score.  Further down: the same for the
accuracy-tracking trainers of NTU RGB+D (search, found stage, tester) and EgoGesture (search)."""
import json
import logging
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from helpers import loop_stubs as stubs

GOLD = os.path.join(HERE, 'golden')


def _args(save, found=False, hip_graph=False):
    class Args:
        pass

    a = Args()
    a.C, a.L, a.drpt = 32, 16, 0.1
    a.num_input_nodes, a.num_keep_edges, a.steps, a.multiplier = 6, 2, 2, 2
    a.node_steps, a.node_multiplier, a.num_outputs = (2, 2, 23) if found else (1, 1, 23)
    a.batchsize, a.epochs = 8, 2
    a.eta_max, a.eta_min, a.Ti, a.Tm = 2e-3, 1e-5, 1, 2
    a.arch_learning_rate, a.arch_weight_decay, a.weight_decay = 3e-3, 1e-3, 1e-4
    a.f1_type = 'weighted'
    a.use_dataparallel = False
    a.hip_graph = bool(hip_graph)
    if hip_graph == 'k2':
        # two weight steps per hipGraph replay (models/search/train_searchable/_loop.py steps_per_replay): the recordings
        # hold phases of 3 / 2 batches with a ragged last one, so k = 2 is what gives every phase a k-step replay AND a tail
        a.steps_per_replay = 2
    a.save = save
    return a


def _loaders(gold, data_cls=stubs.MMIMDBData):
    return {k: DataLoader(data_cls(n, gold['seed_data'] + i), batch_size=8, shuffle=False, drop_last=False)
            for i, (k, n) in enumerate(gold['sizes'].items())}


CENTRAL = {'mmimdb': dict(GP_VGG=stubs.StubVGG, MaxOut_MLP=stubs.StubMLP),
           'ntu': dict(Visual=stubs.StubVisual, Skeleton=stubs.StubSkeleton),
           'ego': dict(get_rgb_model=stubs.ego_rgb_model, get_depth_model=stubs.ego_depth_model)}


def _install(monkeypatch, task='mmimdb'):
    central = types.ModuleType('models.central')
    fake = types.ModuleType('models.central.' + task)
    for k, v in CENTRAL[task].items():
        setattr(fake, k, v)
    setattr(central, task, fake)
    monkeypatch.setitem(sys.modules, 'models.central', central)
    monkeypatch.setitem(sys.modules, 'models.central.' + task, fake)
    import importlib
    drv = importlib.import_module('models.search.%s_darts_searchable' % task)
    import models.search.train_searchable._loop as loop
    rec = stubs.Recorder()

    def observer(event, **k):
        if event == 'batch':
            # (a k-step replay reports the rates each of its batches was STAGED with, decoded from the Adam scalars)
            lr = k['lr'][0] if k.get('lr') is not None else float(k['optimizer'].param_groups[0]['lr'])
            rec.batch(k['epoch'], k['phase'], k['learn'], k['loss'].detach(), k['output'], lr)
        else:
            rec.phase(k['epoch'], k['phase'], k['loss'], k['metric'], k['genotype'])

    monkeypatch.setattr(loop.run, 'observer', observer, raising=False)
    return drv, loop, rec


def _compare_batches(got, want, rel, label):
    assert len(got) == len(want), (label, len(got), len(want))
    worst = 0.0
    for i, (g, w) in enumerate(zip(got, want)):
        assert g[:3] == w[:3], (label, i, g[:3], w[:3])                       # epoch, phase, learn
        assert abs(g[3] - w[3]) <= rel * max(1.0, abs(w[3])), (label, 'loss', i, g[3], w[3])
        # logits: l2 within rel, the sum within rel * l2 * sqrt(n) (n = batch * 23 <= 184)
        assert abs(g[5] - w[5]) <= rel * w[5], (label, 'logits l2', i, g[5], w[5])
        assert abs(g[4] - w[4]) <= rel * w[5] * 14.0, (label, 'logits sum', i, g[4], w[4])
        if w[2] and w[6] >= 0:
            assert abs(g[6] - w[6]) <= 1e-9 + 2e-6 * w[6], (label, 'lr', i, g[6], w[6])   # (k-step: decoded from fp32)
        worst = max(worst, abs(g[3] - w[3]) / max(1.0, abs(w[3])), abs(g[5] - w[5]) / w[5])
    print(f'{label}: {len(got)} batches, worst relative deviation {worst:.2e}')


def _compare_phases(rec, gold, label, f1_tol=1e-9):
    assert [p[:2] for p in rec.phases] == [p[:2] for p in gold['phases']], label
    for g, w in zip(rec.phases, gold['phases']):
        assert abs(g[2] - w[2]) <= 1e-4 + 5e-5, (label, 'epoch loss', g, w)      # the golden has the logged 4 decimals
        assert abs(g[3] - w[3]) <= f1_tol, (label, 'F1', g, w)
    assert rec.genotypes == gold['genotypes'], label


def _compare_state(model, want, rel, label):
    sd = model.state_dict()
    for k, w in want.items():
        if k == '_nbt':
            for kk, v in w.items():
                assert int(sd[kk]) == v, (label, kk)
            continue
        g = stubs.summary(sd[k])
        n = sd[k].numel()
        l2 = max(abs(w[1]), 1e-12)
        if k.endswith('conv.bias'):
            # a conv bias in front of a train-mode BatchNorm has a mathematically zero gradient: what Adam normalises
            # there is weight decay + round-off, element by element — only the tensor as a whole is comparable
            assert abs(g[1] - w[1]) <= 1e-2 * l2, (label, k, 'l2', g[1], w[1])
            continue
        assert abs(g[1] - w[1]) <= rel * l2 + 1e-7, (label, k, 'l2', g[1], w[1])
        assert abs(g[0] - w[0]) <= rel * l2 * np.sqrt(n) + 1e-6, (label, k, 'sum', g[0], w[0])
        assert np.all(np.abs(np.array(g[2:]) - np.array(w[2:])) <= rel * (np.abs(w[2:]) + l2 / np.sqrt(n)) + 1e-7), \
            (label, k, g[2:], w[2:])


def _gold(name, seed_data=21):
    with open(os.path.join(GOLD, name)) as f:
        g = json.load(f)
    g['seed_data'] = seed_data               # make_golden_r04.SEED (21) / _ntu_ego.SEED (33): loaders are seeded SEED + i
    return g


@pytest.mark.parametrize('hip_graph', [False, True, 'k2'], ids=['eager', 'graph', 'graph-k2'])
def test_search_loop_reproduces_the_reference_trainer(tmp_path, monkeypatch, hip_graph):
    gold = _gold('loop_mmimdb_search.json')
    drv, loop, rec = _install(monkeypatch)
    from models.search.darts.utils import create_exp_dir
    made = []

    class Pinned(drv.Searchable_Image_Text_Net):
        def __init__(self, args, criterion):
            super().__init__(args, criterion)
            stubs.fill_state(self, gold['seed'])
            made.append(self)

    monkeypatch.setattr(drv, 'Searchable_Image_Text_Net', Pinned)
    a = _args(str(tmp_path / 'exp'), hip_graph=hip_graph)
    create_exp_dir(a.save)
    best_f1, genotype = drv.train_darts_model(_loaders(gold), a, torch.device('cuda:0'), logging.getLogger('bmnas-test'))
    label = 'search/' + ('graph' if hip_graph else 'eager')
    if hip_graph:
        # (k2: the two full batches of each epoch's train phase go out as ONE replay, and the first batch the single-step
        # path then sees is the ragged one — it captures THAT shape, so the tails replay too instead of running eagerly)
        want_replays = 6 if hip_graph == 'k2' else 4
        assert loop.run.stats['graph_replays'] == want_replays and loop.run.stats['forward_replays'] >= 2, loop.run.stats
        if hip_graph == 'k2':
            assert loop.run.stats['k_step_replays'] == 2, loop.run.stats
    else:
        assert loop.run.stats['graph_replays'] == 0, loop.run.stats
    _compare_batches(rec.batches, gold['batches'], 2e-4, label)
    _compare_phases(rec, gold, label)
    assert abs(best_f1 - gold['best_f1']) <= 1e-9
    assert str(genotype) == gold['best_genotype']
    with open(os.path.join(a.save, 'best', 'best_genotype.pkl'), 'rb') as f:
        assert str(pickle.load(f)) == gold['best_genotype']
    model = made[0]
    for p, w in zip(model.arch_parameters(), gold['arch']):
        want = np.asarray(w, dtype=np.float64)
        assert np.abs(p.detach().cpu().double().numpy() - want).max() <= 2e-4 * np.abs(want).max(), label
    _compare_state(model, gold['final'], 5e-4, label + ' final')
    ckpt = torch.load(os.path.join(a.save, 'best', 'best_model.pt'))
    for k, w in gold['best_ckpt'].items():
        g = stubs.summary(ckpt[k])
        assert abs(g[1] - w[1]) <= 5e-4 * max(abs(w[1]), 1e-12) + 1e-7, (label, 'best checkpoint', k, g[1], w[1])


@pytest.mark.parametrize('hip_graph', [False, True, 'k2'], ids=['eager', 'graph', 'graph-k2'])
def test_found_stage_reproduces_the_reference_trainer_and_tester(tmp_path, monkeypatch, hip_graph):
    gold = _gold('loop_mmimdb_found.json')
    drv, loop, rec = _install(monkeypatch)
    import models.auxiliary.scheduler as sc
    import models.search.train_searchable.mmimdb as tr
    from bmnas import nn as bnn
    from bmnas.optim import Adam
    from models.search.darts.genotypes import Genotype, StepGenotype
    from models.search.darts.utils import create_exp_dir
    from models.search.plot_genotype import Plotter
    genotype = Genotype(
        edges=[('skip', 1), ('skip', 4), ('skip', 0), ('skip', 5)],
        steps=[StepGenotype(inner_edges=[('skip', 0), ('skip', 1), ('skip', 2), ('skip', 0)],
                            inner_steps=['ScaleDotAttn', 'LinearGLU'], inner_concat=[2, 3]),
               StepGenotype(inner_edges=[('skip', 1), ('skip', 0), ('skip', 1), ('skip', 2)],
                            inner_steps=['ConcatFC', 'Sum'], inner_concat=[2, 3])],
        concat=[6, 7])
    a = _args(str(tmp_path / 'exp'), found=True, hip_graph=hip_graph)
    create_exp_dir(a.save)
    device = torch.device('cuda:0')
    criterion = bnn.BCEWithLogitsLoss()
    model = stubs.fill_state(drv.Found_Image_Text_Net(a, criterion, genotype), gold['seed'])
    lds = _loaders(gold)
    sizes = {k: len(v.dataset) for k, v in lds.items()}
    model.to(device)
    optimizer = Adam(model.parameters(), lr=a.eta_max, weight_decay=1e-4)
    scheduler = sc.LRCosineAnnealingScheduler(a.eta_max, a.eta_min, a.Ti, a.Tm, sizes['train'] / a.batchsize)
    logger = logging.getLogger('bmnas-test')
    test_f1, test_genotype = tr.train_mmimdb_track_f1(model, None, criterion, optimizer, scheduler, lds, sizes, device,
                                                      a.epochs, False, logger, Plotter(a), a, a.f1_type, 0.0, 0.3,
                                                      'eval')
    label = 'found/' + ('graph' if hip_graph else 'eager')
    _compare_batches(rec.batches, gold['batches'], 2e-4, label)
    _compare_phases(rec, gold, label)
    assert abs(test_f1 - gold['test_f1']) <= 1e-9 and str(test_genotype) == gold['test_genotype']
    _compare_state(model, gold['final'], 5e-4, label + ' final')
    # the tester on the reloaded best-test weights (main_darts_found_mmimdb.py:131-139)
    model2 = drv.Found_Image_Text_Net(a, criterion, genotype)
    model2.load_state_dict(torch.load(os.path.join(a.save, 'best', 'best_test_model.pt')))
    model2.to(device)
    rec.batches.clear()
    got = tr.test_mmimdb_track_f1(model2, criterion, lds, sizes, device, False, logger, a, a.f1_type, init_f1=0.0,
                                  th_fscore=0.3)
    assert abs(got - gold['tester_f1']) <= 1e-9


# ---------------------------------------------------------------------------------------------------------------------
# The accuracy-tracking trainers: NTU RGB+D (search, found stage + tester) and EgoGesture (search), against
# tests/golden/loop_{ntu_search,ntu_found,ego_search}.json (tests/golden/make_golden_r04_ntu_ego.py ran the reference's
# ntu_darts_searchable.py:21-72 / train_searchable/ntu.py:12-227 / ego_darts_searchable.py:20-69 /
# train_searchable/ego.py:13-223 on the same stand-in data).

def _acc_args(tmp_path, ns, nm, nout, hip_graph):
    a = _args(str(tmp_path / 'exp'), hip_graph=hip_graph)
    a.C, a.L, a.drpt = 32, 8, 0.2
    a.num_input_nodes = 8
    a.node_steps, a.node_multiplier, a.num_outputs = ns, nm, nout
    a.parallel = False
    a.checkpointdir = str(tmp_path)
    a.ske_cp, a.rgb_cp, a.depth_cp = 'ske.pt', 'rgb.pt', 'depth.pt'
    for name in (a.ske_cp, a.rgb_cp, a.depth_cp):
        torch.save({}, os.path.join(a.checkpointdir, name))
    return a


def _check_search(gold, a, rec, loop, made, best_acc, genotype, label, hip_graph, arch_rel=2e-4):
    if hip_graph:
        # (k2: the two full batches of each epoch's train phase go out as ONE replay, and the first batch the single-step
        # path then sees is the ragged one — it captures THAT shape, so the tails replay too instead of running eagerly)
        want_replays = 6 if hip_graph == 'k2' else 4
        assert loop.run.stats['graph_replays'] == want_replays and loop.run.stats['forward_replays'] >= 2, loop.run.stats
        if hip_graph == 'k2':
            assert loop.run.stats['k_step_replays'] == 2, loop.run.stats
    else:
        assert loop.run.stats['graph_replays'] == 0, loop.run.stats
    _compare_batches(rec.batches, gold['batches'], 2e-4, label)
    _compare_phases(rec, gold, label)
    assert abs(float(best_acc) - gold['best_acc']) <= 1e-9
    assert str(genotype) == gold['best_genotype']
    with open(os.path.join(a.save, 'best', 'best_genotype.pkl'), 'rb') as f:
        assert str(pickle.load(f)) == gold['best_genotype']
    model = made[0]
    for p, w in zip(model.arch_parameters(), gold['arch']):
        want = np.asarray(w, dtype=np.float64)
        assert np.abs(p.detach().cpu().double().numpy() - want).max() <= arch_rel * np.abs(want).max(), label
    _compare_state(model, gold['final'], 5e-4, label + ' final')
    ckpt = torch.load(os.path.join(a.save, 'best', 'best_model.pt'))
    for k, w in gold['best_ckpt'].items():
        g = stubs.summary(ckpt[k])
        assert abs(g[1] - w[1]) <= 5e-4 * max(abs(w[1]), 1e-12) + 1e-7, (label, 'best checkpoint', k, g[1], w[1])


def _pin(monkeypatch, drv, cls_name, seed):
    made = []
    base = getattr(drv, cls_name)

    class Pinned(base):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            stubs.fill_state(self, seed)
            made.append(self)

    monkeypatch.setattr(drv, cls_name, Pinned)
    return made


@pytest.mark.parametrize('hip_graph', [False, True, 'k2'], ids=['eager', 'graph', 'graph-k2'])
def test_ntu_search_loop_reproduces_the_reference_trainer(tmp_path, monkeypatch, hip_graph):
    gold = _gold('loop_ntu_search.json', 33)
    drv, loop, rec = _install(monkeypatch, 'ntu')
    from models.search.darts.utils import create_exp_dir
    made = _pin(monkeypatch, drv, 'Searchable_Skeleton_Image_Net', gold['seed'])
    a = _acc_args(tmp_path, 2, 2, stubs.NTU_CLASSES, hip_graph)
    create_exp_dir(a.save)
    best_acc, genotype = drv.train_darts_model(_loaders(gold, stubs.NTUData), a, torch.device('cuda:0'),
                                               logging.getLogger('bmnas-test'))
    _check_search(gold, a, rec, loop, made, best_acc, genotype, 'ntu search/' + ('graph' if hip_graph else 'eager'),
                  hip_graph)


@pytest.mark.parametrize('hip_graph', [False, True, 'k2'], ids=['eager', 'graph', 'graph-k2'])
def test_ego_search_loop_reproduces_the_reference_trainer(tmp_path, monkeypatch, hip_graph):
    gold = _gold('loop_ego_search.json', 43)
    drv, loop, rec = _install(monkeypatch, 'ego')
    from models.search.darts.utils import create_exp_dir
    made = _pin(monkeypatch, drv, 'Searchable_RGB_Depth_Net', gold['seed'])
    a = _acc_args(tmp_path, 3, 3, stubs.EGO_CLASSES, hip_graph)
    create_exp_dir(a.save)
    best_acc, genotype = drv.train_darts_model(_loaders(gold, stubs.EgoData), a, None, torch.device('cuda:0'),
                                               logging.getLogger('bmnas-test'))
    _check_search(gold, a, rec, loop, made, best_acc, genotype, 'ego search/' + ('graph' if hip_graph else 'eager'),
                  hip_graph)


@pytest.mark.parametrize('hip_graph', [False, True, 'k2'], ids=['eager', 'graph', 'graph-k2'])
def test_ntu_found_stage_reproduces_the_reference_trainer_and_tester(tmp_path, monkeypatch, hip_graph):
    gold = _gold('loop_ntu_found.json', 33)
    drv, loop, rec = _install(monkeypatch, 'ntu')
    import models.auxiliary.scheduler as sc
    import models.search.train_searchable.ntu as tr
    from bmnas import nn as bnn
    from bmnas.optim import Adam
    from models.search.darts.genotypes import Genotype, StepGenotype
    from models.search.darts.utils import create_exp_dir
    from models.search.plot_genotype import Plotter
    genotype = Genotype(
        edges=[('skip', 2), ('skip', 7), ('skip', 4), ('skip', 8)],
        steps=[StepGenotype(inner_edges=[('skip', 0), ('skip', 1), ('skip', 2), ('skip', 1)],
                            inner_steps=['LinearGLU', 'ScaleDotAttn'], inner_concat=[2, 3]),
               StepGenotype(inner_edges=[('skip', 1), ('skip', 0), ('skip', 2), ('skip', 0)],
                            inner_steps=['Sum', 'ConcatFC'], inner_concat=[2, 3])],
        concat=[8, 9])
    a = _acc_args(tmp_path, 2, 2, stubs.NTU_CLASSES, hip_graph)
    create_exp_dir(a.save)
    device = torch.device('cuda:0')
    criterion = bnn.CrossEntropyLoss()
    model = stubs.fill_state(drv.Found_Skeleton_Image_Net(a, criterion, genotype), gold['seed'])
    lds = _loaders(gold, stubs.NTUData)
    sizes = {k: len(v.dataset) for k, v in lds.items()}
    model.to(device)
    optimizer = Adam(model.parameters(), lr=a.eta_max, weight_decay=1e-4)
    scheduler = sc.LRCosineAnnealingScheduler(a.eta_max, a.eta_min, a.Ti, a.Tm, sizes['train'] / a.batchsize)
    logger = logging.getLogger('bmnas-test')
    test_acc, test_genotype = tr.train_ntu_track_acc(model, None, criterion, optimizer, scheduler, lds, sizes,
                                                     device=device, num_epochs=a.epochs, parallel=False, logger=logger,
                                                     plotter=Plotter(a), args=a, status='eval')
    label = 'ntu found/' + ('graph' if hip_graph else 'eager')
    _compare_batches(rec.batches, gold['batches'], 2e-4, label)
    _compare_phases(rec, gold, label)
    assert abs(float(test_acc) - gold['test_acc']) <= 1e-9 and str(test_genotype) == gold['test_genotype']
    _compare_state(model, gold['final'], 5e-4, label + ' final')
    model2 = drv.Found_Skeleton_Image_Net(a, criterion, genotype)
    model2.load_state_dict(torch.load(os.path.join(a.save, 'best', 'best_test_model.pt')))
    model2.to(device)
    rec.batches.clear()
    got = tr.test_ntu_track_acc(model2, lds, criterion, genotype, sizes, device, logger, a)
    assert abs(float(got) - gold['tester_acc']) <= 1e-9
    assert len(rec.batches) in (0, len(gold['tester_batches']))
    for g, w in zip(rec.batches, gold['tester_batches']):
        assert abs(g[3] - w[3]) <= 2e-4 * max(1.0, abs(w[3])) and abs(g[5] - w[5]) <= 2e-4 * w[5], (label, g, w)


@pytest.mark.parametrize('hip_graph', [False, True, 'k2'], ids=['eager', 'graph', 'graph-k2'])
def test_ego_found_stage_reproduces_the_reference_trainer_and_tester(tmp_path, monkeypatch, hip_graph):
    """tests/golden/loop_ego_found.json (make_golden_r05_ego_found.py ran the reference's Found_RGB_Depth_Net through
    train_ego_track_acc(status='eval') and test_ego_track_acc as main_darts_found_ego.py:118-153 does): the last
    trainer / tester call chain of the mains without a recording (missing 4).  node_steps 3 /
    node_multiplier 3, reshape layers only where the genotype reads an input, every parameter optimised."""
    gold = _gold('loop_ego_found.json', 63)
    drv, loop, rec = _install(monkeypatch, 'ego')
    import models.auxiliary.scheduler as sc
    import models.search.train_searchable.ego as tr
    from bmnas import nn as bnn
    from bmnas.optim import Adam
    from models.search.darts.genotypes import Genotype, StepGenotype
    from models.search.darts.utils import create_exp_dir
    from models.search.plot_genotype import Plotter
    gg = gold['genotype']
    genotype = Genotype(edges=[tuple(e) for e in gg['edges']],
                        steps=[StepGenotype(inner_edges=[tuple(e) for e in st['inner_edges']],
                                            inner_steps=list(st['inner_steps']), inner_concat=list(st['inner_concat']))
                               for st in gg['steps']],
                        concat=list(gg['concat']))
    a = _acc_args(tmp_path, 3, 3, stubs.EGO_CLASSES, hip_graph)
    create_exp_dir(a.save)
    device = torch.device('cuda:0')
    criterion = bnn.CrossEntropyLoss()
    model = stubs.fill_state(drv.Found_RGB_Depth_Net(a, None, criterion, genotype), gold['seed'])
    lds = _loaders(gold, stubs.EgoData)
    sizes = {k: len(v.dataset) for k, v in lds.items()}
    model.to(device)
    optimizer = Adam(model.parameters(), lr=a.eta_max, weight_decay=1e-4)
    scheduler = sc.LRCosineAnnealingScheduler(a.eta_max, a.eta_min, a.Ti, a.Tm, sizes['train'] / a.batchsize)
    logger = logging.getLogger('bmnas-test')
    test_acc, test_genotype = tr.train_ego_track_acc(model, None, criterion, optimizer, scheduler, lds, sizes, device,
                                                     a.epochs, False, logger, Plotter(a), a, 'eval')
    label = 'ego found/' + ('graph' if hip_graph else 'eager')
    _compare_batches(rec.batches, gold['batches'], 2e-4, label)
    _compare_phases(rec, gold, label)
    assert abs(float(test_acc) - gold['test_acc']) <= 1e-9 and str(test_genotype) == gold['test_genotype']
    _compare_state(model, gold['final'], 5e-4, label + ' final')
    model2 = drv.Found_RGB_Depth_Net(a, None, criterion, genotype)
    model2.load_state_dict(torch.load(os.path.join(a.save, 'best', 'best_test_model.pt')))
    model2.to(device)
    rec.batches.clear()
    got = tr.test_ego_track_acc(model2, lds, criterion, genotype, sizes, device, logger, a)
    assert abs(float(got) - gold['tester_acc']) <= 1e-9
    assert len(rec.batches) in (0, len(gold['tester_batches']))
    for g, w in zip(rec.batches, gold['tester_batches']):
        assert abs(g[3] - w[3]) <= 2e-4 * max(1.0, abs(w[3])) and abs(g[5] - w[5]) <= 2e-4 * w[5], (label, g, w)
