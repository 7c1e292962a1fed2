"""CPU: host-side behaviour of the nn.Module mirror that needs no kernel — genotype() against
the reference's golden answers, state_dict / pickle compatibility, arch-parameter handling,
the stacked parameter storage of NodeMixedOp, scheduler arithmetic."""
import copy
import io
import json
import pickle

import numpy as np
import pytest
import torch

from oracle import fusion_oracle as fo
from oracle import synth
from util import golden_files


class Args:
    def __init__(self, cfg):
        self.C, self.L, self.drpt = cfg.C, cfg.L, cfg.drpt
        self.num_input_nodes, self.num_keep_edges = cfg.N, 2
        self.node_steps, self.node_multiplier = cfg.ns, cfg.nm
        self.steps, self.multiplier = cfg.S, cfg.M
        self.parallel, self.weight_decay = False, 1e-4


def make_net(cfg):
    from models.search.darts.model_search import FusionNetwork
    return FusionNetwork(cfg.S, cfg.M, cfg.N, 2, Args(cfg))


def test_genotype_matches_reference_golden():
    with open(golden_files('genotypes.json')[0]) as f:
        cases = json.load(f)
    for c in cases:
        cfg = fo.Cfg({**c['cfg'], 'C': 16, 'L': 8})
        net = make_net(cfg)
        for dst, src in zip(net.arch_parameters(), c['arch']):
            dst.data.copy_(torch.tensor(src))
        got = fo.genotype_to_jsonable(net.genotype())
        assert got == c['genotype'], (c['kind'], c['seed'])


def test_genotype_exhausted_pairs_raises_like_reference():
    net = make_net(fo.make_cfg(N=2, C=16, L=8, S=2, M=2))
    with pytest.raises(IndexError):
        net.genotype()


@pytest.mark.parametrize('name', ['mmimdb', 'ntu', 'ego'])
def test_state_dict_keys_shapes_and_param_counts(name):
    cfg = fo.CONFIGS[name]
    net = make_net(cfg)
    sd, want = net.state_dict(), fo.param_shapes(cfg)
    assert set(sd) == set(want)
    for k in sd:
        assert tuple(sd[k].shape) == tuple(want[k]), k
    n_params = sum(p.numel() for p in net.parameters())
    assert n_params == {'mmimdb': 482688, 'ntu': 480512, 'ego': 716288}[name]
    assert [tuple(a.shape) for a in net.arch_parameters()] == fo.arch_shapes(cfg)
    # alphas/betas/gammas are NOT parameters / state (reference: unregistered leaf tensors)
    ids = {id(p) for p in net.parameters()}
    assert all(id(a) not in ids and a.requires_grad and a.is_leaf for a in net.arch_parameters())
    assert float(net.alphas_edges.abs().max()) < 0.01                                # 1e-3 * randn init


def test_load_state_dict_roundtrip_and_stacked_storage():
    """NodeMixedOp keeps LinearGLU/ConcatFC conv+BN tensors as views of stacked buffers; this
    must be invisible: load_state_dict, deepcopy and optimizers see ordinary parameters."""
    cfg = fo.make_cfg(N=3, C=16, L=8, ns=2, nm=2)
    net = make_net(cfg)
    p = synth.make_params(cfg, 3)
    net.load_state_dict(p)
    op = net.cell._step_nodes[0].node_cell.node_ops[1]
    pk = op.pack()
    C = cfg.C
    glu, cfc = op._ops[2], op._ops[3]
    assert glu.conv.weight.data_ptr() == pk.stack_W.data_ptr()
    assert cfc.conv.weight.data_ptr() == pk.stack_W[2 * C:].data_ptr()
    key = 'cell._step_nodes.0.node_cell.node_ops.1._ops'
    assert torch.equal(pk.stack_W[:2 * C].view(2 * C, 2 * C, 1), p[f'{key}.2.conv.weight'])
    assert torch.equal(pk.stack_W[2 * C:].view(C, 2 * C, 1), p[f'{key}.3.conv.weight'])
    assert torch.equal(pk.stack_rv[2 * C:], p[f'{key}.3.bn.running_var'])
    # in-place optimizer-style updates go through the views
    with torch.no_grad():
        cfc.conv.weight.add_(1.0)
    assert torch.equal(op.pack().stack_W[2 * C:].view(C, 2 * C, 1), p[f'{key}.3.conv.weight'] + 1.0)
    # state_dict round trip through a fresh network
    buf = io.BytesIO()
    torch.save(net.state_dict(), buf)
    buf.seek(0)
    net2 = make_net(cfg)
    net2.load_state_dict(torch.load(buf))
    for k, v in net.state_dict().items():
        assert torch.equal(v, net2.state_dict()[k]), k
    # deepcopy breaks the sharing; pack() re-establishes it without changing values
    net3 = copy.deepcopy(net)
    op3 = net3.cell._step_nodes[0].node_cell.node_ops[1]
    pk3 = op3.pack()
    assert op3._ops[2].conv.weight.data_ptr() == pk3.stack_W.data_ptr() != pk.stack_W.data_ptr()
    assert torch.equal(pk3.stack_W, op.pack().stack_W)


def test_genotype_pickle_uses_reference_module_path():
    net = make_net(fo.make_cfg(N=3, C=16, L=8))
    g = net.genotype()
    blob = pickle.dumps(g)
    assert b'models.search.darts.genotypes' in blob
    g2 = pickle.loads(blob)
    assert g2 == g and type(g2).__name__ == 'Genotype' and g2._fields == ('edges', 'steps', 'concat')
    assert g2.steps[0]._fields == ('inner_edges', 'inner_steps', 'inner_concat')


def test_arch_tensors_follow_module_apply_and_keep_identity():
    """The reference creates the arch optimizer BEFORE model.to(device): tensor identity must
    survive .to()/.double() so the optimizer keeps stepping the live tensors."""
    net = make_net(fo.make_cfg(N=3, C=16, L=8))
    arch = list(net.arch_parameters())
    opt = torch.optim.Adam(arch, lr=3e-4, betas=(0.5, 0.999), weight_decay=1e-3)
    net.double()
    assert all(a is b for a, b in zip(arch, net.arch_parameters()))
    assert all(a.dtype == torch.float64 for a in arch)
    net.float()
    for a in arch:
        a.grad = torch.ones_like(a)
    before = [a.detach().clone() for a in arch]
    opt.step()
    assert all(not torch.equal(a, b) for a, b in zip(arch, before))


def test_registries_and_public_names():
    from models.search.darts import genotypes, node_operations, operations
    assert genotypes.PRIMITIVES == ['none', 'skip'] == genotypes.STEP_EDGE_PRIMITIVES
    assert genotypes.STEP_STEP_PRIMITIVES == ['Sum', 'ScaleDotAttn', 'LinearGLU', 'ConcatFC']
    assert set(operations.OPS) == {'none', 'fc_relu', 'fc_mish', 'skip'}
    assert set(node_operations.STEP_STEP_OPS) == {'Sum', 'ScaleDotAttn', 'LinearGLU', 'ConcatFC'}
    from models.search.darts.architect import Architect        # noqa: F401
    from models.search.darts.model import Found_FusionNetwork   # noqa: F401
    from models.search.darts.node import Found_FusionNode, Found_NodeCell   # noqa: F401
    from models.search.darts.utils import (count_parameters, create_exp_dir, load, load_pickle,   # noqa: F401
                                           save, save_pickle)


def test_found_network_state_dict_keys():
    from models.search.darts.model import Found_FusionNetwork
    cfg = fo.make_cfg(N=4, C=16, L=8, ns=3, nm=2)
    net = make_net(cfg)
    for a in net.arch_parameters():
        a.data.normal_()
    g = net.genotype()
    f = Found_FusionNetwork(cfg.S, cfg.M, cfg.N, 2, Args(cfg), None, g)
    og = fo.genotype_from_jsonable(fo.genotype_to_jsonable(g))
    assert set(f.state_dict()) == set(fo.found_param_shapes(cfg, og))
    assert f.get_genotype() is g


def test_graph_step_switch_precedence(monkeypatch):
    """GraphedTrainStep.enabled: args.hip_graph beats BMNAS_HIP_GRAPH beats the default (on for a
    single process); never on without a GPU."""
    import torch
    from bmnas.graph import GraphedTrainStep

    class A:
        pass

    monkeypatch.delenv('BMNAS_HIP_GRAPH', raising=False)
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    a = A()
    assert GraphedTrainStep.enabled(a) is True                 # default: single process -> on
    monkeypatch.setenv('BMNAS_HIP_GRAPH', '0')
    assert GraphedTrainStep.enabled(a) is False
    a.hip_graph = True
    assert GraphedTrainStep.enabled(a) is True                 # explicit argument wins over the environment
    a.hip_graph = False
    monkeypatch.setenv('BMNAS_HIP_GRAPH', '1')
    assert GraphedTrainStep.enabled(a) is False
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: False)
    a.hip_graph = True
    assert GraphedTrainStep.enabled(a) is False                # no GPU, no graphs


def test_live_autograd_graph_detector():
    """The guard in front of every capture: non-leaf tensors that are still referenced count."""
    import torch
    from bmnas.graph import GraphedTrainStep
    dev = torch.device('cpu')
    base = GraphedTrainStep._live_graph_tensors(dev)
    w = torch.ones(3, requires_grad=True)
    held = (w * 2).sum()                                       # a loss someone keeps
    assert GraphedTrainStep._live_graph_tensors(dev) == base + 1
    held = held.detach()
    assert GraphedTrainStep._live_graph_tensors(dev) == base


def test_trainer_loop_shards_unsharded_batches(monkeypatch):
    """Data parallelism through the reference's call chain: a loader without a DistributedSampler yields
    the GLOBAL batch on every rank; the trainer loop keeps this rank's contiguous slice, cut the way
    DataParallel's scatter (Tensor.chunk) cuts it."""
    import torch
    import models.search.train_searchable._loop as loop
    from bmnas import dist as bdist
    x = (torch.arange(9).float().view(9, 1), torch.arange(18).float().view(9, 2))
    y = torch.arange(9)
    try:
        for rank in range(3):
            monkeypatch.setattr(loop, '_world', lambda: 3)
            monkeypatch.setattr(loop, '_rank', lambda r=rank: r)
            (a, b), lab = loop._shard_batch(x, y)
            assert lab.tolist() == [3 * rank, 3 * rank + 1, 3 * rank + 2]
            assert a.shape == (3, 1) and b.shape == (3, 2) and float(a[0, 0]) == 3 * rank
            assert bdist.shard_weight() == 1.0                     # equal shards: plain mean of means
    finally:
        bdist.set_shard_weight(1.0)


def test_uneven_global_batch_is_scattered_like_dataparallel(monkeypatch):
    """nn.DataParallel's scatter (mmimdb_darts_searchable.py:36-37 -> comm.scatter -> Tensor.chunk) hands a global batch
    of 10 to 3 replicas as chunks of 4 / 4 / 2, 100 over 8 as 13 x 7 + 9, 9 over 8 as 2, 2, 2, 2, 1 + three idle replicas,
    and the criterion takes ONE mean over all n gathered outputs.  The loop cuts the same slices — checked against
    torch.chunk itself — and weights each rank's shard mean by n_rank * world / n, so that the average over ranks IS that
    mean; no sample is dropped."""
    import torch
    import models.search.train_searchable._loop as loop
    from bmnas import dist as bdist
    try:
        for n, world in ((10, 3), (100, 8), (9, 8), (128, 8), (7, 2), (1, 4)):
            x, y = torch.arange(n).float().view(n, 1), torch.arange(n)
            want = [c.tolist() for c in y.chunk(world)]
            want += [[] for _ in range(world - len(want))]
            monkeypatch.setattr(loop, '_world', lambda w=world: w)
            kept, wsum = [], 0.0
            vals = torch.randn(n, dtype=torch.float64)
            mean_of_weighted = 0.0
            for rank in range(world):
                monkeypatch.setattr(loop, '_rank', lambda r=rank: r)
                xs, lab = loop._shard_batch(x, y)
                assert lab.tolist() == want[rank] and xs.shape[0] == len(want[rank])
                assert bdist.uneven_bounds(n, rank, world)[1] == len(want[rank])
                w = bdist.shard_weight()
                assert abs(w - len(want[rank]) * world / n) < 1e-12
                kept += lab.tolist()
                wsum += w
                if len(want[rank]):
                    mean_of_weighted += w * float(vals[lab].mean()) / world
            assert kept == list(range(n))                          # every sample, once, in order
            assert abs(wsum - world) < 1e-9
            assert abs(mean_of_weighted - float(vals.mean())) < 1e-12
    finally:
        bdist.set_shard_weight(1.0)
    # the epoch metrics divide by what the ranks processed together, not by the dataset size (run(): n = all-reduced `seen`)
    import inspect
    assert '_all_sum(torch.tensor(float(seen)' in inspect.getsource(loop.run)


def test_step_arena_slices_have_the_requested_length():
    """LinearFn views its arena slice as (b, O); 6 x 83 = 498 floats must come back as 498 (the cursor still
    advances by 500 so that the next slice stays 16-byte aligned)."""
    from bmnas.functions import _StepArena
    a = _StepArena(torch.zeros(1024))
    v = a.take(6 * 83)
    assert v.numel() == 498 and v.view(6, 83).shape == (6, 83) and a.off == 500
    w = a.take(8)
    assert w.data_ptr() == a.buf.data_ptr() + 500 * 4 and w.data_ptr() % 16 == 0
    assert a.take(1024) is None and a.off == 508                   # too large: nothing handed out, cursor unchanged


def test_adam_fast_activate_waits_for_the_staging_buffer_of_copy_node_plans(monkeypatch):
    """the no-op fast path of Adam.activate() must still wait for the previous replay's H2D copy node when the
    captured plan is NOT in poke mode (capture_safe() without poke, or more than 8 scalar rows): prepare_replay() rewrites
    the pinned staging buffer that node reads."""
    from bmnas.optim import Adam
    opt = Adam([torch.nn.Parameter(torch.zeros(2))], lr=1e-3)
    waits = []
    monkeypatch.setattr(opt, 'wait_staging', lambda: waits.append(1))
    for poke, want in ((False, 1), (True, 0)):
        del waits[:]
        plan = dict(gen=opt._gen, stamp=opt._touch, poke=poke)
        opt._plan = plan
        opt.activate(plan)
        assert len(waits) == want


def test_deferred_affine_launches_a_repeated_or_non_leaf_layernorm_at_once(monkeypatch):
    """a deferred LayerNorm-affine job hands unfilled (dweight, dbias) to autograd; that is only safe for a
    leaf weight seen once per pass.  A second job for the same weight launches both at once; a non-leaf weight is never
    deferred."""
    from bmnas import functions as F
    launched = []
    monkeypatch.setattr(F.lib, 'ln_affine_bwd', lambda *a: launched.append(a[7]))      # a[7] = dln_w
    monkeypatch.setattr(F.lib, 'ln_affine_bwd_multi', lambda probs, b, L: launched.extend(p['dln_w'] for p in probs))
    t = torch.zeros(1)
    with F.deferred_affine():
        F._ln_affine(t, [t], None, t, t, t, 'dw_a', 'db', 2, 4, 8, False, False, key=11, leaf=True)
        assert launched == []                                          # first use of a leaf: deferred
        F._ln_affine(t, [t], None, t, t, t, 'dw_b', 'db', 2, 4, 8, False, False, key=22, leaf=False)
        assert launched == ['dw_b']                                    # not a leaf: at once
        F._ln_affine(t, [t], None, t, t, t, 'dw_a2', 'db', 2, 4, 8, False, False, key=11, leaf=True)
        assert launched == ['dw_b', 'dw_a', 'dw_a2']                   # repeated: the pending job first, then this one
        F._ln_affine(t, [t], None, t, t, t, 'dw_c', 'db', 2, 4, 8, False, False, key=33, leaf=True)
    assert launched == ['dw_b', 'dw_a', 'dw_a2', 'dw_c']               # the one still pending goes out at the exit


def test_forward_graph_cache_holds_its_criterion():
    """the captured-forward cache is keyed by id(criterion); an id re-used by a NEW criterion (the old one
    freed, e.g. rebuilt per stage with another pos_weight) must not get the old graphs, whose criterion is baked in."""
    import models.search.train_searchable._loop as loop

    class M:
        pass

    class Args:
        graph_step = False

    model, c1 = M(), torch.nn.BCEWithLogitsLoss()
    fg1 = loop._ForwardGraphs.of(model, c1, Args())
    assert loop._ForwardGraphs.of(model, c1, Args()) is fg1                   # same object: same graphs
    cache = model.__dict__['_bmnas_forward_graphs']
    c2 = torch.nn.BCEWithLogitsLoss(pos_weight=torch.ones(3))
    cache[id(c2)] = cache.pop(id(c1))                                          # simulate the id collision
    fg2 = loop._ForwardGraphs.of(model, c2, Args())
    assert fg2 is not fg1 and cache[id(c2)][0] is c2


def test_world_size_alone_requests_data_parallelism(monkeypatch):
    from models.search._common import data_parallel_world, parallel_flag

    class A:
        use_dataparallel = False

    monkeypatch.delenv('WORLD_SIZE', raising=False)
    assert data_parallel_world(A()) == 1
    monkeypatch.setenv('WORLD_SIZE', '8')
    assert data_parallel_world(A()) == 8 and parallel_flag(A()) is False


def test_head_state_resolves_deferred_and_given_gradients():
    import torch
    from bmnas.cell import HeadState, StatArena
    h = HeadState(marker=torch.zeros(4, 3))
    g = torch.ones(4, 3)
    assert h.resolve(g)[0] == 0 and h.resolve(g)[1] is g
    labels = torch.zeros(4, 3)
    h.deferred = ('bce', labels)
    # a criterion was deferred into the head's backward launch, yet another gradient reached the logits
    # (autograd summed it with the marker into a new tensor): there is no dlogits to add it to — refuse loudly
    from bmnas.lib import BmnasError
    with pytest.raises(BmnasError, match='besides the deferred'):
        h.resolve(g)
    mode, gt, gs, lab = h.resolve(h.marker)
    assert (mode, gt, gs) == (1, None, None) and lab is labels
    h.deferred = ('ce', labels)
    assert h.resolve(h.marker)[0] == 2
    a = StatArena(torch.zeros(1), [576, 576, 128])
    v1, v2, v3 = a.take(576), a.take(576), a.take(128)
    assert v1.numel() == v2.numel() and v3.numel() * 576 == v1.numel() * 128 and a.buf.numel() % 4 == 0
    with pytest.raises(Exception):
        a.take(16)


def test_zero_pool_one_fill_per_backward_pass():
    """bmnas.functions._ZeroPool: forwards announce, the first backward take allocates for all of them,
    slices are disjoint, zero and 16-byte aligned; the next forward starts a new pass; unexpected takes
    get their own fill."""
    from bmnas.functions import _ZeroPool
    dev = torch.device('cpu')
    pool = _ZeroPool()
    for n in (10, 7, 33):
        pool.announce(n)
    a, b, c = pool.take(33, dev), pool.take(7, dev), pool.take(10, dev)
    base = pool.chunk.data_ptr()
    assert pool.chunk.numel() == 12 + 8 + 36
    assert [t.data_ptr() - base for t in (a, b, c)] == [0, 36 * 4, 44 * 4]
    assert all(float(t.abs().sum()) == 0.0 and t.data_ptr() % 16 == 0 for t in (a, b, c))
    extra = pool.take(5, dev)                          # beyond what was announced (second backward, ...)
    assert extra.numel() == 8 and not (base <= extra.data_ptr() < base + pool.chunk.numel() * 4)
    pool.announce(4)                                   # next step's forward
    assert not pool.in_backward and pool.pending == 4
    d = pool.take(4, dev)
    assert pool.chunk.numel() == 4 and d.data_ptr() == pool.chunk.data_ptr()
    lone = _ZeroPool().take(6, dev)                    # a module that never announced
    assert lone.numel() == 8


def test_forward_stat_pool_sizes_a_pass_by_the_previous_one():
    from bmnas import cell
    from bmnas.functions import _FwdStatPool
    pool = _FwdStatPool()
    pool.device = torch.device('cpu')
    n = lambda M: (cell.STAT_SHARDS * M * 2 + 3) // 4 * 4
    first = [pool.take(16), pool.take(32)]             # first pass: nothing known, one fill each
    assert pool.chunk is None and [t.numel() for t in first] == [n(16), n(32)]
    pool.close()                                       # a backward pass began
    assert pool.last_total == n(16) + n(32)
    a, b = pool.take(16), pool.take(32)                # second pass: one chunk
    assert pool.chunk.numel() == n(16) + n(32)
    assert a.data_ptr() == pool.chunk.data_ptr() and b.data_ptr() == a.data_ptr() + n(16) * 4
    c = pool.take(64)                                  # more than last time: own fill, remembered for the next pass
    assert c.numel() == n(64) and float(c.abs().sum()) == 0.0
    pool.close()
    assert pool.last_total == n(16) + n(32) + n(64)


def test_searcher_facades_build_loaders_and_hand_over(monkeypatch):
    """models/darts_searchable.py (reference :25-90): MMIMDB_Searcher / NTUSearcher / Ego_Searcher build the
    datasets' DataLoaders and call the per-dataset train_darts_model.  The datasets and models.utils are out of
    scope (reference checkout): stubs stand in; what is checked is the wiring — splits, batch size, shuffling,
    the arguments handed to train_darts_model — and the per-rank loaders under WORLD_SIZE > 1."""
    import sys
    import types
    from torch.utils.data import Dataset
    from torch.utils.data.distributed import DistributedSampler

    class DS(Dataset):
        def __init__(self, *a, **k):
            self.stage = k.get('stage')

        def __len__(self):
            return 10

        def __getitem__(self, i):
            return i

    tv = types.ModuleType('torchvision')
    tvt = types.ModuleType('torchvision.transforms')
    tvt.Compose = lambda ts: ts
    tv.transforms = tvt
    ds = types.ModuleType('datasets')
    mm = types.ModuleType('datasets.mmimdb')
    mm.ToTensor, mm.MM_IMDB = (lambda: 'tt'), DS
    nt = types.ModuleType('datasets.ntu')
    nt.NormalizeLen = nt.ToTensor = nt.AugCrop = lambda: 't'
    nt.NTU = DS
    eg = types.ModuleType('datasets.ego')
    eg.get_train_loader = lambda opt, args: ('train', opt)
    eg.get_dev_loader = lambda opt, args: ('dev', opt)
    eg.get_test_loader = lambda opt, args: ('test', opt)
    mu = types.ModuleType('models.utils')
    mu.parse_opts = lambda args: 'OPT'
    for name, mod in (('torchvision', tv), ('torchvision.transforms', tvt), ('datasets', ds),
                      ('datasets.mmimdb', mm), ('datasets.ntu', nt), ('datasets.ego', eg), ('models.utils', mu)):
        monkeypatch.setitem(sys.modules, name, mod)
    import models.darts_searchable as S
    import models.search.ego_darts_searchable as ego
    import models.search.mmimdb_darts_searchable as mmimdb
    import models.search.ntu_darts_searchable as ntu
    calls = []
    monkeypatch.setattr(mmimdb, 'train_darts_model', lambda *a: calls.append(('mmimdb', a)) or 'r1')
    monkeypatch.setattr(ntu, 'train_darts_model', lambda *a: calls.append(('ntu', a)) or 'r2')
    monkeypatch.setattr(ego, 'train_darts_model', lambda *a: calls.append(('ego', a)) or 'r3')

    class A:
        datadir, batchsize, num_workers = '/nowhere', 4, 0

    monkeypatch.delenv('WORLD_SIZE', raising=False)
    s = S.MMIMDB_Searcher(A(), 'dev0', 'log')
    assert set(s.dataloaders) == {'train', 'dev', 'test'} and s.dataloaders['train'].batch_size == 4
    assert s.dataloaders['dev'].dataset.stage == 'dev'
    assert s.search() == 'r1' and calls[-1][0] == 'mmimdb' and calls[-1][1][0] is s.dataloaders
    n = S.NTUSearcher(A(), 'dev0', 'log')
    assert n.dataloaders['train'].dataset.stage == 'train_exp' and n.search() == 'r2'
    assert calls[-1][1][1:] == (n.args, 'dev0', 'log')
    e = S.Ego_Searcher(A(), 'dev0', 'log')
    assert e.opt == 'OPT' and e.dataloaders['dev'] == ('dev', 'OPT') and e.search() == 'r3'
    assert calls[-1][1][2] == 'OPT'                       # ego's train_darts_model(dataloaders, args, opt, device, logger)
    # data parallel launch: per-rank loaders with a DistributedSampler and batchsize // world
    monkeypatch.setenv('WORLD_SIZE', '2')
    import torch.distributed as dist
    monkeypatch.setattr(dist, 'is_available', lambda: True)
    monkeypatch.setattr(dist, 'is_initialized', lambda: True)
    monkeypatch.setattr(dist, 'get_world_size', lambda group=None: 2)
    monkeypatch.setattr(dist, 'get_rank', lambda group=None: 1)
    s2 = S.MMIMDB_Searcher(A(), 'dev0', 'log')
    ld = s2.dataloaders['train']
    assert isinstance(ld.sampler, DistributedSampler) and ld.batch_size == 2 and len(list(ld.sampler)) == 5


def test_bench_line_is_the_median_region_of_the_fastest_shape():
    """bench.headline (host logic of the JSON line): value / ms_per_step come from the MEDIAN timed region of the
    fastest measured step shape, every shape is reported, `steps` stays the per-region count."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench as B

    class A:
        config, tier, batch, steps, warmup, mode = 'mmimdb', 'F', 128, 20, 5, 'graph'

    c = B.CONFIGS['mmimdb']
    shapes = {'host': [0.0050, 0.0046, 0.0048, 0.0047, 0.0100], 'graph': [0.0040, 0.0041, 0.0039, 0.0040, 0.0042]}
    d = B.headline(A(), c, 8, shapes, 'graph', 1.4, {'ranks': 8})
    assert d['n_gpus'] == 8 and d['steps'] == 20 and d['scaling'] == 'weak' and d['unit'] == 'steps/s'
    assert abs(d['ms_per_step'] - 0.2) < 1e-9 and abs(d['value'] - 8 * 20 / 0.0040) < 1e-6
    assert d['timed_regions'] == {'n': 5, 'steps_each': 20, 'headline': 'median',
                                  'ms_per_step': [0.2, 0.205, 0.195, 0.2, 0.21]}
    assert d['step_shapes']['headline'] == 'graph' and d['step_shapes']['host']['ms_per_step_median'] == 0.24
    assert 'inside one hipGraph' in d['config']['step'] and d['config']['global_batch'] == 1024
    assert d['rccl'] == {'ranks': 8} and d['vs_baseline'] is None and d['dtype'] == 'f32'
    single = B.headline(A(), c, 1, {'single': [0.0033] * 5}, 'single', None, None)
    assert 'step_shapes' not in single and 'rccl' not in single and single['config']['parallelism'] == 'dp1'


def test_captured_step_identifies_the_phase_by_its_optimizer_tensors():
    """bmnas.graph.classify_targets: the architecture step (every optimizer tensor is one of model.arch_parameters():
    architect.py:14-18) skips the weight gradients, the weight step (none is: train_searchable/*.py build their optimizer
    over model.parameters()) skips the architecture gradients; anything mixed, or a model without architecture tensors
    (the found stage), keeps the full backward."""
    import torch
    from bmnas.graph import classify_targets

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(3))
            self.v = torch.nn.Parameter(torch.zeros(2))
            self._arch = [torch.zeros(4, 2, requires_grad=True), torch.zeros(2, 2, requires_grad=True)]

        def arch_parameters(self):
            return self._arch

    net = Net()
    assert classify_targets(net, net.arch_parameters()) == (True, False)
    assert classify_targets(net, net.parameters()) == (False, True)
    assert classify_targets(net, [net.w, net.arch_parameters()[0]]) == (False, False)
    assert classify_targets(net, []) == (False, False)
    # an equal-valued copy is not the model's tensor: identity decides
    assert classify_targets(net, [net.arch_parameters()[0].clone()]) == (False, True)
    found = torch.nn.Linear(2, 2)                               # no arch_parameters(): nothing identified
    assert classify_targets(found, found.parameters()) == (False, False)
