#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference, read-only, imported with
sys.dont_write_bytecode so nothing is written there).  Nothing from the reference
is copied: the fixtures hold inputs' seeds and the reference's numeric OUTPUTS.
Inputs/params are regenerated bit-identically from oracle/synth.py seeds by the tests.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz / *.json

Recipe: stub IPython (every hot-path module does
``from IPython import embed``), import model_search before node_search (import cycle),
dropout made an identity for train-mode vectors via drpt=1e-12 and
ScaledDotAttn.dropout.p = 0 (never drpt=0: aliasing + in-place add breaks autograd).
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import fusion_oracle as fo
from oracle import synth


def import_reference():
    ipy = types.ModuleType('IPython')
    ipy.embed = lambda *a, **k: None
    sys.modules['IPython'] = ipy
    tv = types.ModuleType('torchvision')
    tvt = types.ModuleType('torchvision.transforms')
    tv.transforms = tvt
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.transforms'] = tvt
    sys.path.insert(0, REF)
    import models.search.darts.model_search as ms      # noqa: must precede node_search
    import models.search.darts.node_search as ns       # noqa
    import models.search.darts.model as mf
    import models.search.darts.architect as arch
    import models.search.darts.genotypes as gt
    return ms, ns, mf, arch, gt


class Args:
    pass


def ref_args(cfg, drpt):
    a = Args()
    a.C, a.L = cfg.C, cfg.L
    a.drpt = drpt
    a.num_input_nodes = cfg.N
    a.num_keep_edges = 2
    a.node_steps, a.node_multiplier = cfg.ns, cfg.nm
    a.steps, a.multiplier = cfg.S, cfg.M
    a.parallel = False
    a.weight_decay = 1e-4
    return a


def set_attn_dropout(model, p):
    for m in model.modules():
        if m.__class__.__name__ == 'ScaledDotAttn':
            m.dropout.p = p


def load_arch(net, arch):
    for dst, src in zip(net.arch_parameters(), arch):
        assert tuple(dst.shape) == tuple(src.shape), (dst.shape, src.shape)
        dst.data.copy_(src)


def build_ref_search(ms, cfg, seed, mode):
    """mode: 'eval' | 'train_nodrop'."""
    args = ref_args(cfg, 1e-12 if mode == 'train_nodrop' else cfg.drpt)
    net = ms.FusionNetwork(cfg.S, cfg.M, cfg.N, 2, args, criterion=None)
    shapes = fo.param_shapes(cfg)
    sd = net.state_dict()
    assert set(sd.keys()) == set(shapes.keys()), set(sd.keys()) ^ set(shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), (k, v.shape, shapes[k])
    net.load_state_dict(synth.make_params(cfg, seed))
    load_arch(net, synth.make_arch(cfg, seed))
    if mode == 'eval':
        net.eval()
    else:
        net.train()
        set_attn_dropout(net, 0.0)
    return net


def summarize(t, nsample=8):
    """[sum, l2, first nsample flat elements] in float64 for big tensors."""
    f = t.detach().double().reshape(-1)
    head = f[:nsample]
    if head.numel() < nsample:
        head = torch.cat([head, torch.zeros(nsample - head.numel(), dtype=torch.float64)])
    return torch.cat([f.sum()[None], f.norm()[None], head]).numpy()


HYPERNET_CASES = [
    # name, cfg, batch, num_outputs, loss, full(=store whole tensors)
    ('tiny_a', fo.make_cfg(N=3, C=16, L=8, S=2, M=2, ns=1, nm=1, drpt=0.1), 4, 5, 'bce', True),
    ('tiny_b', fo.make_cfg(N=3, C=16, L=8, S=2, M=2, ns=2, nm=2, drpt=0.2), 5, 7, 'ce', True),
    ('tiny_c', fo.make_cfg(N=4, C=32, L=16, S=2, M=2, ns=3, nm=3, drpt=0.1), 3, 4, 'ce', True),
    ('tiny_d', fo.make_cfg(N=2, C=16, L=16, S=3, M=3, ns=2, nm=1, drpt=0.1), 6, 3, 'bce', True),
    ('tiny_e', fo.make_cfg(N=5, C=48, L=8, S=1, M=1, ns=3, nm=2, drpt=0.1), 1, 3, 'ce', True),
    ('mmimdb_b8', fo.CONFIGS['mmimdb'], 8, 23, 'bce', False),
    ('ntu_b8', fo.CONFIGS['ntu'], 8, 60, 'ce', False),
    ('ego_b6', fo.CONFIGS['ego'], 6, 83, 'ce', False),
]


def run_hypernet_case(ms, name, cfg, batch, nout, loss_kind, full, mode, seed=7):
    net = build_ref_search(ms, cfg, seed, mode)
    cls = torch.nn.Linear(cfg.M * cfg.C * cfg.L, nout)
    cw, cb = synth.make_classifier(cfg, nout, seed)
    cls.weight.data.copy_(cw)
    cls.bias.data.copy_(cb)
    xs = [x.requires_grad_(True) for x in synth.make_inputs(cfg, batch, seed)]
    y = synth.make_labels(loss_kind, batch, nout, seed)
    crit = torch.nn.BCEWithLogitsLoss() if loss_kind == 'bce' else torch.nn.CrossEntropyLoss()
    # eval mode + node_multiplier != 1: Dropout(eval) aliases the ReLU output and the
    # reference's in-place ``out += x`` (node_search.py:67) then breaks autograd, so the
    # reference itself can only run that combination forward-only.
    has_grads = not (mode == 'eval' and cfg.nm != 1)
    with torch.set_grad_enabled(has_grads):
        feat = net(xs)
        logits = cls(feat)
        loss = crit(logits, y)
    if has_grads:
        loss.backward()
    out = {'meta': json.dumps(dict(name=name, cfg=dict(cfg), batch=batch, num_outputs=nout,
                                   loss=loss_kind, mode=mode, seed=seed, full=full,
                                   has_grads=has_grads))}
    out['logits'] = logits.detach().numpy()
    out['loss'] = loss.detach().numpy()
    keep = (lambda t: t.detach().numpy()) if full else summarize
    out['feat'] = keep(feat)
    for k, v in (net.named_parameters() if has_grads else []):
        out['grad:' + k] = keep(v.grad)
    if has_grads:
        out['grad:central_classifier.weight'] = keep(cls.weight.grad)
        out['grad:central_classifier.bias'] = keep(cls.bias.grad)
        for i, a in enumerate(net.arch_parameters()):
            out[f'grad:arch.{i}'] = a.grad.detach().numpy()       # always full (<= 94 floats)
        for i, x in enumerate(xs):
            out[f'grad:input.{i}'] = keep(x.grad)
    for k, v in net.state_dict().items():
        if fo.is_buffer(k):
            out['buf:' + k] = v.detach().numpy() if (full or v.dim() == 0) else summarize(v)
    return out


def make_hypernet(ms):
    for name, cfg, batch, nout, loss_kind, full in HYPERNET_CASES:
        for mode in ('eval', 'train_nodrop'):
            if mode == 'train_nodrop' and batch * cfg.L < 2:
                continue
            out = run_hypernet_case(ms, name, cfg, batch, nout, loss_kind, full, mode)
            path = os.path.join(HERE, f'hypernet_{name}_{mode}.npz')
            np.savez_compressed(path, **out)
            print('wrote', path, os.path.getsize(path))


# ------------------------------------------------------------------- genotype cases
def make_genotypes(ms):
    cases = []
    cfgs = [fo.CONFIGS['mmimdb'], fo.CONFIGS['ntu'], fo.CONFIGS['ego'],
            fo.make_cfg(N=3, C=16, L=8, S=2, M=2, ns=2, nm=1),
            fo.make_cfg(N=5, C=16, L=8, S=3, M=2, ns=3, nm=3),
            fo.make_cfg(N=2, C=16, L=8, S=1, M=1, ns=1, nm=1)]
    for ci, cfg in enumerate(cfgs):
        tiny = fo.make_cfg(**{**cfg, 'C': 16, 'L': 8})     # genotype() ignores C, L
        args = ref_args(tiny, 0.1)
        for seed in range(6):
            net = ms.FusionNetwork(cfg.S, cfg.M, cfg.N, 2, args, criterion=None)
            kind = ['randn', 'randn', 'randn', 'zeros', 'quantized', 'init'][seed]
            if kind == 'zeros':          # every comparison ties -> pure tie-breaking order
                arch = [torch.zeros(s) for s in fo.arch_shapes(cfg)]
            elif kind == 'quantized':    # few distinct values -> many engineered ties
                arch = [torch.round(a * 2) / 2 for a in synth.make_arch(cfg, 100 + seed, 1.0)]
            elif kind == 'init':         # reference-like 1e-3 scale
                arch = synth.make_arch(cfg, 100 + seed, 1e-3)
            else:
                arch = synth.make_arch(cfg, 100 + seed, 1.0)
            load_arch(net, arch)
            g = net.genotype()
            cases.append({'cfg': dict(cfg), 'kind': kind, 'seed': 100 + seed,
                          'arch': [a.numpy().tolist() for a in arch],
                          'genotype': fo.genotype_to_jsonable(g)})
    path = os.path.join(HERE, 'genotypes.json')
    with open(path, 'w') as f:
        json.dump(cases, f)
    print('wrote', path, len(cases), 'cases')


# ---------------------------------------------------------------------- found nets
def make_found(ms, mf, gt):
    cases = [
        ('found_a', fo.make_cfg(N=3, C=16, L=8, S=2, M=2, ns=1, nm=1, drpt=0.1), 4, 11),
        ('found_b', fo.make_cfg(N=4, C=16, L=8, S=2, M=2, ns=2, nm=2, drpt=0.1), 5, 12),
        ('found_c', fo.make_cfg(N=4, C=32, L=16, S=2, M=2, ns=3, nm=3, drpt=0.1), 3, 13),
        ('found_d', fo.make_cfg(N=4, C=16, L=8, S=2, M=2, ns=3, nm=2, drpt=0.1), 4, 14),
    ]
    for name, cfg, batch, gseed in cases:
        # a genotype produced by the reference's own genotype() on random arch params
        args = ref_args(cfg, 0.1)
        snet = ms.FusionNetwork(cfg.S, cfg.M, cfg.N, 2, args, criterion=None)
        load_arch(snet, synth.make_arch(cfg, gseed, 1.0))
        g = snet.genotype()
        gj = fo.genotype_to_jsonable(g)
        for mode in ('eval', 'train_nodrop'):
            args = ref_args(cfg, 1e-12 if mode == 'train_nodrop' else cfg.drpt)
            net = mf.Found_FusionNetwork(cfg.S, cfg.M, cfg.N, 2, args, None, g)
            shapes = fo.found_param_shapes(cfg, g)
            sd = net.state_dict()
            assert set(sd.keys()) == set(shapes.keys()), set(sd.keys()) ^ set(shapes.keys())
            net.load_state_dict(synth.make_params(cfg, gseed, shapes))
            if mode == 'eval':
                net.eval()
            else:
                net.train()
                set_attn_dropout(net, 0.0)
            xs = [x.requires_grad_(True) for x in synth.make_inputs(cfg, batch, gseed)]
            has_grads = not (mode == 'eval' and cfg.nm != 1)     # see run_hypernet_case
            with torch.set_grad_enabled(has_grads):
                feat = net(xs)
            out = {'meta': json.dumps(dict(name=name, cfg=dict(cfg), batch=batch, mode=mode,
                                           seed=gseed, genotype=gj, has_grads=has_grads)),
                   'feat': feat.detach().numpy()}
            if has_grads:
                w = torch.from_numpy(np.random.Generator(np.random.PCG64(gseed))
                                     .standard_normal(tuple(feat.shape)).astype(np.float32))
                (feat * w).sum().backward()
                for k, v in net.named_parameters():
                    # unused branches (inner step outside inner_concat and unreferenced)
                    # leave .grad None in the reference: recorded as zeros
                    out['grad:' + k] = (v.grad.detach().numpy() if v.grad is not None
                                        else np.zeros(tuple(v.shape), np.float32))
                for i, x in enumerate(xs):
                    out[f'grad:input.{i}'] = (x.grad.detach().numpy() if x.grad is not None
                                              else np.zeros(tuple(x.shape), np.float32))
            for k, v in net.state_dict().items():
                if fo.is_buffer(k):
                    out['buf:' + k] = v.detach().numpy()
            path = os.path.join(HERE, f'{name}_{mode}.npz')
            np.savez_compressed(path, **out)
            print('wrote', path, os.path.getsize(path))


# ---------------------------------------------------------------------- trajectory
class _RefSearchNet(torch.nn.Module):
    """fusion_net + central_classifier wired like Searchable_* (mmimdb_darts_searchable.py:
    76-83,113-114) minus backbones/reshape layers (out of the hot path)."""

    def __init__(self, ms, cfg, nout, args):
        super().__init__()
        self.fusion_net = ms.FusionNetwork(cfg.S, cfg.M, cfg.N, 2, args, criterion=None)
        self.central_classifier = torch.nn.Linear(cfg.M * cfg.C * cfg.L, nout)

    def forward(self, xs):
        return self.central_classifier(self.fusion_net(list(xs)))

    def arch_parameters(self):
        return self.fusion_net.arch_parameters()


def make_trajectory(ms, architect_mod):
    """3 iterations of: w-step on a train batch (Adam lr 1e-3 wd 1e-4,
    mmimdb_darts_searchable.py:28) then Architect.step on a dev batch (Adam lr 3e-4
    betas (0.5,0.999) wd 1e-3, :32-33; architect.py:21-29).  Dropout identity."""
    for name, cfg, nout, loss_kind in [
            ('traj_a', fo.make_cfg(N=3, C=16, L=8, S=2, M=2, ns=1, nm=1, drpt=0.1), 5, 'bce'),
            ('traj_b', fo.make_cfg(N=3, C=16, L=8, S=2, M=2, ns=2, nm=2, drpt=0.1), 6, 'ce')]:
        seed, batch, iters = 21, 6, 3
        args = ref_args(cfg, 1e-12)
        model = _RefSearchNet(ms, cfg, nout, args)
        model.fusion_net.load_state_dict(synth.make_params(cfg, seed))
        load_arch(model.fusion_net, synth.make_arch(cfg, seed, 1e-3))
        cw, cb = synth.make_classifier(cfg, nout, seed)
        model.central_classifier.weight.data.copy_(cw)
        model.central_classifier.bias.data.copy_(cb)
        model.train()
        set_attn_dropout(model, 0.0)
        crit = torch.nn.BCEWithLogitsLoss() if loss_kind == 'bce' else torch.nn.CrossEntropyLoss()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        aopt = torch.optim.Adam(model.arch_parameters(), lr=3e-4, betas=(0.5, 0.999),
                                weight_decay=1e-3)
        architect = architect_mod.Architect(model, args, crit, aopt)
        out = {'meta': json.dumps(dict(name=name, cfg=dict(cfg), batch=batch, num_outputs=nout,
                                       loss=loss_kind, seed=seed, iters=iters))}
        for it in range(iters):
            xs = synth.make_inputs(cfg, batch, seed + 10 * it)
            y = synth.make_labels(loss_kind, batch, nout, seed + 10 * it)
            opt.zero_grad()
            logits = model(xs)
            loss = crit(logits, y)
            loss.backward()
            opt.step()
            out[f'train_logits.{it}'] = logits.detach().numpy()
            xv = synth.make_inputs(cfg, batch, seed + 10 * it + 5)
            yv = synth.make_labels(loss_kind, batch, nout, seed + 10 * it + 5)
            architect.step(xv, yv, None)
            with torch.no_grad():
                out[f'dev_logits.{it}'] = model(xv).numpy()
        for i, a in enumerate(model.arch_parameters()):
            out[f'arch.{i}'] = a.detach().numpy()
        for k, v in model.state_dict().items():
            out['sd:' + k] = v.detach().numpy()
        out['genotype'] = json.dumps(fo.genotype_to_jsonable(model.fusion_net.genotype()))
        path = os.path.join(HERE, f'{name}.npz')
        np.savez_compressed(path, **out)
        print('wrote', path, os.path.getsize(path))


# ------------------------------------------------ reshape layers + scheduler ("next" rows)
def make_aux():
    import models.auxiliary.aux_models as aux_ref
    import models.auxiliary.scheduler as sc_ref
    out = {}
    cases = [('mm_vec', 'ReshapeInputLayer_MMIMDB', 64, 32, 16, (5, 64)),
             ('mm_map', 'ReshapeInputLayer_MMIMDB', 128, 32, 16, (3, 128, 10, 12)),
             ('nt_vec', 'ReshapeInputLayer', 256, 32, 8, (4, 256)),
             ('nt_vid', 'ReshapeInputLayer', 64, 16, 8, (3, 64, 6, 5, 5)),
             ('nt_ske', 'ReshapeInputLayer', 128, 16, 8, (4, 128, 4, 4))]
    meta = []
    for name, cls, c_in, C, L, shape in cases:
        for mode in ('eval', 'train_nodrop'):
            a = Args()
            a.drpt = 1e-12 if mode == 'train_nodrop' else 0.1
            layer = getattr(aux_ref, cls)(c_in, C, L, a)
            rng = np.random.Generator(np.random.PCG64(77))
            sd = {'conv.weight': (rng.uniform(-1, 1, (C, c_in, 1)) / np.sqrt(c_in)).astype(np.float32),
                  'conv.bias': (0.1 * rng.standard_normal(C)).astype(np.float32),
                  'bn.weight': (1 + 0.1 * rng.standard_normal(C)).astype(np.float32),
                  'bn.bias': (0.1 * rng.standard_normal(C)).astype(np.float32),
                  'bn.running_mean': (0.1 * rng.standard_normal(C)).astype(np.float32),
                  'bn.running_var': (1 + 0.2 * np.abs(rng.standard_normal(C))).astype(np.float32),
                  'bn.num_batches_tracked': np.zeros((), np.int64)}
            layer.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
            layer.train(mode != 'eval')
            x = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).requires_grad_(True)
            y = layer(x)
            w = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
            (y * w).sum().backward()
            key = f'{name}_{mode}'
            out[key + ':y'] = y.detach().numpy()
            out[key + ':dx'] = x.grad.numpy()
            out[key + ':dconv_w'] = layer.conv.weight.grad.numpy()
            out[key + ':dbn_w'] = layer.bn.weight.grad.numpy()
            out[key + ':dbn_b'] = layer.bn.bias.grad.numpy()
            out[key + ':rm'] = layer.bn.running_mean.numpy().copy()
            out[key + ':rv'] = layer.bn.running_var.numpy().copy()
            meta.append(dict(key=key, cls=cls, c_in=c_in, C=C, L=L, shape=list(shape), mode=mode))
    sched = sc_ref.LRCosineAnnealingScheduler(1e-3, 1e-6, 1, 2, 7.5)
    out['sched'] = np.array([sched.step() for _ in range(60)], dtype=np.float64)
    out['meta'] = json.dumps(meta)
    path = os.path.join(HERE, 'aux_layers.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    ms, ns, mf, architect_mod, gt = import_reference()
    make_hypernet(ms)
    make_genotypes(ms)
    make_found(ms, mf, gt)
    make_trajectory(ms, architect_mod)
    make_aux()


if __name__ == '__main__':
    main()
