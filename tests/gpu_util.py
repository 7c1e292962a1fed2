"""Helpers for the -m gpu parity tests: build the product modules on cuda:0 from the same
deterministic synthetic tensors (oracle/synth.py) the golden generator used."""
import numpy as np
import torch
import torch.nn as nn

from oracle import fusion_oracle as fo
from oracle import synth


class Args:
    def __init__(self, cfg, drpt=None):
        self.C, self.L = cfg.C, cfg.L
        self.drpt = cfg.drpt if drpt is None else drpt
        self.num_input_nodes = cfg.N
        self.num_keep_edges = 2
        self.node_steps, self.node_multiplier = cfg.ns, cfg.nm
        self.steps, self.multiplier = cfg.S, cfg.M
        self.parallel = False
        self.weight_decay = 1e-4


def dev():
    return torch.device('cuda:0')


def set_mode(module, mode):
    """'eval' -> eval(); 'train_nodrop' -> train() with every dropout an identity (the golden
    vectors' drpt=1e-12 / ScaledDotAttn.dropout.p=0 recipe); 'train' -> train()."""
    if mode == 'eval':
        module.eval()
        return
    module.train()
    if mode == 'train_nodrop':
        for m in module.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0


def build_search_net(cfg, seed, mode, arch_scale=0.5):
    from models.search.darts.model_search import FusionNetwork
    net = FusionNetwork(cfg.S, cfg.M, cfg.N, 2, Args(cfg), criterion=None)
    net.load_state_dict(synth.make_params(cfg, seed))
    for dst, src in zip(net.arch_parameters(), synth.make_arch(cfg, seed, arch_scale)):
        dst.data.copy_(src)
    net.to(dev())
    set_mode(net, mode)
    return net


def build_found_net(cfg, genotype, seed, mode):
    from models.search.darts.genotypes import Genotype, StepGenotype
    from models.search.darts.model import Found_FusionNetwork
    g = Genotype(edges=[tuple(e) for e in genotype.edges],
                 steps=[StepGenotype(inner_edges=[tuple(e) for e in s.inner_edges],
                                     inner_steps=list(s.inner_steps), inner_concat=list(s.inner_concat))
                        for s in genotype.steps],
                 concat=list(genotype.concat))
    net = Found_FusionNetwork(cfg.S, cfg.M, cfg.N, 2, Args(cfg), None, g)
    net.load_state_dict(synth.make_params(cfg, seed, fo.found_param_shapes(cfg, genotype)))
    net.to(dev())
    set_mode(net, mode)
    return net


def scale_tol(want, rel=1e-4, floor=1e-6):
    """absolute tolerance = rel * (largest magnitude of the expected tensor)"""
    w = np.asarray(want, dtype=np.float64)
    return max(floor, rel * float(np.abs(w).max()) if w.size else floor)


def assert_close_scaled(name, got, want, rel=1e-4, floor=1e-6):
    """|got - want| <= rel * (|want| + max|want|): fp32 parity at 1e-4 of the tensor's scale."""
    got = np.asarray(got.detach().cpu().numpy() if torch.is_tensor(got) else got, dtype=np.float64)
    want = np.asarray(want.detach().cpu().numpy() if torch.is_tensor(want) else want, dtype=np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    if want.size == 0:
        return
    tol = rel * np.abs(want) + scale_tol(want, rel, floor)
    err = np.abs(got - want)
    if not np.isfinite(got).all() or not (err <= tol).all():
        i = int(np.argmax(err - tol))
        raise AssertionError(f'{name}: max|err|={np.nanmax(err):.3e} (scale {np.abs(want).max():.3e}) at flat {i}: '
                             f'got {got.reshape(-1)[i]!r} want {want.reshape(-1)[i]!r}')


def assert_summary_scaled(name, got, want, rel=2e-4):
    """Compare a tensor with a stored [sum, l2, first-k elements] float64 summary (the fixture form
    of production-size tensors): l2 within rel, the sum within rel * l2 * sqrt(n) (cancellation),
    the leading elements within rel * (|want| + l2 / sqrt(n))."""
    g = got.detach().double().reshape(-1).cpu()
    k = len(want) - 2
    n = max(g.numel(), 1)
    head = g[:k].numpy()
    if head.size < k:
        head = np.concatenate([head, np.zeros(k - head.size)])
    l2 = max(abs(float(want[1])), 1e-12)
    assert np.isfinite(g.numpy()).all(), name
    assert abs(float(g.norm()) - want[1]) <= rel * l2 + 1e-7, (name, 'l2', float(g.norm()), want[1])
    assert abs(float(g.sum()) - want[0]) <= rel * l2 * np.sqrt(n) + 1e-6, (name, 'sum', float(g.sum()), want[0])
    assert np.all(np.abs(head - want[2:]) <= rel * (np.abs(want[2:]) + l2 / np.sqrt(n)) + 1e-7), \
        (name, 'head', head, want[2:])


def assert_close_of_scale(name, got, want, rel=1e-4, floor=1e-6):
    """|got - want| <= rel * max|want| for every element: "within 1e-4 of scale", the bound BASELINE.json's
    north_star states for logits (no per-element relative term on top)."""
    got = np.asarray(got.detach().cpu().numpy() if torch.is_tensor(got) else got, dtype=np.float64)
    want = np.asarray(want.detach().cpu().numpy() if torch.is_tensor(want) else want, dtype=np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    if want.size == 0:
        return
    tol = scale_tol(want, rel, floor)
    err = np.abs(got - want)
    if not np.isfinite(got).all() or not (err <= tol).all():
        i = int(np.argmax(err))
        raise AssertionError(f'{name}: max|err|={np.nanmax(err):.3e} > {rel:g} of scale {np.abs(want).max():.3e} at '
                             f'flat {i}: got {got.reshape(-1)[i]!r} want {want.reshape(-1)[i]!r}')


class EitherLog:
    """Per-tensor comparison against the fp32 oracle, with the float64 evaluation of the same oracle as a
    second legitimate answer (a ReLU decision on a pre-activation within round-off of zero may fall either way and
    moves whole gradient tensors by ~1e-2 of their scale: DESIGN.md section 4, tools/diag_ntu250.py).  Unlike a
    bare "either" it RECORDS which evaluation matched and fails when more than `max_rescued` tensors needed the
    second one — a real regression in one branch cannot hide behind the rescue."""

    def __init__(self, max_rescued):
        self.max_rescued = max_rescued
        self.first, self.rescued = [], []

    def check(self, name, got, want_a, want_b, rel=1e-4, of_scale=False):
        """want_b: the second evaluation, or a callable that produces it (only computed when needed)."""
        close = assert_close_of_scale if of_scale else assert_close_scaled
        try:
            close(name, got, want_a, rel)
            self.first.append(name)
            return 'fp32'
        except AssertionError as first:
            try:
                close(name, got, want_b() if callable(want_b) else want_b, rel)
            except AssertionError as second:
                raise AssertionError(f'{first}\n   and against the float64 evaluation: {second}') from None
            self.rescued.append(name)
            return 'fp64'

    def finish(self):
        if len(self.rescued) > self.max_rescued:
            raise AssertionError(f'{len(self.rescued)} tensors matched only the float64 evaluation (allowed: '
                                 f'{self.max_rescued}): {self.rescued}')
        return self.rescued
