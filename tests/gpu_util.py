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


def _np64(t):
    return np.asarray(t.detach().cpu().numpy() if torch.is_tensor(t) else t, dtype=np.float64)


OUTCOMES = []          # (label, outcome) of every match_step call of the session; tests/conftest.py prints them
_TABLE = None


def _allowed_flips(label):
    """tests/golden/match_step_table.json: label -> number of ReLU decisions a committed GPU run needed flipped
    (0: it matched the fp32 or the float64 evaluation as they stand).  A case may need at most that many + 1 here
    — one element within round-off of zero falling the other way from run to run (atomic summation order) is what
    the matcher exists for; a drift from 0 to many is a regression and fails.  Unknown labels are not capped."""
    global _TABLE
    if _TABLE is None:
        import json
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'match_step_table.json')
        try:
            with open(path) as f:
                _TABLE = json.load(f)
        except OSError:
            _TABLE = {}
    n = _TABLE.get(label)
    return None if n is None else int(n) + 1


_DET_TABLE = None


def _check_deterministic_kind(label, outcome):
    """Labels that start with 'det: ' come from BMNAS_DETERMINISTIC runs (tests/test_deterministic_gpu.py): there the
    result is bit-reproducible, so WHICH evaluation of the reference math it matches is reproducible too, and any
    change against tests/golden/match_step_table_det.json (label -> 'fp32' | 'fp64' | 'fp64+Nflips', written by
    tools/match_table.py from a GPU run) is a real change of the arithmetic — no '+1 flip' allowance as in the default
    (atomics-order) mode.  Unknown labels are recorded, not judged."""
    global _DET_TABLE
    if not label.startswith('det: '):
        return
    if _DET_TABLE is None:
        import json
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'match_step_table_det.json')
        try:
            with open(path) as f:
                _DET_TABLE = json.load(f)
        except OSError:
            _DET_TABLE = {}
    want = _DET_TABLE.get(label)
    if want is not None and want != outcome:
        raise AssertionError(f'[{label}] deterministic mode now matches the reference math as {outcome!r}; the committed '
                             f'table (tests/golden/match_step_table_det.json) recorded {want!r}: the arithmetic changed')


def _record(label, outcome):
    import json
    import os
    OUTCOMES.append((label, outcome))
    _check_deterministic_kind(label, outcome)
    root = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:                                   # gpurun merges gpurun_out/ back: the outcomes of a GPU run can be read later
        os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(root, 'gpurun_out', 'match_step_outcomes.jsonl'), 'a') as f:
            f.write(json.dumps({'label': label, 'outcome': outcome}) + '\n')
    except OSError:
        pass
    return outcome


def match_step(got, specs, evaluate, label='', near=2e-5, max_ambiguous=96):
    """Does the HIP result of a whole step equal the reference math?  `got`: name -> tensor; `specs`: name ->
    (rel, of_scale) tolerance; `evaluate(double, flips, near)` -> (name -> expected tensor, ambiguous ReLU inputs)
    runs the CPU oracle (oracle.fusion_oracle.relu_decisions).

    Every tensor must match ONE evaluation of the oracle at full tolerance:
      1. the fp32 op sequence (the reference's own arithmetic), or
      2. the same in float64, or
      3. float64 with an explicit set of ReLU decisions flipped, all of them on inputs within `near` of zero.
    (3) exists because a ReLU input within round-off of zero falls on either side depending on the summation
    order, and one such element moves whole gradient tensors by ~1e-2 of their scale (a sample's term enters or
    leaves a batch reduction): at 250 samples the two CPU evaluations already disagree with EACH OTHER that way
    (DESIGN.md section 4).  The flipped set is not guessed: every ambiguous element's effect on the result is
    measured by one oracle evaluation, the residual (got - float64) is decomposed over those effect vectors by
    least squares, and the decomposition is then VERIFIED by an exact evaluation under that assignment.
    Returns (and prints, when it is not plain fp32) which evaluation matched."""
    def failures(want):
        out = []
        for k, g in got.items():
            rel, of_scale = specs[k]
            try:
                (assert_close_of_scale if of_scale else assert_close_scaled)(k, g, want[k], rel)
            except AssertionError as e:
                out.append(str(e))
        return out

    want32, _ = evaluate(False, (), 0.0)
    f32 = failures(want32)
    if not f32:
        return _record(label, 'fp32')
    want64, amb = evaluate(True, (), near)
    f64 = failures(want64)
    if not f64:
        print(f'[{label}] matches the float64 evaluation ({len(f32)} tensors differ from the fp32 one: {f32[0][:120]})')
        return _record(label, 'fp64')
    if not amb or len(amb) > max_ambiguous:
        raise AssertionError(f'[{label}] {len(f64)} tensors match neither evaluation and {len(amb)} ReLU inputs are within '
                             f'{near:g} of zero:\n  fp32: ' + '\n  fp32: '.join(f32[:4]) + '\n  fp64: ' + '\n  fp64: '.join(f64[:4]))
    # residual and per-element effect vectors over the (small) parameter / arch tensors, each in units of its scale
    keys = [k for k in got if got[k].numel() <= (1 << 20)]
    scale = {k: max(float(np.abs(_np64(want64[k])).max()), 1e-30) for k in keys}
    vec = lambda d: np.concatenate([(_np64(d[k]).reshape(-1)) / scale[k] for k in keys])
    base = vec(want64)
    resid = vec(got) - base
    cols = []
    for site, idx, val in amb:
        w, _ = evaluate(True, [(site, idx)], 0.0)
        cols.append(vec(w) - base)
    coef = np.linalg.lstsq(np.stack(cols, 1), resid, rcond=None)[0]
    flips = [(s_, i_) for (s_, i_, _), c in zip(amb, coef) if c > 0.5]
    wantf, _ = evaluate(True, flips, 0.0)
    ff = failures(wantf)
    desc = ', '.join(f'site {s_} elem {i_} (input {v:+.1e}, weight {c:.2f})' for (s_, i_, v), c in zip(amb, coef) if c > 0.5)
    if ff:
        raise AssertionError(f'[{label}] no assignment of the {len(amb)} ambiguous ReLU decisions reproduces the result '
                             f'(tried flipping: {desc or "none"}; weights {np.round(coef, 2).tolist()}):\n  ' + '\n  '.join(ff[:4])
                             + '\n  against plain float64: ' + f64[0])
    print(f'[{label}] matches float64 with {len(flips)} of {len(amb)} ambiguous ReLU decisions on the other side: {desc}')
    cap = _allowed_flips(label)
    _record(label, f'fp64+{len(flips)}flips')
    if cap is not None and len(flips) > cap:
        raise AssertionError(f'[{label}] needs {len(flips)} ReLU decisions flipped to match the reference math; the '
                             f'committed table (tests/golden/match_step_table.json) allows {cap}: {desc}')
    return f'fp64+{len(flips)}flips'


def compare_search_step(cfg, batch, nout, loss_kind, net, cls, input_grads, logits, loss, masks=None, seed=31,
                        label='', attn_drop=None, max_ambiguous=96):
    """Every tensor of one search step (logits, loss, every weight / arch / input gradient) against the oracle,
    through match_step; BatchNorm running statistics against the fp32 oracle.  masks: the dropout multipliers of
    the step's live sites in issue order (None: dropout is an identity, attn_drop must then be 0)."""
    import contextlib
    from oracle import fusion_oracle as fo

    def evaluate(double, flips, near):
        f = (lambda t: t.double() if t.is_floating_point() else t) if double else (lambda t: t)
        p = {k: f(v) for k, v in synth.make_params(cfg, seed).items()}
        cw, cb = synth.make_classifier(cfg, nout, seed)
        inj = fo.injected_masks(masks) if masks is not None else contextlib.nullcontext()
        with inj, fo.relu_decisions(near, flips) as rd:
            lg, ls, grads = fo.search_step([f(x) for x in synth.make_inputs(cfg, batch, seed)],
                                           synth.make_labels(loss_kind, batch, nout, seed),
                                           [f(a) for a in synth.make_arch(cfg, seed)], p, f(cw), f(cb), cfg, loss_kind,
                                           training=True, **({} if attn_drop is None else {'attn_drop': attn_drop}))
        if masks is not None:
            assert inj.used == len(masks)
        want = {'logits': lg, 'loss': ls}
        for k, v in grads.items():
            want['grad:' + k] = v
        want['_params'] = p
        return want, rd.ambiguous

    got, specs = {'logits': logits, 'loss': loss}, {'logits': (1e-4, True), 'loss': (1e-4, True)}
    for k, v in net.named_parameters():
        if k.endswith('conv.bias'):
            assert float(v.grad.abs().max()) < 1e-4, k       # mathematically zero (BN removes the mean)
        else:
            got['grad:' + k] = v.grad
    for i, a in enumerate(net.arch_parameters()):
        got[f'grad:arch.{i}'] = a.grad
    for i, g in enumerate(input_grads):
        got[f'grad:input.{i}'] = g
    for k in ('weight', 'bias'):
        got['grad:central_classifier.' + k] = getattr(cls, k).grad
    for k in got:
        specs.setdefault(k, (2e-4, False))
    how = match_step(got, specs, evaluate, label, max_ambiguous=max_ambiguous)
    p32 = evaluate(False, (), 0.0)[0]['_params']
    for k, v in net.state_dict().items():
        if fo.is_buffer(k):
            assert_close_scaled('buf:' + k, v.float(), p32[k].float())
    return how
