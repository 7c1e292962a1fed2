#!/usr/bin/env python3
"""bench.py — search-steps/s of the BM-NAS fusion hypernet (fwd + bwd) on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = forward + loss + backward of FusionNetwork + central_classifier on one batch of
synthetic MM-IMDB-shaped features (6 x (128, 192, 16) fp32 = relu(N(0,1)), 23-way multi-hot
labels, BCEWithLogits), gradients for every weight, alpha/beta/gamma and the 6 inputs, train
mode, dropout on — BASELINE.json configs[1] (batch 128 on one MI355X).  Inputs are resident
in HBM before the timed region.  With N > 1 every rank processes its own 128-sample batch
(weak scaling, configs[2] = 1024 over 8 GPUs) and the timed step also averages the weight
and architecture gradients with one flat RCCL all-reduce each; value = N*K / max-rank time.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel: launch-inclusive device
durations from a rocprofv3 kernel trace of the hipGraph replay, taken by a child process of this script
after the timed regions), "roofline_kernels" (every kernel of the step), "k1" (the cell-level mixed sums
under SURVEY.md 8(d)'s literal bytes), "full_search_step" (w-step, alpha-step and the dev phase's metric
forward as replays), "cpu_baseline" (the CPU oracle — a port of the reference — timed on this host's
cores); under N > 1 "rccl" (what the collective saw) and "dp_cost" (compute-only step, exposed all-reduce,
the efficiency bound it implies).
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'bm-nas_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 MFMA

CONFIGS = {
    # N, C, L, steps, multiplier, node_steps, node_multiplier, drpt, num_outputs, loss
    'mmimdb': dict(N=6, C=192, L=16, S=2, M=2, ns=1, nm=1, drpt=0.1, nout=23, loss='bce'),
    'ntu': dict(N=8, C=128, L=8, S=2, M=2, ns=2, nm=2, drpt=0.2, nout=60, loss='ce'),
    'ego': dict(N=8, C=128, L=8, S=2, M=2, ns=3, nm=3, drpt=0.0, nout=83, loss='ce'),
}


class Args:
    pass


def make_args(c):
    a = Args()
    a.C, a.L, a.drpt = c['C'], c['L'], c['drpt']
    a.num_input_nodes, a.num_keep_edges = c['N'], 2
    a.node_steps, a.node_multiplier = c['ns'], c['nm']
    a.steps, a.multiplier = c['S'], c['M']
    a.parallel = False
    a.weight_decay = 1e-4
    return a


# C_in of the backbone features each reshape layer receives (mmimdb_darts_searchable.py:86,
# ntu_darts_searchable.py:104, ego_darts_searchable.py:104)
C_INS = {'mmimdb': [512, 512, 512, 512, 64, 128],
         'ntu': [512, 1024, 2048, 2048, 128, 256, 1024, 512],
         'ego': [512, 1024, 2048, 2048, 512, 1024, 2048, 2048]}


class HyperNet(torch.nn.Module):
    """[reshape_layers ->] fusion_net -> central_classifier, wired like Searchable_* minus the
    backbones.  tier 'F' (headline): the inputs ARE the (b, C, L) features.  tier 'R': the inputs
    are pooled backbone features (b, C_in_i, L) and go through the Conv1d(k=1) + BatchNorm + ReLU +
    Dropout tail of the reshape layers first (aux_models.py:61-76, 100-115; the adaptive max pool in
    front of it is the backbone side of the boundary)."""

    def __init__(self, c, tier='F', cname='mmimdb'):
        super().__init__()
        from models.search.darts.model_search import FusionNetwork
        args = make_args(c)
        self.tier = tier
        if tier == 'R':
            import models.auxiliary.aux_models as aux
            cls = aux.ReshapeInputLayer_MMIMDB if cname == 'mmimdb' else aux.ReshapeInputLayer
            self.reshape_layers = torch.nn.ModuleList(cls(ci, c['C'], c['L'], args) for ci in C_INS[cname])
        self.fusion_net = FusionNetwork(c['S'], c['M'], c['N'], 2, args, criterion=None)
        from bmnas import nn as bnn
        self.central_classifier = bnn.Linear(c['M'] * c['C'] * c['L'], c['nout'])

    def forward(self, xs):
        if self.tier == 'R':
            import models.auxiliary.aux_models as aux
            xs = aux.reshape_tails(list(self.reshape_layers), xs)
        # = central_classifier(fusion_net(xs)), what Searchable_*.forward does (HyperNetBase.fuse)
        return self.fusion_net.forward_classified(xs, self.central_classifier)

    def arch_parameters(self):
        return self.fusion_net.arch_parameters()


def synth_batch(c, batch, device, seed, tier='F', cname='mmimdb'):
    g = torch.Generator(device='cpu').manual_seed(seed)
    widths = [c['C']] * c['N'] if tier == 'F' else C_INS[cname]
    xs = [torch.relu(torch.randn(batch, ci, c['L'], generator=g)).to(device).requires_grad_(True)
          for ci in widths]
    if c['loss'] == 'bce':
        y = (torch.rand(batch, c['nout'], generator=g) < 0.2).float().to(device)
    else:
        y = torch.randint(0, c['nout'], (batch,), generator=g).to(device)
    return xs, y


class _K1Units(float):
    """algorithmic bytes of a K1 launch that also remember the n_in of its cell-level sum (roofline 'k1' object)"""
    n_in = None


def _k1(units, n_in):
    u = _K1Units(units)
    u.n_in = n_in
    return u


def algo_table(C, L):
    """wrapper name -> (bound, algorithmic units of one launch); SURVEY.md section 8(d)."""
    T = lambda t: t.numel() * 4
    return {
        'mixsum_fwd': lambda xs, w, ws, out, *_: ('hbm', _k1((len(xs) + 1) * T(out), len(xs))),
        # reads: g (+ g2) + the n_in inputs + the destinations that accumulate; writes: the destinations
        'mixsum_bwd': lambda xs, dxs, w, ws, g, dw, acc, sh=1, st=0, g2=None:
            ('hbm', _k1((1 + (g2 is not None) + len(xs) + bin(acc).count('1')
                         + sum(d is not None for d in dxs)) * T(g), len(xs))),
        'mixsum_pair_fwd': lambda xs, w, ws, w2, ws2, out, *_: ('hbm', _k1((len(xs) + 2) * T(out), len(xs))),
        # the first pair sum with the cell prologue's jobs in the same launch: K1's bytes + the folds
        'cell_prologue_pair': lambda al, ol, Ws, We, M, Cc, step, scrub, xs, a, bt, h, z:
            ('hbm', _k1((len(xs) + 2) * T(h) + sum(T(w) + T(e) for w, e in zip(Ws, We))
                        + (0 if scrub is None else T(scrub)), len(xs))),
        'mixsum_pair_bwd': lambda xs, dxs, w, ws, w2, ws2, h, gh, gz, dw, dw2, acc, sh=1, st=0, gz2=None:
            ('hbm', _k1((2 + (gh is not None) + (gz2 is not None) + len(xs) + bin(acc).count('1')
                         + sum(d is not None for d in dxs)) * T(gz), len(xs))),
        # the lazy-LayerNorm forms (csrc/lazyln.hip).  K1 forward with the previous node normalised in the fetch: the
        # n_in plain inputs + pre read, n + h + z written (+ the LayerNorm affine)
        'mixsum_pair_fwd_lazy': lambda xs, w, ws, w2, ws2, lz, nout, sums, out, out2, b, Cc, L_:
            ('hbm', _k1((len(xs) + 1 + 3) * T(out) + 2 * 4 * Cc * L_, len(xs) + 1)),
        # reads gz (+ gz2) (+ gh) + h + the n_in inputs + pre of the lazy ones + accumulating destinations; writes the
        # destinations (+ G)
        'mixsum_pair_bwd_lazy': lambda xs, dxs, w, ws, w2, ws2, h, gh, gz, dw, dw2, acc, lzs, lnp, strides, b, Cc, L_,
                                       sh=1, st=0, gz2=None, g_full=None:
            ('hbm', _k1((2 + (gh is not None) + (gz2 is not None) + len(xs) + len(lzs) + bin(acc).count('1')
                         + sum(d is not None for d in dxs) + (g_full is not None)) * T(gz), len(xs))),
        'mixsum_pair_bwd_x': lambda xs, dxs, w, ws, w2, ws2, h, gh, gz, dw, dw2, acc, g_more, w_more, sh=1, st=0,
                                    gz2=None:
            ('hbm', _k1((2 + (gh is not None) + (gz2 is not None) + len(xs) + len(g_more) + bin(acc).count('1')
                         + sum(d is not None for d in dxs)) * T(gz), len(xs))),
        # K2 + residual, un-normalised: U (3T) + z + p1 + resid read, pre written (x is y: one read) + the affine
        'node_mix_pre_fwd': lambda x, y, p1, U, ch, gm, resid, w, b_, pre, rec, prm, b, Cc, L_, dg, df, fin=None:
            ('hbm', T(U) + (3 + (x.data_ptr() != y.data_ptr()) + 1) * T(pre) + 2 * T(w)),
        # LayerNorm backward from partial sums + K2 backward: gy, pre, U, x, p1 read; g_in, dresid, dx, dV written
        'node_mix_lnp_bwd': lambda gy, pre, w, st, l0, l1, g_in, dres, racc, x, y, p1, U, ch, gm, dgm, dx, dy, m, dV,
                                   *_:
            ('hbm', 2 * T(U) + (4 + (g_in is not None) + (dres is not None) + bool(racc) + (dx is not None)
                                + bin(m).count('1')) * T(x) + T(w)),
        'head_fwd_lazy': lambda srcs, sums, lq, lz, lw, lb, W, bias, hb, st, b, Cc, L_, O, *_:
            ('mfma', 2.0 * b * O * len(srcs) * Cc * L_),
        'head_bwd_lazy': lambda lzs, lnp, ds, m, lw, lb, W, hb, st, mode, g, gs, lab, loss, part, b, Cc, L_, O, *_:
            ('mfma', 4.0 * b * O * len(lzs) * Cc * L_),
        'cat_ln_fwd': lambda srcs, resid, w, b_, out, *_:
            ('hbm', (len(srcs) + (1 if resid is not None else 0)) * T(srcs[0]) + T(out) + 2 * T(w)),
        'cat_ln_bwd': lambda g, srcs, resid, w, *_:
            ('hbm', T(g) + (2 * len(srcs) + (2 if resid is not None else 0)) * T(srcs[0]) + 2 * T(w)),
        'ln_affine_bwd': lambda g, gs, srcs, resid, *_:
            ('hbm', T(g) + (len(srcs) + (1 if resid is not None else 0)) * T(srcs[0])),
        'ln_affine_bwd_multi': lambda probs, b, L_: ('hbm', sum(
            T(p['g']) + (len(p['srcs']) + (1 if p['resid'] is not None else 0)) * T(p['srcs'][0]) for p in probs)),
        'backward_epilogue': lambda probs, b, L_, *_: ('hbm', sum(
            T(p['g']) + (len(p['srcs']) + (1 if p['resid'] is not None else 0)) * T(p['srcs'][0]) for p in probs)),
        'sdpa_ln_fwd': lambda x, y, w, b_, out, *_:
            ('hbm', (3 if x.data_ptr() == y.data_ptr() else 4) * T(x) + 2 * T(w)),
        'sdpa_ln_bwd': lambda g, gs, x, y, w, xhat, st, dx, dy, *_:
            ('hbm', (4 if dy is None else 6) * T(x) + T(w)),
        'conv1x1_fwd': lambda srcs, Cs, W, ldw, bias, U, part, b, L_, M, *_:
            ('mfma', 2.0 * M * len(srcs) * Cs * b * L_),
        'conv1x1_bwd_data': lambda dU, W, ldw, ds, Cs, m, b, L_, M, *_:
            ('mfma', 2.0 * M * len(ds) * Cs * b * L_),
        'conv1x1_fwd_sdpa': lambda srcs, Cs, W, ldw, bias, U, part, b, L_, M, *_:
            ('mfma', 2.0 * M * len(srcs) * Cs * b * L_ + 4.0 * b * L_ * L_ * Cs),
        'conv1x1_bwd_all_sdpa': lambda dU, W, ldw, ds, Cs, m, b, L_, M, *_:
            ('mfma', 4.0 * M * len(ds) * Cs * b * L_ + 12.0 * b * L_ * L_ * Cs),
        'conv1x1_bwd_all': lambda dU, W, ldw, ds, Cs, m, b, L_, M, *_:
            ('mfma', 4.0 * M * len(ds) * Cs * b * L_),
        'conv1x1_bwd_weight': lambda dU, srcs, Cs, dW, ldw, db, dup, b, L_, M, *_:
            ('mfma', 2.0 * M * len(srcs) * Cs * b * L_),
        'node_mix_fwd': lambda x, y, p1, U, ch, gm, out, b, Cc, L_, dg, df, fin=None, nxt=None:
            ('hbm', T(U) + 3 * T(out) + (0 if nxt is None else (len(nxt[0]) + 1) * T(out))),
        'node_mix_ln_fwd': lambda x, y, p1, U, ch, gm, resid, w, b_, pre, out, *_:
            ('hbm', T(U) + 5 * T(out) + 2 * T(w)),
        # the mix as producer of out_conv's last operand: the GEMM's product (the mix bytes ride along)
        'node_mix_conv_fwd': lambda x, y, p1, U, ch, gm, out, dg, df, fin, srcs, W, ldw, bias, V, st, sh, b, Cc, L_:
            ('mfma', 2.0 * Cc * (len(srcs) + 1) * Cc * b * L_),
        'node_mix_bwd': lambda g, x, y, p1, U, ch, gm, dgm, dx, dy, m, dV, bg, b, Cc, L_, dg, df, sh=1, st=0, nxt=None:
            ('hbm', 2 * T(U) + 4 * T(x) + (0 if nxt is None else (2 * len(nxt[0]) + 4) * T(x))),
        # K6 backward + K2 backward: gy, pre, x, p1, U read; g_in, dresid, dx, dV written (rmw where accumulated)
        'node_mix_ln_bwd': lambda g, pre, w, st, g_in, dres, racc, x, y, p1, U, ch, gm, dgm, dx, dy, m, dV, *_:
            ('hbm', 2 * T(U) + (4 + (g_in is not None) + (dres is not None) + bool(racc) + (dx is not None)
                                + bin(m).count('1')) * T(x) + T(w)),
        'bn_relu_fwd': lambda U, *_: ('hbm', 2 * T(U)),
        'bn_relu_bwd': lambda g, U, *_: ('hbm', 3 * T(U)),
        'bn_bwd_apply': lambda dV, U, *_: ('hbm', 3 * T(U)),
        'bn_relu_ln_fwd': lambda U, ch, resid, w, *_: ('hbm', 4 * T(U) + 2 * T(w)),
        'bn_relu_ln_bwd': lambda g, o, resid, w, st, U, *_: ('hbm', 6 * T(U) + T(w)),
        'fold_weight': lambda W, We, *_: ('hbm', T(W) + T(We)),
        # K7 + classifier: the ALGORITHMIC classifier product 2 b M C L out; the kernel executes
        # three accumulator sets (logits, A, B), i.e. 3x that, which is its own business
        'head_fwd': lambda srcs, sums, lw, lb, W, bias, hb, st, b, Cc, L_, O: ('mfma', 2.0 * b * O * len(srcs) * Cc * L_),
        'head_bwd': lambda srcs, sums, ds, m, lw, lb, W, hb, st, mode, g, gs, lab, loss, part, b, Cc, L_, O, *_:
            ('mfma', 4.0 * b * O * len(srcs) * Cc * L_),
        # the N reshape layers as grouped launches (row f1): sum over the layers of 2 M C_in b L (fwd), twice that bwd
        'conv1x1_fwd_group': lambda srcs, Ws, bs, Us, st, sh, b, L_, M: ('mfma', sum(2.0 * M * x.shape[1] * b * L_ for x in srcs)),
        'conv1x1_bwd_group': lambda dVs, Ws, srcs, dsrcs, dWs, dbs, bU, bc, bg, tr, b, L_, M:
            ('mfma', sum((2.0 + 2.0 * (d is not None)) * M * x.shape[1] * b * L_ for x, d in zip(srcs, dsrcs))),
        'bn_relu_fwd_group': lambda Us, chans, outs, fins, drops, b, M, L_: ('hbm', sum(2 * T(U) for U in Us)),
        'bn_relu_bwd_group': lambda gs, Us, chans, dVs, bgs, drops, b, M, L_: ('hbm', sum(3 * T(U) for U in Us)),
        'linear_fwd': lambda feat, W, bias, out, b, O, Kd, *_, **__: ('mfma', 2.0 * b * O * Kd),
        'linear_bwd': lambda g, gs, feat, W, df, dW, db, b, O, Kd: ('mfma', 4.0 * b * O * Kd),
    }


def full_search_step(model, crit, params, arch, xs, y, c, a, world, device, log, pairs=100):
    """SURVEY.md section 8(d) secondary figure: the search loop's two optimisation steps with
    everything in them — weight phase (fwd + criterion + bwd + [RCCL all-reduce] + Adam on w,
    train_searchable/mmimdb.py:85-101) and architecture phase (the same for alpha/beta/gamma,
    architect.py:21-29) — each one hipGraph replay (bmnas.graph.GraphedTrainStep,
    bmnas.optim.Adam).  Not the headline metric; reported next to it."""
    from bmnas import dist as bdist
    from bmnas.graph import GraphedTrainStep
    from bmnas.optim import Adam
    w_opt = Adam(params, lr=1e-3, weight_decay=1e-4)
    a_opt = Adam(arch, lr=3e-4, betas=(0.5, 0.999), weight_decay=1e-3)
    selftest = bool(getattr(a, 'dp_selftest', False)) and world == 1
    dp_phase = None
    if world > 1 or selftest:
        # The TRAINERS' buckets, phase by phase: the weight step reduces its optimizer's tensors only
        # (no alpha / beta / gamma), the architecture step the 42-94 architecture floats alone.  Each phase's step is
        # timed first WITHOUT any collective (optimizers that nobody attached a reducer to: compute only — the replicas
        # drift apart for these few steps, which no later figure depends on), then with the step the trainers run.
        # One GPU (--dp-selftest): the same through a world-size-1 communicator — the collective is in the graph, it
        # just has nobody to talk to.
        try:
            dp_phase = {}
            w0 = Adam(params, lr=1e-3, weight_decay=1e-4)
            a0 = Adam(arch, lr=3e-4, betas=(0.5, 0.999), weight_decay=1e-3)
            for name, opt0 in (('w_step', w0), ('alpha_step', a0)):
                g0 = GraphedTrainStep(model, crit, opt0, xs, y)
                for _ in range(200):
                    g0(xs, y)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(pairs):
                    g0(xs, y)
                torch.cuda.synchronize()
                dp_phase[name] = {'compute_only_ms': round((time.perf_counter() - t0) / pairs * 1e3, 4)}
                del g0
            del w0, a0
        except Exception as e:                                   # noqa: BLE001
            dp_phase = {'error': f'{type(e).__name__}: {e}'[:200]}
        rw, _ = bdist.attach(w_opt, selftest=selftest)
        ra, _ = bdist.attach(a_opt, selftest=selftest)
    xv, yv = synth_batch(c, a.batch, device, 1000 + (torch.distributed.get_rank() if world > 1 else 0), a.tier,
                         a.config)
    gw = GraphedTrainStep(model, crit, w_opt, xs, y)
    ga = GraphedTrainStep(model, crit, a_opt, xv, yv)

    def timed(fn, n):
        # (the captures above left the chip idle for a second: a fixed number of untimed calls first, see measure())
        for _ in range(300):
            fn()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            bdist.all_reduce(t, torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt / n * 1e3

    def pair():
        for g in w_opt.param_groups:
            g['lr'] *= 0.999                      # a per-batch schedule, as the cosine rule applies
        gw(xs, y)
        ga(xv, yv)

    ms_pair = timed(pair, pairs)
    ms_w = timed(lambda: gw(xs, y), pairs)
    ms_a = timed(lambda: ga(xv, yv), pairs)
    if dp_phase is not None and 'error' not in dp_phase:
        for name, ms, red in (('w_step', ms_w, rw), ('alpha_step', ms_a, ra)):
            d = dp_phase[name]
            d['with_allreduce_ms'] = round(ms, 4)
            d['exposed_us_per_step'] = round(max(0.0, ms - d['compute_only_ms']) * 1e3, 1)
            d['bucket_bytes'] = int(red.flat.numel() * 4) if red.flat is not None else None
            d['plan'] = red.plan()
            try:                                                  # the bucket's all-reduce alone, on this communicator
                for _ in range(20):
                    red.reduce_bucket()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(100):
                    red.reduce_bucket()
                torch.cuda.synchronize()
                d['allreduce_alone_us'] = round((time.perf_counter() - t0) / 100 * 1e6, 1)
            except Exception as e:                                # noqa: BLE001
                d['allreduce_alone_error'] = f'{type(e).__name__}: {e}'[:120]
        dp_phase['in_graph'] = bool(gw.native and ga.native)
        dp_phase['note'] = ('exposed = (step with its collective) - (the same step with no reducer attached), both '
                            'hipGraph replays timed in this run; the all-reduce alone is host-issued back to back')
    # what the batch copy into the graph's static tensors costs: the same replay handed its own static tensors
    # (GraphedTrainStep.static_batch(): a producer that writes the batch there skips the copy launch)
    sx, sy = gw.static_batch()
    ms_w_nocopy = timed(lambda: gw(sx, sy), pairs)
    # host side of one replayed step: the time Python needs to ISSUE a call (no synchronisation inside the loop); a loop
    # is host-bound once this exceeds the step's device time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        gw(xs, y)
    host_us = (time.perf_counter() - t0) / 50 * 1e6
    torch.cuda.synchronize()
    # the dev phase's metric pass (a gradient-free forward after every architect.step, train mode): one replay
    ms_f = None
    from bmnas.graph import GraphedForward
    gf = GraphedForward.try_build(model, crit, xv, yv)
    if gf:
        ms_f = round(timed(lambda: gf(xv, yv), pairs), 4)
    # round 5: the architecture step and that metric forward as ONE replay over one copy of the batch (what the trainer
    # loop runs in the dev phase, models/search/darts/architect.py) -> the loop's cost per (train batch, dev batch)
    ms_am = ms_loop = None
    if world == 1 and not selftest:
        try:
            gam = GraphedTrainStep(model, crit, a_opt, xv, yv, metric_forward=True)
            ms_am = round(timed(lambda: gam(xv, yv), pairs), 4)

            def loop_pair():
                for g in w_opt.param_groups:
                    g['lr'] *= 0.999
                gw(xs, y)
                gam(xv, yv)
            ms_loop = round(timed(loop_pair, pairs), 4)
        except Exception as e:                                   # noqa: BLE001
            log(f'merged alpha + metric step not captured: {type(e).__name__}: {e}')
    # round 6: the trainer's steps_per_replay = 4 — four weight steps (four resident batches, per-step learning rates) as
    # ONE replay; the phases of the search loop are sequential (a whole train phase, then a whole dev phase), so the
    # cost per (train batch, dev batch) is (w4 + alpha4-with-metric-forward) / 4
    ms_w4 = ms_am4 = ms_loop4 = None
    if world == 1 and not selftest:
        try:
            gw4 = GraphedTrainStep(model, crit, w_opt, xs, y, k=4)
            more = [synth_batch(c, a.batch, device, 2000 + i, a.tier, a.config) for i in range(3)]
            b4 = [([t.detach() for t in xs], y)] + [([t.detach() for t in xi], yi) for xi, yi in more]

            def w4():
                for j, (xj, yj) in enumerate(b4):
                    for g in w_opt.param_groups:
                        g['lr'] *= 0.999
                    gw4.stage(j, xj, yj)
                gw4.replay_staged()
            ms_w4 = round(timed(w4, max(1, pairs // 4)) / 4, 4)
            gam4 = GraphedTrainStep(model, crit, a_opt, xv, yv, metric_forward=True, k=4)
            moreb = [synth_batch(c, a.batch, device, 3000 + i, a.tier, a.config) for i in range(3)]
            bv4 = [([t.detach() for t in xv], yv)] + [([t.detach() for t in xi], yi) for xi, yi in moreb]

            def am4():
                for j, (xj, yj) in enumerate(bv4):
                    gam4.stage(j, xj, yj)
                gam4.replay_staged()
            ms_am4 = round(timed(am4, max(1, pairs // 4)) / 4, 4)
            ms_loop4 = round(ms_w4 + ms_am4, 4)
        except Exception as e:                                   # noqa: BLE001
            log(f'k = 4 weight step not captured: {type(e).__name__}: {e}')
    log(f'full search step: {ms_pair:.4f} ms per (w-step + alpha-step) pair')
    return {'ms_per_pair': round(ms_pair, 4), 'w_step_ms': round(ms_w, 4), 'alpha_step_ms': round(ms_a, 4),
            'metric_forward_ms': ms_f, 'alpha_step_with_metric_forward_ms': ms_am,
            'loop_ms_per_train_dev_batch_pair': ms_loop,
            'w_step_ms_at_4_steps_per_replay': ms_w4,
            'alpha_step_with_metric_forward_ms_at_4_steps_per_replay': ms_am4,
            'loop_ms_per_train_dev_batch_pair_at_4_steps_per_replay': ms_loop4,
            'w_step_without_input_copy_ms': round(ms_w_nocopy, 4),
            'input_copy_us': round((ms_w - ms_w_nocopy) * 1e3, 1),
            'host_issue_us_per_w_step': round(host_us, 1),
            'pairs_per_s': round(world * 1e3 / ms_pair, 1), 'pairs_timed': pairs,
            'dp_cost_trainer_buckets': dp_phase,
            'includes': 'w-step: fwd + criterion + bwd w.r.t. the weight optimizer\'s tensors (the network and classifier '
                        'weights; input-feature gradients as far as the model asks for them; no alpha/beta/gamma '
                        'gradient: nothing reads it in this phase) + Adam(w, wd 1e-4); '
                        'alpha-step: fwd + criterion + bwd w.r.t. alpha/beta/gamma only + Adam(betas (0.5, 0.999), wd 1e-3); '
                        + ('each step = one hipGraph replay' if world == 1 else
                           'hipGraph replay + one flat RCCL all-reduce + one-launch Adam per step')}


def cpu_baseline(cname, c, batch, max_seconds=20.0, tier='F'):
    """The CPU oracle (a port of the reference's path: same aten op sequence, pinned against
    the reference by tests/golden) on this host's cores; same synthetic batch, dropout on.
    tier 'R': the reshape layers' conv -> BatchNorm -> ReLU -> dropout stacks on pooled (b, C_in, L)
    features in front of the hypernet (oracle.reshape_layer minus its pooling), gradients for them too."""
    from oracle import fusion_oracle as fo, synth
    cfg = fo.CONFIGS[cname]
    p = synth.make_params(cfg, 2)
    arch = synth.make_arch(cfg, 2, 1e-3)
    cw, cb = synth.make_classifier(cfg, c['nout'], 2)
    xs = synth.make_inputs(cfg, batch, 0)
    y = synth.make_labels(c['loss'], batch, c['nout'], 0)
    if tier == 'R':
        g = torch.Generator().manual_seed(0)
        raws = [torch.relu(torch.randn(batch, ci, cfg.L, generator=g)) for ci in C_INS[cname]]
        rp = []
        for i, ci in enumerate(C_INS[cname]):
            shapes = {'conv.weight': (cfg.C, ci, 1), 'conv.bias': (cfg.C,), 'bn.weight': (cfg.C,), 'bn.bias': (cfg.C,),
                      'bn.running_mean': (cfg.C,), 'bn.running_var': (cfg.C,), 'bn.num_batches_tracked': ()}
            rp.append(synth.make_params(cfg, 100 + i, shapes))

    def one():
        t0 = time.perf_counter()
        if tier == 'R':
            leaves = [[v.detach().requires_grad_(True) if v.is_floating_point() and v.dim() > 0 and 'running' not in k
                       else v for k, v in q.items()] for q in rp]
            qs = [dict(zip(q.keys(), lv)) for q, lv in zip(rp, leaves)]
            xr = [x.detach().requires_grad_(True) for x in raws]
            feats = [fo._dropout(fo._relu(fo._conv_bn(x, q['conv.weight'], q['conv.bias'], q['bn.weight'], q['bn.bias'],
                                                      q['bn.running_mean'], q['bn.running_var'], True)), cfg.drpt, True)
                     for x, q in zip(xr, qs)]
            pp = {k: (v if fo.is_buffer(k) else v.detach().requires_grad_(True)) for k, v in p.items()}
            a = [t.detach().requires_grad_(True) for t in arch]
            cwl, cbl = cw.detach().requires_grad_(True), cb.detach().requires_grad_(True)
            fo.loss_fn(c['loss'])(fo.hypernet_logits(feats, a, pp, cwl, cbl, cfg, True), y).backward()
        else:
            fo.search_step(xs, y, arch, p, cw, cb, cfg, c['loss'], training=True)
        return time.perf_counter() - t0

    # the op mix is ~1000 small aten calls: more threads than a few cores make it SLOWER
    # (measured on the 256-core GPU host: 8 thr 35 ms, 32 thr 94 ms, 128 thr 500 ms), so the
    # baseline uses the fastest thread count of a short probe, and says which.
    ncpu = os.cpu_count() or 1
    probe = {}
    for nt in sorted({1, 4, 8, 16, min(32, ncpu)}):
        if nt > ncpu:
            continue
        torch.set_num_threads(nt)
        one()
        probe[nt] = min(one(), one())
    # the two fastest thread counts of the probe are both timed properly (the probe is 2 steps and
    # a noisy host can mislead it); the better median is the baseline
    finalists = sorted(probe, key=probe.get)[:2]
    runs = {}
    for nt in finalists:
        torch.set_num_threads(nt)
        ts = []
        t_start = time.time()
        for i in range(30):
            ts.append(one())
            if time.time() - t_start > max_seconds / len(finalists) and len(ts) >= 8:
                break
        runs[nt] = ts
    best = min(runs, key=lambda k: statistics.median(runs[k][3:]))
    torch.set_num_threads(best)
    times = runs[best]
    timed = times[3:]
    med = statistics.median(timed)
    return {'value': round(1.0 / med, 3), 'unit': 'steps/s', 'cores': torch.get_num_threads(),
            'kind': 'port', 'ms_per_step': round(med * 1e3, 3),
            'samples_per_s': round(batch / med, 1),
            'host_cpus': ncpu,
            'thread_probe_ms': {str(k): round(v * 1e3, 2) for k, v in probe.items()},
            'sample': f'{len(timed)} timed fwd+bwd steps (after {len(times) - len(timed)} warm-up) of the '
                      f'same {cname} batch-{batch} synthetic workload' + (' incl. the reshape layers (tier R)' if tier == 'R' else '') + ', torch CPU fp32, median; threads = '
                      
                      f'faster of the two best of a probe over {sorted(probe)} on a {ncpu}-cpu host'}



# ----------------------------------------------------------------------------------------------- row f3: found stage
# Fixed genotypes of the found stage (what a search on these datasets returns in shape: the ones the reference-
# generated fixtures tests/golden/found_{mm,nt}_*.npz were computed with).
FOUND_GENOTYPES = {
    'mmimdb': {'edges': [['skip', 3], ['skip', 4], ['skip', 2], ['skip', 4]], 'concat': [6, 7],
               'steps': [{'inner_edges': [['skip', 0], ['skip', 1]], 'inner_steps': ['Sum'], 'inner_concat': [2]},
                         {'inner_edges': [['skip', 1], ['skip', 0]], 'inner_steps': ['ScaleDotAttn'],
                          'inner_concat': [2]}]},
    'ntu': {'edges': [['skip', 0], ['skip', 1], ['skip', 1], ['skip', 5]], 'concat': [8, 9],
            'steps': [{'inner_edges': [['skip', 1], ['skip', 0], ['skip', 0], ['skip', 2]],
                       'inner_steps': ['LinearGLU', 'LinearGLU'], 'inner_concat': [2, 3]},
                      {'inner_edges': [['skip', 0], ['skip', 1], ['skip', 1], ['skip', 2]],
                       'inner_steps': ['LinearGLU', 'ConcatFC'], 'inner_concat': [2, 3]}]},
}


def found_genotype(cname):
    from models.search.darts.genotypes import Genotype, StepGenotype
    g = FOUND_GENOTYPES[cname]
    return Genotype(edges=[tuple(e) for e in g['edges']],
                    steps=[StepGenotype(inner_edges=[tuple(e) for e in st['inner_edges']],
                                        inner_steps=list(st['inner_steps']), inner_concat=list(st['inner_concat']))
                           for st in g['steps']],
                    concat=list(g['concat']))


class FoundNet(torch.nn.Module):
    """fusion_net (Found_FusionNetwork, x != y kernels) -> central_classifier, wired like Found_*_Net minus the
    backbones and reshape layers (models/search/mmimdb_darts_searchable.py:128-190, darts/model.py:133-160)."""

    def __init__(self, c, cname):
        super().__init__()
        from bmnas import nn as bnn
        from models.search.darts.model import Found_FusionNetwork
        self.fusion_net = Found_FusionNetwork(c['S'], c['M'], c['N'], 2, make_args(c), None, found_genotype(cname))
        self.central_classifier = bnn.Linear(c['M'] * c['C'] * c['L'], c['nout'])

    def forward(self, xs):
        return self.fusion_net.forward_classified(list(xs), self.central_classifier)      # as models/search/_common.py


def found_cpu_baseline(cname, c, batch, max_seconds=15.0):
    """The oracle's found network (oracle.found_cell: the reference's discrete cell, model.py / node.py) + classifier
    + criterion, forward and backward, on this host's cores."""
    from oracle import fusion_oracle as fo, synth
    cfg = fo.CONFIGS[cname]
    geno = fo.genotype_from_jsonable(FOUND_GENOTYPES[cname])
    p = synth.make_params(cfg, 2, fo.found_param_shapes(cfg, geno))
    cw, cb = synth.make_classifier(cfg, c['nout'], 2)
    xs = synth.make_inputs(cfg, batch, 0)
    y = synth.make_labels(c['loss'], batch, c['nout'], 0)

    def one():
        t0 = time.perf_counter()
        pp = {k: (v if fo.is_buffer(k) else v.detach().requires_grad_(True)) for k, v in p.items()}
        cwl, cbl = cw.detach().requires_grad_(True), cb.detach().requires_grad_(True)
        feat = fo.found_cell([x.detach().requires_grad_(True) for x in xs], geno, pp, cfg, True)
        fo.loss_fn(c['loss'])(torch.nn.functional.linear(feat.reshape(batch, -1), cwl, cbl), y).backward()
        return time.perf_counter() - t0

    ncpu = os.cpu_count() or 1
    probe = {}
    for nt in sorted({1, 4, 8, 16}):
        if nt <= ncpu:
            torch.set_num_threads(nt)
            one()
            probe[nt] = min(one(), one())
    best = min(probe, key=probe.get)
    torch.set_num_threads(best)
    ts, t_start = [], time.time()
    while len(ts) < 30 and (time.time() - t_start < max_seconds or len(ts) < 8):
        ts.append(one())
    med = statistics.median(ts[3:])
    return {'value': round(1.0 / med, 3), 'unit': 'steps/s', 'cores': best, 'kind': 'port',
            'ms_per_step': round(med * 1e3, 3), 'host_cpus': ncpu,
            'sample': f'{len(ts) - 3} timed fwd+bwd steps (after 3 warm-up) of the same {cname} batch-{batch} found '
                      f'network (oracle.found_cell + classifier + criterion), torch CPU fp32, median; threads = the '
                      f'fastest of a probe over {sorted(probe)}'}


def found_main(a, log):
    """bench.py --stage found: the found-stage training step (main_darts_found_*.py -> train_*_track_*(status='eval'):
    forward + criterion + backward + Adam over EVERY parameter, one hipGraph replay per step) of a fixed genotype on
    synthetic (b, C, L) features, and the evaluation forward (test_*_track_*: GraphedForward)."""
    from bmnas import lib, nn as bnn
    from bmnas.graph import GraphedForward, GraphedTrainStep
    from bmnas.optim import Adam
    lib.load()
    if a.config not in FOUND_GENOTYPES:
        raise SystemExit(f'--stage found: no fixed genotype for {a.config}')
    device = torch.device('cuda', 0)
    torch.cuda.set_device(device)
    c = CONFIGS[a.config]
    torch.manual_seed(2)
    model = FoundNet(c, a.config).to(device).train()
    crit = bnn.BCEWithLogitsLoss() if c['loss'] == 'bce' else bnn.CrossEntropyLoss()
    xs, y = synth_batch(c, a.batch, device, 0)
    xs = [x.detach() for x in xs]
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)      # main_darts_found_mmimdb.py:120

    def step():
        opt.zero_grad()
        loss = crit(model(xs), y)
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t0) / 10 * 1e3
    if a.mode == 'graph':
        g = GraphedTrainStep(model, crit, opt, xs, y)
        run = lambda: g(xs, y)
        for _ in range(400):
            run()
    else:
        run = step
    torch.cuda.synchronize()
    for _ in range(a.warmup):
        run()
    times = []
    for _ in range(a.regions):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            run()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    result = {
        'metric': f'found-network train-steps/sec (fwd + criterion + bwd + Adam of the discrete fusion network) on '
                  f'{a.config} synthetic',
        'value': round(a.steps / dt, 3), 'unit': 'steps/s', 'n_gpus': 1, 'steps': a.steps, 'warmup': a.warmup,
        'ms_per_step': round(dt / a.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'{a.config} found network (fixed genotype, x != y kernels), batch {a.batch}, '
                               f'N{c["N"]} C{c["C"]} L{c["L"]} steps{c["S"]} node_steps{c["ns"]} '
                               f'node_multiplier{c["nm"]}, train mode, dropout {c["drpt"]}/0.1(attn)',
                   'genotype': FOUND_GENOTYPES[a.config], 'stage': 'found', 'mode': a.mode,
                   'step': 'fwd + criterion + bwd + Adam(all parameters, wd 1e-4) as one hipGraph replay'
                           if a.mode == 'graph' else 'the same issued from Python',
                   'global_batch': a.batch, 'parallelism': 'dp1'},
        'samples_per_s': round(a.steps * a.batch / dt, 1),
        'timed_regions': {'n': len(times), 'steps_each': a.steps, 'headline': 'median',
                          'ms_per_step': [round(t / a.steps * 1e3, 4) for t in times]},
        'eager_ms_per_step': round(eager_ms, 4),
    }
    log(f'found stage: {result["ms_per_step"]} ms/step (eager {eager_ms:.3f})')
    # the evaluation pass (test_*_track_*: model.eval(), no gradients): one replay per batch
    # (not in the profiled child: its kernel trace has to END with the training step's replays)
    try:
        if os.environ.get('BMNAS_BENCH_CHILD'):
            raise StopIteration
        model.eval()
        gf = GraphedForward.try_build(model, crit, xs, y)
        if gf:
            for _ in range(50):
                gf(xs, y)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                gf(xs, y)
            torch.cuda.synchronize()
            result['eval_forward_ms'] = round((time.perf_counter() - t0) / 200 * 1e3, 4)
        model.train()
    except StopIteration:
        pass
    except Exception as e:                           # noqa: BLE001
        result['eval_forward_ms'] = {'error': f'{type(e).__name__}: {e}'[:200]}
    if not a.no_roofline:
        def plain():
            for p_ in model.parameters():
                p_.grad = None
            crit(model(xs), y).backward()
        try:
            result.update(roofline_report(a, c, plain, result['ms_per_step'], log))
        except Exception as e:                       # noqa: BLE001
            result['roofline_error'] = f'{type(e).__name__}: {e}'[:300]
    if not a.no_cpu_baseline:
        try:
            result['cpu_baseline'] = found_cpu_baseline(a.config, c, a.batch)
            result['speedup_vs_cpu_baseline'] = round(result['value'] / result['cpu_baseline']['value'], 1)
        except Exception as e:                       # noqa: BLE001
            result['cpu_baseline'] = {'error': f'{type(e).__name__}: {e}'[:300]}
    print(json.dumps(result), flush=True)


# wrapper (bmnas.lib) -> substrings of the kernel symbols it may launch; used to attach the
# algorithmic units of an instrumented eager step to the kernels of a profiled graph replay
KERNELS_OF = {
    'mixsum_fwd': ('mixsum_fwd_k',), 'mixsum_bwd': ('mixsum_bwd_k',),
    'mixsum_pair_fwd': ('mixsum_pair_fwd_k',), 'mixsum_pair_bwd': ('mixsum_pair_bwd_k',),
    'cat_ln_fwd': ('cat_ln_fwd_k',), 'cat_ln_bwd': ('cat_ln_bwd_k',),
    'ln_affine_bwd': ('ln_affine_bwd_k',), 'ln_affine_bwd_multi': ('ln_affine_bwd_multi_k', 'ln_affine_bwd_k'),
    'backward_epilogue': ('backward_epilogue_k',),
    'sdpa_ln_fwd': ('sdpa_ln_fwd_k',), 'sdpa_ln_bwd': ('sdpa_ln_bwd_k',),
    'conv1x1_fwd': ('conv_pipe_fwd_k', 'conv_ksplit_k', 'conv_ksplit_multi_k', 'conv_lds_k', 'conv_fwd_k'),
    'conv1x1_bwd_data': ('conv_pipe_bwd_k', 'conv_ksplit_k', 'conv_ksplit_multi_k', 'conv_lds_k', 'conv_bwd_k'),
    'conv1x1_bwd_weight': ('conv_w_k',),
    'conv1x1_fwd_sdpa': ('conv_pipe_fwd_sdpa_k', 'conv_fwd_sdpa_k'),
    'conv1x1_bwd_all_sdpa': ('conv_bwd_all_pipe_k', 'conv_bwd_all_k'),
    # (large grids run as bn_bwd_apply + data + weight launches: the units go to the data-gradient kernel)
    'conv1x1_bwd_all': ('conv_bwd_pair_k', 'conv_pipe_bwd_k', 'conv_ksplit_k', 'conv_ksplit_multi_k', 'conv_lds_k', 'conv_bwd_k'),
    'node_mix_fwd': ('node_mix_fwd_k',), 'node_mix_ln_fwd': ('node_mix_ln_fwd_k',),
    'node_mix_conv_fwd': ('mix_conv_fwd_k',),
    'node_mix_bwd': ('node_mix_bwd_k',), 'node_mix_ln_bwd': ('node_mix_ln_bwd_k',),
    'bn_relu_fwd': ('bn_relu_fwd_k',), 'bn_relu_bwd': ('bn_relu_bwd_k',),
    'bn_relu_ln_fwd': ('bn_relu_ln_fwd_k',), 'bn_relu_ln_bwd': ('bn_relu_ln_bwd_k',),
    'bn_glu_fwd': ('bn_glu_fwd_k',), 'bn_glu_bwd': ('bn_glu_bwd_k',),
    'bn_bwd_apply': ('bn_bwd_apply_k',), 'bn_finalize': ('bn_finalize_k',), 'fold_weight': ('fold_weight_k',),
    'linear_fwd': ('linear_fwd_k',), 'linear_bwd': ('linear_bwd_k',),
    'bce_logits': ('bce_logits_k',), 'cross_entropy': ('ce_rows_k', 'ce_small_k'),
    'head_fwd': ('head_fwd_k',), 'head_bwd': ('head_bwd_k',), 'head_loss_bwd': ('head_loss_bwd_k',),
    'cell_prologue': ('cell_prologue_k',), 'cell_prologue_pair': ('cell_prologue_pair_k',),
    'node_mix_pre_fwd': ('node_mix_pre_fwd_k',), 'node_mix_lnp_bwd': ('node_mix_lnp_bwd_k',),
    'mixsum_pair_fwd_lazy': ('mixsum_pair_fwd_lazy_k',), 'mixsum_pair_bwd_lazy': ('mixsum_pair_bwd_lazy_k',),
    'mixsum_pair_bwd_x': ('mixsum_pair_bwd_x_k', 'mixsum_pair_bwd_k'),
    'head_fwd_lazy': ('head_fwd_k',), 'head_bwd_lazy': ('head_bwd_k',),
    'adam_multi': ('adam_multi_k',),
    'conv1x1_fwd_group': ('conv_fwd_group_k',), 'conv1x1_bwd_group': ('conv_bwd_group_k',),
    'bn_relu_fwd_group': ('bn_relu_fwd_group_k',), 'bn_relu_bwd_group': ('bn_relu_bwd_group_k',),
}


def short_kernel_name(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    i = n.find('(')
    return n if i < 0 else n[:i]


def profile_graph_replay(a, log, steps=60):
    """Run THIS bench command (same config / batch, headline loop only) as a child process under
    `rocprofv3 --kernel-trace` and return the kernels of its hipGraph replays:
    [[(name, start_ns, end_ns), ... one replay], ...].  Launch-inclusive device durations — the
    figures a committed profiles/*_kernel_stats.csv of the same command shows."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(rp):
        raise RuntimeError('rocprofv3 not found')
    out = tempfile.mkdtemp(prefix='bmnas_prof_', dir='/tmp')
    cmd = [rp, '--kernel-trace', '--output-format', 'csv', '-d', out, '--', sys.executable,
           os.path.abspath(__file__), '--config', a.config, '--batch', str(a.batch), '--tier', a.tier,
           '--steps', str(steps), '--warmup', '5', '--no-cpu-baseline', '--no-roofline', '--no-full-step',
           '--stage', getattr(a, 'stage', 'search')]
    env = dict(os.environ, TMPDIR='/tmp', BMNAS_BENCH_CHILD='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True, timeout=600)
        files = glob.glob(os.path.join(out, '**', '*kernel_trace.csv'), recursive=True)
        if r.returncode != 0 or not files:
            raise RuntimeError(f'rocprofv3 child failed (rc {r.returncode}): {r.stdout[-400:]}')
        child_ms = None
        for line in r.stdout.splitlines():
            if line.startswith('{') and '"ms_per_step"' in line:
                child_ms = json.loads(line)['ms_per_step']
        rows = []
        for f in files:
            with open(f) as fh:
                rows += [(x['Kernel_Name'], int(x['Start_Timestamp']), int(x['End_Timestamp']))
                         for x in csv.DictReader(fh)]
    finally:
        shutil.rmtree(out, ignore_errors=True)
    rows.sort(key=lambda x: x[1])
    names = [x[0] for x in rows]
    # the replays are the periodic tail of the trace.  Every step starts with exactly one cell prologue
    # launch: the distance between its last occurrences is the period (a step that repeats an identical
    # sub-sequence — six reshape layers in a row — would fool a shortest-repeat search); fallback: the
    # shortest period whose last three repetitions agree
    period = None
    marks = [i for i, n in enumerate(names) if 'cell_prologue_k' in n or 'cell_prologue_pair_k' in n]
    if len(marks) >= 4 and marks[-1] - marks[-2] == marks[-2] - marks[-3]:
        period = marks[-1] - marks[-2]        # (the trace ENDS with the last replay: cut from the end)
    if period is None:
        for per in range(4, 400):
            if len(names) > 3 * per + 3 and names[-per:] == names[-2 * per:-per] == names[-3 * per:-2 * per]:
                period = per
                break
    if period is None:
        raise RuntimeError('no periodic replay pattern in the kernel trace')
    n_rep = 0
    while (n_rep + 2) * period <= len(names) and names[-(n_rep + 1) * period - period:-(n_rep + 1) * period] == names[-period:]:
        n_rep += 1
    n_rep = min(n_rep, steps)
    tail = rows[-n_rep * period:]
    # rotate so that a replay starts at its first kernel (the trace tail may be cut anywhere: it is not)
    replays = [tail[i * period:(i + 1) * period] for i in range(n_rep)]
    log(f'rocprofv3 child: {len(rows)} dispatches, {period} kernels per replay, {n_rep} replays used')
    return replays, child_ms


def roofline_report(a, c, step, ms_per_step, log):
    """roofline / roofline_kernels: every kernel of the step with its launch-inclusive device
    duration from a rocprofv3 kernel trace of the hipGraph replay (child process, same command) and
    the algorithmic bytes / FLOPs of that launch (units from the arguments
    of an instrumented eager step).  achieved = units / duration; frac = achieved / peak."""
    from bmnas import lib
    algo = algo_table(c['C'], c['L'])
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    lib.profile_begin(algo, events=False)
    step()
    calls = lib.profile_end_calls()              # [(wrapper, bound, units)] of ONE step, in launch order
    source = 'rocprofv3 --kernel-trace of this command (child process), hipGraph replays'
    try:
        replays, child_ms = profile_graph_replay(a, log)
    except Exception as e:                           # noqa: BLE001
        log(f'rocprofv3 child unavailable ({e}); falling back to HIP-event brackets')
        return roofline_from_events(a, c, step, ms_per_step, log, algo)
    period = len(replays[0])
    # mean duration per position of the replay, and the launch spacing (start to next start)
    dur = [sum(r[i][2] - r[i][1] for r in replays) / len(replays) / 1e3 for i in range(period)]
    span = sum(r[-1][2] - r[0][1] for r in replays) / len(replays) / 1e3
    names = [short_kernel_name(replays[0][i][0]) for i in range(period)]
    # attach the wrappers' units to the kernels, in order
    units = [None] * period
    ptr = 0
    skipped = []
    for i, n in enumerate(names):
        # a wrapper whose kernel is not in the table must not derail everything after it: look a few calls ahead
        for ahead in range(4):
            q = ptr + ahead
            if q < len(calls) and any(k in n for k in KERNELS_OF.get(calls[q][0], ())):
                skipped += [cname for cname, _, _ in calls[ptr:q]]
                units[i] = calls[q]
                ptr = q + 1
                break
    unmatched = skipped + [cname for cname, _, _ in calls[ptr:]]
    tpath = next((q for q in (os.path.join(ROOT, 'profiles', f'r0{r}_traffic.json') for r in (6, 5, 4, 3, 2))
                  if os.path.exists(q)), '')
    traffic = {}
    if a.config == 'mmimdb' and a.batch == 128 and a.tier == 'F' and os.path.exists(tpath):
        with open(tpath) as f:
            traffic = {k: v.get('traffic_bytes') for k, v in json.load(f).items()}
    # MFMA utilisation and cache behaviour of the GEMM launches from the hardware counters:
    # profiles/r06_pmc_gemm.json = tools/pmc_summary.py over four rocprofv3 --pmc passes of this same step
    pmc = {}
    ppath = os.path.join(ROOT, 'profiles', 'r06_pmc_gemm.json')
    if a.config == 'mmimdb' and a.batch == 128 and a.tier == 'F' and os.path.exists(ppath):
        with open(ppath) as f:
            pmc = json.load(f)
    agg = {}
    for i, n in enumerate(names):
        g = agg.setdefault(n, {'us': 0.0, 'n': 0, 'units': 0.0, 'bound': None, 'wrapper': None})
        g['us'] += dur[i]
        g['n'] += 1
        if units[i] is not None:
            g['wrapper'], g['bound'] = units[i][0], units[i][1]
            g['units'] += units[i][2]
    rows = []
    for n, g in agg.items():
        row = {'kernel': n, 'wrapper': g['wrapper'], 'launches_per_step': g['n'],
               'avg_us': round(g['us'] / g['n'], 2), 'us_per_step': round(g['us'], 2)}
        if g['bound'] is not None:
            per = g['units'] / g['n']
            sec = g['us'] / g['n'] * 1e-6
            if g['bound'] == 'hbm':
                ach, peak, unit = per / sec / 1e9, HBM_PEAK_GBS, 'GB/s'
            else:
                ach, peak, unit = per / sec / 1e12, MFMA_F32_PEAK_TFLOPS, 'TFLOP/s'
            row.update({'bound': g['bound'], 'achieved': round(ach, 2), 'peak': peak, 'unit': unit,
                        'frac': round(ach / peak, 4), 'algorithmic_units_per_launch': round(per),
                        'traffic': traffic.get(n)})
        else:
            row.update({'bound': 'latency', 'achieved': None, 'peak': None, 'unit': None, 'frac': None,
                        'algorithmic_units_per_launch': None, 'traffic': traffic.get(n)})
        if traffic.get(n):
            # the same launch priced by the bytes the PMC passes counted for it (profiles/*traffic.json): what an
            # MFMA-priced launch of a few hundred MFLOP is really bound by (the head pair: VERDICT r04 item 2)
            row['frac_of_hbm_by_traffic'] = round(traffic[n] / (g['us'] / g['n'] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        if n in pmc:
            # SQ_VALU_MFMA_BUSY_CYCLES (cycles the matrix pipes were busy, summed over the 1024 SIMDs) against the launch
            # duration measured HERE at the 2.4 GHz engine clock; hit rates and the wave-time split as collected
            d, cnt = pmc[n]['derived'], pmc[n]['counters']
            row['pmc'] = {'source': 'profiles/r06_pmc_gemm.json (rocprofv3 --pmc, separate passes)',
                          'mfma_util': round(cnt['SQ_VALU_MFMA_BUSY_CYCLES'] /
                                             (1024 * g['us'] / g['n'] * 1e-6 * 2.4e9), 4),
                          'mfma_gflop_executed': d['mfma_gflop'], 'l2_hit_rate': d['l2_hit_rate'],
                          'l1_hit_rate': d['l1_hit_rate'],
                          'wave_time_waiting_for_any_instruction': d['wave_time_waiting_for_any_instruction'],
                          'lds_bank_conflict_share': d['lds_bank_conflict_share']}
        rows.append(row)
    rows.sort(key=lambda r: -r['us_per_step'])
    out = {'roofline_kernels': rows,
           'roofline_check': {'kernels_per_step': period, 'sum_kernel_us_per_step': round(sum(dur), 2),
                              'replay_span_us': round(span, 2), 'headline_us_per_step': round(ms_per_step * 1e3, 2),
                              'profiled_child_us_per_step': None if child_ms is None else round(child_ms * 1e3, 2),
                              'wrappers_without_kernel': unmatched}}
    # K1 (the >= 40 % HBM target of north_star) under SURVEY.md 8(d)'s LITERAL byte counts — forward (n_in + 1) T,
    # backward (2 n_in + 1) T per cell-level sum, whatever else the launch carries (the pair kernels also form /
    # differentiate the node's first inner sum, the first one also the cell prologue: not counted here)
    T_bytes = a.batch * c['C'] * c['L'] * 4
    k1 = {'formula': 'SURVEY.md 8(d): fwd (n_in+1)*T, bwd (2*n_in+1)*T per cell-level mixed sum; T = b*C*L*4',
          'T_bytes': T_bytes, 'peak_GBs': HBM_PEAK_GBS, 'launches': []}
    tot = {'fwd': [0.0, 0.0], 'bwd': [0.0, 0.0]}
    for i, n in enumerate(names):
        u = units[i]
        n_in = None if u is None else getattr(u[2], 'n_in', None)
        if n_in is None or n_in <= 3:                  # cell-level sums only (the inner sums have 2-4 inputs)
            continue
        d = 'bwd' if 'bwd' in u[0] else 'fwd'
        by = ((2 * n_in + 1) if d == 'bwd' else (n_in + 1)) * T_bytes
        k1['launches'].append({'kernel': n, 'dir': d, 'n_in': n_in, 'bytes': by, 'avg_us': round(dur[i], 2),
                               'frac': round(by / (dur[i] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)})
        tot[d][0] += by
        tot[d][1] += dur[i]
    for d in ('fwd', 'bwd'):
        if tot[d][1] > 0:
            k1[d + '_frac'] = round(tot[d][0] / (tot[d][1] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
    out['k1'] = k1
    top = next((r for r in rows if r['frac'] is not None), None)
    if top is not None:
        top = dict(top)
        top['measured'] = (source + f'; mean over {len(replays)} replays of End - Start per dispatch; '
                           'cross-check: profiles/r06_kernel_stats_<config>_b<batch>.csv (rocprofv3 --kernel-trace --stats of the same command)')
        out['roofline'] = top
    return out


def roofline_from_events(a, c, step, ms_per_step, log, algo):
    """Fallback when rocprofv3 cannot run: HIP-event brackets around every launch of instrumented
    eager steps queued behind a GPU-side spin blocker.  The bracket INCLUDES the two event records
    (~4-5 us on MI355X); nothing is subtracted, so these durations are upper bounds."""
    from bmnas import lib
    n_prof = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.cuda._sleep(20_000_000)
    e1.record()
    torch.cuda.synchronize()
    cycles_per_ms = 20_000_000 / max(e0.elapsed_time(e1), 1e-3)
    recs = {}
    for park_ms in (80, 200, 500):
        torch.cuda._sleep(int(cycles_per_ms * park_ms))
        guard = torch.cuda.Event()
        guard.record()
        lib.profile_begin(algo)
        for _ in range(n_prof):
            step()
        queued_in_time = not guard.query()
        recs, ev_ms = lib.profile_end()
        if queued_in_time:
            break
    rows = []
    for name, rr in recs.items():
        tot_ms = sum(r[0] for r in rr)
        units = sum(r[2] for r in rr)
        bound = rr[0][1]
        if bound == 'hbm':
            ach, peak, unit = units / (tot_ms * 1e-3) / 1e9, HBM_PEAK_GBS, 'GB/s'
        else:
            ach, peak, unit = units / (tot_ms * 1e-3) / 1e12, MFMA_F32_PEAK_TFLOPS, 'TFLOP/s'
        rows.append({'kernel': name, 'wrapper': name, 'bound': bound, 'achieved': round(ach, 2), 'peak': peak,
                     'unit': unit, 'frac': round(ach / peak, 4), 'traffic': None,
                     'algorithmic_units_per_launch': round(units / len(rr)),
                     'launches_per_step': len(rr) / n_prof, 'avg_us': round(tot_ms / len(rr) * 1e3, 2),
                     'us_per_step': round(tot_ms / n_prof * 1e3, 2)})
    rows.sort(key=lambda r: -r['us_per_step'])
    out = {'roofline_kernels': rows}
    if rows:
        top = dict(rows[0])
        top['measured'] = ('HIP events on the launch stream around every launch (rocprofv3 was unavailable); the '
                           f'bracket includes the event records (empty bracket {ev_ms * 1e3:.2f} us), nothing subtracted')
        out['roofline'] = top
    return out


def one_gpu_full_batch(a, c, device, log, replays=200):
    """The same step on ONE GPU holding the whole global batch of --scaling strong (hipGraph replay, local: no
    collective), for the line of a sharded run to be read against."""
    from bmnas import nn as bnn
    from bmnas.functions import unit_grad
    from bmnas.graph import GraphedStep
    torch.manual_seed(2)
    model = HyperNet(c, a.tier, a.config).to(device).train()
    crit = bnn.BCEWithLogitsLoss() if c['loss'] == 'bce' else bnn.CrossEntropyLoss()
    xs, y = synth_batch(c, a.global_batch, device, 0, a.tier, a.config)
    leaves = list(model.parameters()) + list(model.arch_parameters()) + xs

    def fn():
        with bnn.fused_criterion():
            loss = crit(model(xs), y)
        grads = torch.autograd.grad(loss, leaves, grad_outputs=unit_grad(device), allow_unused=True)
        for t, g in zip(leaves, grads):
            t.grad = g
        return loss

    g = GraphedStep(fn)
    for _ in range(400):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(replays):
        g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / replays * 1e3
    log(f'one GPU, whole global batch {a.global_batch}: {ms:.4f} ms/step')
    return {'global_batch': a.global_batch, 'ms_per_step': round(ms, 4), 'steps_per_s': round(1e3 / ms, 1),
            'what': 'fwd+bwd of the same hypernet on ONE GPU holding the whole global batch (hipGraph replay)'}


class DPStep:
    """The benchmarked step in its data-parallel shapes.  Every gradient that is averaged across ranks (weights +
    alpha/beta/gamma) lives in ONE flat fp32 bucket; the captured step writes into it, so the per-step
    communication is RCCL all-reduce(avg) on that bucket and nothing else (no flatten / unflatten / scale
    kernels).  The bucket is laid out with the LAST cell step's conv / BatchNorm gradients first: they are final
    half-way through the backward, and the 'overlap' shape reduces them on a forked stream while the first
    step's backward runs (bmnas.cell.NODE_DONE_HOOK)."""

    def __init__(self, model, c, crit, xs, y, params, arch, device, comm, loss_scale, bucket=True):
        self.model, self.c, self.crit, self.xs, self.y = model, c, crit, xs, y
        self.device, self.comm, self.loss_scale = device, comm, loss_scale
        self.shared = params + arch
        self.leaves = self.shared + xs
        self.last_node = model.fusion_net.cell._step_nodes[c['S'] - 1].node_cell
        early = [p_ for n_, p_ in self.last_node.named_parameters() if '.ln.' not in n_ and not n_.startswith('ln.')]
        self.early_ids = {id(p_) for p_ in early}
        order = early + [t for t in self.shared if id(t) not in self.early_ids]
        self.flat, self.view_of, self.n_early = None, {}, 0
        if bucket:
            self.flat = torch.zeros(sum(t.numel() for t in order), device=device)
            off = 0
            for t in order:
                self.view_of[id(t)] = self.flat[off:off + t.numel()].view(t.shape)
                off += t.numel()
                if id(t) in self.early_ids:
                    self.n_early = off
        self.flat_views = [self.view_of[id(t)] for t in self.shared] if bucket else []
        self.side = torch.cuda.Stream(device) if comm is not None else None

    def _node_done(self, i, NG):
        if i != self.c['S'] - 1:
            return
        gs = self.last_node.grads_in_param_order(NG)
        pairs = [(self.view_of[id(p_)], g_) for p_, g_ in zip(self.last_node.param_list(), gs)
                 if id(p_) in self.early_ids]
        self.side.wait_stream(torch.cuda.current_stream())   # fork (a capture turns it into a graph branch)
        with torch.cuda.stream(self.side):
            torch._foreach_copy_([v for v, _ in pairs], [g_ for _, g_ in pairs])
            self.comm.all_reduce(self.flat[:self.n_early], average=True)

    def make_step(self, shape):
        """shape: 'single' (no bucket), 'host' (gradients into the bucket; the all-reduce is issued by the host
        after the replay), 'graph' (the all-reduce is the last launch of the captured step), 'overlap' (two
        all-reduces inside the captured step, the first on a forked stream as soon as its gradients are final)."""
        from bmnas import cell as K
        from bmnas import nn as bnn
        from bmnas.functions import unit_grad
        model, crit, xs, y, leaves, shared = self.model, self.crit, self.xs, self.y, self.leaves, self.shared

        def fn():
            # gradients are taken with autograd.grad (no AccumulateGrad nodes, whose streams are pinned at
            # creation and do not follow the capture stream) and then attached as .grad — the tensors are
            # static, replays refresh them in place
            K.NODE_DONE_HOOK = self._node_done if shape == 'overlap' else None
            try:
                with bnn.fused_criterion():
                    loss = crit(model(xs), y)
                if shape != 'single' and self.loss_scale != 1.0:
                    grads = torch.autograd.grad(loss * self.loss_scale, leaves, allow_unused=True)
                else:
                    grads = torch.autograd.grad(loss, leaves, grad_outputs=unit_grad(self.device), allow_unused=True)
            finally:
                K.NODE_DONE_HOOK = None
            if shape == 'single':
                for t, g in zip(leaves, grads):
                    t.grad = g
                return loss
            late = [(v, g) for t, v, g in zip(shared, self.flat_views, grads)
                    if not (shape == 'overlap' and id(t) in self.early_ids)]
            torch._foreach_copy_([v for v, _ in late], [g for _, g in late])
            if shape == 'graph':
                self.comm.all_reduce(self.flat, average=True)            # a launch on the capture stream
            elif shape == 'overlap':
                self.comm.all_reduce(self.flat[self.n_early:], average=True)
                torch.cuda.current_stream().wait_stream(self.side)       # join
            for t, v in zip(shared, self.flat_views):
                t.grad = v
            for t, g in zip(xs, grads[len(shared):]):
                t.grad = g
            return loss
        return fn


def dp_shapes_selfcheck(cname, batch):
    """tests/test_optim_gpu.py: the bucket after a 'graph'-shape step and after an 'overlap'-shape step (each
    captured and replayed once) on the same model and batch, dropout off, world-size-1 communicator.
    -> {parameter name: (plain, overlapped)}"""
    from bmnas import dist as bdist
    from bmnas import nn as bnn
    from bmnas.graph import GraphedStep
    c = dict(CONFIGS[cname], drpt=0.0)
    device = torch.device('cuda', torch.cuda.current_device())
    torch.manual_seed(2)
    model = HyperNet(c, 'F', cname).to(device).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    crit = bnn.BCEWithLogitsLoss() if c['loss'] == 'bce' else bnn.CrossEntropyLoss()
    xs, y = synth_batch(c, batch, device, 0, 'F', cname)
    params, arch = list(model.parameters()), list(model.arch_parameters())
    dp = DPStep(model, c, crit, xs, y, params, arch, device, bdist.NativeComm.get(), 1.0)
    names = [n for n, _ in model.named_parameters()] + [f'arch.{i}' for i in range(len(arch))]
    state = {k: v.clone() for k, v in model.state_dict().items()}
    out = {}
    for shape in ('graph', 'overlap'):
        g = GraphedStep(dp.make_step(shape), warmup=1)
        model.load_state_dict(state)
        dp.flat.fill_(float('nan'))
        g.replay()
        torch.cuda.synchronize()
        for n, v in zip(names, dp.flat_views):
            out.setdefault(n, []).append(v.detach().clone())
    return {k: tuple(v) for k, v in out.items()}


def headline(a, c, world, shapes, best, eager_ms, rccl, note=None):
    """The JSON line.  shapes: shape name -> list of timed-region durations (s), each region = --steps steps
    bracketed by barrier + synchronize, max over ranks; the headline is the MEDIAN region of the best shape."""
    dt = statistics.median(shapes[best])
    step_desc = {'single': 'fwd+bwd, one hipGraph replay',
                 'k4': 'fwd+bwd, FOUR consecutive steps (each over a resident batch of its own, fresh dropout masks) per '
                       'hipGraph replay: K steps = K / 4 replays',
                 'eager': 'fwd+bwd issued from Python, then flatten + RCCL all-reduce + copy back',
                 'host': 'fwd+bwd as one hipGraph replay writing every w- and arch-gradient into one flat bucket, '
                         'then ONE host-issued RCCL all-reduce(avg) of the bucket',
                 'graph': 'fwd+bwd + ONE RCCL all-reduce(avg) of the flat gradient bucket, all inside one hipGraph '
                          'replay (bmnas_allreduce_f32 through the C ABI)',
                 'graph_k4': 'FOUR consecutive steps per hipGraph replay, each fwd+bwd over a resident batch of its own + ONE '
                             'RCCL all-reduce(avg) of its flat gradient bucket inside the replay: K steps = K / 4 replays',
                 'overlap': 'fwd+bwd + RCCL all-reduce(avg) of the flat gradient bucket in two parts inside one '
                            'hipGraph replay: the last cell step\'s conv / BatchNorm gradients on a forked stream '
                            'while the first step\'s backward runs, the rest at the end'}[best]
    result = {
        'metric': ('search-steps/sec (fwd+bwd of fusion hypernet) on MM-IMDB synthetic'
                   if a.config == 'mmimdb' else
                   f'search-steps/sec (fwd+bwd of fusion hypernet) on {a.config} synthetic')
                  + (' [tier R: reshape layers + hypernet]' if a.tier == 'R' else ''),
        # weak: every rank steps its own --batch samples (N steps' worth of samples per step interval); strong: a step
        # IS the global batch, whatever the number of ranks
        'value': round((1 if getattr(a, 'scaling', 'weak') == 'strong' else world) * a.steps / dt, 3),
        'unit': 'steps/s',
        'n_gpus': world,
        'steps': a.steps,
        'warmup': a.warmup,
        'ms_per_step': round(dt / a.steps * 1e3, 4),
        'higher_is_better': True,
        'scaling': getattr(a, 'scaling', 'weak'),
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': f'{a.config} fusion search, batch {a.batch} per GPU x {world} GPU(s), '
                               f'N{c["N"]} C{c["C"]} L{c["L"]} steps{c["S"]} node_steps{c["ns"]} '
                               f'node_multiplier{c["nm"]}, train mode, dropout {c["drpt"]}/0.1(attn)',
                   'inputs': ('(b, C, L) features' if a.tier == 'F' else
                              f'pooled backbone features (b, C_in, L), C_in = {C_INS[a.config]}, through the reshape layers'),
                   'global_batch': a.batch * world, 'per_gpu_batch': a.batch,
                   'parallelism': f'dp{world}', 'mode': a.mode, 'step': step_desc},
        'samples_per_s': round(world * a.steps * a.batch / dt, 1),
        'timed_regions': {'n': len(shapes[best]), 'steps_each': a.steps, 'headline': 'median',
                          'ms_per_step': [round(t / a.steps * 1e3, 4) for t in shapes[best]]},
    }
    if len(shapes) > 1:
        result['step_shapes'] = {k: {'ms_per_step_median': round(statistics.median(v) / a.steps * 1e3, 4),
                                     'ms_per_step': [round(t / a.steps * 1e3, 4) for t in v]}
                                 for k, v in shapes.items()}
        result['step_shapes']['headline'] = best
    if rccl is not None:
        result['rccl'] = rccl
    if note:
        result['note'] = note
    if eager_ms is not None:
        result['eager_ms_per_step'] = round(eager_ms, 4)
    return result


def dp_report(bdist, comm, comm_err, flat, n_early, device, world, rank, host_reduce, log, base=None):
    """The N > 1 line's evidence that the collective really spans the ranks: what the process group and the
    C-ABI communicator report, every rank's device (all-gathered), and the bucket's all-reduce timed alone."""
    import torch.distributed as tdist
    props = torch.cuda.get_device_properties(device)
    me = {'rank': rank, 'device': str(device), 'name': props.name,
          'pci': ':'.join(f'{getattr(props, k, -1):02x}' if isinstance(getattr(props, k, None), int) else '?'
                          for k in ('pci_domain_id', 'pci_bus_id', 'pci_device_id')),
          'uuid': str(getattr(props, 'uuid', ''))}
    if comm is not None:
        me['comm'] = comm.info
    ranks = [me]
    if world > 1:
        ranks = [None] * world
        tdist.all_gather_object(ranks, me)
    out = {'ranks': world, 'backend': tdist.get_backend() if world > 1 else 'none (single process)',
           'comm_ranks': None if comm is None else comm.info['ranks'],
           'rccl_version': None if comm is None else comm.info['rccl_version'],
           'devices': ranks, 'distinct_devices': len({(r['pci'], r['uuid']) for r in ranks}),
           'allreduce_bytes': int(flat.numel() * 4), 'allreduce_early_bytes': int(n_early * 4)}
    if comm_err:
        out['native_comm_error'] = comm_err

    def timed(fn, n=30):
        for _ in range(5):
            fn()
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = torch.tensor([e0.elapsed_time(e1) / n * 1e3], device=device, dtype=torch.float64)
        if world > 1:
            bdist.all_reduce(t, tdist.ReduceOp.MAX)
        return round(float(t.item()), 2)

    us = {}
    if base is not None:
        us = dict(base.get('allreduce_us', {}))          # the torch.distributed figure of the first call
    elif host_reduce is not None:
        us['torch_distributed'] = timed(host_reduce)
    if comm is not None:
        us['c_abi'] = timed(lambda: comm.all_reduce(flat, average=True))
        us['c_abi_early_part'] = timed(lambda: comm.all_reduce(flat[:n_early], average=True))
    out['allreduce_us'] = us
    log(f'rccl: {world} ranks, {out["distinct_devices"]} distinct devices, bucket {out["allreduce_bytes"]} B, '
        f'all-reduce alone {us} us')
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', default='mmimdb', choices=sorted(CONFIGS))
    ap.add_argument('--batch', type=int, default=128, help='per-GPU batch')
    ap.add_argument('--mode', default='graph', choices=['graph', 'eager'])
    ap.add_argument('--tier', default='F', choices=['F', 'R'],
                    help='F: (b, C, L) features straight into the fusion cell (headline); R: pooled raw '
                         'features (b, C_in_i, L) through the reshape layers first (SURVEY.md 8(d))')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-full-step', action='store_true')
    ap.add_argument('--regions', type=int, default=5,
                    help='timed regions of --steps steps each; the headline is their median (each region is the '
                         'contract\'s measurement: barrier + synchronize on both sides, max over ranks)')
    ap.add_argument('--stage', default='search', choices=['search', 'found'],
                    help='search: the hypernet step (headline); found: the found-stage training step of a fixed '
                         'genotype (SURVEY.md 8 row f3), one GPU')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='N > 1: weak = --batch samples per GPU (headline); strong = the GLOBAL batch of the BASELINE '
                         'config (mmimdb 1024, ntu 64, ego 48) split over the ranks, with the one-GPU time of the '
                         'whole batch measured beside it')
    ap.add_argument('--dp-selftest', action='store_true',
                    help='one GPU: build a world-size-1 RCCL communicator and run the N > 1 step shapes (bucket, '
                         'in-graph all-reduce, forked-stream overlap) through it')
    a = ap.parse_args()
    if a.stage == 'found':
        if int(os.environ.get('WORLD_SIZE', '1')) > 1:
            raise SystemExit('--stage found is a one-GPU line')

        def flog(msg):
            print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)
        return found_main(a, flog)

    if a.scaling == 'strong':
        # the BASELINE configs' GLOBAL batches (BASELINE.json configs 3-5), split over the ranks
        a.global_batch = {'mmimdb': 1024, 'ntu': 64, 'ego': 48}[a.config]
        w_env = max(1, int(os.environ.get('WORLD_SIZE', '1')))
        if a.global_batch % w_env:
            raise SystemExit(f'--scaling strong: global batch {a.global_batch} does not divide over {w_env} ranks')
        a.batch = a.global_batch // w_env

    from bmnas import dist as bdist
    from bmnas import lib
    rank, local, world = bdist.init_from_env('nccl')
    if world != a.gpus and world > 1:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}')
    lib.load()                                   # fail loudly if the HIP library is missing
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    c = CONFIGS[a.config]

    torch.manual_seed(2)                         # the mains' default --seed 2
    model = HyperNet(c, a.tier, a.config).to(device).train()
    from bmnas import nn as bnn
    from bmnas.functions import unit_grad
    crit = bnn.BCEWithLogitsLoss() if c['loss'] == 'bce' else bnn.CrossEntropyLoss()
    xs, y = synth_batch(c, a.batch, device, rank, a.tier, a.config)
    params = [p for p in model.parameters()]
    arch = list(model.arch_parameters())
    leaves = params + arch + xs

    def log(msg):
        if rank == 0:
            print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)

    def step():
        for t in leaves:
            t.grad = None
        with bnn.fused_criterion():              # the loss is read only after backward
            loss = crit(model(xs), y)
        loss.backward()
        return loss

    # N > 1: every gradient that is averaged across ranks (weights + alpha/beta/gamma) lives in ONE
    # flat fp32 bucket; the captured step writes into it, so the per-step communication is RCCL
    # all-reduce(avg) on that bucket and nothing else (no flatten / unflatten / scale kernels).  The bucket is
    # laid out with the LAST cell step's conv / BatchNorm gradients first: they are final half-way through the
    # backward (the 'overlap' shape reduces them on a forked stream while the first step's backward runs).
    shared = params + arch
    # RCCL can average in the collective (ReduceOp.AVG / ncclAvg): the captured step is then exactly the
    # single-GPU one; without it (gloo test harness) the loss is pre-scaled and the bucket summed
    use_avg = world > 1 and bdist.avg_supported(device)
    loss_scale = 1.0 if (world == 1 or use_avg) else 1.0 / world
    # RCCL through the C ABI (bmnas_allreduce_f32): a plain launch on the current stream, so it can be captured
    # INSIDE the step's hipGraph.  Tried whenever the process group is RCCL; every rank must succeed.
    # (built AFTER a complete host-issued measurement exists and under the watchdog, see below: a communicator
    # that never comes up on some node must not cost the line)
    comm, comm_err = None, None
    want_native = a.mode == 'graph' and ((world > 1 and torch.distributed.get_backend() == 'nccl') or a.dp_selftest)
    dp = DPStep(model, c, crit, xs, y, params, arch, device, comm, loss_scale, bucket=world > 1 or a.dp_selftest)
    flat, n_early, make_step = dp.flat, dp.n_early, dp.make_step

    if world > 1:
        bdist.broadcast_state(model, arch)
        red_all = bdist.FlatGradAllReducer(shared)          # eager mode: flatten -> all-reduce -> copy back

    def host_reduce():
        bdist.all_reduce(flat, torch.distributed.ReduceOp.AVG if use_avg else torch.distributed.ReduceOp.SUM)

    def measure(run, regions):
        """W untimed warm-up steps, then `regions` timed regions of EXACTLY a.steps steps, each bracketed by a
        barrier + synchronize on both sides, MAX over ranks per region -> list of region times (s)."""
        # the chip's clocks need some tens of ms of continuous work to settle after the capture (timed regions of 20
        # steps right after a 20-step warm-up read 0.170, 0.172, 0.220, 0.170, 0.163 ms/step; after 400 replays (~65 ms)
        # all five read 0.163): untimed replays first, THEN the contract's W warm-up steps and the timed regions
        # (a fixed COUNT, not a duration: under N > 1 `run` contains a collective and every rank must issue the same
        # number of them)
        if a.mode == 'graph':
            for _ in range(400):
                run()
            torch.cuda.synchronize()
        for _ in range(a.warmup):
            run()
        out = []
        for _ in range(regions):
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                run()
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], device=device, dtype=torch.float64)
                bdist.all_reduce(t, torch.distributed.ReduceOp.MAX)
                dt = float(t.item())
            out.append(dt)
        return out

    eager_ms = None
    shapes = {}
    from bmnas.graph import GraphedStep
    if a.mode == 'graph':
        # short eager measurement for reference, then capture
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t0) / 10 * 1e3
        log(f'eager {eager_ms:.3f} ms/step; capturing hipGraph')
        graphed = GraphedStep(make_step('single' if world == 1 else 'host'))
        log('captured')
        run_local = graphed.replay
    else:
        run_local = step

    def run():
        run_local()
        if world > 1:
            if a.mode == 'graph':
                host_reduce()
            else:
                red_all()

    times = measure(run, a.regions)
    log('timed regions (ms/step): ' + ', '.join(f'{t / a.steps * 1e3:.4f}' for t in times))
    first = 'single' if world == 1 else ('host' if a.mode == 'graph' else 'eager')
    shapes[first] = times
    best = first
    # N > 1: the same replay WITHOUT its collective (no rank waits for another: no barrier semantics needed beyond
    # measure()'s own) — what the all-reduce costs on top is then a difference of two measured numbers, not a model
    compute_only = None
    if (world > 1 or a.dp_selftest) and a.mode == 'graph':
        compute_only = statistics.median(measure(run_local, 3)) / a.steps * 1e3
        log(f'compute only (no all-reduce): {compute_only:.4f} ms/step')

    rccl = None
    guard = None
    if world > 1 or a.dp_selftest:
        rccl = dp_report(bdist, None, None, flat, n_early, device, world, rank, host_reduce if world > 1 else None, log)
    if want_native:
        # everything that goes through the C-ABI communicator — its rendezvous, its all-reduce timed alone, the
        # in-graph step shapes — runs AFTER a complete, valid measurement exists and under a watchdog: should the
        # communicator not come up or a captured RCCL launch wedge on some node, every rank prints / exits with
        # what has been measured
        pending = {'line': headline(a, c, world, shapes, best, eager_ms, rccl,
                                    note='C-ABI RCCL communicator / in-graph all-reduce shapes did not finish '
                                         '(watchdog): host-issued all-reduce only')}

        def bail():
            # a rank wedged inside a collective: the line printed is the complete host-issued measurement, marked
            # "watchdog": true so that no harness takes it for a finished N > 1 run, and the exit status says so too
            # (never restart or re-exec from here: a retry has to be a fresh child process)
            if rank == 0:
                print(json.dumps(dict(pending['line'], watchdog=True)), flush=True)
            os._exit(3)

        import threading
        guard = threading.Timer(float(os.environ.get('BMNAS_BENCH_WATCHDOG_S', '120')), bail)
        guard.daemon = True
        guard.start()
        try:
            comm = bdist.NativeComm.get()
        except Exception as e:                       # noqa: BLE001
            comm_err = f'{type(e).__name__}: {e}'[:200]
        if not bdist.all_ranks_agree(comm is not None, device):
            comm = None
        if comm is not None:
            dp.comm, dp.side = comm, torch.cuda.Stream(device)
            rccl = dp_report(bdist, comm, comm_err, flat, n_early, device, world, rank,
                             host_reduce if world > 1 else None, log, base=rccl)
            pending['line'] = headline(a, c, world, shapes, best, eager_ms, rccl,
                                       note='in-graph RCCL shapes did not finish (watchdog): host-issued all-reduce only')
        elif comm_err:
            rccl['native_comm_error'] = comm_err
        for shape in (('graph', 'overlap') if comm is not None else ()):
            try:
                g2 = GraphedStep(make_step(shape), warmup=1)
                ok = True
            except Exception as e:                   # noqa: BLE001
                ok = False
                rccl[f'{shape}_error'] = f'{type(e).__name__}: {e}'[:200]
            if not bdist.all_ranks_agree(ok, device):
                continue
            shapes[shape] = measure(g2.replay, a.regions)
            log(f'{shape}: ' + ', '.join(f'{t / a.steps * 1e3:.4f}' for t in shapes[shape]) + ' ms/step')
            if statistics.median(shapes[shape]) < statistics.median(shapes[best]):
                best = shape
        if ('graph' in shapes and a.steps % 4 == 0 and a.steps >= 4 and os.environ.get('BMNAS_BENCH_K4', '1') != '0'):
            # the N > 1 counterpart of the one-GPU `k4` shape: FOUR consecutive steps per replay, each over a resident
            # batch of its own, each ending in its own in-graph all-reduce of its own flat bucket (what a data-parallel
            # trainer with steps_per_replay = 4 replays).  Still under the watchdog; tried only when the one-step
            # `graph` shape captured and ran on every rank.
            ok, g4 = True, None
            try:
                ones = [make_step('graph')]
                for i in range(1, 4):
                    xs_i, y_i = synth_batch(c, a.batch, device, 100 + i + 10 * rank, a.tier, a.config)
                    d_i = DPStep(model, c, crit, xs_i, y_i, params, arch, device, comm, loss_scale)
                    ones.append(d_i.make_step('graph'))

                def four_graph():
                    out = None
                    for f in ones:
                        out = f()
                    return out
                g4 = GraphedStep(four_graph, warmup=1)
            except Exception as e:                   # noqa: BLE001
                ok = False
                rccl['graph_k4_error'] = f'{type(e).__name__}: {e}'[:200]
            if bdist.all_ranks_agree(ok, device):
                saved_steps, a.steps = a.steps, a.steps // 4
                try:
                    shapes['graph_k4'] = measure(g4.replay, a.regions)
                finally:
                    a.steps = saved_steps
                log('graph_k4: ' + ', '.join(f'{t / a.steps * 1e3:.4f}' for t in shapes['graph_k4']) + ' ms/step')
                if statistics.median(shapes['graph_k4']) < statistics.median(shapes[best]):
                    best = 'graph_k4'
        if world > 1:
            torch.distributed.barrier()
        guard.cancel()

    if (world == 1 and a.mode == 'graph' and not a.dp_selftest and a.steps % 4 == 0 and a.steps >= 4
            and not os.environ.get('BMNAS_BENCH_CHILD') and os.environ.get('BMNAS_BENCH_K4', '1') != '0'):
        # The same K steps with FOUR consecutive steps per hipGraph replay (K / 4 replays): each step is the full fwd + bwd
        # over a synthetic batch of its OWN (four resident batches), fresh dropout masks per step — what a trainer that
        # keeps k batches resident runs (models/search/train_searchable/_loop.py `steps_per_replay`,
        # bmnas.graph.GraphedTrainStep(k=...)).  What separates two replays is paid once per four steps.
        try:
            ones = [make_step('single')]
            for i in range(1, 4):
                xs_i, y_i = synth_batch(c, a.batch, device, 100 + i, a.tier, a.config)
                ones.append(DPStep(model, c, crit, xs_i, y_i, params, arch, device, None, 1.0,
                                   bucket=False).make_step('single'))

            def four():
                out = None
                for f in ones:
                    out = f()
                return out
            g4 = GraphedStep(four, warmup=1)
            saved_steps, a.steps = a.steps, a.steps // 4
            try:
                shapes['k4'] = measure(g4.replay, a.regions)        # (each region: K / 4 replays = K steps)
            finally:
                a.steps = saved_steps
            log('k4: ' + ', '.join(f'{t / a.steps * 1e3:.4f}' for t in shapes['k4']) + ' ms/step')
            if statistics.median(shapes['k4']) < statistics.median(shapes[best]):
                best = 'k4'
        except Exception as e:                       # noqa: BLE001 — a secondary shape must not cost the line
            shapes.pop('k4', None)
            log(f'k4 shape not measured: {type(e).__name__}: {e}')

    result = headline(a, c, world, shapes, best, eager_ms, rccl)
    dt = statistics.median(shapes[best])
    log(f'timed region done: {dt / a.steps * 1e3:.4f} ms/step')
    if compute_only is not None and rccl is not None:
        # what the exchange step costs and what that allows (the driver computes the scaling
        # efficiency itself from the per-N values — these are the ingredients, measured in this run)
        step_ms = dt / a.steps * 1e3
        ar = rccl.get('allreduce_us', {})
        ar_us = ar.get('c_abi', ar.get('torch_distributed'))
        exposed_us = max(0.0, (step_ms - compute_only) * 1e3)
        result['dp_cost'] = {
            'bucket': 'weights + alpha/beta/gamma in ONE flat fp32 bucket (the headline step differentiates both); the '
                      'trainers reduce per phase: the w-step its optimizer\'s tensors only, the alpha-step the 42-94 '
                      'architecture floats alone (bmnas.dist.attach(optimizer))',
            'bucket_bytes': rccl.get('allreduce_bytes'),
            'compute_only_ms_per_step': round(compute_only, 4),
            'step_ms_per_step': round(step_ms, 4),
            'allreduce_alone_us': ar_us,
            'allreduce_share_of_step': None if ar_us is None else round(ar_us / (step_ms * 1e3), 3),
            'exposed_us_per_step': round(exposed_us, 1),
            'weak_scaling_efficiency_vs_compute_only': round(compute_only / step_ms, 3),
            'speedup_bound_over_one_gpu': round(world * compute_only / step_ms, 2),
            'exposed_us_allowed_for_6x_at_8_gpus': round((8.0 / 6.0 - 1.0) * compute_only * 1e3, 1)}
    if world == 1 and a.mode == 'graph' and a.tier == 'R' and not a.dp_selftest:
        # The reference FREEZES its backbones (`p.requires_grad = False`: mmimdb_darts_searchable.py:68-71,
        # ntu_darts_searchable.py:86-90, ego_darts_searchable.py:85-89): the features that enter the reshape layers never
        # require a gradient, and autograd forms no data gradient for those layers.  The tier R line above keeps the input
        # gradients (comparable with earlier rounds, and what SURVEY.md 8(d) literally lists); this is the same step as
        # the reference's training loop runs it — no data-gradient tiles in the grouped backward launch.
        try:
            dpf = DPStep(model, c, crit, [x.detach() for x in xs], y, params, arch, device, None, 1.0, bucket=False)
            dpf.leaves = dpf.shared
            gF = GraphedStep(dpf.make_step('single'), warmup=1)
            tF = measure(gF.replay, 3)
            result['tier_R_frozen_backbones'] = {
                'ms_per_step': round(statistics.median(tF) / a.steps * 1e3, 4),
                'note': 'raw features without requires_grad (the reference freezes its backbones): no input gradient of '
                        'the reshape layers is formed; every weight / arch gradient as in the headline'}
            del gF
        except Exception as e:                       # noqa: BLE001 — diagnostics must not cost the headline
            result['tier_R_frozen_backbones'] = {'error': f'{type(e).__name__}: {e}'[:200]}
    if not a.no_full_step:
        # secondary figure: never allowed to take the headline line down with it — neither by raising nor (N > 1: it
        # contains collectives) by never returning: past the watchdog every rank leaves with the line as it stands
        import threading

        def bail2():
            if rank == 0:
                print(json.dumps(dict(result, full_search_step={'error': 'did not finish (watchdog)'}, watchdog=True)),
                      flush=True)
            os._exit(3)

        guard2 = threading.Timer(float(os.environ.get('BMNAS_BENCH_WATCHDOG_S', '120')), bail2)
        guard2.daemon = True
        guard2.start()
        try:
            result['full_search_step'] = full_search_step(model, crit, params, arch, xs, y, c, a, world, device,
                                                          log)
        except Exception as e:                       # noqa: BLE001
            result['full_search_step'] = {'error': f'{type(e).__name__}: {e}'[:300]}
            log(f'full search step failed: {e}')
        if world > 1:
            torch.distributed.barrier()
        guard2.cancel()
    if rank == 0 and not a.no_roofline and world == 1:
        try:
            result.update(roofline_report(a, c, step, result['ms_per_step'], log))
        except Exception as e:                       # noqa: BLE001 — diagnostics must not cost the headline
            result['roofline_error'] = f'{type(e).__name__}: {e}'[:300]
            log(f'roofline pass failed: {e}')
    log('roofline pass done')
    if rank == 0 and not a.no_cpu_baseline and world == 1:
        try:
            result['cpu_baseline'] = cpu_baseline(a.config, c, a.batch, tier=a.tier)
            result['speedup_vs_cpu_baseline'] = round(result['value'] / result['cpu_baseline']['value'], 1)
        except Exception as e:                       # noqa: BLE001
            result['cpu_baseline'] = {'error': f'{type(e).__name__}: {e}'[:300]}
    if rank == 0 and a.scaling == 'strong':
        # what ONE GPU takes for the whole global batch, measured here beside the sharded figure: at 8 / 6 samples per
        # GPU configs 4 / 5 are launch-count-bound, and a scaling record read without this number would be misread
        try:
            result['strong_scaling'] = one_gpu_full_batch(a, c, device, log)
            result['strong_scaling']['speedup_over_one_gpu_full_batch'] = round(
                result['strong_scaling']['ms_per_step'] / result['ms_per_step'], 3)
        except Exception as e:                       # noqa: BLE001
            result['strong_scaling'] = {'error': f'{type(e).__name__}: {e}'[:300]}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()          # ranks leave together (rank 0 ran the roofline pass)
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
