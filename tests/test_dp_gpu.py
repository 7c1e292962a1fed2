"""-m gpu: data parallelism through the reference's call chain with the REAL HIP hypernet
(SURVEY.md section 8e).  Two fresh child processes (tests/helpers/dp_child.py) share the one GPU
of the box (BMNAS_FORCE_DEVICE=0) and talk over gloo (RCCL refuses two ranks on one device); what
is under test is everything around the collective: device placement from LOCAL_RANK although the
mains pass cuda:0, replica broadcast, minibatch sharding, the captured step writing into the flat
bucket, ONE all-reduce, identical Adam updates.

Parity rule (8e): the all-reduced gradients equal the CPU-oracle gradients computed shard by
shard with shared weights (per-shard BatchNorm statistics, like nn.DataParallel) and averaged."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from oracle import fusion_oracle as fo
from oracle import synth
from gpu_util import assert_close_scaled

pytestmark = pytest.mark.gpu
CHILD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers', 'dp_child.py')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(mode, out, world=2, timeout=900):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), BMNAS_FORCE_DEVICE='0', BMNAS_DIST_BACKEND='gloo',
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        env.pop('BMNAS_HIP_GRAPH', None)
        procs.append(subprocess.Popen([sys.executable, CHILD, mode, str(out)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    for r, (p, o) in enumerate(zip(procs, logs)):
        assert p.returncode == 0, f'rank {r} failed:\n{o[-4000:]}'
    return logs


def test_two_ranks_allreduced_grads_match_oracle_shard_by_shard(tmp_path):
    sys.path.insert(0, os.path.dirname(CHILD))
    import dp_child as ch
    _launch('grads', tmp_path)
    d0 = torch.load(tmp_path / 'grads_rank0.pt')
    d1 = torch.load(tmp_path / 'grads_rank1.pt')
    assert d0['device'] == d1['device'] == 'cuda:0'
    cfg = ch.cfg_small()
    X = synth.make_inputs(cfg, ch.GLOBAL_BATCH, ch.SEED)
    Y = synth.make_labels('bce', ch.GLOBAL_BATCH, ch.NOUT, ch.SEED)
    cw, cb = synth.make_classifier(cfg, ch.NOUT, ch.SEED)
    acc, world = None, 2
    per = ch.GLOBAL_BATCH // world
    for r in range(world):
        _, _, g = fo.search_step([x[r * per:(r + 1) * per] for x in X], Y[r * per:(r + 1) * per],
                                 synth.make_arch(cfg, ch.SEED, 0.5), synth.make_params(cfg, ch.SEED), cw, cb,
                                 cfg, 'bce', training=True, attn_drop=0.0)
        acc = g if acc is None else {k: acc[k] + v for k, v in g.items()}
    want = {k: v / world for k, v in acc.items()}
    checked = 0
    for k, v in d0.items():
        if k.startswith('wgrad:'):
            name = k[len('wgrad:'):]
            key = name[len('fusion_net.'):] if name.startswith('fusion_net.') else name
            if key.endswith('conv.bias') or key.endswith('out_conv.bias'):
                # zero in exact arithmetic (a bias in front of a train-mode BatchNorm)
                assert float(v.abs().max()) < 1e-4
            else:
                assert_close_scaled(k, v, want[key], rel=5e-4)
            checked += 1
        elif k.startswith('agrad:'):
            assert_close_scaled(k, v, want['arch.' + k.split(':')[1]], rel=5e-4)
            checked += 1
    assert checked > 20
    # replicas: identical reduced gradients, and bit-identical weights / arch tensors / BN-free state
    # after further steps (graph, eager ragged, graph)
    for k in d0:
        if k == 'device':
            continue
        if k.startswith('state:') and ('running_' in k or 'num_batches' in k):
            continue                        # BatchNorm statistics are per replica, as with DataParallel
        assert torch.equal(d0[k], d1[k]), k


def test_two_ranks_uneven_global_batch_matches_one_mean_over_all_samples(tmp_path):
    """nn.DataParallel scatters 7 samples over 2 replicas as 4 + 3 (Tensor.chunk) and its criterion takes ONE mean over the
    7 gathered outputs (mmimdb_darts_searchable.py:36-37, :114).  The loop's `_shard_batch` cuts the same slices and the
    reducer weights each rank's shard mean by n_rank * world / n: the all-reduced weight and architecture gradients equal
    the CPU oracle's shard gradients (per-replica BatchNorm statistics) combined with weights 4/7 and 3/7 — captured and
    eager steps alike."""
    sys.path.insert(0, os.path.dirname(CHILD))
    import dp_child as ch
    _launch('uneven', tmp_path)
    d0 = torch.load(tmp_path / 'uneven_rank0.pt')
    d1 = torch.load(tmp_path / 'uneven_rank1.pt')
    cfg = ch.cfg_small()
    n = ch.UNEVEN_BATCH
    X = [x[:n] for x in synth.make_inputs(cfg, ch.GLOBAL_BATCH, ch.SEED)]
    Y = synth.make_labels('bce', ch.GLOBAL_BATCH, ch.NOUT, ch.SEED)[:n]
    cw, cb = synth.make_classifier(cfg, ch.NOUT, ch.SEED)
    want = None
    for lo, hi in ((0, 4), (4, 7)):
        _, _, g = fo.search_step([x[lo:hi] for x in X], Y[lo:hi], synth.make_arch(cfg, ch.SEED, 0.5),
                                 synth.make_params(cfg, ch.SEED), cw, cb, cfg, 'bce', training=True, attn_drop=0.0)
        w = (hi - lo) / n
        g = {k: v for k, v in g.items() if not k.startswith('input.')}       # (per-sample gradients: a shard's own)
        want = {k: w * v for k, v in g.items()} if want is None else {k: want[k] + w * v for k, v in g.items()}
    checked = 0
    for k, v in d0.items():
        if k.startswith(('wgrad:', 'egrad:')):
            name = k.split(':', 1)[1]
            key = name[len('fusion_net.'):] if name.startswith('fusion_net.') else name
            if key.endswith('conv.bias') or key.endswith('out_conv.bias'):
                assert float(v.abs().max()) < 1e-4
            else:
                assert_close_scaled(k, v, want[key], rel=5e-4)
            checked += 1
        elif k.startswith('agrad:'):
            assert_close_scaled(k, v, want['arch.' + k.split(':')[1]], rel=5e-4)
            checked += 1
        assert torch.equal(v, d1[k]), k          # both ranks hold the same reduced gradients
    assert checked > 40


def test_two_ranks_run_the_reference_call_chain(tmp_path):
    _launch('driver', tmp_path)
    r0 = torch.load(tmp_path / 'driver_rank0.pt')
    r1 = torch.load(tmp_path / 'driver_rank1.pt')
    assert r0['genotype'] == r1['genotype']
    assert r0['best_f1'] == r1['best_f1']          # metrics are reduced over ranks
    assert 0.0 <= r0['best_f1'] <= 1.0
    assert 'best_model.pt' in r0['files'] and 'best_genotype.pkl' in r0['files']
    assert r1['files'] == []                        # only rank 0 writes checkpoints
    # 2 epochs x (2 full + 1 ragged) train batches; each rank sees half of every batch
    assert r0['stats']['graph_replays'] == 4 and r1['stats']['graph_replays'] == 4, (r0['stats'], r1['stats'])


def test_bench_two_ranks_weak_scaling_line(tmp_path):
    """bench.py's N > 1 path (what the driver launches with torch.distributed.run --nproc-per-node N):
    every rank captures its step writing into the flat bucket, one all-reduce per step, barrier +
    max-over-ranks timing, rank 0 prints ONE JSON line.  Two ranks on the one GPU of the box over
    gloo stand in for two GPUs over RCCL."""
    import json
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py')
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), BMNAS_FORCE_DEVICE='0', BMNAS_DIST_BACKEND='gloo',
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, bench, '--gpus', '2', '--steps', '5', '--warmup', '2',
                                       '--batch', '32', '--no-full-step'], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f'rank {r}:\n{e[-3000:]}'
    lines = [l for l in outs[0][0].splitlines() if l.startswith('{')]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith('{')]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 5 and d['scaling'] == 'weak' and d['value'] > 0
    assert d['config']['global_batch'] == 64 and d['config']['parallelism'] == 'dp2'
    assert abs(d['value'] - 2 * 5 / (d['ms_per_step'] * 5e-3)) / d['value'] < 1e-3
