"""-m gpu: every HIP kernel of libbmnas_hip.so (through the C ABI) against the CPU oracle
on the same seeded inputs.  Tolerance: 1e-4 of the expected tensor's scale (fp32)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fusion_oracle as fo
from gpu_util import Args, assert_close_scaled, dev, set_mode

pytestmark = pytest.mark.gpu


def _rand(gen, *shape):
    return torch.from_numpy(gen.standard_normal(shape).astype(np.float32))


def _gen(seed):
    return np.random.Generator(np.random.PCG64(seed))


def test_library_loads_and_reports_version():
    from bmnas import lib
    assert lib.version() >= 100


@pytest.mark.parametrize('n_in,shape', [(1, (3, 16, 8)), (2, (5, 16, 8)), (6, (7, 192, 16)),
                                        (9, (4, 128, 8)), (16, (2, 16, 4))])
def test_mixsum_fwd_bwd(n_in, shape):
    from bmnas.functions import MixSumFn
    g = _gen(n_in)
    xs = [_rand(g, *shape) for _ in range(n_in)]
    W = torch.softmax(_rand(g, n_in, 2), -1)
    go = _rand(g, *shape)
    # oracle
    xo = [x.clone().requires_grad_(True) for x in xs]
    Wo = W.clone().requires_grad_(True)
    ref = fo.mixed_edge_sum(xo, Wo, 0)
    ref.backward(go)
    # HIP
    xd = [x.to(dev()).requires_grad_(True) for x in xs]
    Wd = W.to(dev()).requires_grad_(True)
    out = MixSumFn.apply(Wd[:, 1], *xd)
    out.backward(go.to(dev()))
    assert_close_scaled('out', out, ref)
    for j in range(n_in):
        assert_close_scaled(f'dx{j}', xd[j].grad, xo[j].grad)
    assert_close_scaled('dW[:,1]', Wd.grad[:, 1], Wo.grad[:, 1])
    assert float(Wd.grad[:, 0].abs().max()) == 0.0        # 'none' weight: exactly zero gradient


@pytest.mark.parametrize('n_src,b,C,L,relu,resid', [(1, 3, 16, 8, False, True), (2, 5, 192, 16, True, False),
                                                     (3, 2, 16, 16, True, False), (1, 4, 128, 8, False, True),
                                                     (2, 1, 512, 16, True, False)])
def test_cat_ln_fwd_bwd(n_src, b, C, L, relu, resid):
    from bmnas.functions import CatLnFn
    g = _gen(100 + n_src + C)
    srcs = [_rand(g, b, C, L) for _ in range(n_src)]
    res = _rand(g, b, C, L) if resid else None
    w = 1 + 0.1 * _rand(g, n_src * C, L)
    bb = 0.1 * _rand(g, n_src * C, L)
    go = _rand(g, b, n_src * C, L)

    def run(tensors, device, fn):
        ts = [t.detach().clone().to(device).requires_grad_(True) for t in tensors]
        out = fn(ts)
        out.backward(go.to(device))
        return out, [t.grad for t in ts]

    tensors = srcs + [w, bb] + ([res] if resid else [])

    def oracle(ts):
        x = torch.cat(ts[:n_src], 1)
        if resid:
            x = x + ts[-1]
        y = F.layer_norm(x, (n_src * C, L), ts[n_src], ts[n_src + 1], fo.EPS)
        return F.relu(y) if relu else y

    def hip(ts):
        return CatLnFn.apply(relu, ts[n_src], ts[n_src + 1], ts[-1] if resid else None, *ts[:n_src])

    ro, rg = run(tensors, 'cpu', oracle)
    ho, hg = run(tensors, dev(), hip)
    assert_close_scaled('out', ho, ro)
    for i, (a, b_) in enumerate(zip(hg, rg)):
        assert_close_scaled(f'grad{i}', a, b_)


@pytest.mark.parametrize('b,C,L,same', [(4, 16, 8, False), (5, 16, 8, True), (3, 32, 16, False),
                                        (7, 48, 4, False), (2, 192, 16, True), (9, 128, 8, False),
                                        (1, 16, 16, True)])
def test_sdpa_ln_fwd_bwd(b, C, L, same):
    from bmnas.functions import SdpaLnFn
    g = _gen(200 + b + C + L)
    x = _rand(g, b, C, L)
    y = x if same else _rand(g, b, C, L)
    w = 1 + 0.1 * _rand(g, C, L)
    bb = 0.1 * _rand(g, C, L)
    go = _rand(g, b, C, L)

    def run(device, fn):
        xd = x.detach().clone().to(device).requires_grad_(True)
        yd = xd if same else y.detach().clone().to(device).requires_grad_(True)
        wd = w.detach().clone().to(device).requires_grad_(True)
        bd = bb.detach().clone().to(device).requires_grad_(True)
        out = fn(xd, yd, wd, bd)
        out.backward(go.to(device))
        return out, [xd.grad, yd.grad, wd.grad, bd.grad]

    ro, rg = run('cpu', lambda a, b_, c, d: fo.op_scaled_dot_attn(a, b_, c, d, False))
    ho, hg = run(dev(), lambda a, b_, c, d: SdpaLnFn.apply(a, b_, c, d, 0.1, False))
    assert_close_scaled('out', ho, ro)
    for n, a, b_ in zip(['dx', 'dy', 'dln_w', 'dln_b'], hg, rg):
        assert_close_scaled(n, a, b_, rel=2e-4)


def _copy_params(dst_module, src_dict, prefix=''):
    sd = dst_module.state_dict()
    for k in sd:
        sd[k] = src_dict[prefix + k].clone()
    dst_module.load_state_dict(sd)


@pytest.mark.parametrize('kind', ['glu', 'fc'])
@pytest.mark.parametrize('b,C,L,training', [(4, 16, 8, True), (5, 16, 8, False), (3, 32, 16, True),
                                            (6, 48, 4, True), (8, 192, 16, True), (9, 128, 8, True)])
def test_conv_bn_act_modules(kind, b, C, L, training):
    """LinearGLU / ConcatFC (x != y): conv GEMM + BN statistics + activation, fwd + bwd,
    including the running-statistics update."""
    from models.search.darts.node_operations import ConcatFC, LinearGLU
    from oracle import synth
    cfg = fo.make_cfg(N=2, C=C, L=L, drpt=0.0)
    g = _gen(300 + b + C + L)
    M = 2 * C if kind == 'glu' else C
    shapes = {'conv.weight': (M, 2 * C, 1), 'conv.bias': (M,), 'bn.weight': (M,), 'bn.bias': (M,),
              'bn.running_mean': (M,), 'bn.running_var': (M,), 'bn.num_batches_tracked': ()}
    p = synth.make_params(cfg, 5 + b, shapes)
    x, y, go = _rand(g, b, C, L), _rand(g, b, C, L), _rand(g, b, C, L)
    # oracle
    po = {k: (v.clone() if fo.is_buffer(k) else v.clone().requires_grad_(True)) for k, v in p.items()}
    xo, yo = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    fn = fo.op_linear_glu if kind == 'glu' else fo.op_concat_fc
    pref = {'op.' + k: v for k, v in po.items()}
    ref = fn(xo, yo, pref, 'op', training, 0.0)
    ref.backward(go)
    # HIP module
    mod = (LinearGLU if kind == 'glu' else ConcatFC)(C, Args(cfg, 0.0))
    _copy_params(mod, p)
    mod.to(dev())
    mod.train(training)
    xd, yd = x.to(dev()).requires_grad_(True), y.to(dev()).requires_grad_(True)
    out = mod(xd, yd)
    out.backward(go.to(dev()))
    assert_close_scaled('out', out, ref)
    assert_close_scaled('dx', xd.grad, xo.grad)
    assert_close_scaled('dy', yd.grad, yo.grad)
    assert_close_scaled('dconv.weight', mod.conv.weight.grad, po['conv.weight'].grad)
    assert_close_scaled('dbn.weight', mod.bn.weight.grad, po['bn.weight'].grad)
    assert_close_scaled('dbn.bias', mod.bn.bias.grad, po['bn.bias'].grad)
    if training:   # conv bias in front of train-mode BN: zero gradient up to round-off
        assert float(mod.conv.bias.grad.abs().max()) < 1e-4
        assert_close_scaled('running_mean', mod.bn.running_mean, po['bn.running_mean'])
        assert_close_scaled('running_var', mod.bn.running_var, po['bn.running_var'])
        assert int(mod.bn.num_batches_tracked) == 1
    else:
        assert_close_scaled('dconv.bias', mod.conv.bias.grad, po['conv.bias'].grad)


@pytest.mark.parametrize('b,C,L,same,training', [(4, 16, 8, True, True), (5, 16, 8, False, True),
                                                 (3, 32, 16, True, False), (6, 192, 16, True, True),
                                                 (7, 128, 8, False, True)])
def test_node_mixed_op(b, C, L, same, training):
    from models.search.darts.node_operations import NodeMixedOp
    from oracle import synth
    cfg = fo.make_cfg(N=2, C=C, L=L, S=1, M=1, ns=1, nm=1, drpt=0.0)
    g = _gen(400 + b + C)
    prefix = 'cell._step_nodes.0.node_cell.node_ops.0._ops'
    full = synth.make_params(cfg, 11)
    p = {k: v for k, v in full.items() if k.startswith(prefix)}
    x = _rand(g, b, C, L)
    y = x if same else _rand(g, b, C, L)
    gam = torch.softmax(_rand(g, 4), -1)
    go = _rand(g, b, C, L)
    # oracle
    po = {k: (v.clone() if fo.is_buffer(k) else v.clone().requires_grad_(True)) for k, v in p.items()}
    xo = x.clone().requires_grad_(True)
    yo = xo if same else y.clone().requires_grad_(True)
    go_ = gam.clone().requires_grad_(True)
    ref = fo.node_mixed_op(xo, yo, go_, po, prefix, training, 0.0, attn_drop=0.0)
    ref.backward(go)
    # HIP
    op = NodeMixedOp(C, L, Args(cfg, 0.0))
    _copy_params(op, p, prefix[:-len('_ops')])
    op.to(dev())
    set_mode(op, 'train_nodrop' if training else 'eval')
    xd = x.to(dev()).requires_grad_(True)
    yd = xd if same else y.to(dev()).requires_grad_(True)
    gd = gam.to(dev()).requires_grad_(True)
    out = op(xd, yd, gd)
    out.backward(go.to(dev()))
    assert_close_scaled('out', out, ref)
    assert_close_scaled('dgamma', gd.grad, go_.grad)
    assert_close_scaled('dx', xd.grad, xo.grad, rel=2e-4)
    if not same:
        assert_close_scaled('dy', yd.grad, yo.grad, rel=2e-4)
    for k, v in op.named_parameters():
        want = po[prefix[:-len('_ops')] + k].grad
        if k.endswith('conv.bias') and training:
            assert float(v.grad.abs().max()) < 1e-4
        else:
            assert_close_scaled('d' + k, v.grad, want, rel=2e-4)
    if training:
        for k, v in op.state_dict().items():
            if k.endswith('num_batches_tracked'):
                assert int(v) == 1            # (the oracle bumps the counter one level up, in node_cell)
            elif fo.is_buffer(k):
                assert_close_scaled(k, v.float(), po[prefix[:-len('_ops')] + k].float())


def test_dropout_keep_fraction_and_fresh_masks():
    """Philox dropout statistics only: zero fraction ~ dead-ReLU + p of the rest, a new mask on every call,
    identity in eval mode.  (That kept values are scaled by 1/(1-p) and that the backward regenerates exactly
    the forward's mask is checked against the oracle under exported masks in tests/test_dropout_gpu.py.)"""
    from models.search.darts.node_operations import ConcatFC
    cfg = fo.make_cfg(N=2, C=64, L=16, drpt=0.25)
    torch.manual_seed(0)
    mod = ConcatFC(64, Args(cfg, 0.25)).to(dev()).train()
    x = torch.randn(64, 64, 16, device=dev(), requires_grad=True)
    y = torch.randn(64, 64, 16, device=dev())
    out = mod(x, y)
    dead_relu = 0.5                    # BN output is ~zero-mean: about half is clipped by ReLU
    zero_frac = float((out == 0).float().mean())
    assert abs(zero_frac - (dead_relu + (1 - dead_relu) * 0.25)) < 0.03, zero_frac
    out2 = mod(x, y)
    assert not torch.equal(out == 0, out2 == 0)            # a new mask every call
    # eval mode: identity dropout
    mod.eval()
    oe = mod(x, y)
    assert abs(float((oe == 0).float().mean()) - dead_relu) < 0.03


@pytest.mark.parametrize('b,O,K', [(128, 23, 6144), (7, 60, 2048), (5, 83, 2048), (3, 5, 256), (33, 128, 64)])
def test_linear_classifier_fwd_bwd(b, O, K):
    from bmnas import nn as bnn
    g = _gen(500 + b + O)
    x, go = _rand(g, b, K), _rand(g, b, O)
    ref = torch.nn.Linear(K, O)
    mod = bnn.Linear(K, O)
    mod.load_state_dict(ref.state_dict())
    mod.to(dev())
    xo = x.clone().requires_grad_(True)
    ro = ref(xo)
    ro.backward(go)
    xd = x.to(dev()).requires_grad_(True)
    out = mod(xd)
    out.backward(go.to(dev()))
    assert_close_scaled('out', out, ro)
    assert_close_scaled('dx', xd.grad, xo.grad)
    assert_close_scaled('dW', mod.weight.grad, ref.weight.grad)
    assert_close_scaled('db', mod.bias.grad, ref.bias.grad)
    assert list(mod.state_dict()) == list(ref.state_dict())


@pytest.mark.parametrize('training', [True, False])
@pytest.mark.parametrize('b,C,L,nm,p', [(8, 128, 8, 2, 0.2), (6, 128, 8, 3, 0.0), (7, 64, 16, 2, 0.3), (5, 128, 4, 2, 0.1),
                                        (3, 192, 8, 4, 0.25), (16, 256, 8, 2, 0.0)])
def test_mix_as_producer_of_out_conv(b, C, L, nm, p, training):
    """bmnas_node_mix_conv_fwd (the last inner step's NodeMixedOp combine inside the out_conv launch, small grids)
    against the two launches it replaces — bmnas_node_mix_fwd, then bmnas_conv1x1_fwd over cat(states, mix) — on the
    same inputs, dropout masks, BatchNorm batch sums and running statistics: the mix output bit for bit (same
    arithmetic, element by element), the conv output and its batch sums to fp32 round-off (another summation
    order), the mixed op's finalised BatchNorm (chan, running statistics, num_batches_tracked) bit for bit."""
    from bmnas import lib
    from bmnas import cell as K
    assert lib.node_mix_conv_fwd_ok(b, C, L, nm - 1)
    g = _gen(900 + b + C + L + nm)
    d = dev()
    z = _rand(g, b, C, L).to(d)
    p1 = _rand(g, b, C, L).to(d)
    prev = [_rand(g, b, C, L).to(d) for _ in range(nm - 1)]
    Wm = (_rand(g, 3 * C, C) / C ** 0.5).to(d)
    bm = _rand(g, 3 * C).to(d)
    bn_w, bn_b = (1 + 0.1 * _rand(g, 3 * C)).to(d), (0.1 * _rand(g, 3 * C)).to(d)
    Wo = (_rand(g, C, nm * C) / (nm * C) ** 0.5).to(d)
    bo = _rand(g, C).to(d)
    gamma = torch.softmax(_rand(g, 4), 0).to(d)
    shards = K.STAT_SHARDS
    # the producer GEMM of the mixed op, with its batch sums
    U = torch.empty(b, 3 * C, L, device=d)
    stat = torch.zeros(shards * 3 * C * 2, device=d) if training else None
    lib.conv1x1_fwd([z], C, Wm, C, bm, U, stat, b, L, 3 * C, stat_shards=shards if training else 0)
    dg = lib.make_dropout(p, 1234, 0, None) if (training and p > 0) else lib.NO_DROP
    df = lib.make_dropout(p, 1234, 10 ** 6, None) if (training and p > 0) else lib.NO_DROP

    def run(fused):
        rm, rv = (0.1 * _rand(_gen(5), 3 * C)).to(d), (1 + 0.1 * _rand(_gen(6), 3 * C).abs()).to(d)
        nbt = torch.full((2,), 7, dtype=torch.int64, device=d)
        fin = lib.make_bn_fin(stat, shards if training else 0, bm, bn_w, bn_b, rm, rv, nbt, training)
        chan = torch.zeros(4 * 3 * C, device=d)
        mix = torch.empty(b, C, L, device=d)
        V = torch.empty(b, C, L, device=d)
        ostat = torch.zeros(shards * C * 2, device=d) if training else None
        osh = shards if training else 0
        if fused:
            lib.node_mix_conv_fwd(z, z, p1, U, chan, gamma, mix, dg, df, fin, prev, Wo, nm * C, bo, V, ostat, osh,
                                  b, C, L)
        else:
            lib.node_mix_fwd(z, z, p1, U, chan, gamma, mix, b, C, L, dg, df, fin)
            lib.conv1x1_fwd(prev + [mix], C, Wo, nm * C, bo, V, ostat, b, L, C, stat_shards=osh)
        torch.cuda.synchronize()
        return mix, V, ostat, chan, rm, rv, nbt

    a, r = run(True), run(False)
    assert torch.equal(a[0], r[0]), float((a[0] - r[0]).abs().max())
    assert_close_scaled('V', a[1], r[1], rel=1e-5)
    if training:
        sa, sr = a[2].view(shards, C, 2).sum(0), r[2].view(shards, C, 2).sum(0)
        assert_close_scaled('out_conv batch sums', sa, sr, rel=1e-5)
    for name, x_, y_ in zip(('chan', 'running_mean', 'running_var', 'num_batches_tracked'), a[3:], r[3:]):
        assert torch.equal(x_, y_), name
    # and against torch, for the pair as a whole
    ch = r[3].view(4, 3 * C)
    Un = U * ch[2][None, :, None] + ch[3][None, :, None]
    ones = torch.ones(b * C * L, device=d)
    m2 = ones if dg.thr == 0 else lib.dropout_mask(dg, b * C * L, d)
    m3 = ones if df.thr == 0 else lib.dropout_mask(df, b * C * L, d)
    glu = Un[:, :C] * torch.sigmoid(Un[:, C:2 * C]) * m2.view(b, C, L)
    fc = torch.relu(Un[:, 2 * C:]) * m3.view(b, C, L)
    s_ref = gamma[0] * (z + z) + gamma[1] * p1 + gamma[2] * glu + gamma[3] * fc
    assert_close_scaled('mix vs torch', a[0], s_ref, rel=1e-5)
    V_ref = torch.einsum('jk,bkl->bjl', Wo.double(), torch.cat(prev + [s_ref], 1).double()) + bo.double()[None, :, None]
    assert_close_scaled('V vs torch', a[1], V_ref.float(), rel=1e-5)


@pytest.mark.parametrize('b,O,K', [(128, 23, 6144), (8, 60, 2048), (6, 83, 2048)])
def test_linear_classifier_under_graph_replay(b, O, K):
    """The classifier that is NOT fused into the cell's tail (BMNAS_FUSE_HEAD=0, a cell that concatenates an input
    state) inside a captured step: its split-K forward adds into a zero-filled output.  The fill used to be a
    hipMemsetAsync; as a hipGraph memset node it cleared the buffer on the first replay only and left 1e21-sized
    values on later ones (ROCm 7.2; tools/memset_node_probe.py), which every replay after the first then returned
    as logits.  Every replay must reproduce the eager result."""
    from bmnas import nn as bnn
    from bmnas.graph import GraphedStep
    g = _gen(540 + b + O)
    mod = bnn.Linear(K, O).to(dev())
    x = _rand(g, b, K).to(dev()).requires_grad_(True)
    y = torch.from_numpy(g.integers(0, O, size=(b,))).to(dev())
    crit = bnn.CrossEntropyLoss()
    tg = [x, mod.weight, mod.bias]

    def fn():
        z = mod(x)
        loss = crit(z, y)
        return (loss, z, *torch.autograd.grad(loss, tg))

    want = [t.detach().clone() for t in fn()]
    step = GraphedStep(fn, warmup=1)
    for replay in range(4):
        got = step.replay()
        torch.cuda.synchronize()
        for name, a, w in zip(('loss', 'logits', 'dx', 'dW', 'db'), got, want):
            assert_close_scaled(f'replay {replay} {name}', a, w, rel=1e-5)


@pytest.mark.parametrize('b,O', [(128, 23), (5, 7), (1, 3)])
def test_bce_with_logits_loss(b, O):
    from bmnas import nn as bnn
    g = _gen(600 + b)
    z = 3 * _rand(g, b, O)
    y = (torch.from_numpy(g.uniform(size=(b, O))) < 0.3).float()
    zo = z.clone().requires_grad_(True)
    lo = torch.nn.BCEWithLogitsLoss()(zo, y)
    (lo * 1.7).backward()
    zd = z.to(dev()).requires_grad_(True)
    ld = bnn.BCEWithLogitsLoss()(zd, y.to(dev()))
    (ld * 1.7).backward()
    assert ld.shape == lo.shape
    assert_close_scaled('loss', ld, lo)
    assert_close_scaled('dz', zd.grad, zo.grad)


@pytest.mark.parametrize('b,O', [(64, 60), (48, 83), (5, 3), (1, 100), (256, 60), (257, 83), (1000, 7)])   # <= 256: one launch
def test_cross_entropy_loss(b, O):
    from bmnas import nn as bnn
    g = _gen(700 + b)
    z = 3 * _rand(g, b, O)
    y = torch.from_numpy(g.integers(0, O, size=(b,)).astype(np.int64))
    zo = z.clone().requires_grad_(True)
    lo = torch.nn.CrossEntropyLoss()(zo, y)
    lo.backward()
    zd = z.to(dev()).requires_grad_(True)
    ld = bnn.CrossEntropyLoss()(zd, y.to(dev()))
    ld.backward()
    assert_close_scaled('loss', ld, lo)
    assert_close_scaled('dz', zd.grad, zo.grad)


@pytest.mark.parametrize('b,C,L,M,bias,stats', [(128, 192, 16, 576, True, True), (100, 192, 16, 576, False, True),
                                              (250, 128, 8, 384, True, True), (509, 64, 4, 208, True, False),
                                              (128, 192, 16, 80, True, True), (400, 96, 4, 288, True, True)])
def test_conv1x1_fwd_large_single_source(b, C, L, M, bias, stats):
    """The production-size forward GEMM through the C ABI: U = W x + bias and the per-16-column
    BatchNorm partials (sum, centred second moment), against float64 on the CPU.  Ragged batches
    and output-channel counts that are not a multiple of the workgroup tile included."""
    from bmnas import lib
    g = _gen(900 + b + C)
    x = _rand(g, b, C, L)
    W = _rand(g, M, C) * 0.1
    bv = _rand(g, M)
    xd, Wd, bd = x.to(dev()), W.to(dev()), bv.to(dev())
    U = torch.full((b, M, L), float('nan'), device=dev())
    n_part = lib.conv1x1_num_partials(b, L)
    part = torch.full((M * n_part * 2,), float('nan'), device=dev()) if stats else None
    lib.conv1x1_fwd([xd], C, Wd, C, bd if bias else None, U, part, b, L, M)
    ref = torch.einsum('mc,bcl->bml', W.double(), x.double())
    if bias:
        ref = ref + bv.double()[None, :, None]
    assert_close_scaled('U', U, ref.float(), rel=2e-5)
    if stats:
        cols = ref.permute(1, 0, 2).reshape(M, b * L)                  # (M, n) in (sample, l) order
        got = part.view(M, n_part, 2).cpu().double()
        for gi in (0, n_part // 2, n_part - 1):
            seg = cols[:, 16 * gi:16 * gi + 16]
            assert_close_scaled(f'sum[{gi}]', got[:, gi, 0].float(), seg.sum(1).float(), rel=5e-5)
            m2 = ((seg - seg.mean(1, keepdim=True)) ** 2).sum(1)
            assert_close_scaled(f'm2[{gi}]', got[:, gi, 1].float(), m2.float(), rel=2e-4)


@pytest.mark.parametrize('b,C,L,n_prev,have_g,alias', [(8, 128, 8, 2, True, True), (6, 128, 8, 3, False, True),
                                                       (64, 128, 8, 2, True, False), (5, 192, 16, 4, True, False),
                                                       (3, 16, 4, 1, False, False), (100, 64, 8, 5, True, True)])
def test_node_mix_next_sum_matches_separate_launches(b, C, L, n_prev, have_g, alias):
    """bmnas_node_mix_fwd_next / _bwd_next (the next inner step's mixed sum inside the mix launch,
    node_search.py:52-57) against the separate launches they replace (bmnas_node_mix_fwd + bmnas_mixsum_fwd;
    bmnas_mixsum_bwd + bmnas_node_mix_bwd), which the oracle tests pin."""
    from bmnas import lib
    g = _gen(9900 + b + C + n_prev)
    d = dev()
    x, p1, U = _rand(g, b, C, L).to(d), _rand(g, b, C, L).to(d), _rand(g, b, 3 * C, L).to(d)
    gamma = torch.softmax(_rand(g, 4), 0).to(d)
    M = 3 * C
    Ud = U.double()
    mean = Ud.mean(dim=(0, 2))
    rstd = 1.0 / torch.sqrt(Ud.var(dim=(0, 2), unbiased=False) + 1e-5)
    bn_w, bn_b = (_rand(g, M) * 0.3 + 1.0).to(d).double(), (_rand(g, M) * 0.2).to(d).double()
    scale = rstd * bn_w
    chan = torch.cat([mean, rstd, scale, bn_b - mean * scale]).float()
    prev = [_rand(g, b, C, L).to(d) for _ in range(n_prev)]
    if alias and n_prev >= 2:
        prev[1] = prev[0]                       # search mode: x is y
    w = torch.softmax(_rand(g, n_prev + 1, 2), -1).to(d)          # (k, 2): column 1 is the edge weight
    nd = lib.NO_DROP
    # forward
    s0, z0 = torch.empty_like(x), torch.empty_like(x)
    lib.node_mix_fwd(x, x, p1, U, chan, gamma, s0, b, C, L, nd, nd)
    lib.mixsum_fwd(prev + [s0], w[:, 1], 2, z0)
    s1, z1 = torch.empty_like(x), torch.empty_like(x)
    lib.node_mix_fwd(x, x, p1, U, chan, gamma, s1, b, C, L, nd, nd, lib.NO_FIN, (prev, w[:, 1], 2, z1))
    assert torch.equal(s0, s1)
    assert_close_scaled('z_next', z1, z0.cpu(), rel=2e-6)
    # backward
    gz, gz2 = _rand(g, b, C, L).to(d), _rand(g, b, C, L).to(d)
    g_prev = _rand(g, b, C, L).to(d) if have_g else None
    old = [_rand(g, b, C, L).to(d) for _ in range(n_prev)]
    if alias and n_prev >= 2:
        old[1] = old[0]
    acc = 0b10101 & ((1 << n_prev) - 1)
    if alias and n_prev >= 2:
        acc |= 2                                 # the second writer of an aliased destination accumulates

    def run(fused):
        dst = [o.clone() for o in old]
        if alias and n_prev >= 2:
            dst[1] = dst[0]
        dw = torch.zeros(n_prev + 1, 2, device=d)
        dgam = torch.zeros(4, device=d)
        dx, dV, bn_grad = torch.empty_like(x), torch.empty(b, M, L, device=d), torch.zeros(2 * M, device=d)
        gs = g_prev.clone() if have_g else torch.empty_like(x)
        if fused:
            lib.node_mix_bwd(gs if have_g else None, x, x, p1, U, chan, gamma, dgam, dx, None, 0, dV, bn_grad, b, C,
                             L, nd, nd, 1, 0, (prev, dst, acc, w[:, 1], 2, dw[:, 1], 1, 0, s0, gz, gz2, gs))
        else:
            sd = torch.empty_like(x)
            lib.mixsum_bwd(prev + [s0], dst + [gs if have_g else sd], w[:, 1], 2, gz, dw[:, 1],
                           acc | ((1 << n_prev) if have_g else 0), 1, 0, gz2)
            gs = gs if have_g else sd
            lib.node_mix_bwd(gs, x, x, p1, U, chan, gamma, dgam, dx, None, 0, dV, bn_grad, b, C, L, nd, nd)
        torch.cuda.synchronize()
        return dict(g=gs, dx=dx, dV=dV, bn_grad=bn_grad, dgamma=dgam, dw=dw,
                    **{f'dprev{j}': t for j, t in enumerate(dst)})

    want, got = run(False), run(True)
    for k in want:
        assert_close_scaled(k, got[k], want[k].cpu(), rel=1e-5)


@pytest.mark.parametrize('b,C,L,n_prev,drop', [(8, 128, 8, 8, True), (6, 128, 8, 9, False), (64, 128, 8, 4, True),
                                               (3, 16, 4, 1, False), (128, 64, 16, 15, True), (5, 32, 16, 2, False)])
def test_bn_relu_ln_fwd_with_next_pair_sum(b, C, L, n_prev, drop):
    """bmnas_bn_relu_ln_fwd_pair (NodeCell tail + the next cell step's K1 pair sum, model_search.py:58 +
    node_search.py:54) against the two launches it replaces: bmnas_bn_relu_ln_fwd + bmnas_mixsum_pair_fwd —
    identical node outputs and, same arithmetic order, identical sums."""
    from bmnas import lib
    assert lib.bn_relu_ln_fwd_pair_ok(b, C, L, n_prev)
    assert not lib.bn_relu_ln_fwd_pair_ok(129, C, L, n_prev) and not lib.bn_relu_ln_fwd_pair_ok(b, 512, 16, n_prev)
    g = _gen(6100 + b + C + n_prev)
    d = dev()
    U, x = _rand(g, b, C, L).to(d), _rand(g, b, C, L).to(d)
    ln_w, ln_b = (_rand(g, C, L) * 0.3 + 1.0).to(d), (_rand(g, C, L) * 0.2).to(d)
    Ud = U.double()
    mean = Ud.mean(dim=(0, 2))
    rstd = 1.0 / torch.sqrt(Ud.var(dim=(0, 2), unbiased=False) + 1e-5)
    bn_w, bn_b = (_rand(g, C) * 0.3 + 1.0).to(d).double(), (_rand(g, C) * 0.2).to(d).double()
    scale = rstd * bn_w
    chan = torch.cat([mean, rstd, scale, bn_b - mean * scale]).float()
    prev = [_rand(g, b, C, L).to(d) for _ in range(n_prev)]
    w = torch.softmax(_rand(g, n_prev + 1, 2), -1).to(d)
    w2 = torch.softmax(_rand(g, 3, 2), -1).to(d)
    dcfg = lib.make_dropout(0.15, 77, 0) if drop else lib.NO_DROP

    def run(fused):
        o, out = torch.empty_like(x), torch.empty_like(x)
        stats, osum = torch.empty(b, 2, device=d), torch.empty(b, 2, device=d)
        h, z = torch.empty_like(x), torch.empty_like(x)
        if fused:
            lib.bn_relu_ln_fwd(U, chan, x, ln_w, ln_b, o, out, stats, b, C, L, dcfg, lib.NO_FIN, osum,
                               (prev, w[:, 1], 2, w2[:, 1], 2, h, z))
        else:
            lib.bn_relu_ln_fwd(U, chan, x, ln_w, ln_b, o, out, stats, b, C, L, dcfg, lib.NO_FIN, osum)
            lib.mixsum_pair_fwd(prev + [out], w[:, 1], 2, w2[:, 1], 2, h, z)
        torch.cuda.synchronize()
        return dict(o=o, out=out, stats=stats, osum=osum, h=h, z=z)

    want, got = run(False), run(True)
    for k in want:
        assert torch.equal(got[k], want[k]), k


@pytest.mark.parametrize('b,C,L,n_prev,have_g,have_gh,have_gz2,acc,drop',
                         [(8, 128, 8, 8, True, True, True, 0b10110101, True), (6, 128, 8, 9, False, False, False, 0, False),
                          (64, 128, 8, 4, True, False, True, 0b1111, True), (3, 16, 4, 1, False, True, False, 1, False),
                          (128, 64, 16, 14, True, True, False, 0x2aaa, True), (5, 32, 16, 2, True, True, True, 0, False)])
def test_bn_relu_ln_bwd_with_next_pair_backward(b, C, L, n_prev, have_g, have_gh, have_gz2, acc, drop):
    """bmnas_bn_relu_ln_bwd_pair (the next cell step's K1 pair backward + the NodeCell tail backward in one
    launch; model_search.py:58 and node_search.py:64-69 backwards) against bmnas_mixsum_pair_bwd +
    bmnas_bn_relu_ln_bwd, which the oracle tests pin: destinations (overwriting / accumulating / absent), the
    completed node-output gradient, arch-weight gradients over shards, dV, BatchNorm reductions, residual."""
    from bmnas import lib
    g = _gen(8200 + b + C + n_prev)
    d = dev()
    U, x, o = _rand(g, b, C, L).to(d), _rand(g, b, C, L).to(d), _rand(g, b, C, L).to(d)
    ln_w = (_rand(g, C, L) * 0.3 + 1.0).to(d)
    Ud = U.double()
    mean = Ud.mean(dim=(0, 2))
    rstd = 1.0 / torch.sqrt(Ud.var(dim=(0, 2), unbiased=False) + 1e-5)
    bn_w, bn_b = (_rand(g, C) * 0.3 + 1.0).to(d).double(), (_rand(g, C) * 0.2).to(d).double()
    scale = rstd * bn_w
    chan = torch.cat([mean, rstd, scale, bn_b - mean * scale]).float()
    pre = (o + x).double()
    stats = torch.stack([pre.mean(dim=(1, 2)), 1.0 / torch.sqrt(pre.var(dim=(1, 2), unbiased=False) + 1e-5)],
                        dim=1).float().contiguous()
    prev = [_rand(g, b, C, L).to(d) for _ in range(n_prev)]
    out_fwd = _rand(g, b, C, L).to(d)                       # the node's forward output (the sum's last input)
    h = _rand(g, b, C, L).to(d)
    w = torch.softmax(_rand(g, n_prev + 1, 2), -1).to(d)
    w2 = torch.softmax(_rand(g, 3, 2), -1).to(d)
    gz, gz2, gh = _rand(g, b, C, L).to(d), _rand(g, b, C, L).to(d), _rand(g, b, C, L).to(d)
    g_part = _rand(g, b, C, L).to(d)
    old = [_rand(g, b, C, L).to(d) for _ in range(n_prev)]
    old_res = _rand(g, b, C, L).to(d)
    skip = {1} if n_prev > 2 else set()                     # one destination absent
    dcfg = lib.make_dropout(0.15, 99, 0) if drop else lib.NO_DROP
    shards, stride = 4, 64

    def run(fused):
        dst = [None if j in skip else old[j].clone() for j in range(n_prev)]
        dw = torch.zeros(shards * stride, device=d)
        dw2 = torch.zeros(shards * stride, device=d)
        dV, bn_grad, dres = torch.empty_like(x), torch.zeros(2 * C, device=d), old_res.clone()
        gfull = g_part.clone() if have_g else torch.empty_like(x)
        if fused:
            lib.bn_relu_ln_bwd(gfull if have_g else None, o, x, ln_w, stats, U, chan, dV, bn_grad, dres, 1, b, C, L,
                               dcfg, (prev, dst, acc, out_fwd, w[:, 1], 2, w2[:, 1], 2, h, gh if have_gh else None,
                                      gz, gz2 if have_gz2 else None, dw, dw2, shards, stride, gfull))
        else:
            lib.mixsum_pair_bwd(prev + [out_fwd], dst + [gfull], w[:, 1], 2, w2[:, 1], 2, h,
                                gh if have_gh else None, gz, dw, dw2, acc | ((1 << n_prev) if have_g else 0),
                                shards, stride, gz2 if have_gz2 else None)
            lib.bn_relu_ln_bwd(gfull, o, x, ln_w, stats, U, chan, dV, bn_grad, dres, 1, b, C, L, dcfg)
        torch.cuda.synchronize()
        res = dict(g_full=gfull, dV=dV, bn_grad=bn_grad, dresid=dres,
                   dw=dw.view(shards, stride).sum(0)[:2 * (n_prev + 1)], dw2=dw2.view(shards, stride).sum(0)[:4])
        for j, t in enumerate(dst):
            if t is not None:
                res[f'dx{j}'] = t
        return res

    want, got = run(False), run(True)
    for k in want:
        assert_close_scaled(k, got[k], want[k].cpu(), rel=2e-5)


@pytest.mark.parametrize('b,C,L,racc,xacc,drop,same', [(128, 192, 16, 0, 0, True, True), (8, 128, 8, 1, 1, False, True),
                                                         (5, 32, 16, 0, 1, True, True), (3, 16, 4, 1, 0, False, False),
                                                         (100, 256, 16, 0, 0, True, True), (7, 512, 16, 1, 1, True, False),
                                                         (1, 64, 8, 0, 0, False, True)])
def test_node_mix_ln_bwd_matches_separate_launches(b, C, L, racc, xacc, drop, same):
    """bmnas_node_mix_ln_bwd (K6 LayerNorm backward + K2 mix backward in one launch, two workgroups per
    sample; node_search.py:55,67-68 backwards) against bmnas_cat_ln_bwd + bmnas_node_mix_bwd, which the
    oracle tests pin.  All three instantiations (C*L/4 <= 512, <= 1024, <= 2048), dropout on and off,
    accumulating and overwriting destinations, x is y and x != y."""
    from bmnas import lib
    assert lib.node_mix_ln_bwd_ok(b, C, L)
    assert not lib.node_mix_ln_bwd_ok(129, C, L) and not lib.node_mix_ln_bwd_ok(4, 1024, 16)
    g = _gen(4300 + b + C + L)
    d = dev()
    x, p1, U = _rand(g, b, C, L).to(d), _rand(g, b, C, L).to(d), _rand(g, b, 3 * C, L).to(d)
    y = x if same else _rand(g, b, C, L).to(d)
    pre, gy = (_rand(g, b, C, L) * 1.5 + 0.2).to(d), _rand(g, b, C, L).to(d)
    ln_w, ln_b = (_rand(g, C, L) * 0.3 + 1.0).to(d), (_rand(g, C, L) * 0.2).to(d)
    gamma = torch.softmax(_rand(g, 4), 0).to(d)
    M = 3 * C
    Ud = U.double()
    mean = Ud.mean(dim=(0, 2))
    rstd = 1.0 / torch.sqrt(Ud.var(dim=(0, 2), unbiased=False) + 1e-5)
    bn_w, bn_b = (_rand(g, M) * 0.3 + 1.0).to(d).double(), (_rand(g, M) * 0.2).to(d).double()
    scale = rstd * bn_w
    chan = torch.cat([mean, rstd, scale, bn_b - mean * scale]).float()
    pd = pre.double()
    smean = pd.mean(dim=(1, 2))
    srstd = 1.0 / torch.sqrt(pd.var(dim=(1, 2), unbiased=False) + 1e-5)
    stats = torch.stack([smean, srstd], dim=1).float().contiguous()
    dglu = lib.make_dropout(0.1, 1234, 0) if drop else lib.NO_DROP
    dfc = lib.make_dropout(0.2, 1234, b * C * L // 4) if drop else lib.NO_DROP
    old_r, old_x, old_y = _rand(g, b, C, L).to(d), _rand(g, b, C, L).to(d), _rand(g, b, C, L).to(d)

    def run(fused):
        dres, dx, dy = old_r.clone(), old_x.clone(), (None if same else old_y.clone())
        gin = torch.empty_like(x)
        dgam = torch.zeros(4, device=d)
        dV, bn_grad = torch.empty(b, M, L, device=d), torch.zeros(2 * M, device=d)
        acc = xacc | (0 if same else (xacc << 1))
        if fused:
            lib.node_mix_ln_bwd(gy, pre, ln_w, stats, gin, dres, racc, x, y, p1, U, chan, gamma, dgam, dx, dy, acc,
                                dV, bn_grad, b, C, L, dglu, dfc)
        else:
            lib.cat_ln_bwd(gy, [pre], None, ln_w, ln_b, stats, [gin], dres, racc << 31, None, None, b, C, L, False)
            lib.node_mix_bwd(gin, x, y, p1, U, chan, gamma, dgam, dx, dy, acc, dV, bn_grad, b, C, L, dglu, dfc)
        torch.cuda.synchronize()
        out = dict(g_in=gin, dresid=dres, dx=dx, dV=dV, bn_grad=bn_grad, dgamma=dgam)
        if dy is not None:
            out['dy'] = dy
        return out

    want, got = run(False), run(True)
    for k in want:
        assert_close_scaled(k, got[k], want[k].cpu(), rel=2e-5)
    if drop:                                                    # the masks really were on, and identical
        assert float((want['dV'] == 0).float().mean()) > 0.05


@pytest.mark.parametrize('b,C,L,acc,sums', [(8, 128, 8, 0, True), (6, 128, 8, 1, False), (64, 128, 8, 1, True),
                                            (5, 192, 16, 0, True), (3, 16, 4, 1, False), (100, 256, 16, 0, True)])
def test_bn_relu_ln_tail(b, C, L, acc, sums):
    """bmnas_bn_relu_ln_fwd / _bwd (NodeCell tail, node_search.py:64-69, dropout off) against
    float64 autograd: o = relu(bn(U)), out = LN(o + x); dV, the BatchNorm reductions, the residual gradient."""
    from bmnas import lib
    g = _gen(7700 + b + C)
    U, x, gy = _rand(g, b, C, L), _rand(g, b, C, L), _rand(g, b, C, L)
    ln_w, ln_b = _rand(g, C, L) * 0.3 + 1.0, _rand(g, C, L) * 0.2
    bn_w, bn_b = _rand(g, C) * 0.3 + 1.0, _rand(g, C) * 0.2
    prev = _rand(g, b, C, L)
    Ud = U.double()
    mean = Ud.mean(dim=(0, 2))
    rstd = 1.0 / torch.sqrt(Ud.var(dim=(0, 2), unbiased=False) + 1e-5)
    scale = rstd * bn_w.double()
    shift = bn_b.double() - mean * scale
    V = (Ud * scale[None, :, None] + shift[None, :, None]).requires_grad_(True)
    xd = x.double().requires_grad_(True)
    o_ref = torch.relu(V)
    out_ref = F.layer_norm(o_ref + xd, [C, L], ln_w.double(), ln_b.double(), 1e-5)
    out_ref.backward(gy.double())
    chan = torch.cat([mean, rstd, scale, shift]).float().to(dev())
    o, out = torch.empty(b, C, L, device=dev()), torch.empty(b, C, L, device=dev())
    stats = torch.empty(b, 2, device=dev())
    osum = torch.empty(b, 2, device=dev()) if sums else None
    nodrop = lib.NO_DROP
    lib.bn_relu_ln_fwd(U.to(dev()), chan, x.to(dev()), ln_w.to(dev()), ln_b.to(dev()), o, out, stats, b, C, L,
                       nodrop, lib.NO_FIN, osum)
    assert_close_scaled('o', o, o_ref.detach().float(), rel=2e-5)
    assert_close_scaled('out', out, out_ref.detach().float(), rel=5e-5)
    if sums:
        want = torch.stack([out_ref.detach().sum(dim=(1, 2)), (out_ref.detach() ** 2).sum(dim=(1, 2))], dim=1)
        assert_close_scaled('out_sums', osum, want.float(), rel=1e-4)
    dV = torch.empty(b, C, L, device=dev())
    bn_grad = torch.zeros(2 * C, device=dev())
    dres = prev.clone().to(dev())
    lib.bn_relu_ln_bwd(gy.to(dev()), o, x.to(dev()), ln_w.to(dev()), stats, U.to(dev()), chan, dV, bn_grad, dres,
                       acc, b, C, L, nodrop)
    assert_close_scaled('dV', dV, V.grad.float(), rel=1e-4)
    want_res = xd.grad + (prev.double() if acc else 0.0)
    assert_close_scaled('dresid', dres, want_res.float(), rel=1e-4)
    xhat = (Ud - mean[None, :, None]) * rstd[None, :, None]
    want_bn = torch.cat([(V.grad * xhat).sum(dim=(0, 2)), V.grad.sum(dim=(0, 2))])
    assert_close_scaled('bn_grad', bn_grad, want_bn.float(), rel=1e-4)


@pytest.mark.parametrize('b,C,L,M,n_src,training,acc', [(8, 128, 8, 128, 2, True, 0), (6, 128, 8, 128, 2, True, 1),
                                                        (64, 128, 8, 128, 2, True, 2), (48, 128, 8, 128, 3, False, 0),
                                                        (7, 64, 4, 64, 2, True, 3), (5, 192, 16, 192, 2, True, 0),
                                                        (128, 192, 16, 192, 2, True, 1), (250, 128, 8, 128, 2, True, 0),
                                                        (128, 512, 16, 192, 1, True, -1), (64, 2048, 8, 128, 1, True, 0),
                                                        (6, 2048, 8, 128, 1, True, -1)])
def test_conv1x1_bwd_all_pair(b, C, L, M, n_src, training, acc):
    """bmnas_conv1x1_bwd_all (out_conv + bn backward, node_search.py:63-66): BatchNorm input gradient,
    data gradient and weight / bias gradient; merged launch at the small grids, three launches otherwise."""
    from bmnas import lib
    g = _gen(4100 + b + C + n_src)
    dV, U = _rand(g, b, M, L), _rand(g, b, M, L) * 1.5 + 0.3
    W = _rand(g, M, n_src * C) * 0.1
    bn_w = _rand(g, M) * 0.3 + 1.0
    srcs = [_rand(g, b, C, L) for _ in range(n_src)]
    prev = [_rand(g, b, C, L) for _ in range(n_src)]
    Ud, dVd = U.double(), dV.double()
    N = b * L
    if training:
        mean = Ud.mean(dim=(0, 2))
        rstd = 1.0 / torch.sqrt(Ud.var(dim=(0, 2), unbiased=False) + 1e-5)
    else:
        mean = _rand(g, M).double() * 0.2
        rstd = 1.0 / torch.sqrt(_rand(g, M).double().abs() + 0.5)
    xhat = (Ud - mean[None, :, None]) * rstd[None, :, None]
    scale = rstd * bn_w.double()
    s_dx, s_d = (dVd * xhat).sum(dim=(0, 2)), dVd.sum(dim=(0, 2))
    if training:
        dU = scale[None, :, None] * (dVd - s_d[None, :, None] / N - xhat * s_dx[None, :, None] / N)
    else:
        dU = scale[None, :, None] * dVd
    chan = torch.cat([mean, rstd, scale, torch.zeros(M, dtype=torch.float64)]).float().to(dev())
    bn_grad = torch.cat([s_dx, s_d]).float().to(dev())
    want_data = acc >= 0                        # acc = -1: no data gradient asked for (frozen inputs)
    acc = max(acc, 0)
    dst = [p.clone().to(dev()) if want_data else None for p in prev]
    dW0, db0 = _rand(g, M, n_src * C), _rand(g, M)
    dW, db = dW0.clone().to(dev()), db0.clone().to(dev())
    dV_dev = dV.to(dev())
    before = lib.conv_family_calls(reset=True)
    lib.conv1x1_bwd_all(dV_dev, W.to(dev()), n_src * C, dst, C, acc, b, L, M, 0, [x.to(dev()) for x in srcs], dW,
                        n_src * C, db, 0, (U.to(dev()), chan, bn_grad, training))
    fam = lib.conv_family_calls(reset=True)
    import os
    pipe_on = os.environ.get('BMNAS_CONV_PIPE', '1') != '0'
    ng, jt = (b * L + 15) // 16, n_src * C // 16
    pipe = pipe_on and M % 48 == 0 and ((ng + 1) // 2) * ((n_src * C + 63) // 64) >= 96
    merged = want_data and not pipe and ((ng + 1) // 2) * ((jt + 1) // 2) < 1024
    assert (fam['bwd_pair'] == 1) == merged, fam
    untouched = merged or pipe or not want_data         # else bn_bwd_apply ran in place first
    ref = torch.einsum('mc,bml->bcl', W.double(), dU)
    for q in range(n_src if want_data else 0):
        want = ref[:, q * C:(q + 1) * C]
        if acc & (1 << q):
            want = want + prev[q].double()
        assert_close_scaled(f'dsrc{q}', dst[q], want.float(), rel=5e-5)
    cat = torch.cat([x.double() for x in srcs], dim=1)
    assert_close_scaled('dW', dW, (dW0.double() + torch.einsum('bml,bkl->mk', dU, cat)).float(), rel=5e-5)
    assert_close_scaled('dbias', db, (db0.double() + dU.sum(dim=(0, 2))).float(), rel=5e-5)
    if untouched:
        assert torch.equal(dV_dev.cpu(), dV)           # the folding launches leave their dV operand alone
    else:
        assert_close_scaled('dU in place', dV_dev, dU.float(), rel=5e-5)


@pytest.mark.parametrize('b,C,L,M,n_dst,acc', [(128, 192, 16, 576, 1, 1), (100, 192, 16, 576, 1, 1),
                                              (128, 192, 16, 576, 1, 0), (250, 128, 8, 384, 2, 3),
                                              (509, 64, 4, 192, 1, 1), (128, 128, 16, 128, 2, 0),
                                              (100, 80, 16, 240, 1, 1), (77, 48, 8, 144, 3, 5)])
def test_conv1x1_bwd_data_large(b, C, L, M, n_dst, acc):
    """The production-size data-gradient GEMM through the C ABI, overwrite and accumulate,
    one and two destinations: d src_q (=|+=) W[:, qC:(q+1)C]^T dU."""
    from bmnas import lib
    g = _gen(950 + b + C)
    dU = _rand(g, b, M, L)
    W = _rand(g, M, n_dst * C) * 0.1
    prev = [_rand(g, b, C, L) for _ in range(n_dst)]
    dst = [p.clone().to(dev()) for p in prev]
    lib.conv1x1_bwd_data(dU.to(dev()), W.to(dev()), n_dst * C, dst, C, acc, b, L, M)
    ref = torch.einsum('mc,bml->bcl', W.double(), dU.double())
    for q in range(n_dst):
        want = ref[:, q * C:(q + 1) * C]
        if acc & (1 << q):
            want = want + prev[q].double()
        assert_close_scaled(f'dsrc{q}', dst[q], want.float(), rel=3e-5)


def test_copy_batch_one_launch_any_dtypes():
    """bmnas_copy_batch (the batch into a captured step's static tensors): features and labels of different dtypes,
    byte counts that are no multiple of 16, a source whose address is not 16-byte aligned, more tensors than one
    launch carries, an empty tensor — bit-equal to torch's copies, neighbours of the destinations untouched."""
    from bmnas import lib
    g = torch.Generator().manual_seed(3)
    shapes = [((128, 192, 16), torch.float32), ((64, 128, 8), torch.float32), ((6,), torch.int64), ((7, 23), torch.float32),
              ((0, 5), torch.float32), ((3,), torch.int32), ((1000003,), torch.uint8)] + [((5, 9, 4), torch.float32)] * 14
    pairs, guards = [], []
    for shp, dt in shapes:
        n = 1
        for s_ in shp:
            n *= s_
        src = (torch.randn(n + 8, generator=g) * 100).to(dt).to(dev())
        buf = torch.full((n + 16,), 7, dtype=dt, device=dev())
        pairs.append((buf[8:8 + n].view(shp), src[1:1 + n].view(shp)))      # src offset by ONE element: unaligned for most
        guards.append(buf)
    lib.copy_batch(pairs)
    torch.cuda.synchronize()
    for (d, s_), buf in zip(pairs, guards):
        assert torch.equal(d, s_), (d.shape, d.dtype)
        n = d.numel()
        assert bool((buf[:8] == 7).all()) and bool((buf[8 + n:] == 7).all()), (d.shape, d.dtype)


def test_copy_batch_blob_by_value():
    """The by-value blob of bmnas_copy_batch (a captured optimizer step's per-replay scalars ride in the launch that
    copies the batch: no H2D node in the graph): stored bit-exactly, with tensors and alone, neighbours untouched."""
    import numpy as np
    from bmnas import lib
    src = torch.arange(1000, dtype=torch.float32, device=dev())
    dst = torch.zeros(1000, dtype=torch.float32, device=dev())
    for nbytes, with_tensors in ((32, True), (256, True), (64, False), (4, False)):
        host = np.random.default_rng(nbytes).integers(0, 255, nbytes, dtype=np.uint8)
        buf = torch.full((nbytes + 32,), 9, dtype=torch.uint8, device=dev())
        dst.zero_()
        lib.copy_batch([(dst, src)] if with_tensors else [], blob=(buf[16:16 + nbytes], host.tobytes()))
        torch.cuda.synchronize()
        assert bool((buf[16:16 + nbytes].cpu() == torch.from_numpy(host)).all()), nbytes
        assert bool((buf[:16] == 9).all()) and bool((buf[16 + nbytes:] == 9).all()), nbytes
        assert torch.equal(dst, src) == with_tensors
    assert lib.copy_blob_max() == 256
    with pytest.raises(lib.BmnasError):
        lib.copy_batch([], blob=(torch.zeros(512, dtype=torch.uint8, device=dev()), bytes(260)))


def test_batch_copier_zero_jobs_and_counter_advance():
    """What rides in front of a captured per-op step's replay (bmnas.lib.BatchCopier): the batch copies, the zero-fill of
    the step's accumulation arena (a NULL source) and the advance of its dropout step counter — one launch."""
    from bmnas import lib
    srcs = [torch.randn(64, 32, 8, device=dev()), torch.arange(64, device=dev())]
    dsts = [torch.zeros_like(t) for t in srcs]
    arena = torch.full((1003,), 7.0, device=dev())          # no multiple of 4 floats: the byte tail is cleared too
    guard = torch.full((8,), 5.0, device=dev())
    counter = torch.full((1,), (1 << 60) + 5, dtype=torch.int64, device=dev())
    cp = lib.BatchCopier(dsts, zero=[arena], advance=(counter, 1234))
    for k in range(3):
        arena.fill_(7.0)
        assert cp(srcs) == []
        torch.cuda.synchronize()
        assert all(torch.equal(d, s_) for d, s_ in zip(dsts, srcs))
        assert float(arena.abs().max()) == 0.0 and bool((guard == 5.0).all())
        assert int(counter.item()) == (1 << 60) + 5 + 1234 * (k + 1)
    # the static tensors themselves as sources (nothing to copy): the zero-fill and the advance still happen
    arena.fill_(3.0)
    cp(dsts)
    torch.cuda.synchronize()
    assert float(arena.abs().max()) == 0.0 and int(counter.item()) == (1 << 60) + 5 + 1234 * 4


@pytest.mark.parametrize('n_in,n_more,shape,have_gh,have_gz2', [(6, 1, (16, 192, 16), True, True), (8, 0, (8, 128, 8), False, False),
                                                               (3, 2, (5, 16, 4), True, False), (15, 0, (2, 32, 8), False, True)])
def test_mixsum_pair_bwd_without_dot_products(n_in, n_more, shape, have_gh, have_gz2):
    """dw = dw2 = NULL (the weight step: nobody differentiates the edge weights): the input gradients are bit-equal to
    the launch that also forms the dot products; one of the two without the other is refused."""
    from bmnas import lib
    g = _gen(n_in * 7 + n_more)
    d = dev()
    xs = [_rand(g, *shape).to(d) for _ in range(n_in)]
    h, gz = _rand(g, *shape).to(d), _rand(g, *shape).to(d)
    gh = _rand(g, *shape).to(d) if have_gh else None
    gz2 = _rand(g, *shape).to(d) if have_gz2 else None
    aw = torch.softmax(_rand(g, n_in, 2), -1).to(d)
    bw = torch.softmax(_rand(g, 2, 2), -1).to(d)
    g_more = [_rand(g, *shape).to(d) for _ in range(n_more)]
    w_more = [torch.softmax(_rand(g, n_in, 2), -1).to(d) for _ in range(n_more)]

    def run(dots):
        dxs = [torch.full(shape, float('nan'), device=d) if j % 4 != 3 else None for j in range(n_in)]
        dxs[0] = torch.ones(shape, device=d)
        da, db = torch.zeros(n_in, 2, device=d), torch.zeros(2, 2, device=d)
        lib.mixsum_pair_bwd_x(xs, dxs, aw[:, 1], 2, bw[:, 1], 2, h, gh, gz, da[:, 1] if dots else None,
                              db[:, 1] if dots else None, 1, g_more, [w[:, 1] for w in w_more], 1, 0, gz2)
        torch.cuda.synchronize()
        return dxs, da, db

    (want, da, db), (got, _, _) = run(True), run(False)
    assert float(da.abs().sum()) > 0 and float(db.abs().sum()) > 0
    for j, (a, e) in enumerate(zip(got, want)):
        assert (a is None) == (e is None)
        if a is not None:
            assert torch.isfinite(a).all() and torch.equal(a, e), j
    with pytest.raises(lib.BmnasError):
        lib.mixsum_pair_bwd_x(xs, [None] * n_in, aw[:, 1], 2, bw[:, 1], 2, h, gh, gz, torch.zeros(n_in, 2, device=d)[:, 1],
                              None, 0, g_more, [w[:, 1] for w in w_more], 1, 0, gz2)
