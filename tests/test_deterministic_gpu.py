"""-m gpu: BMNAS_DETERMINISTIC — two runs of the same search step give bit-identical results.

The reference's CPU path is run-to-run deterministic (aten's CPU reductions have a fixed order); the default HIP path
accumulates its batch reductions (BatchNorm sums, head logits, weight / BatchNorm-affine / LayerNorm-affine /
architecture gradients, the criterion) with fp32 atomics and is reproducible only to ~1e-6 ... 1e-3 of scale.
In deterministic mode every such reduction is a set of plain-store partials summed in a fixed order
(include/bmnas_hip.h, "deterministic mode"; bmnas.cell.DETERMINISTIC)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fusion_oracle as fo
from gpu_util import assert_close_scaled
from test_lazy_ln_gpu import _step


@pytest.fixture
def deterministic():
    from bmnas import cell as K
    if not K.FUSE_HEAD:
        pytest.skip('BMNAS_FUSE_HEAD=0 (switch matrix): the deterministic mode covers the fused-head path only')
    prev = K.DETERMINISTIC
    K.DETERMINISTIC = True
    K.apply_deterministic()
    yield K
    K.DETERMINISTIC = prev
    K.apply_deterministic()


@pytest.mark.parametrize('batch', [32, 128, 37])
@pytest.mark.parametrize('mode', ['train', 'train_nodrop'])
def test_two_runs_are_bit_identical(deterministic, batch, mode):
    cfg = fo.Cfg({**fo.CONFIGS['mmimdb'], 'drpt': 0.1 if mode == 'train' else 0.0})
    a = _step(cfg, batch, 11, 23, True, 'bce', mode)
    b = _step(cfg, batch, 11, 23, True, 'bce', mode)
    diff = [k for k in a if not torch.equal(a[k], b[k])]
    assert not diff, (len(diff), diff[:6])


@pytest.mark.parametrize('head', ['fused', 'deferred'])
@pytest.mark.parametrize('batch', [128, 100, 32])
def test_deterministic_step_matches_the_oracle_the_recorded_way(deterministic, batch, head):
    """in the default mode WHICH evaluation of the reference math a whole step matches (fp32,
    float64, float64 with one ReLU decision flipped) moves between runs with the atomics' order, so the table can only
    cap the flips.  In deterministic mode the outcome is reproducible: it is compared with the recorded kind exactly
    (gpu_util._check_deterministic_kind, tests/golden/match_step_table_det.json)."""
    from gpu_util import compare_search_step
    from test_network_gpu import _run_search_case
    cfg = fo.Cfg({**fo.CONFIGS['mmimdb'], 'drpt': 0.0})
    meta = dict(cfg=dict(cfg), seed=31, batch=batch, num_outputs=23, loss='bce', mode='train_nodrop', has_grads=True)
    net, cls, xs, feat, logits, loss = _run_search_case(meta, head)
    compare_search_step(cfg, batch, 23, 'bce', net, cls, [x.grad for x in xs], logits, loss, masks=None, seed=31,
                        label=f'det: mmimdb b{batch} head={head}', attn_drop=0.0)


def test_small_cell_with_three_steps_is_bit_identical(deterministic):
    cfg = fo.make_cfg(N=3, C=64, L=16, S=3, M=2, ns=1, nm=1, drpt=0.1)
    a = _step(cfg, 10, 5, 7, True, 'ce', 'train')
    b = _step(cfg, 10, 5, 7, True, 'ce', 'train')
    diff = [k for k in a if not torch.equal(a[k], b[k])]
    assert not diff, (len(diff), diff[:6])


def test_deterministic_mode_gives_the_same_numbers_as_the_default(deterministic):
    K = deterministic
    cfg = fo.Cfg({**fo.CONFIGS['mmimdb'], 'drpt': 0.0})
    det = _step(cfg, 32, 11, 23, True, 'bce', 'train_nodrop')
    K.DETERMINISTIC = False
    K.apply_deterministic()
    ref = _step(cfg, 32, 11, 23, True, 'bce', 'train_nodrop')
    for k in ref:
        if k.endswith('conv.bias'):
            continue                         # mathematically zero in front of a train-mode BatchNorm
        if k in ('logits', 'loss'):
            assert_close_scaled(k, det[k], ref[k], rel=2e-5)
            continue
        # (gradients: as in test_lazy_ln_gpu — the default run's ReLU decisions carry atomics-order noise)
        l2 = float((det[k].double() - ref[k].double()).norm() / ref[k].double().norm().clamp_min(1e-30))
        assert l2 <= 1e-2, (k, 'relative l2', l2)
        assert_close_scaled(k, det[k], ref[k], rel=0.3)


def test_configurations_outside_the_mode_are_refused(deterministic):
    from bmnas import lib
    cfg = fo.make_cfg(N=3, C=32, L=8, S=2, M=2, ns=2, nm=2, drpt=0.0)       # node_multiplier != 1
    with pytest.raises(lib.BmnasError, match='BMNAS_DETERMINISTIC'):
        _step(cfg, 6, 5, 7, True, 'ce', 'train_nodrop')
    # node_steps == 2 with node_multiplier == 1 takes the lazy-LayerNorm path, but its inner step's mix backward
    # accumulates with atomics from many workgroups: refused too, not silently non-reproducible
    cfg = fo.make_cfg(N=3, C=64, L=16, S=2, M=2, ns=2, nm=1, drpt=0.0)
    with pytest.raises(lib.BmnasError, match='BMNAS_DETERMINISTIC'):
        _step(cfg, 6, 5, 7, True, 'ce', 'train_nodrop')


def test_the_default_mode_is_not_bit_reproducible():
    """What the mode is for: with the atomic reductions two runs of the MM-IMDB b128 step differ in the last bits of
    most gradient tensors (printed, not required: an accidental match would not be a failure of anything)."""
    cfg = fo.Cfg({**fo.CONFIGS['mmimdb'], 'drpt': 0.0})
    a = _step(cfg, 128, 11, 23, True, 'bce', 'train_nodrop')
    b = _step(cfg, 128, 11, 23, True, 'bce', 'train_nodrop')
    diff = [k for k in a if not torch.equal(a[k], b[k])]
    worst = max((float((a[k] - b[k]).abs().max() / a[k].abs().max().clamp_min(1e-30)) for k in diff
                 if not k.endswith('conv.bias')), default=0.0)
    print(f'default mode: {len(diff)} of {len(a)} tensors differ between two runs; worst {worst:.1e} of scale')
