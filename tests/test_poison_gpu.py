"""-m gpu: stale-read check (was tools/poison_check.py, MM-IMDB only).

Every fresh float32 device allocation of the step — torch.empty / empty_like / new_empty, what the kernel sequencing
uses for outputs, saved tensors and gradient slots — is filled with NaN before use; then one search step (forward,
criterion, backward) runs at each dataset's configuration.  A kernel that reads a destination it was told to
overwrite (an accumulate flag set on a fresh slot, an old value fetched before an aliasing store, a node output that
the lazy-LayerNorm path "never writes" but somebody still reads) turns results into NaN; a zero-filled allocator
would hide exactly these.  Covers the streaming LayerNorm path and the per-sample one (MM-IMDB, node_multiplier 1),
the out_conv tails of NTU / Ego (node_multiplier 2 / 3), fused head and the reference composition, dropout on."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fusion_oracle as fo, synth
from gpu_util import build_search_net, dev


@pytest.fixture
def poisoned(monkeypatch):
    nan = float('nan')
    real_empty, real_like, real_new = torch.empty, torch.empty_like, torch.Tensor.new_empty
    count = [0]

    def fill(r):
        if torch.is_tensor(r) and r.is_cuda and r.dtype == torch.float32 and r.numel():
            r.fill_(nan)
            count[0] += 1
        return r

    monkeypatch.setattr(torch, 'empty', lambda *a, **k: fill(real_empty(*a, **k)))
    monkeypatch.setattr(torch, 'empty_like', lambda *a, **k: fill(real_like(*a, **k)))
    monkeypatch.setattr(torch.Tensor, 'new_empty', lambda self, *a, **k: fill(real_new(self, *a, **k)))
    yield count


def _step(name, batch, nout, kind, head, lazy, drpt):
    from bmnas import cell as K, nn as bnn
    cfg = fo.Cfg({**fo.CONFIGS[name], 'drpt': drpt})
    prev = K.LAZY_LN
    K.LAZY_LN = lazy
    try:
        net = build_search_net(cfg, 7, 'train')
        cls = (bnn.Linear if head else torch.nn.Linear)(cfg.M * cfg.C * cfg.L, nout).to(dev())
        cw, cb = synth.make_classifier(cfg, nout, 7)
        cls.weight.data.copy_(cw)
        cls.bias.data.copy_(cb)
        xs = [x.to(dev()).requires_grad_(True) for x in synth.make_inputs(cfg, batch, 7)]
        y = synth.make_labels(kind, batch, nout, 7).to(dev())
        if head:
            crit = bnn.BCEWithLogitsLoss() if kind == 'bce' else bnn.CrossEntropyLoss()
            with bnn.fused_criterion(head == 'deferred'):
                logits = net.forward_classified(xs, cls)
                loss = crit(logits, y)
        else:
            crit = torch.nn.BCEWithLogitsLoss() if kind == 'bce' else torch.nn.CrossEntropyLoss()
            logits = cls(net(xs))
            loss = crit(logits, y)
        loss.backward()
    finally:
        K.LAZY_LN = prev
    out = {'logits': logits.detach(), 'loss': loss.detach()}
    for i, a in enumerate(net.arch_parameters()):
        out[f'arch.{i}'] = a.grad
    for i, x in enumerate(xs):
        out[f'input.{i}'] = x.grad
    for n, p in net.named_parameters():
        out['p.' + n] = p.grad
    for n, v in net.state_dict().items():
        if fo.is_buffer(n):
            out['buf.' + n] = v
    out['cls.w'], out['cls.b'] = cls.weight.grad, cls.bias.grad
    torch.cuda.synchronize()
    return out


CASES = [('mmimdb', 128, 23, 'bce'), ('mmimdb', 37, 23, 'bce'), ('ntu', 8, 60, 'ce'), ('ntu', 64, 60, 'ce'),
         ('ego', 6, 83, 'ce'), ('ego', 48, 83, 'ce')]


@pytest.mark.parametrize('head', [None, 'fused', 'deferred'])
@pytest.mark.parametrize('name,batch,nout,kind', CASES)
def test_no_kernel_reads_a_fresh_allocation(poisoned, name, batch, nout, kind, head):
    from bmnas import cell as K
    if head and not K.FUSE_HEAD:
        pytest.skip('BMNAS_FUSE_HEAD=0')
    for lazy in ((True, False) if name == 'mmimdb' and head else (True,)):
        out = _step(name, batch, nout, kind, head, lazy, 0.1)
        assert poisoned[0] > 0                                   # the poison did reach the step's allocations
        bad = [k for k, v in out.items() if v is not None and not torch.isfinite(v.float()).all()]
        assert not bad, (name, batch, head, 'lazy' if lazy else 'per-sample', bad[:8], len(bad))
