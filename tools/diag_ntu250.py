import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/bm-nas_amd'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from oracle import fusion_oracle as fo, synth
from gpu_util import build_search_net, dev
name, batch = sys.argv[1], int(sys.argv[2]); nout, loss_kind = (23, "bce") if name == "mmimdb" else (60, "ce")
cfg = fo.Cfg({**fo.CONFIGS[name], 'drpt': 0.0}); seed = 31
net = build_search_net(cfg, seed, 'train_nodrop')
cls = torch.nn.Linear(cfg.M*cfg.C*cfg.L, nout)
cw, cb = synth.make_classifier(cfg, nout, seed); cls.weight.data.copy_(cw); cls.bias.data.copy_(cb); cls.to(dev())
xs = [x.to(dev()).requires_grad_(True) for x in synth.make_inputs(cfg, batch, seed)]
y = synth.make_labels(loss_kind, batch, nout, seed).to(dev())
loss = (torch.nn.BCEWithLogitsLoss() if loss_kind == "bce" else torch.nn.CrossEntropyLoss())(cls(net(xs)), y); loss.backward()
ol, olo, og = fo.search_step(synth.make_inputs(cfg, batch, seed), synth.make_labels(loss_kind, batch, nout, seed),
    synth.make_arch(cfg, seed), synth.make_params(cfg, seed), cw, cb, cfg, loss_kind, training=True, attn_drop=0.0)
# float64 oracle for reference
p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in synth.make_params(cfg, seed).items()}
_, _, og64 = fo.search_step([x.double() for x in synth.make_inputs(cfg, batch, seed)], synth.make_labels(loss_kind, batch, nout, seed),
    [a.double() for a in synth.make_arch(cfg, seed)], p64, cw.double(), cb.double(), cfg, loss_kind, training=True, attn_drop=0.0)
for k, v in net.named_parameters():
    if k.endswith('conv.bias'): continue
    g = v.grad.detach().cpu().double().numpy(); w = og[k].double().numpy(); w64 = og64[k].numpy()
    sc = np.abs(w64).max()
    e_gpu = np.abs(g - w64).max() / sc; e_cpu = np.abs(w - w64).max() / sc
    print(f'{k:60s} scale {sc:.3e}  gpu-vs-f64 {e_gpu:.2e}  cpu32-vs-f64 {e_cpu:.2e}')
