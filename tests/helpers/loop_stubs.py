"""What the outer-loop goldens are built from, shared by the generator (tests/golden/make_golden_r04.py, which runs the
REFERENCE's trainers on the CPU) and the GPU test (tests/test_outer_loop_golden_gpu.py, which runs this repo's):
in-memory deterministic MM-IMDB-, NTU- and EgoGesture-shaped datasets, parameter-free stand-ins for the unimodal backbones (out of
scope) that are pure functions of the batch, a deterministic state for any model with the
reference's state_dict keys, and the recorder both sides fill.  Nothing here comes from the reference."""
import numpy as np
import torch
from torch.utils.data import Dataset

from oracle import synth

PHASES = {'train': 0, 'dev': 1, 'test': 2}


class MMIMDBData(Dataset):
    """{'image': (3, 16, 16), 'text': (300,), 'label': (23,)} — values from a counter-keyed numpy generator."""

    def __init__(self, n, seed):
        rng = np.random.Generator(np.random.PCG64(seed + 7000003))
        self.img = torch.from_numpy(rng.standard_normal((n, 3, 16, 16)).astype(np.float32))
        self.txt = torch.from_numpy(rng.standard_normal((n, 300)).astype(np.float32))
        # labels correlated with the text so that two epochs of training move the F1
        w = rng.standard_normal((300, 23)).astype(np.float32)
        score = self.txt.numpy() @ w / np.sqrt(300.0)
        self.lab = torch.from_numpy((score > 0.8).astype(np.float32))

    def __len__(self):
        return len(self.lab)

    def __getitem__(self, i):
        return {'image': self.img[i], 'text': self.txt[i], 'label': self.lab[i]}


class StubVGG(torch.nn.Module):
    """GP_VGG stand-in: four feature maps (b, 512, h, w) + a 23-vector, a fixed projection of the image."""

    def __init__(self, args):
        super().__init__()
        rng = np.random.Generator(np.random.PCG64(11))
        self.register_buffer('proj', torch.from_numpy((rng.standard_normal((3 * 16 * 16, 512)) / 16.0)
                                                      .astype(np.float32)), persistent=False)

    def forward(self, image):
        f = (image.flatten(1) @ self.proj).relu()
        mk = lambda h, w: f[:, :, None, None] * torch.linspace(0.5, 1.5, h * w, device=f.device).view(1, 1, h, w)
        return [mk(20, 32), mk(20, 32), mk(10, 16), mk(5, 8), f[:, :23]]


class StubMLP(torch.nn.Module):
    """MaxOut_MLP stand-in: slices of the text vector."""

    def __init__(self, args):
        super().__init__()

    def forward(self, text):
        return [text[:, :64].relu(), text[:, 100:228].relu(), text[:, :23]]


def _proj(seed, n_in, n_out):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((rng.standard_normal((n_in, n_out)) / np.sqrt(n_in)).astype(np.float32))


def _maps(f, shapes):
    """Feature maps (b, C_i, *spatial_i): channel c of map i is f[:, (c + 17 i) mod width] times a fixed ramp over the
    spatial positions — a pure function of f, no parameters."""
    outs = []
    for i, (C, sp) in enumerate(shapes):
        base = f.roll(-17 * i, dims=1)[:, :C]
        if not sp:
            outs.append(base)
            continue
        n = int(np.prod(sp))
        ramp = torch.linspace(0.5, 1.5, n, device=f.device).view(1, 1, *sp)
        outs.append(base.view(base.shape[0], C, *([1] * len(sp))) * ramp)
    return outs


NTU_CLASSES, EGO_CLASSES = 6, 5


class NTUData(Dataset):
    """{'rgb': (4, 6, 6, 3), 'ske': (3, 8, 5, 2), 'label': int64} — the keys train_ntu_track_acc reads
    (reference train_searchable/ntu.py:59); labels follow the skeleton so that two epochs move the accuracy."""

    def __init__(self, n, seed):
        rng = np.random.Generator(np.random.PCG64(seed + 7000019))
        self.rgb = torch.from_numpy(rng.standard_normal((n, 4, 6, 6, 3)).astype(np.float32))
        self.ske = torch.from_numpy(rng.standard_normal((n, 3, 8, 5, 2)).astype(np.float32))
        w = rng.standard_normal((240, NTU_CLASSES)).astype(np.float32)
        self.lab = torch.from_numpy((self.ske.numpy().reshape(n, -1) @ w).argmax(1).astype(np.int64))

    def __len__(self):
        return len(self.lab)

    def __getitem__(self, i):
        return {'rgb': self.rgb[i], 'ske': self.ske[i], 'label': self.lab[i]}


class StubVisual(torch.nn.Module):
    """models.central.ntu.Visual stand-in: six outputs, [-5:-1] are the visual features (512, 1024, 2048 channel maps and
    a pooled 2048-vector; reference ntu_darts_searchable.py:122-125)."""

    def __init__(self, args):
        super().__init__()
        self.nout = args.num_outputs
        self.register_buffer('proj', _proj(31, 4 * 6 * 6 * 3, 2048), persistent=False)

    def forward(self, image):
        f = (image.flatten(1) @ self.proj).relu()
        return [f[:, :64]] + _maps(f, [(512, (4, 3, 3)), (1024, (4, 2, 2)), (2048, (2, 2, 2)), (2048, ())]) + \
            [f[:, :self.nout]]


class StubSkeleton(torch.nn.Module):
    """models.central.ntu.Skeleton stand-in: (hidden list, logits); hidden[-4:] are the skeleton features (128 and 256
    channel maps, a 1024- and a 512-vector; reference ntu_darts_searchable.py:128-129)."""

    def __init__(self, args):
        super().__init__()
        self.nout = args.num_outputs
        self.register_buffer('proj', _proj(32, 3 * 8 * 5 * 2, 1024), persistent=False)

    def forward(self, ske):
        f = (ske.flatten(1) @ self.proj).relu()
        return [f[:, :8]] + _maps(f, [(128, (4, 5)), (256, (2, 3)), (1024, ()), (512, ())]), f[:, :self.nout]


class EgoData(Dataset):
    """(clip (4, 4, 6, 6), label): RGB in channels 0:3, depth in 3: (reference train_searchable/ego.py:61-64)."""

    def __init__(self, n, seed):
        rng = np.random.Generator(np.random.PCG64(seed + 7000033))
        self.clip = torch.from_numpy(rng.standard_normal((n, 4, 4, 6, 6)).astype(np.float32))
        w = rng.standard_normal((144, EGO_CLASSES)).astype(np.float32)
        self.lab = torch.from_numpy((self.clip[:, 3].numpy().reshape(n, -1) @ w).argmax(1).astype(np.int64))

    def __len__(self):
        return len(self.lab)

    def __getitem__(self, i):
        return self.clip[i], self.lab[i]


class StubEgoNet(torch.nn.Module):
    """models.central.ego.get_{rgb,depth}_model stand-in: net(x)[0:-1] are four feature maps of 512, 1024, 2048, 2048
    channels (reference ego_darts_searchable.py:122-131)."""

    def __init__(self, cin, salt):
        super().__init__()
        self.register_buffer('proj', _proj(40 + salt, cin * 4 * 6 * 6, 2048), persistent=False)

    def forward(self, x):
        f = (x.flatten(1) @ self.proj).relu()
        return _maps(f, [(512, (4, 3, 3)), (1024, (2, 2, 2)), (2048, (2, 2, 2)), (2048, (1, 1, 1))]) + [f[:, :EGO_CLASSES]]


def ego_rgb_model(opt):
    return StubEgoNet(3, 0)


def ego_depth_model(opt):
    return StubEgoNet(1, 1)


def fill_state(model, seed, arch_scale=0.05):
    """Deterministic parameters / buffers for every state_dict key (alphabetical key order, so that two
    implementations with the same keys get the same values whatever their construction order), the arch
    parameters likewise, and every nn.Dropout turned into an identity."""
    sd = model.state_dict()
    rng = np.random.Generator(np.random.PCG64(seed))
    new = {}
    for k in sorted(sd):
        shape = tuple(sd[k].shape)
        if k.endswith('central_classifier.weight'):
            v = (rng.uniform(-1.0, 1.0, shape) / np.sqrt(shape[1])).astype(np.float32)
        elif k.endswith('central_classifier.bias'):
            v = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        else:
            v = synth._fill(rng, k, shape)
        new[k] = torch.from_numpy(np.asarray(v)).reshape(shape)
    model.load_state_dict(new)
    try:
        arch = list(model.arch_parameters())
    except AttributeError:                     # a found network has no architecture parameters
        arch = []
    if arch:
        rng = np.random.Generator(np.random.PCG64(seed + 1000003))
        for p in arch:
            p.data.copy_(torch.from_numpy((arch_scale * rng.standard_normal(tuple(p.shape))).astype(np.float32)))
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return model


def summary(t, k=6):
    f = t.detach().double().reshape(-1).cpu()
    head = f[:k]
    if head.numel() < k:
        head = torch.cat([head, torch.zeros(k - head.numel(), dtype=torch.float64)])
    return [float(f.sum()), float(f.norm())] + [float(v) for v in head]


class Recorder:
    """batches: [epoch, phase, learn, loss, logits sum, logits l2, lr of the weight optimizer at that step (learn) or
    -1]; phases: [epoch, phase, epoch loss, epoch metric] + the genotype printed after the phase."""

    def __init__(self):
        self.batches, self.phases, self.genotypes = [], [], []

    def batch(self, epoch, phase, learn, loss, output, lr):
        s = summary(output, 0)
        self.batches.append([epoch, PHASES[phase], int(bool(learn)), float(loss), s[0], s[1], lr if learn else -1.0])

    def phase(self, epoch, phase, loss, metric, genotype):
        self.phases.append([epoch, PHASES[phase], float(loss), float(metric)])
        self.genotypes.append(str(genotype))
