"""What the outer-loop goldens are built from, shared by the generator (tests/golden/make_golden_r04.py, which runs the
REFERENCE's trainers on the CPU) and the GPU test (tests/test_outer_loop_golden_gpu.py, which runs this repo's):
an in-memory deterministic MM-IMDB-shaped dataset, parameter-free stand-ins for the unimodal backbones (out of
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
