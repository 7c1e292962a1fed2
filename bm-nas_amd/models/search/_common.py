"""Shared pieces of the per-dataset search drivers (the reference repeats them three times in
{mmimdb,ntu,ego}_darts_searchable.py).  Not part of the reference's public surface."""
import torch
import torch.nn as nn
import torch.optim as op
from bmnas.optim import Adam          # torch.optim.Adam semantics, one HIP launch per step

import models.auxiliary.scheduler as sc
from bmnas import dist as bdist
from bmnas import nn as bnn

from .darts.architect import Architect
from .darts.model import Found_FusionNetwork
from .darts.model_search import FusionNetwork
from .plot_genotype import Plotter


def parallel_flag(args):
    """The reference's library reads args.parallel while two of its mains define
    --use_dataparallel; accept either."""
    return bool(getattr(args, 'parallel', getattr(args, 'use_dataparallel', False)))


def data_parallel_world(args=None):
    """Number of data-parallel replicas.  The reference turns nn.DataParallel on with
    `args.parallel` when several GPUs are visible (mmimdb_darts_searchable.py:36); here the
    replicas are processes, so the launch itself is the request: WORLD_SIZE > 1 (torchrun /
    torch.distributed.run) means data parallel, whatever the flag says — N independent full
    searches writing the same checkpoint directory are never what the launch meant."""
    return bdist.env_world()


class HyperNetBase(nn.Module):
    """backbones (set by the subclass) -> reshape_layers -> fusion_net -> central_classifier.
    Attribute names are the reference's: trainers reach into .reshape_layers / .fusion_net."""

    param_group_order = ('reshape_layers', 'fusion_net', 'central_classifier')

    def _build_head(self, args, criterion, reshape_layers, num_input_nodes, num_keep_edges,
                    genotype=None, logger=None):
        self.args = args
        self.criterion = criterion
        self._criterion = criterion
        self.reshape_layers = reshape_layers
        self.multiplier = args.multiplier
        self.steps = args.steps
        self.parallel = parallel_flag(args)
        self.num_input_nodes = num_input_nodes
        self.num_keep_edges = num_keep_edges
        if genotype is None:
            self.fusion_net = FusionNetwork(steps=self.steps, multiplier=self.multiplier,
                                            num_input_nodes=num_input_nodes, num_keep_edges=num_keep_edges,
                                            args=args, criterion=criterion, logger=logger)
        else:
            self._genotype = genotype
            self.fusion_net = Found_FusionNetwork(steps=self.steps, multiplier=self.multiplier,
                                                  num_input_nodes=num_input_nodes,
                                                  num_keep_edges=num_keep_edges, args=args,
                                                  criterion=criterion, genotype=genotype)
        # nn.Linear subclass (same parameters / state_dict keys) running on the MFMA kernels
        self.central_classifier = bnn.Linear(args.C * args.L * self.multiplier, args.num_outputs)

    @staticmethod
    def make_reshape_layers(layer_cls, C_ins, args, genotype=None):
        """One reshape layer per backbone feature; a found net keeps only those its genotype
        uses (the others become parameter-free nn.ReLU placeholders, as in the reference)."""
        used = None if genotype is None else {e[1] for e in genotype.edges}
        layers = nn.ModuleList()
        for i, c_in in enumerate(C_ins):
            if used is None or i in used:
                layers.append(layer_cls(c_in, args.C, args.L, args))
            else:
                layers.append(nn.ReLU())
        return layers

    def reshape_input_features(self, input_features):
        # = [layer(f) for layer, f in zip(self.reshape_layers, input_features)], the conv stacks of all
        # modalities in one grouped set of launches (models.auxiliary.aux_models.reshape_all)
        import models.auxiliary.aux_models as aux
        return aux.reshape_all(list(self.reshape_layers), list(input_features))

    def fuse(self, raw_features):
        feats = self.reshape_input_features(list(raw_features))
        if hasattr(self.fusion_net, 'forward_classified'):        # search hypernet: K7 + classifier fused
            return self.fusion_net.forward_classified(feats, self.central_classifier)
        return self.central_classifier(self.fusion_net(feats))

    def genotype(self):
        if hasattr(self, '_genotype'):
            return self._genotype
        return self.fusion_net.genotype()

    def central_params(self):
        return [{'params': getattr(self, name).parameters()} for name in self.param_group_order]

    def _loss(self, input_features, labels):
        return self._criterion(self(input_features), labels)

    def arch_parameters(self):
        return self.fusion_net.arch_parameters()


def search_setup(model, args, criterion, device, num_batches_per_epoch, weight_decay):
    """Optimizers, scheduler, device placement, data parallelism and the Architect — the body of
    the reference's train_darts_model between model construction and the trainer call
    (mmimdb_darts_searchable.py:26-40).  nn.DataParallel is replaced by per-process replicas
    whose optimizers average gradients over RCCL right before step() (bmnas.dist)."""
    optimizer = Adam(model.central_params(), lr=args.eta_max, weight_decay=weight_decay)
    scheduler = sc.LRCosineAnnealingScheduler(args.eta_max, args.eta_min, args.Ti, args.Tm,
                                              num_batches_per_epoch)
    arch_optimizer = Adam(model.arch_parameters(), lr=args.arch_learning_rate, betas=(0.5, 0.999),
                             weight_decay=args.arch_weight_decay)
    if data_parallel_world(args) > 1:
        # one process per GPU: the unchanged mains pass cuda:0 to every rank
        # (main_darts_searchable_mmimdb.py:86), so the rank's own device comes from LOCAL_RANK,
        # BEFORE the model moves; the trainer loops follow the model's device (_loop.run)
        _, local, _ = bdist.init_from_env()
        if torch.device(device).type == 'cuda':
            device = torch.device('cuda', local)
            torch.cuda.set_device(device)
    model.to(device)
    if data_parallel_world(args) > 1:
        bdist.broadcast_state(model, model.arch_parameters())
        bdist.attach(optimizer)
        bdist.attach(arch_optimizer)
    architect = Architect(model, args, criterion, arch_optimizer)
    return optimizer, scheduler, architect, Plotter(args)
