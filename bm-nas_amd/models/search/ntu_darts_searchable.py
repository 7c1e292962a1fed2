"""NTU RGB+D search driver and hypernet wrappers (reference models/search/ntu_darts_searchable.py:
train_darts_model :21-72, Searchable_Skeleton_Image_Net :74-176, Found_Skeleton_Image_Net
:178-273).  The paper-ablation nets further down the reference file (:275-680) are out of scope."""
import os

import torch

import models.auxiliary.aux_models as aux
import models.search.train_searchable.ntu as tr

from bmnas import nn as bnn

from ._common import HyperNetBase, parallel_flag, search_setup

C_INS = [512, 1024, 2048, 2048, 128, 256, 1024, 512]


def train_darts_model(dataloaders, args, device, logger):
    dataset_sizes = {x: len(dataloaders[x].dataset) for x in ['train', 'dev', 'test']}
    num_batches_per_epoch = dataset_sizes['train'] / args.batchsize
    criterion = bnn.CrossEntropyLoss()          # torch criterion subclass on the HIP loss kernel
    model = Searchable_Skeleton_Image_Net(args, criterion, logger)
    model.skenet.load_state_dict(torch.load(os.path.join(args.checkpointdir, args.ske_cp)))
    model.rgbnet.load_state_dict(torch.load(os.path.join(args.checkpointdir, args.rgb_cp)))
    optimizer, scheduler, architect, plotter = search_setup(model, args, criterion, device,
                                                            num_batches_per_epoch, args.weight_decay)
    return tr.train_ntu_track_acc(model, architect, criterion, optimizer, scheduler, dataloaders,
                                  dataset_sizes, device=device, num_epochs=args.epochs,
                                  parallel=parallel_flag(args), logger=logger, plotter=plotter, args=args)


class _SkeletonImageBase(HyperNetBase):
    def _build_backbones(self, args):
        import models.central.ntu as ntu              # reference checkout (out of scope here)
        self.rgbnet = ntu.Visual(args)
        self.skenet = ntu.Skeleton(args)

    def forward(self, tensor_tuple):
        skeleton, image = tensor_tuple[1], tensor_tuple[0]
        visual_features = self.rgbnet(image)[-5:-1]
        skel_features, _ = self.skenet(skeleton)
        return self.fuse(list(visual_features) + list(skel_features[-4:]))


class Searchable_Skeleton_Image_Net(_SkeletonImageBase):
    # the reference leaves the reshape layers OUT of the optimised parameters here
    param_group_order = ('fusion_net', 'central_classifier')

    def __init__(self, args, criterion, logger):
        super().__init__()
        self.logger = logger
        self._build_backbones(args)
        self._build_head(args, criterion, self.create_reshape_layers(args), 8, 2, logger=logger)

    def create_reshape_layers(self, args):
        return self.make_reshape_layers(aux.ReshapeInputLayer, C_INS, args)


class Found_Skeleton_Image_Net(_SkeletonImageBase):
    def __init__(self, args, criterion, genotype):
        super().__init__()
        self._build_backbones(args)
        self._genotype = genotype
        self._build_head(args, criterion, self.create_reshape_layers(args), 8, 2, genotype=genotype)

    def create_reshape_layers(self, args):
        return self.make_reshape_layers(aux.ReshapeInputLayer, C_INS, args, self._genotype)
