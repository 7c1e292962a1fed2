"""EgoGesture search driver and hypernet wrappers (reference models/search/ego_darts_searchable.py:
train_darts_model :20-69, Searchable_RGB_Depth_Net :71-178, Found_RGB_Depth_Net :181-285)."""
import os

import torch

import models.auxiliary.aux_models as aux
import models.search.train_searchable.ego as tr

from bmnas import nn as bnn

from ._common import HyperNetBase, parallel_flag, search_setup

C_INS = [512, 1024, 2048, 2048, 512, 1024, 2048, 2048]


def train_darts_model(dataloaders, args, opt, device, logger):
    dataset_sizes = {x: len(dataloaders[x].dataset) for x in ['train', 'dev', 'test']}
    num_batches_per_epoch = dataset_sizes['train'] / args.batchsize
    criterion = bnn.CrossEntropyLoss()          # torch criterion subclass on the HIP loss kernel
    model = Searchable_RGB_Depth_Net(args, opt, criterion)
    rgb_path = os.path.join(args.checkpointdir, args.rgb_cp)
    depth_path = os.path.join(args.checkpointdir, args.depth_cp)
    model.rgb_net.load_state_dict(torch.load(rgb_path))
    logger.info("Loading rgb checkpoint: " + rgb_path)
    model.depth_net.load_state_dict(torch.load(depth_path))
    logger.info("Loading depth checkpoint: " + depth_path)
    optimizer, scheduler, architect, plotter = search_setup(model, args, criterion, device,
                                                            num_batches_per_epoch, 1e-4)
    return tr.train_ego_track_acc(model, architect, criterion, optimizer, scheduler, dataloaders,
                                  dataset_sizes, device=device, num_epochs=args.epochs,
                                  parallel=parallel_flag(args), logger=logger, plotter=plotter, args=args)


class _RGBDepthBase(HyperNetBase):
    param_group_order = ('fusion_net', 'central_classifier', 'reshape_layers')

    def _build_backbones(self, opt):
        import models.central.ego as ego              # reference checkout (out of scope here)
        self.opt = opt
        self.rgb_net = ego.get_rgb_model(opt)
        self.depth_net = ego.get_depth_model(opt)

    def forward(self, inputs):
        rgb, depth = inputs
        self.rgb_net.eval()                           # frozen feature extractors
        self.depth_net.eval()
        rgb_features = self.rgb_net(rgb)[0:-1]
        depth_features = self.depth_net(depth)[0:-1]
        return self.fuse(list(rgb_features) + list(depth_features))


class Searchable_RGB_Depth_Net(_RGBDepthBase):
    def __init__(self, args, opt, criterion):
        super().__init__()
        self._build_backbones(opt)
        self._build_head(args, criterion, self.create_reshape_layers(args), args.num_input_nodes,
                         args.num_keep_edges)

    def create_reshape_layers(self, args):
        return self.make_reshape_layers(aux.ReshapeInputLayer, C_INS, args)


class Found_RGB_Depth_Net(_RGBDepthBase):
    def __init__(self, args, opt, criterion, genotype):
        super().__init__()
        self._build_backbones(opt)
        self._genotype = genotype
        self._build_head(args, criterion, self.create_reshape_layers(args), args.num_input_nodes,
                         args.num_keep_edges, genotype=genotype)

    def create_reshape_layers(self, args):
        return self.make_reshape_layers(aux.ReshapeInputLayer, C_INS, args, self._genotype)
