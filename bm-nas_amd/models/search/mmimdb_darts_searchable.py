"""MM-IMDB search driver and hypernet wrappers (reference models/search/mmimdb_darts_searchable.py:
train_darts_model :19-55, Searchable_Image_Text_Net :57-133, Found_Image_Text_Net :135-221).
Backbones (models.central.mmimdb: VGG19 + MaxOut MLP) are out of scope and come from the reference
checkout on sys.path."""
import torch

import models.auxiliary.aux_models as aux
import models.search.train_searchable.mmimdb as tr

from bmnas import nn as bnn

from ._common import HyperNetBase, parallel_flag, search_setup

C_INS = [512, 512, 512, 512, 64, 128]


def train_darts_model(dataloaders, args, device, logger):
    dataset_sizes = {x: len(dataloaders[x].dataset) for x in ['train', 'dev', 'test']}
    num_batches_per_epoch = dataset_sizes['train'] / args.batchsize
    criterion = bnn.BCEWithLogitsLoss()          # torch criterion subclass on the HIP loss kernel
    model = Searchable_Image_Text_Net(args, criterion)
    optimizer, scheduler, architect, plotter = search_setup(model, args, criterion, device,
                                                            num_batches_per_epoch, args.weight_decay)
    return tr.train_mmimdb_track_f1(model, architect, criterion, optimizer, scheduler, dataloaders,
                                    dataset_sizes, device=device, num_epochs=args.epochs,
                                    parallel=parallel_flag(args), logger=logger, plotter=plotter,
                                    args=args, f1_type=args.f1_type, init_f1=0.0, th_fscore=0.3)


class _ImageTextBase(HyperNetBase):
    def _build_backbones(self, args):
        import models.central.mmimdb as mmimdb        # reference checkout (out of scope here)
        self.imagenet = mmimdb.GP_VGG(args)
        self.textnet = mmimdb.MaxOut_MLP(args)

    def forward(self, tensor_tuple):
        text, image = tensor_tuple
        image_features = self.imagenet(image)[0:-1]
        text_features = self.textnet(text)[0:-1]
        return self.fuse(list(image_features) + list(text_features))


class Searchable_Image_Text_Net(_ImageTextBase):
    def __init__(self, args, criterion):
        super().__init__()
        self._build_backbones(args)
        self._build_head(args, criterion, self.create_reshape_layers(args), args.num_input_nodes,
                         args.num_keep_edges)

    def create_reshape_layers(self, args):
        return self.make_reshape_layers(aux.ReshapeInputLayer_MMIMDB, C_INS, args)


class Found_Image_Text_Net(_ImageTextBase):
    def __init__(self, args, criterion, genotype):
        super().__init__()
        self._build_backbones(args)
        self._genotype = genotype
        self._build_head(args, criterion, self.create_reshape_layers(args), args.num_input_nodes,
                         args.num_keep_edges, genotype=genotype)

    def create_reshape_layers(self, args):
        return self.make_reshape_layers(aux.ReshapeInputLayer_MMIMDB, C_INS, args, self._genotype)
