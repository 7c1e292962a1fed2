"""Searcher facades (reference models/darts_searchable.py:25-90): build the datasets / DataLoaders
and hand over to the per-dataset train_darts_model.  Datasets (`datasets/*`) and `models.utils` are
out of scope and are imported from the reference checkout on sys.path.  With data parallelism
(any launch under torch.distributed.run, WORLD_SIZE > 1) every rank draws its own shard through a
DistributedSampler instead of DataParallel's scatter."""
import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from bmnas import dist as bdist
from models.search._common import data_parallel_world


def _loaders(datasets, args):
    world = data_parallel_world(args)
    loaders = {}
    for split, ds in datasets.items():
        if world > 1:
            sampler = DistributedSampler(ds, shuffle=True, drop_last=False)
            loaders[split] = DataLoader(ds, batch_size=max(1, args.batchsize // world), sampler=sampler,
                                        num_workers=args.num_workers, drop_last=False)
        else:
            loaders[split] = DataLoader(ds, batch_size=args.batchsize, shuffle=True,
                                        num_workers=args.num_workers, drop_last=False)
    return loaders


class MMIMDB_Searcher():
    def __init__(self, args, device, logger):
        import torchvision.transforms as transforms
        from datasets import mmimdb as mmimdb_data
        self.args, self.device, self.logger = args, device, logger
        tf = transforms.Compose([mmimdb_data.ToTensor()])
        datasets = {stage: mmimdb_data.MM_IMDB(args.datadir, transform=tf, stage=stage, feat_dim=300, args=args)
                    for stage in ('train', 'dev', 'test')}
        self.dataloaders = _loaders(datasets, args)

    def search(self):
        import models.search.mmimdb_darts_searchable as mmimdb
        return mmimdb.train_darts_model(self.dataloaders, self.args, self.device, self.logger)


class NTUSearcher():
    def __init__(self, args, device, logger):
        import torchvision.transforms as transforms
        from datasets import ntu as ntu_data
        self.args, self.device, self.logger = args, device, logger
        tf_val = transforms.Compose([ntu_data.NormalizeLen(), ntu_data.ToTensor()])
        tf_tra = transforms.Compose([ntu_data.AugCrop(), ntu_data.NormalizeLen(), ntu_data.ToTensor()])
        datasets = {'train': ntu_data.NTU(args.datadir, transform=tf_tra, stage='train_exp', args=args),
                    'dev': ntu_data.NTU(args.datadir, transform=tf_val, stage='dev', args=args),
                    'test': ntu_data.NTU(args.datadir, transform=tf_val, stage='test', args=args)}
        self.dataloaders = _loaders(datasets, args)

    def search(self):
        import models.search.ntu_darts_searchable as ntu
        return ntu.train_darts_model(self.dataloaders, self.args, self.device, self.logger)


class Ego_Searcher():
    def __init__(self, args, device, logger):
        from datasets import ego as ego_data
        from models.utils import parse_opts
        self.args, self.device, self.logger = args, device, logger
        self.opt = parse_opts(args)
        self.dataloaders = {'train': ego_data.get_train_loader(self.opt, args),
                            'dev': ego_data.get_dev_loader(self.opt, args),
                            'test': ego_data.get_test_loader(self.opt, args)}

    def search(self):
        import models.search.ego_darts_searchable as ego
        return ego.train_darts_model(self.dataloaders, self.args, self.opt, self.device, self.logger)
