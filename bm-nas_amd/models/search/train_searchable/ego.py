"""EgoGesture search / retrain loop tracking accuracy (reference train_searchable/ego.py:
train_ego_track_acc :13-177, test_ego_track_acc :179-223).  A batch is (inputs, labels) with RGB in
channels 0:3 and depth in 3: of a (b, 4, T, H, W) clip."""
from . import _loop


def _unpack(data, device):
    inputs, labels = data
    return (inputs[:, 0:3].to(device), inputs[:, 3:].to(device)), labels.to(device)


def train_ego_track_acc(model, architect, criterion, optimizer, scheduler, dataloaders, dataset_sizes,
                        device=None, num_epochs=200, parallel=False, logger=None, plotter=None, args=None,
                        status='search'):
    r = _loop.run(model, architect, criterion, optimizer, scheduler, dataloaders, dataset_sizes, device,
                  num_epochs, logger, plotter, args, status, _unpack, _loop.AccuracyMeter(),
                  eval_phases=['train', 'test'], better=lambda new, old: new >= old, task='ego', init=0)
    if status == 'search':
        return (r['best_dev'] or 0), r['best_dev_genotype']
    return (r['best_test'] or 0), r['best_dev_genotype']


def test_ego_track_acc(model, dataloaders, criterion, genotype, dataset_sizes, device, logger, args):
    return _loop.evaluate(model, criterion, dataloaders['test'], dataset_sizes['test'], device, logger, args,
                          _unpack, _loop.AccuracyMeter())
