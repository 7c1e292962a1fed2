"""NTU RGB+D search / retrain loop tracking accuracy (reference train_searchable/ntu.py:
train_ntu_track_acc :12-182, test_ntu_track_acc :184-227)."""
from . import _loop


def _unpack(data, device):
    return (data['rgb'].to(device), data['ske'].to(device)), data['label'].to(device)


def train_ntu_track_acc(model, architect, criterion, optimizer, scheduler, dataloaders, dataset_sizes,
                        device=None, num_epochs=200, verbose=False, parallel=False, logger=None,
                        plotter=None, args=None, status='search'):
    r = _loop.run(model, architect, criterion, optimizer, scheduler, dataloaders, dataset_sizes, device,
                  num_epochs, logger, plotter, args, status, _unpack, _loop.AccuracyMeter(),
                  eval_phases=['train', 'test'], better=lambda new, old: new >= old, init=0)
    if status == 'search':
        return (r['best_dev'] or 0), r['best_dev_genotype']
    return (r['best_test'] or 0), r['best_dev_genotype']


def test_ntu_track_acc(model, dataloaders, criterion, genotype, dataset_sizes, device, logger, args):
    return _loop.evaluate(model, criterion, dataloaders['test'], dataset_sizes['test'], device, logger, args,
                          _unpack, _loop.AccuracyMeter())
