"""MM-IMDB search / retrain loop tracking F1 (reference train_searchable/mmimdb.py:
train_mmimdb_track_f1 :10-205, test_mmimdb_track_f1 :207-285)."""
from . import _loop


def _unpack(data, device):
    image, text, label = data['image'].to(device), data['text'].to(device), data['label'].to(device)
    return (text, image), label


def train_mmimdb_track_f1(model, architect, criterion, optimizer, scheduler, dataloaders, dataset_sizes,
                          device, num_epochs, parallel, logger, plotter, args, f1_type='weighted',
                          init_f1=0.0, th_fscore=0.3, status='search'):
    overloops = 0
    while True:
        r = _loop.run(model, architect, criterion, optimizer, scheduler, dataloaders, dataset_sizes, device,
                      num_epochs, logger, plotter, args, status, _unpack, _loop.F1Meter(f1_type, th_fscore),
                      eval_phases=['train', 'dev', 'test'], better=lambda new, old: new > old,
                      task='mmimdb', nan_escape=True, init=init_f1)
        best_f1 = init_f1 if r['best_dev'] is None else max(init_f1, r['best_dev'])
        if r['nan_abort']:
            return best_f1                     # the reference returns the bare scalar on a NaN loss
        # the reference trains one extra epoch when a one-epoch run ends with a NaN F1
        if best_f1 != best_f1 and num_epochs == 1 and overloops < 1:
            logger.info('Recording a NaN F1, training for one more epoch.')
            overloops += 1
            continue
        break
    if best_f1 != best_f1:
        best_f1 = 0.0
    if status == 'search':
        return best_f1, r['best_dev_genotype']
    best_test = init_f1 if r['best_test'] is None else max(init_f1, r['best_test'])
    return best_test, r['best_test_genotype']


def test_mmimdb_track_f1(model, criterion, dataloaders, dataset_sizes, device, parallel, logger, args,
                         f1_type='weighted', init_f1=0.0, th_fscore=0.3):
    f1 = _loop.evaluate(model, criterion, dataloaders['test'], dataset_sizes['test'], device, logger, args,
                        _unpack, _loop.F1Meter(f1_type, th_fscore))
    logger.info(str(model.genotype()))
    return f1                                  # the reference returns the bare score (mmimdb.py:285)
