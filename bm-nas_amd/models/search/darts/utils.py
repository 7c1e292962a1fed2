"""Helpers used by the trainers (reference models/search/darts/utils.py:80-127)."""
import os
import pickle
import shutil

import numpy as np
import torch


def count_parameters_in_MB(model):
    return count_parameters(model) / 1e6


def count_parameters(model):
    return int(np.sum([np.prod(v.size()) for name, v in model.named_parameters() if "auxiliary" not in name]))


def save(model, model_path):
    torch.save(model.state_dict(), model_path)


def load(model, model_path):
    model.load_state_dict(torch.load(model_path))


def save_pickle(obj, obj_path):
    with open(obj_path, "wb") as f:
        pickle.dump(obj, f)


def load_pickle(obj_path):
    with open(obj_path, "rb") as f:
        return pickle.load(f)


def create_exp_dir(path, scripts_to_save=None):
    if not os.path.exists(path):
        os.makedirs(path)
    print('Experiment dir : {}'.format(path))
    if scripts_to_save is not None:
        os.mkdir(os.path.join(path, 'scripts'))
        for script in scripts_to_save:
            shutil.copyfile(script, os.path.join(path, 'scripts', os.path.basename(script)))
    os.mkdir(os.path.join(path, 'architectures'))
    os.mkdir(os.path.join(path, 'best'))
