"""COMPUTE_PARAM / seed_all (src/func_util/cluster_mngt.py).  Determinism of this build does not
depend on seeds or cuDNN flags: the kernels have a fixed accumulation order."""
import random

import numpy
import torch

COMPUTE_PARAM = {'device': 'cuda:0', 'flag_gpu': True}


def set_compute_param(key, value):
    COMPUTE_PARAM[key] = value


def seed_all(seed=666):
    torch.manual_seed(seed)
    numpy.random.seed(seed)
    random.seed(seed)
