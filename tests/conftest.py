import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


class Golden:
    """Lazy access to tests/golden/<name>.npz as torch tensors."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name + '.npz'))

    def __getitem__(self, key):
        return torch.from_numpy(self._z[key])

    def keys(self):
        return list(self._z.keys())


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get


def gpu_available():
    return torch.cuda.is_available()


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) where no GPU is visible; on the GPU box they all run.
    if gpu_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
