import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """-> (state_dict of torch tensors, dict of numpy arrays, cfg dict) from tests/golden/<name>.npz"""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    cfg = {k[4:]: (z[k].tolist() if z[k].ndim else z[k].item()) for k in z.files if k.startswith("cfg_")}
    arrs = {k: z[k] for k in z.files if not k.startswith("sd/") and not k.startswith("cfg_")}
    return sd, arrs, cfg


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get
