import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_err(a, b):
    """max|a-b| / max|b| per tensor -- the parity metric of SURVEY §7/§8(c)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.abs(b).max()
    if den == 0:
        return float(np.abs(a).max())
    return float(np.abs(a - b).max() / den)


@pytest.fixture(scope="session")
def golden():
    z = np.load(os.path.join(GOLDEN, "small_scene.npz"))
    return {k: z[k] for k in z.files}


def golden_scene(gold):
    """Unpack the fixture into oracle-style inputs (torch CPU tensors)."""
    grids = {k[len("grid/"):]: torch.from_numpy(v) for k, v in gold.items() if k.startswith("grid/")}
    params = {k[len("param/"):]: torch.from_numpy(v) for k, v in gold.items() if k.startswith("param/")}
    bound = torch.from_numpy(gold["bound"])
    return grids, params, bound
