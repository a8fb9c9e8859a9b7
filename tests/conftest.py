import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Fixtures produced by oracle/gen_golden.py from the reference implementation."""
    path = os.path.join(ROOT, "tests", "golden", "bkcore_golden.npz")
    data = np.load(path, allow_pickle=False)
    names = [str(s) for s in data["__names__"]]
    cases = {}
    for n in names:
        prefix = n + "/"
        cases[n] = {k[len(prefix):]: data[k] for k in data.files if k.startswith(prefix)}
    return cases


def max_rel_err(x, ref, floor=0.0):
    """max |x - ref| / (|ref| + floor) over all elements (complex or real numpy arrays)."""
    x = np.asarray(x)
    ref = np.asarray(ref)
    if ref.size == 0:
        return 0.0
    return float(np.max(np.abs(x - ref) / (np.abs(ref) + floor)))
