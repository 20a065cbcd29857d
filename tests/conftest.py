import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The oracle's many small BLAS calls crawl when every one of them fans out over all the cores of a big host (the GPU
# box has 256): keep the thread pools small for the whole test session.
try:
    from threadpoolctl import threadpool_limits
    _blas_limit = threadpool_limits(limits=min(8, os.cpu_count() or 1))
except Exception:      # threadpoolctl missing: the tests still run, only slower
    _blas_limit = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test (still part of the default CPU suite)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def sa19_golden():
    return load_golden("sa19_female_default.npz")


@pytest.fixture(scope="session")
def sa19_signal():
    from scipy.io import wavfile
    fs, x = wavfile.read(os.path.join(GOLDEN, "SA19.WAV"))
    return fs, x / 32768.0


def unpack_records(g, a, with_fm=True):
    """Expand the sparse frame-centre records stored by make_golden.py for adaptation `a`."""
    shape = tuple(int(v) for v in g["rec%d_shape" % a])
    mask = np.unpackbits(g["rec%d_mask" % a])[: shape[0] * shape[1]].astype(bool).reshape(shape)
    out = {"mask": mask, "a0": g["rec%d_a0" % a]}
    for name in ("am", "ph") + (("fm",) if with_fm else ()):
        arr = np.zeros(shape)
        arr[mask] = g["rec%d_%s" % (a, name)]
        out[name] = arr
    return out


def record_measurement(name, **values):
    """Numbers a GPU test measured next to what the reference gave (not assertions: evidence).  Collected in
    gpurun_out/parity_measurements.json on the GPU box; the round's copy is committed under profiles/."""
    import json
    path = os.path.join(ROOT, "gpurun_out", "parity_measurements.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = {}
    if os.path.exists(path):
        try:
            data = json.load(open(path))
        except ValueError:
            data = {}
    data[name] = values
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
