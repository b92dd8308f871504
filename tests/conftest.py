import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "audio-mamba-aum_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    """max |a-b| / (max|b| + tiny): scale-aware max error used by every parity test."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def rms_err(a, b):
    """rms(a-b) / (rms(b) + tiny): the error relative to the TYPICAL element, next to rel_err's error relative to the LARGEST one
    (a tensor whose bulk is far below its maximum can be wrong everywhere but at the peak and pass rel_err)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b ** 2)) + 1e-30))


def rel_err_by(a, b, axis, floor=1e-3):
    """worst slice of rel_err taken slice by slice along `axis` (each state column of dA, each state row of dB / dC against ITS OWN
    largest element); slices whose largest element is below `floor` x the global maximum are measured against that floor."""
    a = np.moveaxis(np.asarray(a, np.float64), axis, 0)
    b = np.moveaxis(np.asarray(b, np.float64), axis, 0)
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    scale = np.maximum(np.abs(b).max(axis=1), floor * np.abs(b).max()) + 1e-30
    return float((np.abs(a - b).max(axis=1) / scale).max())
