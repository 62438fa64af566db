import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

PKG = "lk-s-2022-estimacija-pokreta_amd"


def pkg(sub=None):
    return importlib.import_module(PKG if sub is None else PKG + "." + sub)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def synth():
    return pkg("synth")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(ROOT, "tests", "golden", "ref_%s.npz" % name))
    return load


GOLDEN_NAMES = ("a40x48_c5x6", "b36x40_c9x8", "c45x35_c9x7", "d45x35_c9x7_unrelated")
