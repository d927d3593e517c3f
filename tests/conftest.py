import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs oracle/_ref (the reference mounted; build container only)")


@pytest.fixture(scope="session")
def T():
    return importlib.import_module("pixel-art-raytracer_amd.types")


@pytest.fixture(scope="session")
def par():
    """The product package (ctypes binding over libpar_raytracer.so)."""
    return importlib.import_module("pixel-art-raytracer_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle.oracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libref_path.so not built (reference not mounted on this machine)")
    return Reference()


@pytest.fixture(scope="session")
def golden_frames(T):
    with open(os.path.join(GOLDEN, "ref_frames.json")) as f:
        meta = json.load(f)
    z = np.load(os.path.join(GOLDEN, "ref_frames.npz"))
    cases = {}
    for name, m in meta.items():
        aabbs = np.ascontiguousarray(z[name + "_aabbs"]).view(T.AABB).reshape(-1)
        light = np.ascontiguousarray(z[name + "_light"]).view(T.LIGHT).reshape(-1)
        cases[name] = (m, aabbs, light)
    return cases


@pytest.fixture(scope="session")
def appendix_b():
    with open(os.path.join(GOLDEN, "appendix_b.json")) as f:
        return json.load(f)
