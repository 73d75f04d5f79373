import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def have_gpu():
    return _have_gpu()


@pytest.fixture(scope="session")
def ctx_small():
    """A GPU context sized for test scans (the HIP library must be built; no fallback)."""
    from sg_slam_amd import api
    cfg = api.preset("kitti", max_points=1 << 18, map_capacity_log2=17)
    c = api.Context(cfg)
    yield c
    c.close()


def make_ctx(**kw):
    from sg_slam_amd import api
    args = dict(max_points=1 << 18, map_capacity_log2=17)
    args.update(kw)
    return api.Context(api.preset("kitti", **args))
