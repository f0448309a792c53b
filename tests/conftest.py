import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One golap context on cuda:0 for the whole GPU session (profiling on)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpu_olap_engine_amd.engine import Context
    torch.cuda.set_device(0)
    c = Context(profiling=True, max_gpu_memory=0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _no_stray_ranks():
    """A multi-process test that fails half-way (one rank raised, the others still wait for it in a collective) must not leave
    its ranks behind: they would hold the GPU and keep the test session from exiting.  Ends exactly the processes this session
    started (multiprocessing's own list of live children), never anything by name."""
    yield
    import multiprocessing
    for p in multiprocessing.active_children():
        p.terminate()
        p.join(10)
        if p.is_alive():
            p.kill()
            p.join(10)
