import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def stacker():
    """One stk_ctx on cuda:0 for the whole GPU session (fails loudly without the HIP library)."""
    from libstacker_rs_amd import Stacker
    s = Stacker(0)
    yield s
    s.close()


@pytest.fixture(scope="session")
def small_stack():
    """4 x 320x240 synthetic BGR u8 frames + ground-truth homographies (seeded)."""
    from libstacker_rs_amd import synth
    frames, G = synth.make_stack(4, 320, 240)
    return frames.numpy(), G
