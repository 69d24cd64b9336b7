import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch's intra-op threads cut down to the cores this process is granted (tts_king_amd/hostcpu.py: on a 1-GPU box the CPU oracle
    # otherwise runs on 128 threads that share 16 cores, and the GPU suite takes 270 s instead of 54)
    from tts_king_amd.hostcpu import fit_torch_threads
    fit_torch_threads(cap=16)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def cfg():
    from tts_king_amd.config import default_config
    return default_config()
