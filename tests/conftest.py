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


def isolated(fn):
    """Run a GPU test in a child pytest process of its own (the parent only checks the child's exit status).

    For the tests that capture RCCL collectives into a hipGraph: on this image (torch 2.10 + RCCL 2.26) the process is occasionally
    ABORTED from a non-Python thread while `capture_end` runs (round 6: one full-suite run in four; `faulthandler` shows no Python
    frame on the aborting thread, the same tests pass in the runs before and after with no code change).  An abort takes the whole
    pytest process — and every test behind it — with it, so these tests get a process to lose: a child that dies of SIGABRT is
    started ONCE more and the event is reported as a warning; any other failure of the child fails the test at once.  The child is
    a child process (never an exec of this one) and at most one runs at a time."""
    import functools
    import inspect
    import subprocess
    import warnings

    @functools.wraps(fn)
    def wrapper(*args, **kw):
        if os.environ.get("TTSK_TEST_CHILD") == "1":
            return fn(*args, **kw)
        node = "%s::%s" % (os.path.relpath(inspect.getsourcefile(fn), ROOT), fn.__name__)
        env = dict(os.environ, TTSK_TEST_CHILD="1")
        for attempt in (1, 2):
            p = subprocess.run([sys.executable, "-m", "pytest", node, "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
            if p.returncode == 0:
                return
            tail = p.stdout.decode(errors="replace")[-3000:]
            aborted = p.returncode in (-6, 134) or "Fatal Python error: Aborted" in tail
            if aborted and attempt == 1:
                warnings.warn("%s: child aborted (SIGABRT from a non-Python thread around hipGraph capture of RCCL collectives); started once more" % node)
                continue
            raise AssertionError("%s failed in its child process (exit %d):\n%s" % (node, p.returncode, tail))
    return wrapper
