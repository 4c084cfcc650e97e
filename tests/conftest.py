import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # Build the CPU oracle NOW, before any test initialises the GPU: building runs `make` in a child
    # process, and a process that has touched the GPU must not spawn/exec other programs on the box.
    import oracle as orc
    orc.build()
    # Same reason: the process-spawning GPU tests (tests/test_gpu_two_ranks.py) take their workers from a fork server
    # that is started here, while this process is still clean -- its children are forks of that clean server, so no
    # program is ever exec'ed from a process that holds the GPU.
    try:
        import multiprocessing.forkserver as fs
        fs.ensure_running()
    except Exception:  # noqa: BLE001  (platforms without a fork server: those tests will say so)
        pass


@pytest.fixture(scope='session')
def oracle():
    """the CPU oracle (test infrastructure): builds oracle/libbvq_oracle.so on first use"""
    import oracle as orc
    orc.build()
    return orc
