import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle (torch / oneDNN) collapses when oversubscribed: on the GPU box's 256 hardware threads one DY3h
    # forward takes 5.6 s with the default thread count and 1.5 s with 32 (bench.py cpu_baseline measured the same)
    try:
        import torch
        torch.set_num_threads(min(32, os.cpu_count() or 1))
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def free_port():
    """A TCP port nobody listens on right now (rendezvous of the multi-process tests: a fixed number collides with a
    previous run's sockets in TIME_WAIT or with a second pytest on the same host)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
