import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


# torch ships its own ROCm runtime libraries; a process that loads /opt/rocm's (through libexa_hip.so) first and imports
# torch later ends up with two HSA runtimes and torch then sees no GPU.  bench.py imports torch first; so do the tests,
# whatever subset of the files is collected.
try:
    import torch  # noqa: F401,E402
except ImportError:
    pass
