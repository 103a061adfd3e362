import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (-m "not gpu": oracle, host logic, emulated kernels, gloo ranks) is 160 independent, single-threaded
    tests: run it on four pytest-xdist workers unless the caller chose otherwise (-n ..., or DLPD_TEST_WORKERS=0) -- 5 minutes
    instead of 17.  The GPU suite is NEVER parallelised here: its tests share one device, its timings and its memory."""
    if os.environ.get("PYTEST_XDIST_WORKER") or os.environ.get("DLPD_TEST_WORKERS") == "0":
        return None
    opt = config.option
    if (getattr(opt, "markexpr", "") or "").strip() != "not gpu" or getattr(opt, "collectonly", False):
        return None
    if not hasattr(opt, "numprocesses") or opt.numprocesses not in (None, 0) or getattr(opt, "dist", "no") != "no":
        return None                                    # (xdist absent, or the caller passed -n / --dist)
    try:
        n = int(os.environ.get("DLPD_TEST_WORKERS", "4"))
    except ValueError:
        n = 4
    n = max(1, min(n, _usable_cores() // 2 or 1))
    if n > 1:
        opt.numprocesses = n
        opt.dist = "load"                              # (what -n implies; xdist's own hook fills in the rest)
    return None


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session")
def emu():
    """CPU-emulated kernel library (tests/emu): same sources, same C ABI, host pointers."""
    from emu_lib import emu_lib
    return emu_lib()


def _usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


try:  # the GPU boxes expose 256 logical CPUs behind a 16-core quota: do not oversubscribe the oracle
    import torch
    torch.set_num_threads(min(_usable_cores(), 16))
except Exception:
    pass
