import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The no-GPU suite (`-m "not gpu"`) spends its time in the emulated kernels -- one OS thread per GPU thread, mostly waiting on
    barriers -- and its tests are independent (the emulator build is behind a file lock, the multi-process tests take OS-assigned
    ports): hand it to pytest-xdist when that is installed and nobody chose a worker count (18.5 min -> 80 s on 8 cores).  Runs that
    select GPU tests are never parallelised (one device); NT_TEST_SERIAL=1 keeps this suite serial too."""
    if os.environ.get("NT_TEST_SERIAL") or not config.pluginmanager.hasplugin("xdist"):
        return None
    if "not gpu" not in (config.getoption("markexpr", "") or "") or os.environ.get("PYTEST_XDIST_WORKER"):
        return None
    explicit = any(a == "-n" or a.startswith("-n") and a[2:3].isdigit() or a.startswith("--numprocesses") or a.startswith("--dist")
                   for a in config.invocation_params.args)  # (the caller chose: -n 0 means serial)
    if not explicit and getattr(config.option, "numprocesses", None) in (None, 0) and getattr(config.option, "dist", "no") == "no":
        n = min(8, max(1, (os.cpu_count() or 1) - 2))
        if n >= 3:
            config.option.numprocesses = n
    return None


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle_bridge

    return oracle_bridge.lib()
