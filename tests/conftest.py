import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _private_jit_cache(tmp_path_factory):
    """compiled kernels persist on disk by default ($XDG_CACHE_HOME/ipcr_hip): the tests use a directory of their own,
    empty at the start of every session, so that every run compiles what it tests"""
    if "IPCR_JIT_CACHE_DIR" not in os.environ:
        os.environ["IPCR_JIT_CACHE_DIR"] = str(tmp_path_factory.mktemp("jit_cache"))
    # a small panel's kernels are built in the background and its first scans take the table-driven kernel: the tests
    # say which kernel they mean to exercise (ipcr_scan_stats.kernel_kind), so every build is synchronous here
    os.environ.setdefault("IPCR_JIT_ASYNC", "0")
    # small launches keep the one-wave-per-block kernel in the tests (the kernel of the headline numbers, and what the suite
    # has always covered); test_small_launches_share_a_block_between_waves turns the segmented form on for its cases
    os.environ.setdefault("IPCR_JIT_SEGMENTS", "1")
    yield
