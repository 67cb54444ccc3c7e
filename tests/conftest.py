import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """librjprt.so is a build artefact (git-ignored): on a fresh checkout compile it once
    before the tests that load it (hipcc cross-compiles gfx950 without a GPU)."""
    lib = os.path.join(ROOT, "rajepy_amd", "librjprt.so")
    if not os.path.exists(lib):
        import shutil
        import subprocess
        if shutil.which("hipcc"):
            subprocess.run(["bash", os.path.join(ROOT, "rajepy_amd", "csrc", "build.sh")],
                           check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
