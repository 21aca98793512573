import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "emul: runs the kernel sources on the CPU wave64 emulator")


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"
