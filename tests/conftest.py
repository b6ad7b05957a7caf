"""pytest configuration.  ``-m "not gpu"`` runs on a CPU-only box; ``-m gpu`` needs one MI355X."""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "sod100k_amd", "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (run on the MI355X box)")


@pytest.fixture(scope="session")
def emu_lib():
    """The csnet kernels compiled for the host by g++ (tests/emu): same sources, CPU fibers for threads."""
    from sod100k_amd import _native as N
    emu_dir = os.path.join(ROOT, "tests", "emu")
    try:
        subprocess.run(["make", "-C", emu_dir, "-j8"], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.PIPE)
    except (OSError, subprocess.CalledProcessError) as e:
        pytest.skip(f"cannot build the CPU emulation of the kernels: {e}")
    return N.bind(ctypes.CDLL(os.path.join(emu_dir, "libcsnet_emu.so")))


@pytest.fixture(scope="session")
def x2_manifest():
    return os.path.join(DATA, "csnet-L-x2.json")


@pytest.fixture(scope="session")
def x1_manifest():
    return os.path.join(DATA, "csnet-L-x1.json")
