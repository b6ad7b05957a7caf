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
def emu_lanes_lib():
    """The emulator's lane-exact mode (tests/emu `make LANES=1`): the kernels' DEVICE paths, with every cross-lane instruction
    (MFMA lane maps, DPP moves, readfirstlane) executed as a rendezvous of the wave's 64 fibers (hip_cpu_shim.h).  Returns
    (bound library, raw CDLL) -- the raw handle exposes csn_emu_lane_ops(kind), the count of executed cross-lane instructions."""
    from sod100k_amd import _native as N
    emu_dir = os.path.join(ROOT, "tests", "emu")
    try:
        subprocess.run(["make", "-C", emu_dir, "LANES=1", "-j8"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    except (OSError, subprocess.CalledProcessError) as e:
        pytest.skip(f"cannot build the lane-exact CPU emulation of the kernels: {e}")
    raw = ctypes.CDLL(os.path.join(emu_dir, "libcsnet_emu_lanes.so"))
    raw.csn_emu_lane_ops.restype = ctypes.c_ulonglong
    raw.csn_emu_lane_ops.argtypes = [ctypes.c_int]
    return N.bind(raw), raw


@pytest.fixture(scope="session")
def x2_manifest():
    return os.path.join(DATA, "csnet-L-x2.json")


@pytest.fixture(scope="session")
def x1_manifest():
    return os.path.join(DATA, "csnet-L-x1.json")
