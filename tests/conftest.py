"""pytest configuration.  ``-m "not gpu"`` runs on a CPU-only box; ``-m gpu`` needs one MI355X."""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The checker (the torch oracle) runs on the host.  torch's default is one thread per core -- 128 on the GPU boxes, whose hosts are
# shared (load 20-40): a B=2 autograd step of the oracle then takes 6.8 s instead of 0.15 s (every small op waits for 128 threads
# at its barrier), a 64-image forward 16.7 s instead of 4.5 s (tools/probes/host_threads.py, lease r6thr: 16 threads are the best
# for both).  Set before torch is imported, so spawned ranks inherit it.
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 16)))

GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "sod100k_amd", "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (run on the MI355X box)")


def _emu_make(lanes):
    """`make` of the CPU emulation unless the library in the tree was built from exactly these sources (tests/emu_build.py)."""
    import emu_build
    return emu_build.emu_make(lanes)


@pytest.fixture(scope="session")
def emu_lib():
    """The csnet kernels compiled for the host by g++ (tests/emu): same sources, CPU fibers for threads."""
    from sod100k_amd import _native as N
    try:
        so = _emu_make(False)
    except (OSError, subprocess.CalledProcessError) as e:
        pytest.skip(f"cannot build the CPU emulation of the kernels: {e}")
    return N.bind(ctypes.CDLL(so))


@pytest.fixture(scope="session")
def emu_lanes_lib():
    """The emulator's lane-exact mode (tests/emu `make LANES=1`): the kernels' DEVICE paths, with every cross-lane instruction
    (MFMA lane maps, DPP moves, readfirstlane) executed as a rendezvous of the wave's 64 fibers (hip_cpu_shim.h).  Returns
    (bound library, raw CDLL) -- the raw handle exposes csn_emu_lane_ops(kind), the count of executed cross-lane instructions."""
    from sod100k_amd import _native as N
    try:
        so = _emu_make(True)
    except (OSError, subprocess.CalledProcessError) as e:
        pytest.skip(f"cannot build the lane-exact CPU emulation of the kernels: {e}")
    raw = ctypes.CDLL(so)
    raw.csn_emu_lane_ops.restype = ctypes.c_ulonglong
    raw.csn_emu_lane_ops.argtypes = [ctypes.c_int]
    return N.bind(raw), raw


@pytest.fixture(scope="session")
def x2_manifest():
    return os.path.join(DATA, "csnet-L-x2.json")


@pytest.fixture(scope="session")
def x1_manifest():
    return os.path.join(DATA, "csnet-L-x1.json")
