"""Builds of the test-side native artefacts under tests/emu (TEST INFRASTRUCTURE, never loaded by the product package), each
skipped when the library in the tree was built from exactly the sources that are there now.  The check is a hash of the sources
next to the library (`<lib>.sources`), not modification times: a copy of the tree -- the GPU box's snapshot -- does not keep
them, and `make` / an mtime test then rebuilds everything (30-60 s for each emulator library, 33 s of hipcc for the lane probe:
two minutes of the GPU suite on a slow host).  `__graft_entry__.build()` runs all three here, so the libraries and their stamps
travel with the snapshot; tests/conftest.py and tests/test_gpu_lane_ops.py call the same functions."""
import glob
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")
CSRC = os.path.join(ROOT, "sod100k_amd", "csrc")


def _digest(files):
    h = hashlib.sha256()
    for f in sorted(files):
        with open(f, "rb") as fi:
            h.update(os.path.basename(f).encode() + b"\0" + fi.read())
    return h.hexdigest()


def _stamped(so, files, cmd):
    want = _digest(files)
    stamp = so + ".sources"
    if os.path.exists(so) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return so
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    tmp = stamp + f".{os.getpid()}"
    with open(tmp, "w") as fo:
        fo.write(want + "\n")
    os.replace(tmp, stamp)
    return so


def emu_sources():
    return (glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inl")) +
            glob.glob(os.path.join(ROOT, "include", "*.h")) + [os.path.join(EMU, f) for f in ("Makefile", "hip_cpu_shim.h", "emu_impl.cpp")])


def emu_make(lanes=False):
    """tests/emu/libcsnet_emu.so (sequential stand-ins for the cross-lane instructions) or, lanes=True, libcsnet_emu_lanes.so (the
    device code paths with lane-exact MFMA / DPP / readfirstlane / shuffle)."""
    so = os.path.join(EMU, "libcsnet_emu_lanes.so" if lanes else "libcsnet_emu.so")
    return _stamped(so, emu_sources(), ["make", "-C", EMU, "-j8"] + (["LANES=1"] if lanes else []))


def lane_probe_make():
    """tests/emu/liblane_probe.so: one wave runs one real cross-lane instruction (gfx950 code object; hipcc cross-compiles it
    without a GPU)."""
    so, src = os.path.join(EMU, "liblane_probe.so"), os.path.join(EMU, "lane_probe.hip")
    return _stamped(so, [src], ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, src])
