"""ctypes binding of the C ABI in ``include/csnet_hip.h`` (libcsnet_hip.so) and its build recipe.

The product path has exactly one backend: the hand-written HIP library compiled for gfx950.
``load()`` raises ``RuntimeError`` if it is missing -- there is no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import shutil
import subprocess
from typing import Optional, Sequence

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# SOD100K_HIP_LIB: developer override (A/B builds of the same sources, e.g. the knock-out variants of profiles/r1_notes.md)
LIB_PATH = os.environ.get("SOD100K_HIP_LIB") or os.path.join(CSRC, "libcsnet_hip.so")
SOURCES = ("csn_plan.hip", "k_misc.hip", "k_goct_pw.hip", "k_ms.hip", "k_train.hip", "k_wgrad.hip", "k_wgrad_c3.hip", "k_wgrad_bf.hip", "k_goct_c3.hip", "k_csf.hip", "k_pw4.hip", "k_c3q.hip", "k_pwq.hip", "k_ilb.hip", "k_head.hip")

ABI_VERSION = 3       # include/csnet_hip.h CSN_ABI_VERSION: checked BEFORE the entry points are bound (a stale .so lacks the new ones)


def sources_sha16() -> str:
    """sha256[:16] over the kernel sources (csrc/*.hip, *.h, *.inl, include/*.h): stamps counter files / bench lines with the tree they belong to
    (the GPU box has no .git)."""
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h", ".inl")):
            h.update(f.encode())
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(fh.read())
    for f in ("csnet_hip.h", "csf_hip.h"):        # the public headers are compiled in too (ABI version, argument structs)
        h.update(f.encode())
        with open(os.path.join(os.path.dirname(HERE), "include", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


MAX_BRANCH = 3
NDIL = 5
UNIT_GOCT, UNIT_DW, UNIT_MS, UNIT_CLS = 1, 2, 3, 4
OPT_FUSE_DW = 1
OPT_GRAPH = 2
OPT_FUSE_CLS = 3
OPT_TILED3 = 4
OPT_FUSE_ILB = 5
OPT_OVERLAP = 6
OPT_TRAIN_BF16 = 7
OPT_PW4 = 8
OPT_C3Q = 9
OPT_SLICE_LANES = 10
OPT_INPUT_GRAD = 11


class ActDesc(C.Structure):
    _fields_ = [("channels", C.c_int32), ("lvl", C.c_int32)]


class BnOff(C.Structure):
    _fields_ = [("weight", C.c_int64), ("bias", C.c_int64), ("running_mean", C.c_int64),
                ("running_var", C.c_int64), ("prelu", C.c_int64)]


class UnitDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_in", C.c_int32), ("n_out", C.c_int32),
                ("cin", C.c_int32 * MAX_BRANCH), ("cout", C.c_int32 * MAX_BRANCH),
                ("in_act", C.c_int32 * MAX_BRANCH), ("out_act", C.c_int32 * MAX_BRANCH),
                ("ksize", C.c_int32), ("stride", C.c_int32),
                ("dil_ch", C.c_int32 * NDIL), ("w_off", C.c_int64 * NDIL), ("bias_off", C.c_int64),
                ("bn", BnOff * MAX_BRANCH)]


class ActInfo(C.Structure):
    _fields_ = [("ws_offset_bytes", C.c_int64), ("channels", C.c_int32), ("height", C.c_int32),
                ("width", C.c_int32), ("batch", C.c_int32)]


def new_unit(kind: int) -> UnitDesc:
    u = UnitDesc()
    u.kind = kind
    u.ksize, u.stride, u.bias_off = 1, 1, -1
    for i in range(MAX_BRANCH):
        u.in_act[i] = -1
        u.out_act[i] = -1
        for f in ("weight", "bias", "running_mean", "running_var", "prelu"):
            setattr(u.bn[i], f, -1)
    for i in range(NDIL):
        u.w_off[i] = -1
    return u


CSF_MAX_BRANCH = 4


class TrainActInfo(C.Structure):
    _fields_ = [("act_offset_bytes", C.c_int64), ("z_offset_bytes", C.c_int64), ("grad_offset_bytes", C.c_int64 * 2),
                ("x16_offset_bytes", C.c_int64), ("n_consumers", C.c_int32), ("bf16", C.c_int32),
                ("dz_offset_bytes", C.c_int64)]


class CsfGnOff(C.Structure):
    _fields_ = [("weight", C.c_int64), ("bias", C.c_int64), ("prelu", C.c_int64)]


class CsfHeadDesc(C.Structure):
    """include/csf_hip.h: csf_head_desc."""
    _fields_ = [("n_branch", C.c_int32), ("gn_groups", C.c_int32),
                ("cin", C.c_int32 * CSF_MAX_BRANCH), ("cmid", C.c_int32 * CSF_MAX_BRANCH),
                ("ms_split", (C.c_int32 * NDIL) * CSF_MAX_BRANCH),
                ("fuse_w", C.c_int64), ("fuse_gn", CsfGnOff * CSF_MAX_BRANCH),
                ("ms_w", (C.c_int64 * NDIL) * CSF_MAX_BRANCH), ("ms_gn", CsfGnOff * CSF_MAX_BRANCH),
                ("fuse1_w", C.c_int64), ("fuse1_gn", CsfGnOff), ("cls_w", C.c_int64), ("cls_b", C.c_int64)]


def hipcc_path() -> Optional[str]:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


HIPCC_FLAGS = ("--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result")
OBJ_DIR = os.path.join(CSRC, "build")      # per-source objects + the key each was compiled under (git-ignored)


def _sha(*parts: bytes) -> str:
    h = hashlib.sha256()
    for p in parts:
        h.update(p)
    return h.hexdigest()[:16]


_BUILD_TAG = b"CSN_BUILD_SOURCES_SHA16="


def built_sources_sha16(path: Optional[str] = None) -> Optional[str]:
    """The kernel-source hash a libcsnet_hip.so was BUILT from (the string behind the exported ``csn_build_sources_sha16``), or
    None when the library is missing / predates the export.  Read from the file's bytes, not through dlopen: a library that is
    already mapped under this path would be handed back by the loader even after the file has been replaced."""
    path = path or LIB_PATH
    if not os.path.exists(path):
        return None
    with open(path, "rb") as fh:
        blob = fh.read()
    i = blob.find(_BUILD_TAG)
    if i < 0:
        return None
    return blob[i + len(_BUILD_TAG):i + len(_BUILD_TAG) + 16].decode("ascii", "replace")


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 into ``csrc/libcsnet_hip.so`` (in-tree, not installed).

    Staleness is decided by CONTENT, not by mtimes (a prebuilt library that travels with the tree must not be
    reused for other sources): the library exports the sha256 of the kernel sources it was built from
    (``csn_build_sources_sha16``, compared again by ``load()``); a mismatch rebuilds.  Every source is its own hipcc job
    (parallel), cached under csrc/build/ by the hash of (that source, every header, the flags)."""
    want = sources_sha16()
    if not force and built_sources_sha16() == want:
        return LIB_PATH
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build libcsnet_hip.so")
    os.makedirs(OBJ_DIR, exist_ok=True)
    # one builder at a time (ranks of a torchrun job, pytest workers): the objects, their keys and the link step share paths
    import fcntl
    with open(os.path.join(OBJ_DIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and built_sources_sha16() == want:     # another process built it while this one waited
                return LIB_PATH
            return _build_locked(hipcc, want, force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(hipcc: str, want: str, force: bool, verbose: bool) -> str:
    hdr = b""
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".h", ".inl")):
            hdr += f.encode() + open(os.path.join(CSRC, f), "rb").read()
    for h in ("csnet_hip.h", "csf_hip.h"):
        hdr += open(os.path.join(os.path.dirname(HERE), "include", h), "rb").read()
    flags = " ".join(HIPCC_FLAGS).encode()

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ_DIR, src[:-4] + ".o")
        key = _sha(open(os.path.join(CSRC, src), "rb").read(), hdr, flags)
        keyf = obj + ".key"
        if not force and os.path.exists(obj) and os.path.exists(keyf) and open(keyf).read() == key:
            return obj
        if os.path.exists(keyf):
            os.remove(keyf)                       # the key never describes an object that is being rewritten
        cmd = [hipcc, *HIPCC_FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
        with open(keyf + ".tmp", "w") as fh:
            fh.write(key)
        os.replace(keyf + ".tmp", keyf)
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    ido = write_build_id_object(hipcc, want, OBJ_DIR)
    tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-o", tmp] + objs + [ido]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


def write_build_id_object(hipcc: str, want: str, out_dir: str) -> str:
    """The build id (``csn_build_sources_sha16``, include/csnet_hip.h) as its own host-only translation unit, generated here:
    compiled as plain C++ (no device pass, no default offload arch).  ``csrc/dev_build.sh`` links the same unit."""
    idc = os.path.join(out_dir, "csn_build_id.cpp")
    with open(idc, "w") as fh:
        fh.write('// generated by sod100k_amd/_native.py\n'
                 f'static const char csn_build_id[] = "CSN_BUILD_SOURCES_SHA16={want}";\n'
                 'extern "C" const char* csn_build_sources_sha16(void) { return &csn_build_id[24]; }\n')
    ido = os.path.join(out_dir, "csn_build_id.o")
    subprocess.run([hipcc, "-x", "c++", "-O1", "-fPIC", "-c", idc, "-o", ido], check=True)
    return ido


def bind(lib: C.CDLL) -> C.CDLL:
    """Attach argument / result types of every entry point declared in include/csnet_hip.h."""
    lib.csn_abi_version.restype = C.c_int
    lib.csn_strerror.restype = C.c_char_p
    lib.csn_strerror.argtypes = [C.c_int]
    lib.csn_last_hip_error.restype = C.c_char_p
    lib.csn_plan_create.restype = C.c_int
    lib.csn_plan_create.argtypes = [C.POINTER(UnitDesc), C.c_int32, C.POINTER(ActDesc), C.c_int32,
                                    C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.csn_plan_destroy.restype = None
    lib.csn_plan_destroy.argtypes = [C.c_void_p]
    lib.csn_plan_set_option.restype = C.c_int
    lib.csn_plan_set_option.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.csn_plan_workspace_bytes.restype = C.c_size_t
    lib.csn_plan_workspace_bytes.argtypes = [C.c_void_p]
    lib.csn_plan_num_units.restype = C.c_int32
    lib.csn_plan_num_units.argtypes = [C.c_void_p]
    lib.csn_plan_train_act_info.restype = C.c_int
    lib.csn_plan_train_act_info.argtypes = [C.c_void_p, C.c_int32, C.POINTER(TrainActInfo)]
    lib.csn_plan_unit_in_slot.restype = C.c_int32
    lib.csn_plan_unit_in_slot.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.csn_plan_act_info.restype = C.c_int
    lib.csn_plan_act_info.argtypes = [C.c_void_p, C.c_int32, C.POINTER(ActInfo)]
    lib.csn_plan_refresh_params.restype = C.c_int
    lib.csn_plan_refresh_params.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.csn_forward.restype = C.c_int
    lib.csn_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.csn_forward_train.restype = C.c_int
    lib.csn_forward_train.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
    lib.csn_plan_enable_training.restype = C.c_int
    lib.csn_plan_enable_training.argtypes = [C.c_void_p]
    lib.csn_backward.restype = C.c_int
    lib.csn_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                 C.POINTER(C.c_float), C.c_float, C.c_void_p]
    lib.csn_bce_with_logits.restype = C.c_int
    lib.csn_bce_with_logits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.csn_adam_step.restype = C.c_int
    lib.csn_adam_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float,
                                  C.c_float, C.c_float, C.c_float, C.c_int32, C.c_void_p]
    lib.csn_saliency_u8.restype = C.c_int
    lib.csn_saliency_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.csn_normalize_nchw.restype = C.c_int
    lib.csn_normalize_nchw.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
    lib.csn_resize_normalize_nchw.restype = C.c_int
    lib.csn_resize_normalize_nchw.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p]
    lib.csn_saliency_resize_u8.restype = C.c_int
    lib.csn_saliency_resize_u8.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p]
    lib.csn_profile_bracket_us.restype = C.c_double
    lib.csn_profile_bracket_us.argtypes = [C.c_void_p]
    lib.csn_resize_bilinear.restype = C.c_int
    lib.csn_resize_bilinear.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p]
    lib.csn_val_mae.restype = C.c_int
    lib.csn_val_mae.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.csn_stream_copy.restype = C.c_int
    lib.csn_stream_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.csn_sal_hist.restype = C.c_int
    lib.csn_sal_hist.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.csn_forward_profile.restype = C.c_int
    lib.csn_forward_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int32, C.POINTER(C.c_float)]
    lib.csn_profile_num_kernels.restype = C.c_int32
    lib.csn_profile_num_kernels.argtypes = [C.c_void_p]
    lib.csn_profile_kernel.restype = C.c_int
    lib.csn_profile_kernel.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                       C.POINTER(C.c_int32)]
    lib.csn_unit_kernel_name.restype = C.c_char_p
    lib.csn_unit_kernel_name.argtypes = [C.c_void_p, C.c_int32]
    lib.csn_unit_algorithmic_bytes.restype = C.c_int64
    lib.csn_unit_algorithmic_bytes.argtypes = [C.c_void_p, C.c_int32]
    # include/csf_hip.h
    lib.csf_head_create.restype = C.c_int
    lib.csf_head_create.argtypes = [C.POINTER(CsfHeadDesc), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                    C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.csf_head_destroy.restype = None
    lib.csf_head_destroy.argtypes = [C.c_void_p]
    lib.csf_head_workspace_bytes.restype = C.c_size_t
    lib.csf_head_workspace_bytes.argtypes = [C.c_void_p]
    lib.csf_head_refresh_params.restype = C.c_int
    lib.csf_head_refresh_params.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.csf_head_forward.restype = C.c_int
    lib.csf_head_forward.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.csf_head_stage_info.restype = C.c_int
    lib.csf_head_stage_info.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.csf_head_macs.restype = C.c_int64
    lib.csf_head_macs.argtypes = [C.c_void_p]
    lib.csf_bn_act.restype = C.c_int
    lib.csf_bn_act.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                               C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    return lib


EXPORTS: Sequence[str] = (
    "csn_abi_version", "csn_strerror", "csn_last_hip_error", "csn_plan_create", "csn_plan_destroy",
    "csn_plan_set_option", "csn_plan_workspace_bytes", "csn_plan_act_info", "csn_plan_train_act_info", "csn_plan_unit_in_slot", "csn_plan_num_units", "csn_plan_refresh_params",
    "csn_forward", "csn_forward_train", "csn_plan_enable_training", "csn_backward", "csn_bce_with_logits",
           "csn_adam_step", "csn_val_mae", "csn_saliency_u8", "csn_normalize_nchw", "csn_resize_normalize_nchw", "csn_saliency_resize_u8", "csn_resize_bilinear", "csn_sal_hist", "csn_stream_copy", "csn_forward_profile", "csn_profile_num_kernels", "csn_profile_bracket_us", "csn_profile_kernel", "csn_unit_kernel_name", "csn_unit_algorithmic_bytes",
    "csf_head_create", "csf_head_destroy", "csf_head_workspace_bytes", "csf_head_refresh_params", "csf_head_forward",
    "csf_head_stage_info", "csf_head_macs", "csf_bn_act", "csn_build_sources_sha16")

_lib: Optional[C.CDLL] = None


def _share_torch_hip_runtime() -> None:
    """One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (soname libamdhip64.so.7, the name
    libcsnet_hip.so needs).  It must be mapped BEFORE our library, otherwise the loader would pull a second copy
    from /opt/rocm and torch's streams / allocations would belong to a different runtime than our launches."""
    import torch
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load() -> C.CDLL:
    """Load libcsnet_hip.so.  Raises if it has not been built: the HIP path is the only path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing.  Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  sod100k_amd has no CPU / PyTorch fallback by design.")
        _share_torch_hip_runtime()
        raw = C.CDLL(LIB_PATH)
        raw.csn_abi_version.restype = C.c_int
        got = raw.csn_abi_version()
        if got != ABI_VERSION:      # before bind(): a stale library fails here, not with an AttributeError on a new symbol
            raise RuntimeError(f"{LIB_PATH}: ABI version {got}, this package needs {ABI_VERSION} -- rebuild "
                               "(python -c 'import __graft_entry__ as g; g.build()')")
        # ... and the library must have been built from THESE kernel sources (the .so travels with the tree, git-ignored: an
        # edit without a rebuild would otherwise run stale kernels under fresh host code).  SOD100K_HIP_LIB (A/B variant builds of
        # the same sources with extra -D flags) is the developer's own responsibility.
        if not os.environ.get("SOD100K_HIP_LIB"):
            try:
                raw.csn_build_sources_sha16.restype = C.c_char_p
                built = raw.csn_build_sources_sha16().decode()
            except AttributeError:
                built = None
            want = sources_sha16()
            if built != want:
                raise RuntimeError(f"{LIB_PATH} was built from kernel sources {built}, the tree holds {want} -- rebuild "
                                   "(python -c 'import __graft_entry__ as g; g.build()')")
        _lib = bind(raw)
    return _lib


def check(lib: C.CDLL, status: int, what: str) -> None:
    if status != 0:
        msg = lib.csn_strerror(status).decode()
        hip = lib.csn_last_hip_error().decode()
        raise RuntimeError(f"{what}: {msg}" + (f" [{hip}]" if hip else ""))
