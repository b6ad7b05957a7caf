"""Host-side runtime around the C ABI: flat parameter arena, plan cache, workspace, launches.

PyTorch is used here for device memory and streams only (``tensor.data_ptr()``,
``torch.cuda.current_stream()``); all arithmetic happens in libcsnet_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _native as N


def _align4(n: int) -> int:
    return (n + 3) & ~3


class ParamArena:
    """All float parameters and BN running buffers of a module in ONE contiguous fp32 tensor.

    ``param.data`` / buffers become views into the arena, so ``load_state_dict``, optimizers and
    ``state_dict`` keep working on the reference's names while the kernels (and, for training, the RCCL
    gradient all-reduce) see a single flat buffer.  Offsets are in floats.
    """

    def __init__(self, module: nn.Module):
        self.module = module
        self.offsets: Dict[str, int] = {}
        self.flat: Optional[torch.Tensor] = None
        self.n_param_floats = 0
        self.rebuild()

    def rebuild(self) -> None:
        items: List[Tuple[str, torch.Tensor, object, str, bool]] = []
        for mname, mod in self.module.named_modules():
            for pname, p in mod._parameters.items():
                if p is not None:
                    items.append((f"{mname}.{pname}" if mname else pname, p, mod, pname, True))
        n_params = len(items)
        for mname, mod in self.module.named_modules():
            for bname, b in mod._buffers.items():
                if b is not None and b.is_floating_point():
                    items.append((f"{mname}.{bname}" if mname else bname, b, mod, bname, False))
        if not items:
            raise ValueError("module has no parameters")
        dev = items[0][1].device
        total = 0
        offs = []
        for k, (_, t, _, _, _) in enumerate(items):
            if k == n_params:
                self.n_param_floats = total
            offs.append(total)
            total += _align4(t.numel())
        if n_params == len(items):
            self.n_param_floats = total
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets = {}
        with torch.no_grad():
            for (name, t, mod, key, is_param), off in zip(items, offs):
                view = flat[off:off + t.numel()].view(t.shape)
                view.copy_(t.detach().to(torch.float32))
                if is_param:
                    t.data = view
                else:
                    mod._buffers[key] = view
                self.offsets[name] = off
        self.flat = flat

    def is_current(self) -> bool:
        """True if the module's tensors still alias the arena (``.cuda()``/``.to()`` break the views)."""
        flat = self.flat
        if flat is None:
            return False
        base = flat.data_ptr()
        for mname, mod in self.module.named_modules():
            for pname, p in mod._parameters.items():
                if p is None:
                    continue
                name = f"{mname}.{pname}" if mname else pname
                if p.device != flat.device or p.data_ptr() != base + 4 * self.offsets.get(name, -1):
                    return False
        return True


class Engine:
    """One compiled plan (fixed B, H, W) + its workspace."""

    def __init__(self, lib: C.CDLL, units: Sequence[N.UnitDesc], acts: Sequence[Tuple[int, int]],
                 B: int, H: int, W: int, device: torch.device, sub_batch: int = 0,
                 unit_names: Optional[Sequence[str]] = None, train: bool = False, slice_lanes: bool = False,
                 train_bf16: bool = False, input_grad: bool = False):
        self.lib = lib
        self.B, self.H, self.W = B, H, W
        self.sub_batch = sub_batch
        self.slice_lanes = bool(slice_lanes) and 0 < sub_batch < B   # batch slices side by side on the plan's stream lanes
        self.device = device
        self.unit_names = list(unit_names) if unit_names is not None else [str(i) for i in range(len(units))]
        ua = (N.UnitDesc * len(units))(*units)
        aa = (N.ActDesc * len(acts))(*[N.ActDesc(c, l) for c, l in acts])
        plan = C.c_void_p()
        N.check(lib, lib.csn_plan_create(ua, len(units), aa, len(acts), B, H, W, sub_batch, C.byref(plan)),
                "csn_plan_create")
        self.plan = plan
        self.train = bool(train)
        self.input_grad = bool(train) and bool(input_grad)
        if self.train:      # z / gradient / scratch buffers + backward weight images (before the workspace query)
            if train_bf16:  # BEFORE the training buffers are laid out: every activation-typed region gets 2-byte elements
                self.set_option(N.OPT_TRAIN_BF16, 1)
            if input_grad:  # ... and so does the gradient buffer of the image batch (autograd's x.grad, CSN_OPT_INPUT_GRAD)
                self.set_option(N.OPT_INPUT_GRAD, 1)
            N.check(lib, lib.csn_plan_enable_training(plan), "csn_plan_enable_training")
        if os.environ.get("CSN_TILED3") is not None:      # A/B switches for measurements
            self.set_option(N.OPT_TILED3, int(os.environ["CSN_TILED3"]))
        if slice_lanes:       # before the workspace query: one workspace region per concurrent slice
            self.set_option(N.OPT_SLICE_LANES, 1)
        if os.environ.get("CSN_C3Q") is not None:
            self.set_option(N.OPT_C3Q, int(os.environ["CSN_C3Q"]))
        if os.environ.get("CSN_PW4") is not None:
            self.set_option(N.OPT_PW4, int(os.environ["CSN_PW4"]))
        if os.environ.get("CSN_OVERLAP") is not None:
            self.set_option(N.OPT_OVERLAP, int(os.environ["CSN_OVERLAP"]))
        self.n_units = len(units)
        self.n_acts = len(acts)
        nbytes = int(lib.csn_plan_workspace_bytes(plan))
        self.workspace = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)

    def __del__(self):
        plan = getattr(self, "plan", None)
        if plan is not None and plan.value:
            self.lib.csn_plan_destroy(plan)
            self.plan = None

    def set_option(self, option: int, value: int) -> None:
        N.check(self.lib, self.lib.csn_plan_set_option(self.plan, option, value), "csn_plan_set_option")

    def _stream(self) -> int:
        if self.device.type == "cuda":
            return torch.cuda.current_stream(self.device).cuda_stream
        return 0

    def refresh(self, arena: torch.Tensor) -> None:
        assert arena.dtype == torch.float32 and arena.is_contiguous() and arena.device == self.workspace.device
        N.check(self.lib, self.lib.csn_plan_refresh_params(self.plan, arena.data_ptr(), arena.numel(),
                                                           self._stream()), "csn_plan_refresh_params")

    def forward(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert x.dtype == torch.float32 and x.shape[0] == self.B and tuple(x.shape[2:]) == (self.H, self.W), \
            (x.shape, x.dtype)
        x = x.contiguous()
        y = out if out is not None else torch.empty((self.B, 1, self.H, self.W), dtype=torch.float32, device=x.device)
        N.check(self.lib, self.lib.csn_forward(self.plan, x.data_ptr(), y.data_ptr(), self.workspace.data_ptr(),
                                               self._stream()), "csn_forward")
        return y

    def forward_train(self, x: torch.Tensor, arena: torch.Tensor, flop_w, penalty: torch.Tensor,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Train-mode forward: batch-stat BN (running stats in ``arena`` updated in place), penalty accumulated
        into the fp64 device scalar ``penalty``.  ``flop_w``: n_units x 3 floats (host)."""
        assert penalty.dtype == torch.float64 and penalty.numel() == 1 and penalty.device == x.device
        x = x.contiguous()
        y = out if out is not None else torch.empty((self.B, 1, self.H, self.W), dtype=torch.float32, device=x.device)
        fw = (C.c_float * (self.n_units * N.MAX_BRANCH))(*[float(v) for v in flop_w])
        N.check(self.lib, self.lib.csn_forward_train(self.plan, x.data_ptr(), y.data_ptr(), self.workspace.data_ptr(),
                                                     arena.data_ptr(), arena.numel(), fw, penalty.data_ptr(),
                                                     self._stream()), "csn_forward_train")
        return y

    def backward(self, x: torch.Tensor, dy: torch.Tensor, arena: torch.Tensor, grad: torch.Tensor, flop_w,
                 pen_scale: float) -> None:
        """Backward of the last ``forward_train``: writes every parameter gradient into ``grad`` (arena offsets)."""
        assert self.train, "engine was created without training buffers"
        assert dy.dtype == torch.float32 and dy.is_contiguous() and dy.numel() == self.B * self.H * self.W
        assert grad.dtype == torch.float32 and grad.is_contiguous() and grad.device == arena.device
        fw = (C.c_float * (self.n_units * N.MAX_BRANCH))(*[float(v) for v in flop_w])
        N.check(self.lib, self.lib.csn_backward(self.plan, x.data_ptr(), dy.data_ptr(), self.workspace.data_ptr(),
                                                arena.data_ptr(), grad.data_ptr(), grad.numel(), fw,
                                                float(pen_scale), self._stream()), "csn_backward")

    def profile(self, x: torch.Tensor, iters: int = 10):
        """Mean milliseconds per unit (HIP events on the launch stream), kernel names, algorithmic bytes."""
        y = torch.empty((self.B, 1, self.H, self.W), dtype=torch.float32, device=x.device)
        ms = (C.c_float * self.n_units)()
        N.check(self.lib, self.lib.csn_forward_profile(self.plan, x.contiguous().data_ptr(), y.data_ptr(),
                                                       self.workspace.data_ptr(), self._stream(), iters, ms),
                "csn_forward_profile")
        names = [self.lib.csn_unit_kernel_name(self.plan, u).decode() for u in range(self.n_units)]
        nbytes = [int(self.lib.csn_unit_algorithmic_bytes(self.plan, u)) for u in range(self.n_units)]
        return list(ms), names, nbytes

    def profile_bracket_us(self) -> float:
        """What an event pair added to one launch in the last ``profile`` call (already subtracted from its times)."""
        return float(self.lib.csn_profile_bracket_us(self.plan))

    def kernel_stats(self):
        """{kernel name: (ms per forward, launches per forward)} of the last ``profile`` call."""
        out = {}
        for i in range(self.lib.csn_profile_num_kernels(self.plan)):
            name, ms, n = C.c_char_p(), C.c_double(), C.c_int32()
            N.check(self.lib, self.lib.csn_profile_kernel(self.plan, i, C.byref(name), C.byref(ms), C.byref(n)),
                    "csn_profile_kernel")
            out[name.value.decode()] = (ms.value, n.value)
        return out

    def activation(self, act_id: int) -> torch.Tensor:
        """View of an internal activation of the LAST processed batch slice (debug / parity probes)."""
        info = N.ActInfo()
        N.check(self.lib, self.lib.csn_plan_act_info(self.plan, act_id, C.byref(info)), "csn_plan_act_info")
        if info.ws_offset_bytes < 0:
            raise ValueError("activation 0 is the caller's input tensor")
        n = info.batch * info.channels * info.height * info.width
        off = info.ws_offset_bytes
        return self.workspace[off:off + 4 * n].view(torch.float32).view(info.batch, info.channels, info.height,
                                                                         info.width)


    # ---- train-step probes (debug / parity): typed views of the workspace ----
    def _ws_tensor(self, off: int, act_id: int, bf16: bool) -> torch.Tensor:
        info = N.ActInfo()
        N.check(self.lib, self.lib.csn_plan_act_info(self.plan, act_id, C.byref(info)), "csn_plan_act_info")
        shape = (info.batch, info.channels, info.height, info.width)
        n = info.batch * info.channels * info.height * info.width
        if bf16:
            return self.workspace[off:off + 2 * n].view(torch.bfloat16).view(shape)
        return self.workspace[off:off + 4 * n].view(torch.float32).view(shape)

    def train_probe(self, act_id: int, what: str) -> torch.Tensor:
        """float32 copy of a train-step tensor of activation `act_id`: "act", "z" (raw conv output after the forward), "dz" (after the backward), "grad0" / "grad1" (gradient contributed by the first / second consumer)."""
        ti = N.TrainActInfo()
        N.check(self.lib, self.lib.csn_plan_train_act_info(self.plan, act_id, C.byref(ti)), "csn_plan_train_act_info")
        bf16 = bool(ti.bf16)
        off = {"act": ti.x16_offset_bytes if (act_id == 0 and bf16) else ti.act_offset_bytes, "z": ti.z_offset_bytes,
               "dz": ti.dz_offset_bytes, "grad0": ti.grad_offset_bytes[0], "grad1": ti.grad_offset_bytes[1]}[what]
        if off < 0:
            raise ValueError(f"activation {act_id} has no '{what}' buffer")
        return self._ws_tensor(off, act_id, bf16).to(torch.float32).clone()

    def n_consumers(self, act_id: int) -> int:
        ti = N.TrainActInfo()
        N.check(self.lib, self.lib.csn_plan_train_act_info(self.plan, act_id, C.byref(ti)), "csn_plan_train_act_info")
        return int(ti.n_consumers)

    def unit_in_slot(self, unit: int, branch: int) -> int:
        return int(self.lib.csn_plan_unit_in_slot(self.plan, unit, branch))


def _stream_of(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


def normalize_nchw(lib: C.CDLL, hwc: torch.Tensor) -> torch.Tensor:
    """B x H x W x 3 float images in [0,1] -> ImageNet-normalised B x 3 x H x W (test.py:68-69,86) on the device."""
    assert hwc.dtype == torch.float32 and hwc.dim() == 4 and hwc.shape[3] == 3
    hwc = hwc.contiguous()
    B, H, W, _ = hwc.shape
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=hwc.device)
    N.check(lib, lib.csn_normalize_nchw(hwc.data_ptr(), out.data_ptr(), B, H, W, _stream_of(hwc)), "csn_normalize_nchw")
    return out


def saliency_u8(lib: C.CDLL, logits: torch.Tensor) -> torch.Tensor:
    """(sigmoid(logits) * 255).astype(uint8) of test.py:92-96 on the device."""
    assert logits.dtype == torch.float32
    logits = logits.contiguous()
    out = torch.empty(logits.shape, dtype=torch.uint8, device=logits.device)
    N.check(lib, lib.csn_saliency_u8(logits.data_ptr(), out.data_ptr(), logits.numel(), _stream_of(logits)),
            "csn_saliency_u8")
    return out


def resize_normalize_nchw(lib: C.CDLL, hwc: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """test.py:72-86 on the device: B x h x w x 3 float images in [0,1] -> bilinear resize (half-pixel centres, no
    anti-aliasing: skimage's resize(mode='reflect', anti_aliasing=False)) to H x W -> normalise -> B x 3 x H x W."""
    assert hwc.dtype == torch.float32 and hwc.dim() == 4 and hwc.shape[3] == 3
    hwc = hwc.contiguous()
    B, h, w, _ = hwc.shape
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=hwc.device)
    N.check(lib, lib.csn_resize_normalize_nchw(hwc.data_ptr(), out.data_ptr(), B, h, w, H, W, _stream_of(hwc)),
            "csn_resize_normalize_nchw")
    return out


def saliency_resize_u8(lib: C.CDLL, logits: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """test.py:91-96 on the device: logits of ONE picture (H x W) -> sigmoid -> bilinear resize to its own h x w ->
    (p * 255) truncated to uint8."""
    assert logits.dtype == torch.float32
    logits = logits.contiguous()
    H, W = int(logits.shape[-2]), int(logits.shape[-1])
    assert logits.numel() == H * W
    out = torch.empty((h, w), dtype=torch.uint8, device=logits.device)
    N.check(lib, lib.csn_saliency_resize_u8(logits.data_ptr(), out.data_ptr(), H, W, h, w, _stream_of(logits)),
            "csn_saliency_resize_u8")
    return out


def resize_bilinear(lib: C.CDLL, x: torch.Tensor, Ho: int, Wo: int) -> torch.Tensor:
    """Planar float tensor [..., Hi, Wi] -> [..., Ho, Wo], bilinear with half-pixel centres (align_corners=False)."""
    assert x.dtype == torch.float32 and x.dim() >= 2
    x = x.contiguous()
    Hi, Wi = int(x.shape[-2]), int(x.shape[-1])
    planes = x.numel() // (Hi * Wi)
    out = torch.empty(tuple(x.shape[:-2]) + (Ho, Wo), dtype=torch.float32, device=x.device)
    N.check(lib, lib.csn_resize_bilinear(x.data_ptr(), out.data_ptr(), planes, Hi, Wi, Ho, Wo, _stream_of(x)),
            "csn_resize_bilinear")
    return out


def val_mae(lib: C.CDLL, logits: torch.Tensor, target: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One picture of the training caller's validation loop (train.py:262-276): logits 1 x 1 x hi x wi (or hi x wi),
    target h x w float at the picture's own size -> fp64 device scalar, ADDED to ``out`` when given."""
    assert logits.dtype == torch.float32 and target.dtype == torch.float32
    logits, target = logits.contiguous(), target.contiguous()
    hi, wi = int(logits.shape[-2]), int(logits.shape[-1])
    h, w = int(target.shape[-2]), int(target.shape[-1])
    assert logits.numel() == hi * wi and target.numel() == h * w
    if out is None:
        out = torch.zeros(1, dtype=torch.float64, device=logits.device)
    N.check(lib, lib.csn_val_mae(logits.data_ptr(), hi, wi, target.data_ptr(), h, w, out.data_ptr(), _stream_of(logits)),
            "csn_val_mae")
    return out
