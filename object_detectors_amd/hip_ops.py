"""Torch-facing wrappers over the C ABI of liblmi_hip.so.

PyTorch is plumbing here (device memory, streams); every computation below happens in the HIP
kernels of `csrc/`.  All functions enqueue on the current torch stream and do not synchronise
unless they say so.  There is no CPU fallback: tensors must live on a HIP device.
"""
from __future__ import annotations

import ctypes as C
import functools
import struct
import threading
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib
from ._lib import LMI_F16, LMI_F32, LMI_U8, check, lmi_frame

_L = _lib.load()


class _WorkspaceCache:
    """One growing byte buffer per (device, stream, tag): the library itself never allocates.  At most `limit` buffers
    are held, the least recently used one is dropped first (a process that keeps creating streams would otherwise pin
    one workspace per stream handle it ever used; a dropped buffer goes back to torch's allocator, which orders its
    reuse after the work enqueued on the stream it was allocated on)."""

    def __init__(self, limit: int = 32):
        self._buf = {}
        self._limit = limit
        self._lock = threading.Lock()       # host threads that drive their own streams share this cache

    def get(self, device: torch.device, tag: str, nbytes: int) -> torch.Tensor:
        key = (device.index, torch.cuda.current_stream(device).cuda_stream, tag)
        with self._lock:
            t = self._buf.pop(key, None)
            if t is None or t.numel() < nbytes:
                t = torch.empty((max(nbytes, 256),), dtype=torch.uint8, device=device)
            self._buf[key] = t              # most recently used last (dicts keep insertion order)
            while len(self._buf) > self._limit:
                self._buf.pop(next(iter(self._buf)))
        return t


_ws = _WorkspaceCache()


class _DescriptorCache:
    """Device copies of small descriptor arrays, by content.  A serving loop that letterboxes the same frame buffers
    again (same pointers, same geometry) hands over byte-identical descriptors: the upload -- a pageable host-to-device
    copy, ~100 us, more than the kernel -- then happens once.  The first use synchronises its stream so that later calls
    may read the copy from any stream (and from inside a graph capture, where neither a copy nor a wait is allowed)."""

    def __init__(self, limit: int = 16):
        self._d = {}
        self._limit = limit
        self._lock = threading.Lock()

    def get(self, device: torch.device, raw: bytes) -> torch.Tensor:
        key = (device.index, raw)
        with self._lock:
            t = self._d.pop(key, None)
        if t is None:   # uploaded outside the lock (it synchronises a stream); two threads racing for the same new
            t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)   # content both upload it, the later entry stays
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream(device).synchronize()
        with self._lock:
            self._d[key] = t
            while len(self._d) > self._limit:
                self._d.pop(next(iter(self._d)))
        return t


_desc = _DescriptorCache()


def _stream(device=None) -> int:
    """The current torch stream of `device` (the tensors' device, not whatever device happens to be current)."""
    return torch.cuda.current_stream(device).cuda_stream


def _need_cuda(t: torch.Tensor, name: str) -> None:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{name} must be a tensor on the GPU (there is no CPU fallback for this path)")


def _dt(fp16: bool):
    return (torch.float16, LMI_F16) if fp16 else (torch.float32, LMI_F32)


def letterbox_geometry(h: int, w: int, dst_h: int, dst_w: int, stretch: bool = False):
    """(new_w, new_h, pad_l, pad_r, pad_t, pad_b) -- host arithmetic only."""
    out = (C.c_int32 * 6)()
    check(_L.lmi_letterbox_geometry(h, w, dst_h, dst_w, 1 if stretch else 0, out))
    return tuple(int(v) for v in out)


_FRAME_STRUCT = struct.Struct("<Qiiiiiiii")   # lmi_frame: src, h, w, row_stride, new_w, new_h, pad_l, pad_t, reserved
_FRAME_BYTES = _FRAME_STRUCT.size
assert _FRAME_BYTES == C.sizeof(lmi_frame)
_FRAME_PACK = _FRAME_STRUCT.pack_into


@functools.lru_cache(maxsize=4096)
def _cached_geometry(h: int, w: int, dst_h: int, dst_w: int, stretch: bool, torch_mode: bool, fit):
    """Letterbox geometry of one frame size (a batch usually holds few distinct sizes)."""
    if torch_mode:
        return torch_letterbox_geometry(h, w, dst_h, dst_w, fit)
    return letterbox_geometry(h, w, dst_h, dst_w, stretch)


def preprocess_u8(im: torch.Tensor, fp16: bool = False) -> torch.Tensor:
    """`Yolov8.preprocess` for one uint8 image (H,W) / (H,W,1) / (H,W,3) -> (1,3,H,W)."""
    _need_cuda(im, "im")
    if im.dtype != torch.uint8:
        raise TypeError("preprocess_u8 expects uint8")
    if im.ndim == 2:
        im = im.unsqueeze(-1)
    h, w, c = im.shape
    if im.stride(2) != 1 or im.stride(1) != c:
        im = im.contiguous()
    tdt, ldt = _dt(fp16)
    out = torch.empty((1, 3, h, w), dtype=tdt, device=im.device)
    check(_L.lmi_preprocess_u8(im.data_ptr(), h, w, c, im.stride(0), out.data_ptr(), ldt, _stream(im.device)))
    return out


def torch_letterbox_geometry(h: int, w: int, dst_h: int, dst_w: int, resize=None):
    """(new_w, new_h, pad_l, pad_r, pad_t, pad_b) of the reference helpers' chain `resize_image(im, H=, W=)` then
    `fit_im_to_size(im, H=dst_h, W=dst_w)` (lmi_utils/gadget_utils/pipeline_utils.py:36-39, :110-131): a missing
    side keeps the aspect ratio with `int()`; pads are centred with the smaller half first, negative = centred crop.
    resize = (H, W) as the v1 runner's `--resize h w`; None = the aspect-preserving fit into (dst_h, dst_w)."""
    rh, rw = (None, None) if resize is None else resize
    if rh is None and rw is None:
        if dst_h * w <= dst_w * h:
            rh = dst_h
        else:
            rw = dst_w
    new_w = int(rw) if rw is not None else int(w * rh / h)
    new_h = int(rh) if rh is not None else int(h * rw / w)

    def split(have, want):
        if want >= have:
            before = (want - have) // 2
            return before, want - have - before
        before = (have - want) // 2
        return -before, -(have - want - before)

    pl, pr = split(new_w, dst_w)
    pt, pb = split(new_h, dst_h)
    return new_w, new_h, pl, pr, pt, pb


def _check_lb_out(out, b, dst_h, dst_w, tdt, dev) -> None:
    if not isinstance(out, torch.Tensor) or tuple(out.shape) != (b, 3, dst_h, dst_w) or out.dtype != tdt or not out.is_contiguous() or out.device != dev:
        raise ValueError(f"letterbox: `out` must be a contiguous {tdt} tensor of shape {(b, 3, dst_h, dst_w)} on the frames' device")


def letterbox_tables(frame_hw: Tuple[int, int], dst_hw: Tuple[int, int], device, stretch: bool = False) -> torch.Tensor:
    """Plan half of the uniform cv2-flavoured letterbox (LMI_LB_TABLES_ONLY): the coefficient tables of one geometry
    -- every frame (h, w), destination dst_hw -- built once on the current stream into a buffer the CALLER owns.  Hand it
    to `letterbox(..., tables=t)` and every such call is one launch instead of two; the tables depend on the sizes
    alone (not on pixels, channel order, pad value or dtype), so one buffer serves every call of that geometry, from any
    stream that is ordered after this call, and inside HIP graphs (the buffer must outlive them)."""
    h, w = int(frame_hw[0]), int(frame_hw[1])
    dst_h, dst_w = int(dst_hw[0]), int(dst_hw[1])
    device = torch.device(device)
    if device.type != "cuda":
        raise TypeError("letterbox_tables: the tables live on the GPU (there is no CPU fallback for this path)")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    t = torch.empty((int(_L.lmi_letterbox_workspace_bytes(1, dst_h, dst_w)),), dtype=torch.uint8, device=device)
    check(_L.lmi_letterbox_u8_uniform(None, 1, h, w, 3 * w, 3 * w * h, None, dst_h, dst_w, LMI_F32, 0, 0, int(stretch),
                                      _lib.LMI_LB_TABLES_ONLY, t.data_ptr(), t.numel(), _stream(device)))
    t._lmi_tables_key = (h, w, dst_h, dst_w, bool(stretch), device.index)
    return t


def letterbox(frames: Union[torch.Tensor, Sequence[torch.Tensor]], dst_hw: Tuple[int, int], pad_value: int = 114,
              swap_rb: bool = False, fp16: bool = False, stretch: bool = False,
              out: Optional[torch.Tensor] = None, mode: str = "cv2", resize=None, torch_threads: Optional[int] = None,
              tables: Optional[torch.Tensor] = None):
    """Fused resize + centred pad + channel swap + /255 + NCHW.

    frames: a (B,H,W,3) uint8 tensor (uniform batch) or a sequence of (H,W,3) uint8 tensors of any
    sizes; grey planes go in as (B,H,W,1) / (H,W) / (H,W,1): they are resampled once and written to the three output
    planes, which is what the callers' GRAY2BGR + resize gives (run_model.py:77-78).  Returns (tensor (B,3,dst_h,dst_w), operators) where operators[b] is the list
    `[{'resize': [new_w,new_h,w,h]}, {'pad': [L,R,T,B]}]` that `revert_to_origin` consumes.
    mode 'cv2': cv2.resize INTER_LINEAR (the callers' default chain); mode 'torch': the reference helpers
    `resize_image` + `fit_im_to_size` (fp32 bilinear, truncating cast; `resize=(H, W)` as the v1 runner's
    `--resize`, either side None; pads may be negative = centred crop; pad value 0 is the helpers' default).
    torch_threads=1 reproduces what those helpers give when torch runs on exactly one thread (aten then picks
    its other CPU kernel for 3-channel frames); any other value: the >= 2-thread result.
    tables: what `letterbox_tables` returned for this geometry (uniform (B,H,W,C) batches, mode 'cv2'): the table pass
    is left out of the call (LMI_LB_TABLES_READY).
    """
    dst_h, dst_w = int(dst_hw[0]), int(dst_hw[1])
    tdt, ldt = _dt(fp16)
    if mode not in ("cv2", "torch"):
        raise ValueError(f"letterbox: unknown mode {mode!r}")
    if resize is not None and mode != "torch":
        raise ValueError("letterbox: resize= belongs to mode='torch' (the cv2 flavour derives its geometry from dst_hw)")
    if isinstance(frames, torch.Tensor):
        if frames.ndim == 2:
            frames = frames[None, :, :, None]
        elif frames.ndim == 3:
            frames = frames.unsqueeze(0)
    if mode == "torch" and isinstance(frames, torch.Tensor):
        frames = list(frames.unbind(0))   # geometry by the helpers' rules: the descriptor form carries it
    if isinstance(frames, torch.Tensor):
        _need_cuda(frames, "frames")
        if frames.dtype != torch.uint8 or frames.ndim != 4 or frames.shape[-1] not in (1, 3):
            raise TypeError("letterbox expects uint8 (B,H,W,3) or, for grey planes, (B,H,W,1)")
        ch = int(frames.shape[-1])
        if frames.stride(3) != 1 or frames.stride(2) != ch:   # frames may start at any byte and lie any distance apart
            frames = frames.contiguous()
        b, h, w, _ = frames.shape
        if out is None:
            out = torch.empty((b, 3, dst_h, dst_w), dtype=tdt, device=frames.device)
        else:
            _check_lb_out(out, b, dst_h, dst_w, tdt, frames.device)
        flags = _lib.LMI_SRC_GRAY if ch == 1 else 0
        if tables is not None:
            if getattr(tables, "_lmi_tables_key", None) != (h, w, dst_h, dst_w, bool(stretch), frames.device.index):
                raise ValueError("letterbox: `tables` were not built by letterbox_tables for this geometry and device")
            ws, flags = tables, flags | _lib.LMI_LB_TABLES_READY
        else:
            ws = _ws.get(frames.device, "letterbox", int(_L.lmi_letterbox_workspace_bytes(1, dst_h, dst_w)))
        check(_L.lmi_letterbox_u8_uniform(frames.data_ptr(), b, h, w, frames.stride(1), frames.stride(0), out.data_ptr(),
                                          dst_h, dst_w, ldt, pad_value, int(swap_rb), int(stretch), flags, ws.data_ptr(),
                                          ws.numel(), _stream(frames.device)))
        g = letterbox_geometry(h, w, dst_h, dst_w, stretch)
        ops = [[{"resize": [g[0], g[1], w, h]}, {"pad": [g[2], g[3], g[4], g[5]]}] for _ in range(b)]
        return out, ops
    if tables is not None:
        raise ValueError("letterbox: tables= serves uniform (B,H,W,C) batches in mode 'cv2'")
    frames = list(frames)
    if not frames:
        raise ValueError("letterbox: empty batch")
    # the per-frame work below is host time on the caller's critical path (64 frames: 235 us before, more than the
    # kernels): geometry from a cache keyed by the frame size, descriptors packed straight into one byte buffer
    keep = []
    nf = len(frames)
    raw = bytearray(_FRAME_BYTES * nf)
    ops = []
    ch = None
    fit = (dst_h, dst_w) if stretch else resize
    fit = None if fit is None else (fit[0], fit[1])
    for i, f in enumerate(frames):
        if not isinstance(f, torch.Tensor) or not f.is_cuda:
            _need_cuda(f, f"frames[{i}]")
        if f.ndim == 2:
            f = f.unsqueeze(-1)
        if f.dtype != torch.uint8 or f.ndim != 3 or f.shape[-1] not in (1, 3):
            raise TypeError("letterbox expects uint8 (H,W,3) frames, or grey planes (H,W) / (H,W,1)")
        h, w, c = f.shape
        if ch is None:
            ch = int(c)
        elif c != ch:
            raise ValueError("letterbox: one batch holds either colour frames or grey planes, not both")
        s0, s1, s2 = f.stride()
        if s2 != 1 or s1 != ch:
            f = f.contiguous()
            s0 = f.stride(0)
        keep.append(f)
        g = _cached_geometry(h, w, dst_h, dst_w, stretch, mode == "torch", fit)
        _FRAME_PACK(raw, _FRAME_BYTES * i, f.data_ptr(), h, w, s0, g[0], g[1], g[2], g[4], 0)
        ops.append([{"resize": [g[0], g[1], w, h]}, {"pad": [g[2], g[3], g[4], g[5]]}])
    desc = (C.c_char * len(raw)).from_buffer(raw)   # the host copy the C ABI validates (lmi_frame[nf], same layout)
    dev = keep[0].device
    ddev = _desc.get(dev, bytes(raw))
    if out is None:
        out = torch.empty((len(keep), 3, dst_h, dst_w), dtype=tdt, device=dev)
    else:
        _check_lb_out(out, len(keep), dst_h, dst_w, tdt, dev)
    ws = _ws.get(dev, "letterbox", int(_L.lmi_letterbox_workspace_bytes(len(keep), dst_h, dst_w)))
    check(_L.lmi_letterbox_u8(C.addressof(desc), ddev.data_ptr(), len(keep), out.data_ptr(), dst_h, dst_w, ldt,
                              pad_value, int(swap_rb), ((2 if torch_threads == 1 else 1) if mode == "torch" else 0) | (_lib.LMI_SRC_GRAY if ch == 1 else 0),
                              ws.data_ptr(), ws.numel(), _stream(dev)))
    ddev.record_stream(torch.cuda.current_stream(dev))
    return out, ops


def _finalize_setup(b: int, max_det: int, nc: int, dev, model_hw, orig_hw, thres, pad_round: bool, with_slots: bool, out, kpt=None):
    """Buffers + the lmi_finalize_args record shared by finalize_detections and the fused nms_v8(finalize=...).
    Returns (result dict, args struct, tensors to keep alive until the call is enqueued)."""
    if out is not None:   # reuse a previous result dict of the same shapes
        keep, keep_count = out["keep"], out["keep_count"]
        mask_slot, slot_map, slot_base = out["mask_slot"], out["slot_map"], out["slot_base"]
        if keep.shape != (b, max_det) or keep.dtype != torch.uint8 or keep.device != dev or not keep.is_contiguous():
            raise ValueError("finalize: `out` buffers do not match this call's shapes")
    else:
        keep = torch.empty((b, max_det), dtype=torch.uint8, device=dev)
        keep_count = torch.empty((b,), dtype=torch.int32, device=dev)
        mask_slot = torch.empty((b, max_det), dtype=torch.int32, device=dev) if with_slots else None
        slot_map = torch.empty((b * max_det,), dtype=torch.int32, device=dev) if with_slots else None
        slot_base = torch.empty((b + 1,), dtype=torch.int32, device=dev) if with_slots else None
    alive = []
    if isinstance(orig_hw, torch.Tensor):
        _need_cuda(orig_hw, "orig_hw")
        ohw = orig_hw.to(torch.int32).contiguous()
        if ohw.shape != (b, 2):
            raise ValueError("orig_hw must be (B,2)")
        alive.append(ohw)
        ohw_ptr, uh, uw = ohw.data_ptr(), 0, 0
    else:
        ohw_ptr, uh, uw = None, int(orig_hw[0]), int(orig_hw[1])
    if isinstance(thres, torch.Tensor):
        _need_cuda(thres, "thres")
        th = thres.to(torch.float64).contiguous()
        if th.numel() != nc:
            raise ValueError("per-class thresholds must have nc entries")
        alive.append(th)
        th_ptr, ut = th.data_ptr(), 0.0
    else:
        th_ptr, ut = None, float(thres)
    p = lambda t: t.data_ptr() if t is not None else None
    args = _lib.lmi_finalize_args(int(model_hw[0]), int(model_hw[1]), uh, uw, ohw_ptr, int(bool(pad_round)), 0, th_ptr, ut,
                                  keep.data_ptr(), keep_count.data_ptr(), p(mask_slot), p(slot_map), p(slot_base),
                                  *((int(kpt[0]), int(kpt[1]), int(kpt[2]), 0) if kpt else (0, 0, 0, 0)))
    res = {"keep": keep, "keep_count": keep_count, "mask_slot": mask_slot, "slot_map": slot_map, "slot_base": slot_base}
    return res, args, alive


def nms_v8(pred: torch.Tensor, conf_thres: float, iou_thres: float = 0.45, agnostic: bool = False, max_det: int = 300,
           nc: int = 0, max_nms: int = 30000, max_wh: float = 7680.0, out=None, rotated: bool = False,
           decode_done: Optional[torch.cuda.Event] = None, finalize: Optional[dict] = None, sweep_only: bool = False):
    """`ops.non_max_suppression` on the head output (B, 4+nc+nm, A), fp32 or fp16 (an fp16 head is read as it is,
    every value promoted to fp32 in the kernels: the result of `pred.float()` without that pass).  rotated=True:
    oriented boxes (xywh + angle as the last extra channel, ultralytics' nms_rotated on probiou), rows stay xywh.
    decode_done: an event recorded on the current stream between the candidate filter (HBM-bound) and the
    sort + suppression pass (latency-bound): work on another stream that waits for it shares the chip with the
    latter (lmi_nms_decode_ex / lmi_nms_suppress_ex instead of the one-call lmi_nms_v8_ex; same result).
    finalize: dict(model_hw=, orig_hw=, thres=[, pad_round=True, with_slots=True, out=None, kpt=(col0, nkpt, ndim)]) -- kpt:
    keypoint models, `scale_coords` on those row columns in the same launch (`Yolov8Pose.postprocess`); otherwise the arguments of
    finalize_detections: scale_boxes + clip, class threshold and survivor ranks then run in the suppression kernel's
    epilogue (one launch less, same results) and the call returns a 4th value, finalize_detections' result dict.

    sweep_only: LMI_NMS_SWEEP_ONLY -- keep the one-workgroup sorted sweep where the suppressor-list kernel would run
    first (same result; measurement and tests).

    Returns (det (B,max_det,6+nm) fp32, count (B,) int32, anchor (B,max_det) int32), all on the
    device; rows >= count[b] are unspecified.  `out` = a previous result tuple of the same shapes
    to write into instead of allocating (serving loops)."""
    _need_cuda(pred, "pred")
    if pred.dtype not in (torch.float32, torch.float16):
        pred = pred.float()
    if not pred.is_contiguous():
        pred = pred.contiguous()
    pdt = LMI_F16 if pred.dtype == torch.float16 else LMI_F32
    b, ch, a = pred.shape
    nc = nc or (ch - 4)
    nm = ch - nc - 4
    if nm < 0:
        raise ValueError(f"nc={nc} does not fit a head with {ch} channels")
    dev = pred.device
    if out is not None:
        det, count, anchor = out
        if det.shape != (b, max_det, 6 + nm) or count.shape != (b,) or anchor.shape != (b, max_det):
            raise ValueError("nms_v8: `out` buffers do not match this call's shapes")
        if det.dtype != torch.float32 or count.dtype != torch.int32 or anchor.dtype != torch.int32:
            raise ValueError("nms_v8: `out` buffers must be (float32, int32, int32)")
        if not (det.is_contiguous() and count.is_contiguous() and anchor.is_contiguous()) or det.device != dev:
            raise ValueError("nms_v8: `out` buffers must be contiguous and on the head's device")
    else:
        det = torch.empty((b, max_det, 6 + nm), dtype=torch.float32, device=dev)
        count = torch.empty((b,), dtype=torch.int32, device=dev)
        anchor = torch.empty((b, max_det), dtype=torch.int32, device=dev)
    fin_res, fin_ptr, alive = None, None, None
    if finalize is not None:
        if rotated:
            raise ValueError("nms_v8: finalize= serves axis-aligned boxes (use finalize_obb for rotated ones)")
        fin_res, fin_args, alive = _finalize_setup(b, max_det, nc, dev, finalize["model_hw"], finalize["orig_hw"], finalize["thres"],
                                                   finalize.get("pad_round", True), finalize.get("with_slots", True), finalize.get("out"),
                                                   finalize.get("kpt"))
        fin_ptr = C.addressof(fin_args)
    nbytes = int(_L.lmi_nms_workspace_bytes(b, a))
    ws = _ws.get(dev, "nms", nbytes)
    st = _stream(dev)
    if decode_done is not None:
        check(_L.lmi_nms_decode_ex(pred.data_ptr(), pdt, b, nc, nm, a, float(conf_thres), ws.data_ptr(), ws.numel(), st))
        decode_done.record()
        check(_L.lmi_nms_suppress_ex(pred.data_ptr(), pdt, b, nc, nm, a, float(iou_thres), int(agnostic), int(max_det), int(max_nms),
                                     float(max_wh), int(rotated) | (_lib.LMI_NMS_SWEEP_ONLY if sweep_only else 0), fin_ptr, det.data_ptr(),
                                     count.data_ptr(), anchor.data_ptr(), ws.data_ptr(), ws.numel(), st))
    else:
        check(_L.lmi_nms_v8_ex(pred.data_ptr(), pdt, b, nc, nm, a, float(conf_thres), float(iou_thres), int(agnostic), int(max_det),
                               int(max_nms), float(max_wh), int(rotated) | (_lib.LMI_NMS_SWEEP_ONLY if sweep_only else 0), fin_ptr, det.data_ptr(),
                               count.data_ptr(), anchor.data_ptr(), ws.data_ptr(), ws.numel(), st))
    if finalize is not None:
        return det, count, anchor, fin_res
    return det, count, anchor


def finalize_obb(det: torch.Tensor, count: torch.Tensor, nc: int, model_hw: Tuple[int, int],
                 orig_hw: Union[Tuple[int, int], torch.Tensor], thres: Union[float, torch.Tensor]):
    """regularize_rboxes + scale_boxes(xywh=True) + xywhr2xyxyxyxy + per-class threshold on rotated NMS rows.
    Returns dict(corners (B,max_det,4,2) fp32, keep (B,max_det) uint8, keep_count (B,)); det is not modified."""
    _need_cuda(det, "det")
    b, max_det, stride = det.shape
    dev = det.device
    corners = torch.empty((b, max_det, 4, 2), dtype=torch.float32, device=dev)
    keep = torch.empty((b, max_det), dtype=torch.uint8, device=dev)
    keep_count = torch.empty((b,), dtype=torch.int32, device=dev)
    if isinstance(orig_hw, torch.Tensor):
        _need_cuda(orig_hw, "orig_hw")
        ohw = orig_hw.to(torch.int32).contiguous()
        ohw_ptr, uh, uw = ohw.data_ptr(), 0, 0
    else:
        ohw_ptr, uh, uw = None, int(orig_hw[0]), int(orig_hw[1])
    if isinstance(thres, torch.Tensor):
        _need_cuda(thres, "thres")
        th = thres.to(torch.float64).contiguous()
        if th.numel() != nc:
            raise ValueError("per-class thresholds must have nc entries")
        th_ptr, ut = th.data_ptr(), 0.0
    else:
        th_ptr, ut = None, float(thres)
    check(_L.lmi_finalize_obb(det.data_ptr(), count.data_ptr(), b, max_det, stride, nc, int(model_hw[0]), int(model_hw[1]),
                              ohw_ptr, uh, uw, th_ptr, ut, corners.data_ptr(), keep.data_ptr(), keep_count.data_ptr(), _stream(det.device)))
    return {"corners": corners, "keep": keep, "keep_count": keep_count}


def finalize_detections(det: torch.Tensor, count: torch.Tensor, nc: int, model_hw: Tuple[int, int],
                        orig_hw: Union[Tuple[int, int], torch.Tensor], thres: Union[float, torch.Tensor],
                        pad_round: bool = True, with_slots: bool = True, out=None):
    """scale_boxes + clip IN PLACE on det[..., :4], per-class threshold, dense survivor ranks.

    orig_hw: (H0, W0) for every image, or an int32 device tensor (B,2).  thres: a float, or a
    float64 device tensor (nc,) of per-class thresholds.
    Returns dict(keep (B,max_det) uint8, keep_count (B,), mask_slot, slot_map, slot_base (B+1,))."""
    _need_cuda(det, "det")
    if det.dtype != torch.float32 or not det.is_contiguous() or det.ndim != 3:
        raise ValueError("finalize_detections: det must be a contiguous float32 (B,max_det,6+nm) tensor")
    b, max_det, stride = det.shape
    res, a, alive = _finalize_setup(b, max_det, nc, det.device, model_hw, orig_hw, thres, pad_round, with_slots, out)
    check(_L.lmi_finalize_detections(det.data_ptr(), count.data_ptr(), b, max_det, stride, nc, a.model_h, a.model_w, a.orig_hw,
                                     a.uniform_h0, a.uniform_w0, a.pad_round, a.thres, a.uniform_thres, a.keep, a.keep_count,
                                     a.mask_slot, a.slot_map, a.slot_base, _stream(det.device)))
    return res


def scale_coords(det: torch.Tensor, count: torch.Tensor, nkpt: int, ndim: int, model_hw: Tuple[int, int],
                 orig_hw: Union[Tuple[int, int], torch.Tensor], col0: int = 6) -> torch.Tensor:
    """`ops.scale_coords` IN PLACE on the keypoint columns det[..., col0 : col0 + nkpt*ndim] of the NMS rows."""
    _need_cuda(det, "det")
    b, max_det, stride = det.shape
    if isinstance(orig_hw, torch.Tensor):
        _need_cuda(orig_hw, "orig_hw")
        ohw = orig_hw.to(torch.int32).contiguous()
        ohw_ptr, uh, uw = ohw.data_ptr(), 0, 0
    else:
        ohw_ptr, uh, uw = None, int(orig_hw[0]), int(orig_hw[1])
    check(_L.lmi_scale_coords(det.data_ptr(), count.data_ptr(), b, max_det, stride, int(col0), int(nkpt), int(ndim),
                              int(model_hw[0]), int(model_hw[1]), ohw_ptr, uh, uw, _stream(det.device)))
    return det


_MASK_DTYPES = {torch.uint8: LMI_U8, torch.bool: LMI_U8, torch.float32: LMI_F32}


def _mask_flags(logits: str, upsample: str) -> int:
    if logits not in ("rows", "mfma") or upsample not in ("split", "single"):
        raise ValueError("process_mask: logits must be 'rows' | 'mfma', upsample 'split' | 'single'")
    return (_lib.LMI_MASK_LOGITS_MFMA if logits == "mfma" else 0) | (_lib.LMI_MASK_UPSAMPLE_SINGLE if upsample == "single" else 0)


def _check_window(mh: int, mw: int, h0: int, w0: int) -> None:
    """ultralytics' scale_masks crops the letterbox pad off the prototype window before interpolating; for an extremely
    thin frame that window is empty and F.interpolate raises -- same exception type and wording here."""
    gain = min(mh / h0, mw / w0)
    pad_x, pad_y = (mw - w0 * gain) / 2, (mh - h0 * gain) / 2
    wh_, ww_ = int(mh - pad_y) - int(pad_y), int(mw - pad_x) - int(pad_x)
    if wh_ <= 0 or ww_ <= 0:
        raise RuntimeError(f"Input and output sizes should be greater than 0, but got input (H: {max(wh_, 0)}, W: {max(ww_, 0)}) "
                           f"output (H: {h0}, W: {w0})")


def _mask_inputs(proto, det, slot_map, slot_base, logits="rows"):
    """Validated inputs -> (prototypes, flag).  The prototypes of an fp16 model (`fp16=True`) are read as they are
    (LMI_MASK_PROTO_F16: promoted to fp32 in the kernel = `protos.float()` without that pass) whenever the row kernel
    serves the shape; any other dtype is converted first."""
    _need_cuda(proto, "proto")
    half = proto.dtype == torch.float16 and logits == "rows" and proto.ndim == 4 and proto.shape[-1] % 2 == 0
    if proto.dtype != torch.float32 and not half:
        proto = proto.float()
    if not proto.is_contiguous():
        proto = proto.contiguous()
    for t, name, dt in ((det, "det", torch.float32), (slot_map, "slot_map", torch.int32), (slot_base, "slot_base", torch.int32)):
        _need_cuda(t, name)
        if t.dtype != dt or not t.is_contiguous() or t.device != proto.device:
            raise ValueError(f"process_mask: {name} must be a contiguous {dt} tensor on the prototypes' device")
    if det.ndim != 3 or det.shape[0] != proto.shape[0] or slot_base.numel() != proto.shape[0] + 1:
        raise ValueError("process_mask: det must be (B,max_det,6+nm) and slot_base (B+1,)")
    return proto, (_lib.LMI_MASK_PROTO_F16 if half else 0)


def process_mask(proto: torch.Tensor, det: torch.Tensor, slot_map: torch.Tensor, slot_base: torch.Tensor,
                 orig_hw: Tuple[int, int], capacity: int, out_dtype: torch.dtype = torch.uint8,
                 out: Optional[torch.Tensor] = None, split_events=None, before_upsample=None, logits: str = "rows",
                 upsample: str = "split"):
    """`process_mask_native` + `masks[M]` for the whole batch (uniform original size).

    Returns (masks (capacity,H0,W0) of uint8|float32 in slot order, overflow (1,) int32).
    split_events: optional (start, end) torch.cuda.Event pair recorded around the upsample stage
    alone (the two halves are then enqueued through lmi_mask_logits / lmi_mask_upsample).
    before_upsample: optional callable run between the two halves (e.g. joining another stream: nothing should run
    beside the upsample's zero fill, see profiles/r01/write_ceiling.md).
    logits='mfma' / upsample='single' select the dense MFMA product / the one-pass upsample kernel (measurement)."""
    proto, pflag = _mask_inputs(proto, det, slot_map, slot_base, logits)
    b, nm, mh, mw = proto.shape
    _, max_det, stride = det.shape
    h0, w0 = int(orig_hw[0]), int(orig_hw[1])
    _check_window(mh, mw, h0, w0)
    ldt = _MASK_DTYPES[out_dtype]   # torch.bool: the kernels' 0/1 bytes ARE bools (same storage as uint8)
    capacity = max(int(capacity), 1)
    if out is None:
        out = torch.empty((capacity, h0, w0), dtype=out_dtype, device=proto.device)
    elif tuple(out.shape) != (capacity, h0, w0) or out.dtype != out_dtype or not out.is_contiguous() or out.device != proto.device:
        raise ValueError(f"process_mask: `out` must be a contiguous {out_dtype} tensor of shape {(capacity, h0, w0)} on the prototypes' device")
    overflow = torch.empty((1,), dtype=torch.int32, device=proto.device)
    nbytes = int(_L.lmi_mask_workspace_bytes(capacity, mh, mw))
    ws = _ws.get(proto.device, "mask", nbytes)
    flags, st = _mask_flags(logits, upsample) | pflag, _stream(proto.device)
    if split_events is None and before_upsample is None:
        check(_L.lmi_process_mask_ex(proto.data_ptr(), b, nm, mh, mw, det.data_ptr(), max_det, stride, slot_map.data_ptr(),
                                     slot_base.data_ptr(), h0, w0, None, None, out.data_ptr(), ldt, capacity, 0, overflow.data_ptr(),
                                     ws.data_ptr(), ws.numel(), flags, st))
    else:
        check(_L.lmi_mask_logits_ex(proto.data_ptr(), b, nm, mh, mw, det.data_ptr(), max_det, stride, slot_map.data_ptr(),
                                    slot_base.data_ptr(), h0, w0, None, None, capacity, ws.data_ptr(), ws.numel(), flags, st))
        if before_upsample is not None:
            before_upsample()
        if split_events is not None:
            split_events[0].record()
        check(_L.lmi_mask_upsample_ex(ws.data_ptr(), b, nm, mh, mw, det.data_ptr(), max_det, stride, slot_map.data_ptr(),
                                      slot_base.data_ptr(), h0, w0, None, None, out.data_ptr(), ldt, capacity, 0, overflow.data_ptr(),
                                      flags, _stream(proto.device)))
        if split_events is not None:
            split_events[1].record()
    return out, overflow


def process_mask_ragged(proto: torch.Tensor, det: torch.Tensor, slot_map: torch.Tensor, slot_base: torch.Tensor,
                        sizes: Sequence[Tuple[int, int]], counts: Sequence[int], out_dtype: torch.dtype = torch.uint8,
                        logits: str = "rows", upsample: str = "split") -> List[torch.Tensor]:
    """`process_mask_native` + `masks[M]` for a batch of DIFFERENTLY sized originals in one launch (the reference loops
    over the images, object_detectors/yolov8_lmi/model.py:215-236).  sizes[b] = (H0, W0); counts[b] = survivors of
    image b as read from slot_base (the caller synchronised to size its outputs anyway).  The masks are packed into
    one flat buffer, every image's block starting on a 16-element boundary; returns the per-image views
    [(counts[b], H0_b, W0_b)], empty images included."""
    proto, pflag = _mask_inputs(proto, det, slot_map, slot_base, logits)
    b, nm, mh, mw = proto.shape
    _, max_det, stride = det.shape
    if len(sizes) != b or len(counts) != b:
        raise ValueError("process_mask_ragged: one size and one count per image")
    ldt = _MASK_DTYPES[out_dtype]   # torch.bool: the kernels' 0/1 bytes ARE bools (same storage as uint8)
    desc = (_lib.lmi_mask_image * b)()
    off = 0
    for i, ((h0, w0), n) in enumerate(zip(sizes, counts)):
        h0, w0, n = int(h0), int(w0), int(n)
        _check_window(mh, mw, h0, w0)
        desc[i] = _lib.lmi_mask_image(h0, w0, n, 0, off)
        off = (off + n * h0 * w0 + 15) & ~15
    total = int(sum(int(c) for c in counts))
    dev = proto.device
    flat = torch.empty((max(off, 16),), dtype=out_dtype, device=dev)
    views = [flat[d.out_offset:d.out_offset + d.n_masks * d.h0 * d.w0].view(d.n_masks, d.h0, d.w0) for d in desc]
    if total == 0:
        return views
    ddev = torch.frombuffer(bytearray(bytes(desc)), dtype=torch.uint8).to(dev)
    ws = _ws.get(dev, "mask", int(_L.lmi_mask_workspace_bytes(total, mh, mw)))
    st = _stream(dev)
    check(_L.lmi_process_mask_ex(proto.data_ptr(), b, nm, mh, mw, det.data_ptr(), max_det, stride, slot_map.data_ptr(),
                                 slot_base.data_ptr(), 0, 0, C.addressof(desc), ddev.data_ptr(), flat.data_ptr(), ldt, total, flat.numel(),
                                 None, ws.data_ptr(), ws.numel(), _mask_flags(logits, upsample) | pflag, st))
    ddev.record_stream(torch.cuda.current_stream(dev))
    return views


def point_ops(operations: Sequence[dict]):
    """Operator list -> (ctypes array of lmi_point_op, n) for lmi_revert_points_f32.  The divisor of a 'resize'
    is the Python-float ratio rounded to fp32, as `torch.tensor([tw/orig_w, th/orig_h])` does
    (lmi_utils/gadget_utils/pipeline_utils.py:447)."""
    rows = []
    for op in operations:
        if 'resize' in op:
            tw, th, ow, oh = op['resize']
            rows.append((_lib.LMI_POINT_DIV, tw / ow, th / oh))
        elif 'pad' in op:
            rows.append((_lib.LMI_POINT_SUB, op['pad'][0], op['pad'][2]))
        elif 'stretch' in op:
            rows.append((_lib.LMI_POINT_DIV, op['stretch'][0], op['stretch'][1]))
        elif 'flip' in op:
            lr, ud, im_w, im_h = op['flip']
            if lr or ud:
                kind = _lib.LMI_POINT_FLIP_XY if (lr and ud) else (_lib.LMI_POINT_FLIP_X if lr else _lib.LMI_POINT_FLIP_Y)
                rows.append((kind, im_w, im_h))
        else:
            raise Exception(f'unsupported operation: {op}')
    arr = (_lib.lmi_point_op * max(len(rows), 1))()
    for i, (k, x, y) in enumerate(rows):
        arr[i].kind, arr[i].x, arr[i].y = k, float(np.float32(x)), float(np.float32(y))
    return arr, len(rows)


def revert_points(pts: torch.Tensor, operations: Sequence[dict]) -> torch.Tensor:
    """`pipeline_utils.revert_to_origin` for an fp32 (N,2) / (N,4) tensor on the device: one launch for the whole
    array whatever the length of the operator list.  Returns a new tensor."""
    _need_cuda(pts, "pts")
    if pts.ndim != 2 or pts.shape[1] not in (2, 4):
        raise Exception(f'pts should be Nx2 or Nx4, got shape: {pts.shape}')
    if pts.dtype != torch.float32:
        raise TypeError("revert_points handles fp32 points (what the detector emits)")
    src = pts.contiguous()
    out = torch.empty_like(src)
    arr, n = point_ops(operations)
    check(_L.lmi_revert_points_f32(src.data_ptr(), src.numel() // 2, C.cast(arr, C.c_void_p), n, out.data_ptr(), _stream(src.device)))
    return out


def mask_stages(h: int, w: int, operations: Sequence[dict]):
    """Split the reverted operator list into fused stages [(fit_h, fit_w, resize_h, resize_w, lr, ud, out_h, out_w)]:
    every stage is at most one centred crop/pad, then one resize, then one flip, in that order."""
    stages, cur = [], None

    def flush():
        nonlocal cur
        if cur is not None:
            stages.append(cur)
            cur = None

    def need(level):  # 1 fit, 2 resize, 3 flip: a stage applies them in ascending order, once each
        nonlocal cur
        if cur is None or cur["level"] >= level:
            flush()
            cur = {"level": 0, "fit": (0, 0), "resize": (0, 0), "lr": False, "ud": False}
        cur["level"] = level

    for op in reversed(list(operations)):
        if 'resize' in op:
            _, _, ow, oh = op['resize']
            need(2)
            cur["resize"] = (int(oh), int(ow))
            h, w = int(oh), int(ow)
        if 'pad' in op:
            left, right, top, bottom = op['pad']
            nh, nw = h - top - bottom, w - left - right
            if nh <= 0 or nw <= 0:
                raise ValueError(f"pad {op['pad']} leaves no pixels of a {h}x{w} mask")
            need(1)
            cur["fit"] = (int(nh), int(nw))
            h, w = int(nh), int(nw)
        if 'flip' in op:
            lr, ud = bool(op['flip'][0]), bool(op['flip'][1])
            if lr or ud:
                need(3)
                cur["lr"], cur["ud"] = lr, ud
        if cur is not None:
            cur["out"] = (h, w)
    flush()
    return [(s["fit"][0], s["fit"][1], s["resize"][0], s["resize"][1], s["lr"], s["ud"], s["out"][0], s["out"][1]) for s in stages]


def revert_masks(masks: torch.Tensor, operations: Sequence[dict]) -> torch.Tensor:
    """`pipeline_utils.revert_masks_to_origin` for a stack (m,H,W) of fp32 or uint8 masks on the device: one launch
    per fused stage for ALL masks (the reference loops over masks in Python, three torch ops each)."""
    _need_cuda(masks, "masks")
    if masks.ndim != 3:
        raise ValueError(f"masks must be (m,H,W), got {tuple(masks.shape)}")
    ldt = {torch.uint8: LMI_U8, torch.float32: LMI_F32}.get(masks.dtype)
    if ldt is None:
        raise TypeError("revert_masks handles fp32 and uint8 masks")
    cur = masks.contiguous()
    m = cur.shape[0]
    for fit_h, fit_w, rs_h, rs_w, lr, ud, oh, ow in mask_stages(cur.shape[1], cur.shape[2], operations):
        out = torch.empty((m, oh, ow), dtype=cur.dtype, device=cur.device)
        check(_L.lmi_revert_masks_stage(cur.data_ptr(), ldt, m, cur.shape[1], cur.shape[2], fit_h, fit_w, rs_h, rs_w,
                                        1 if lr else 0, 1 if ud else 0, out.data_ptr(), _stream(cur.device)))
        cur = out
    return cur


def paste_masks(masks: torch.Tensor, boxes: torch.Tensor, image_shape: Tuple[int, int], threshold: float = 0.5) -> torch.Tensor:
    """`rescale_masks` (lmi_utils/postprocess_utils/mask_utils.py:66-140): (N,Hm,Wm) soft masks pasted into their xyxy
    boxes of an (H,W) image, `>= threshold`.  Returns a bool tensor (N,H,W) like the reference; for threshold < 0, like
    the reference, the sampled values themselves as uint8 (value * 255, truncated)."""
    _need_cuda(masks, "masks")
    _need_cuda(boxes, "boxes")
    if masks.ndim != 3 or masks.shape[-1] != masks.shape[-2]:
        raise AssertionError("Only square mask predictions are supported")
    n = masks.shape[0]
    if boxes.shape[0] != n:
        raise AssertionError(tuple(boxes.shape))
    h, w = int(image_shape[0]), int(image_shape[1])
    out = torch.empty((n, h, w), dtype=torch.uint8, device=masks.device)
    soft = threshold < 0    # the reference's visualisation mode: (value * 255).to(uint8) instead of a bool mask
    if n == 0:
        return out if soft else out.bool()
    m = masks.float().contiguous()
    b = boxes.float().contiguous()
    check(_L.lmi_paste_masks(m.data_ptr(), n, m.shape[1], m.shape[2], b.data_ptr(), b.shape[1], h, w, float(threshold),
                             out.data_ptr(), _stream(m.device)))
    return out if soft else out.view(torch.bool)
