"""Operator-list geometry helpers of the YOLO path (drop-in for the hot-path subset of the
reference's lmi_utils/gadget_utils/pipeline_utils.py; plotting and gadget file IO are out of scope).

An *operator list* records what was done to a frame before inference, in order:
  {'resize':  [resized_w, resized_h, orig_w, orig_h]}
  {'pad':     [left, right, top, bottom]}            (negative = crop)
  {'stretch': [sx, sy]}
  {'flip':    [left_right, up_down, im_w, im_h]}
`revert_*` undoes the list back to front; `apply_operations` applies it front to back.

Reference semantics followed (file:line in lmi_utils/gadget_utils/pipeline_utils.py):
  resize_image :23-68, fit_im_to_size :72-142, fit_array_to_size :145-173,
  revert_mask_to_origin :374-405, revert_masks_to_origin :408-418, revert_to_origin :422-476,
  apply_operations :479-509.  Pinned by tests/golden/l1_geometry.json (outputs of those functions).
"""
import logging

import numpy as np
import torch
from torch.nn import functional as F

logging.basicConfig()
logger = logging.getLogger(__name__)
logger.setLevel(logging.INFO)


def _to_tensor(x):
    """(tensor, kind) with kind in {'tensor','numpy','other'}; numpy arrays are wrapped, not copied."""
    if isinstance(x, torch.Tensor):
        return x, "tensor"
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x), "numpy"
    return torch.as_tensor(x), "other"


@torch.inference_mode()
def resize_image(im, W=None, H=None, mode='bilinear'):
    """Resize an (H,W) / (H,W,C) image with torch interpolation; a missing side keeps the aspect
    ratio (`int(w*H/h)`).  Integer images are interpolated in fp32 and cast back (truncation)."""
    if W is None and H is None:
        return im
    h, w = im.shape[:2]
    if W is None:
        W = int(w * H / h)
    elif H is None:
        H = int(h * W / w)
    t, kind = _to_tensor(im)
    flat = t.ndim == 2
    if flat:
        t = t[..., None]
    src_dtype, integer = t.dtype, not t.is_floating_point()
    if integer:
        t = t.float()
    out = F.interpolate(t.permute(2, 0, 1)[None], size=(H, W), mode=mode)[0].permute(1, 2, 0)
    if integer:
        out = out.to(src_dtype)
    if flat:
        out = out[..., 0]
    return out.numpy() if kind == "numpy" else out


def _axis_pads(have, want):
    """(before, after): centred pad when want >= have, else centred crop reported as negatives."""
    if want >= have:
        before = (want - have) // 2
        return before, want - have - before
    before = (have - want) // 2
    return -before, -(have - want - before)


@torch.inference_mode()
def fit_im_to_size(im, W=None, H=None, value=0):
    """Pad (constant `value`) or crop an image, centred, to (H, W).
    Returns (image, pad_left, pad_right, pad_top, pad_bottom); negative pads are crops."""
    if W is None and H is None:
        return im, 0, 0, 0, 0
    h, w = im.shape[:2]
    W = w if W is None else W
    H = h if H is None else H
    t, kind = _to_tensor(im)
    flat = t.ndim == 2
    if flat:
        t = t[..., None]
    t = t.permute(2, 0, 1)
    pl, pr = _axis_pads(w, W)
    if pr >= 0:   # a crop by one pixel is (0, -1): the sign of the second pad tells pad from crop
        t = F.pad(t, (pl, pr, 0, 0), value=value)
    else:
        t = t[:, :, -pl:w + pr]
    pt, pb = _axis_pads(h, H)
    if pb >= 0:
        t = F.pad(t, (0, 0, pt, pb), value=value)
    else:
        t = t[:, -pt:h + pb, :]
    t = t.permute(1, 2, 0)
    if flat:
        t = t[..., 0]
    return (t.numpy() if kind == "numpy" else t), pl, pr, pt, pb


def fit_array_to_size(im, W=None, H=None, value=0):
    """numpy twin of fit_im_to_size (the reference uses cv2.copyMakeBorder, BORDER_CONSTANT)."""
    h, w = im.shape[:2]
    W = w if W is None else W
    H = h if H is None else H
    pl, pr = _axis_pads(w, W)
    pt, pb = _axis_pads(h, H)
    extra = [(0, 0)] * (im.ndim - 2)
    if pr >= 0:
        im = np.pad(im, [(0, 0), (pl, pr)] + extra, constant_values=value) if (pl or pr) else im
    else:
        im = im[:, -pl:w + pr]
    if pb >= 0:
        im = np.pad(im, [(pt, pb), (0, 0)] + extra, constant_values=value) if (pt or pb) else im
    else:
        im = im[-pt:h + pb, :]
    return im, pl, pr, pt, pb


@torch.inference_mode()
def revert_mask_to_origin(mask, operations: list):
    """Undo the operator list on one mask image (H,W) or (H,W,C)."""
    numpy_in = isinstance(mask, np.ndarray)
    for op in reversed(operations):
        if 'resize' in op:
            _, _, ow, oh = op['resize']
            mask = resize_image(mask, ow, oh)
        if 'pad' in op:
            h, w = mask.shape[:2]
            left, right, top, bottom = op['pad']
            mask, *_ = fit_im_to_size(mask, w - left - right, h - top - bottom)
        if 'flip' in op:
            lr, ud = op['flip'][0], op['flip'][1]
            t = torch.from_numpy(mask) if numpy_in else mask
            dims = [d for d, on in ((1, lr), (0, ud)) if on]
            if dims:
                t = torch.flip(t, dims)
            mask = t.numpy() if numpy_in else t
    return mask


def _hip_ops():
    from object_detectors_amd import hip_ops  # raises if liblmi_hip.so is missing: no silent fallback on a GPU box
    return hip_ops


def revert_masks_to_origin(masks, operations: list):
    """Undo the operator list on a stack / list of masks; the container kind follows the input.
    A stack of fp32 / uint8 masks that lives on the GPU is reverted there, all masks per launch
    (object_detectors_amd/csrc/revert.hip); host arrays keep the reference's per-mask torch-CPU route."""
    if len(masks) == 0:
        return []
    if isinstance(masks, torch.Tensor) and masks.is_cuda and masks.ndim == 3 and masks.dtype in (torch.float32, torch.uint8):
        return _hip_ops().revert_masks(masks, operations)
    out = [revert_mask_to_origin(m, operations) for m in masks]
    if isinstance(masks[0], torch.Tensor):
        return torch.stack(out)
    return np.stack(out) if isinstance(masks, np.ndarray) else out


@torch.inference_mode()
def revert_to_origin(pts, operations: list):
    """Map Nx2 points / Nx4 boxes back to the original image: operators undone last to first, then
    round (half to even) and clamp at 0.  Returns the same container kind it was given."""
    t, kind = _to_tensor(pts)
    if t.ndim != 2 or t.shape[1] not in (2, 4):
        raise Exception(f'pts should be Nx2 or Nx4, got shape: {t.shape}')
    if kind == "tensor" and t.is_cuda and t.dtype == torch.float32:
        return _hip_ops().revert_points(t, operations)  # one launch, same fp32 arithmetic (csrc/revert.hip)
    reps = t.shape[1] // 2

    def per_xy(vals):
        return torch.tensor(vals, device=t.device).repeat(reps)[None]

    for op in reversed(operations):
        if 'resize' in op:
            tw, th, ow, oh = op['resize']
            t = t / per_xy([tw / ow, th / oh])
        elif 'pad' in op:
            t = t - per_xy([op['pad'][0], op['pad'][2]])
        elif 'stretch' in op:
            t = t / per_xy(list(op['stretch']))
        elif 'flip' in op:
            lr, ud, im_w, im_h = op['flip']
            if lr:
                t[:, 0::2] = im_w - t[:, 0::2]
            if ud:
                t[:, 1::2] = im_h - t[:, 1::2]
        else:
            raise Exception(f'unsupported operation: {op}')
    t = t.round().clamp(min=0)
    if kind == "tensor":
        return t
    return t.numpy() if kind == "numpy" else t.tolist()


def apply_operations(pts, operations: list):
    """Map points / boxes from the original image into the processed image (the inverse of
    revert_to_origin): every operator is inverted and the inverted list is reverted."""
    inverse = []
    for op in operations:
        if 'resize' in op:
            tw, th, ow, oh = op['resize']
            inverse.append({'resize': [ow, oh, tw, th]})
        elif 'pad' in op:
            inverse.append({'pad': [-p for p in op['pad']]})
        elif 'stretch' in op:
            inverse.append({'stretch': [1 / s for s in op['stretch']]})
        elif 'flip' in op:
            inverse.append({'flip': list(op['flip'])})
        else:
            raise Exception(f'unsupported operation: {op}')
    return revert_to_origin(pts, inverse[::-1])
