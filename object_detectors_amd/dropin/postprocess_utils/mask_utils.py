"""Drop-in for the mask-paste helper of lmi_utils/postprocess_utils/mask_utils.py (`rescale_masks`, :66-140):
same name, arguments and result (bool (N,H,W); uint8 0..255 for threshold < 0); runs in HIP kernels (object_detectors_amd/csrc/revert.hip).
The convex-hull / polygon helpers of that module (numba, cv2) are outside the accelerated path."""
import torch

import _paths  # noqa: F401
from object_detectors_amd import hip_ops


def rescale_masks(masks: torch.Tensor, boxes: torch.Tensor, image_shape, threshold: float = 0.5):
    assert masks.shape[-1] == masks.shape[-2], "Only square mask predictions are supported"
    if not isinstance(boxes, torch.Tensor):
        boxes = boxes.tensor
    if len(masks) == 0:
        return masks.new_empty((0,) + tuple(image_shape), dtype=torch.uint8)
    assert len(boxes) == len(masks), boxes.shape
    if not masks.is_cuda:
        raise RuntimeError("rescale_masks: this build pastes masks in a HIP kernel and has no CPU fallback")
    return hip_ops.paste_masks(masks, boxes.to(masks.device), tuple(image_shape), threshold)
