"""v0 ultralytics wrapper `Yolov8` (drop-in for object_detectors/yolov8_lmi/model.py:41-388):
same registry keys, constructor `(weights, device='gpu', data=None, fp16=False)`, methods and
result dict; implementation in object_detectors_amd/wrapper.py (HIP kernels underneath)."""
import logging

import _paths  # noqa: F401
from od_base import ODBase
from object_detector import ObjectDetector
from object_detectors_amd.wrapper import Yolov8Core, Yolov8ObbCore, Yolov8PoseCore, to_numpy  # noqa: F401

__all__ = ["Yolov8", "Yolov8Obb", "Yolov8Pose", "to_numpy"]


@ObjectDetector.register(metadata=dict(versions=['v0'], model_names=['yolov8'], tasks=['od', 'seg'], frameworks=['ultralytics']))
class Yolov8(Yolov8Core, ODBase):
    logger = logging.getLogger(__name__)

    def __init__(self, weights: str, device='gpu', data=None, fp16=False) -> None:
        self._init(weights, device, data, fp16)


@ObjectDetector.register(metadata=dict(versions=['v0'], model_names=['yolov8'], tasks=['pose'], frameworks=['ultralytics']))
class Yolov8Pose(Yolov8PoseCore, ODBase):
    """Drop-in for object_detectors/yolov8_lmi/model.py:535-657."""
    logger = logging.getLogger(__name__)

    def __init__(self, weights: str, device='gpu', data=None, fp16=False) -> None:
        self._init(weights, device, data, fp16)


@ObjectDetector.register(metadata=dict(versions=['v0'], model_names=['yolov8'], tasks=['obb'], frameworks=['ultralytics']))
class Yolov8Obb(Yolov8ObbCore, ODBase):
    """Drop-in for object_detectors/yolov8_lmi/model.py:391-531."""
    logger = logging.getLogger(__name__)

    def __init__(self, weights: str, device='gpu', data=None, fp16=False) -> None:
        self._init(weights, device, data, fp16)
