"""v1 ultralytics wrapper `Yolo` (drop-in for object_detectors/ultralytics_lmi/yolo/model.py:41-388):
registered for yolov8 and yolov11, constructor `(model_path, device='gpu', data=None, fp16=False,
**kwargs)`; bodies shared with the v0 class (object_detectors_amd/wrapper.py)."""
import logging

import _paths  # noqa: F401
from od_base import ODBase
from object_detector import ObjectDetector
from object_detectors_amd.wrapper import Yolov8Core, Yolov8ObbCore, Yolov8PoseCore, to_numpy  # noqa: F401

__all__ = ["Yolo", "YoloObb", "YoloPose", "to_numpy"]


@ObjectDetector.register(metadata=dict(versions=['v1'], model_names=['yolov8', 'yolov11'], tasks=['od', 'seg'], frameworks=['ultralytics']))
class Yolo(Yolov8Core, ODBase):
    logger = logging.getLogger(__name__)

    def __init__(self, model_path: str, device='gpu', data=None, fp16=False, **kwargs) -> None:
        self._init(model_path, device, data, fp16)


@ObjectDetector.register(metadata=dict(versions=['v1'], model_names=['yolov8', 'yolov11'], tasks=['pose'], frameworks=['ultralytics']))
class YoloPose(Yolov8PoseCore, ODBase):
    """Drop-in for object_detectors/ultralytics_lmi/yolo/model.py `YoloPose`."""
    logger = logging.getLogger(__name__)

    def __init__(self, model_path: str, device='gpu', data=None, fp16=False, **kwargs) -> None:
        self._init(model_path, device, data, fp16)


@ObjectDetector.register(metadata=dict(versions=['v1'], model_names=['yolov8', 'yolov11'], tasks=['obb'], frameworks=['ultralytics']))
class YoloObb(Yolov8ObbCore, ODBase):
    """Drop-in for object_detectors/ultralytics_lmi/yolo/model.py `YoloObb`."""
    logger = logging.getLogger(__name__)

    def __init__(self, model_path: str, device='gpu', data=None, fp16=False, **kwargs) -> None:
        self._init(model_path, device, data, fp16)
