"""Importing this module registers every wrapper of the path with ObjectDetector (drop-in for the
reference's object_detectors/od_models.py:1-3, restricted to the YOLOv8 hot path)."""
from yolov8_lmi.model import *  # noqa: F401,F403
from ultralytics_lmi.yolo.model import *  # noqa: F401,F403
