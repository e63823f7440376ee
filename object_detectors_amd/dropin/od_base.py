"""Abstract interface every detector wrapper implements (drop-in for the reference's
object_detectors/od_base.py:6-37: same class name, same five abstract methods)."""
from abc import ABC, abstractmethod


class ODBase(ABC):
    """warmup / preprocess / forward / postprocess / predict -- predict chains the other three."""

    def __init__(self):
        super().__init__()

    @abstractmethod
    def warmup(self):
        ...

    @abstractmethod
    def preprocess(self):
        ...

    @abstractmethod
    def forward(self):
        ...

    @abstractmethod
    def postprocess(self):
        ...

    @abstractmethod
    def predict(self):
        ...
