"""MI355X-native YOLOv8 pre/post-processing path (letterbox, NMS, box rescale, masks).

Host code is Python; the arithmetic runs in hand-written gfx950 HIP kernels behind the C ABI
declared in ``include/lmi_hip.h`` (``liblmi_hip.so``, loaded with ctypes).  There is no CPU
fallback: importing :mod:`object_detectors_amd.hip_ops` fails loudly when the library is missing.
"""

__version__ = "0.1.0"
