"""Phase times inside nms_sort_suppress (workgroup 0, wall clock 100 MHz) from a -DLMI_NMS_STAMPS build
(scratch/variants/lib_stamps.so copied over the library on the box: `cp scratch/variants/lib_stamps.so
object_detectors_amd/liblmi_hip.so && python scratch/nms_stamps.py`)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import hip_ops as ops, synth, _lib
L = C.CDLL(_lib.LIB_PATH)
for objects, cluster, nc, agn, tag in [(40, (5, 16), 80, False, "typical per-class"), (40, (5, 16), 80, True, "typical agnostic"),
                                       (110, (8, 16), 80, False, "dense per-class"), (110, (8, 16), 80, True, "dense agnostic"),
                                       (100, (5, 16), 2, False, "dense 2 classes")]:
    rng = np.random.default_rng(1)
    p = torch.from_numpy(synth.head(rng, 1, nc, 0, 640, objects, cluster=cluster)).cuda()
    acc = np.zeros(64)
    for it in range(30):
        L.lmi_debug_stamps_clear()
        ops.nms_v8(p, 0.25, 0.45, nc=nc, agnostic=agn); torch.cuda.synchronize()
        buf = (C.c_ulonglong * 64)(); L.lmi_debug_stamps(buf)
        if it >= 10: acc += np.array(list(buf), dtype=np.float64)
    t = acc / 20
    us = lambda a, b: (t[b] - t[a]) * 10 / 1000
    print(f"{tag}: gather {us(0,1):.1f}  class-parallel {us(1,2):.1f}  sort {us(2,3):.1f}  boxes {us(3,4):.1f}  sweep {us(4,5):.1f}  emit {us(5,6):.1f}  total {us(0,6):.1f} us  (sort / boxes / sweep only on the sorted path)")
