"""Phase times inside nms_pair_lists / the list path of nms_sort_suppress (workgroup 0, wall clock 100 MHz) from a
-DLMI_NMS_STAMPS build: cp scratch/variants/lib_stamps.so object_detectors_amd/liblmi_hip.so && python scratch/nms_pair_stamps.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import hip_ops as ops, synth, _lib
L = C.CDLL(_lib.LIB_PATH)
for B, objects, cluster, nc, agn, conf, tag in [(1, 40, (5, 16), 80, True, 0.25, "typical agnostic"), (1, 110, (8, 16), 80, True, 0.25, "dense agnostic"),
                                              (16, 110, (8, 16), 80, True, 0.25, "dense agnostic B=16"), (1, 40, (5, 16), 80, True, 0.05, "conf 0.05"),
                                              (64, 40, (5, 16), 2, False, 0.25, "2 classes B=64")]:
    rng = np.random.default_rng(1)
    p = torch.from_numpy(synth.head(rng, B, nc, 0, 640, objects, cluster=cluster)).cuda()
    for so in (False, True):
        acc = np.zeros(64)
        for it in range(30):
            L.lmi_debug_stamps_clear()
            ops.nms_v8(p, conf, 0.45, nc=nc, agnostic=agn, sweep_only=so); torch.cuda.synchronize()
            buf = (C.c_ulonglong * 64)(); L.lmi_debug_stamps(buf)
            if it >= 10: acc += np.array(list(buf), dtype=np.float64)
        t = acc / 20
        us = lambda a, b: (t[b] - t[a]) * 10 / 1000 if t[a] and t[b] else float('nan')
        print(f"{tag} sweep_only={so}: K2 gather {us(10,11):.1f} boxes {us(11,12):.1f} pairs {us(12,13):.1f} tail {us(13,14):.1f} | K2 start -> K3 start {us(10,0):.1f} | "
              f"K3 gather {us(0,1):.1f} lists {us(1,20):.1f} rounds {us(20,21):.1f} compact+sort {us(21,22):.1f} overflow-> {us(20,25):.1f} | sweep: cp {us(1,2):.1f} sort {us(2,3):.1f} boxes {us(3,4):.1f} sweep {us(4,5):.1f} | emit {us(5,6):.1f} total K3 {us(0,6):.1f}")
