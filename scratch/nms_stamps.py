import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import hip_ops as ops, synth, _lib
L = _lib.load()
for objects, cluster, tag in [(40, (5, 16), "typical"), (110, (8, 16), "dense")]:
    rng = np.random.default_rng(1)
    p = torch.from_numpy(synth.head(rng, 1, 80, 0, 640, objects, cluster=cluster)).cuda()
    acc = np.zeros(16)
    for it in range(30):
        ops.nms_v8(p, 0.25, 0.45, nc=80); torch.cuda.synchronize()
        buf = (C.c_ulonglong * 16)(); L.lmi_debug_stamps(buf)
        if it >= 10: acc += np.array(list(buf), dtype=np.float64)
    t = acc / 20
    names = ["sort", "decode boxes+extent", "(to cp)", "group", "pair tests", "fixpoint", "compaction", "emit"]
    print(tag, " ".join(f"{n}={(t[i+1]-t[i])*10/1000:.1f}us" for i, n in enumerate(names)), f"total={(t[8]-t[0])*10/1000:.1f}us")
