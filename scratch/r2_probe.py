import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import hip_ops as ops, synth
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a,b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t=sorted(a.elapsed_time(b) for a,b in e)
    return t[len(t)//2]*1e3
rng = np.random.default_rng(1)
p = torch.from_numpy(synth.head(rng, 64, 80, 0, 640, 40)).cuda()
out = ops.nms_v8(p, 0.25, 0.45, nc=80)
print("nms_v8 B=64: %.1f us" % timeit(lambda: ops.nms_v8(p, 0.25, 0.45, nc=80, out=out)))
fin = dict(model_hw=(640,640), orig_hw=(1080,1920), thres=0.25)
r = ops.nms_v8(p, 0.25, 0.45, nc=80, finalize=fin)
fin["out"] = r[3]
print("nms_v8+fused finalize B=64: %.1f us" % timeit(lambda: ops.nms_v8(p, 0.25, 0.45, nc=80, out=r[:3], finalize=fin)))
def two():
    ops.nms_v8(p, 0.25, 0.45, nc=80, out=out)
    ops.finalize_detections(out[0], out[1], 80, (640,640), (1080,1920), 0.25, out=r[3])
print("nms_v8 then finalize_detections B=64: %.1f us" % timeit(two))
for (h,w,S,B) in [(1080,1920,640,64),(2160,3840,640,64)]:
    fs=[torch.randint(0,256,(B,h,w,3),dtype=torch.uint8,device='cuda') for _ in range(3)]
    o=torch.empty((B,3,S,S),device='cuda')
    print(f"letterbox {h}x{w} same frames: %.1f us" % timeit(lambda: ops.letterbox(fs[0],(S,S),out=o)))
    k=[0]
    def rot():
        k[0]=(k[0]+1)%3
        ops.letterbox(fs[k[0]],(S,S),out=o)
    print(f"letterbox {h}x{w} rotating 3 batches: %.1f us" % timeit(rot))
