"""Letterbox kernel alone on cold frames (3 rotating source batches), HIP events, median of 30."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import hip_ops as ops
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a,b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t=sorted(a.elapsed_time(b) for a,b in e)
    return t[len(t)//2]*1e3
for (h,w,S,B) in [(1080,1920,640,64),(2160,3840,640,64),(640,640,640,64),(1080,1920,1280,32),(720,1280,640,64)]:
    fs=[torch.randint(0,256,(B,h,w,3),dtype=torch.uint8,device='cuda') for _ in range(3)]
    o=torch.empty((B,3,S,S),device='cuda')
    k=[0]
    def rot():
        k[0]=(k[0]+1)%3
        ops.letterbox(fs[k[0]],(S,S),out=o,pad_value=114,swap_rb=True)
    print(f"{B} x {h}x{w} -> {S}: {timeit(rot):.1f} us", flush=True)
    del fs
print("torch flavour (resize_image + fit_im_to_size arithmetic):")
for (h,w,S,B) in [(1080,1920,640,64),(2160,3840,640,64),(480,640,640,64)]:
    fs=[list(torch.randint(0,256,(B,h,w,3),dtype=torch.uint8,device='cuda').unbind(0)) for _ in range(3)]
    k=[0]
    def rot():
        k[0]=(k[0]+1)%3
        ops.letterbox(fs[k[0]],(S,S),pad_value=0,swap_rb=False,mode="torch")
    print(f"{B} x {h}x{w} -> {S}: {timeit(rot):.1f} us", flush=True)
    del fs
print("grey planes (LMI_SRC_GRAY):")
for (h,w,S,B) in [(1080,1920,640,64),(2160,3840,640,64),(1080,1920,1280,32)]:
    fs=[torch.randint(0,256,(B,h,w,1),dtype=torch.uint8,device='cuda') for _ in range(3)]
    o=torch.empty((B,3,S,S),device='cuda')
    k=[0]
    def rot():
        k[0]=(k[0]+1)%3
        ops.letterbox(fs[k[0]],(S,S),out=o,pad_value=114)
    print(f"{B} x {h}x{w} -> {S}: {timeit(rot):.1f} us", flush=True)
    del fs
