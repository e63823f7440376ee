import os, sys, subprocess, json
# runs bench-like upsample timing under several knob settings (each in a fresh process: knobs are read at call time anyway)
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from object_detectors_amd import hip_ops as ops
cfg=bench.WORKLOADS['seg640']
dev=torch.device('cuda',0)
frames,head,proto=bench.make_inputs(cfg,2,dev)
pipe=bench.Pipeline(cfg,frames,head,proto)
out=pipe.step(); torch.cuda.synchronize()
def timeit(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a,b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t=sorted(a.elapsed_time(b) for a,b in e)
    return t[len(t)//2], t[0]
nbytes=pipe.mask_out.numel()
print('mask bytes', nbytes/1e9)
m,mi=timeit(lambda: pipe.mask_out.zero_()); print(f'torch zero_: med {m:.3f} ms min {mi:.3f} -> {nbytes/m/1e6:.0f} GB/s')
src=torch.empty_like(pipe.mask_out)
m,mi=timeit(lambda: pipe.mask_out.copy_(src)); print(f'torch copy_: med {m:.3f} ms -> {2*nbytes/m/1e6:.0f} GB/s (r+w)')
det,count,anchor,fin,masks=out
for rpb in (16,8,32,64,135,270):
  for nt in (0,1):
    os.environ['LMI_UP_RPB']=str(rpb); os.environ['LMI_UP_NT']=str(nt)
    def f():
        ev=(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True))
        ops.process_mask(proto,det,fin['slot_map'],fin['slot_base'],(cfg['h'],cfg['w']),pipe.capacity,torch.uint8,out=pipe.mask_out,split_events=ev)
        return ev
    for _ in range(3): f()
    torch.cuda.synchronize()
    evs=[f() for _ in range(15)]
    torch.cuda.synchronize()
    t=sorted(a.elapsed_time(b) for a,b in evs)
    print(f'rpb {rpb:4d} nt {nt}: upsample med {t[len(t)//2]:.3f} ms min {t[0]:.3f} -> {nbytes/t[len(t)//2]/1e6:.0f} GB/s')
