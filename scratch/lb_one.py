import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd import hip_ops as ops
h,w,S,B=1080,1920,640,64
f=torch.randint(0,256,(B,h,w,3),dtype=torch.uint8,device='cuda')
out=torch.empty((B,3,S,S),device='cuda')
for _ in range(3): ops.letterbox(f,(S,S),out=out)
torch.cuda.synchronize()
