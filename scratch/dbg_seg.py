import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'object_detectors_amd/dropin')
from object_detectors_amd.nets.yolov8 import YoloV8
from object_detectors_amd import synth
NAMES={0:'a',1:'b',2:'c',3:'d'}
for seg in (False,True):
    torch.manual_seed(7)
    m=YoloV8('n',4,seg,NAMES)
    for conv in m.head.cv3:
        conv[-1].bias.data.normal_(-2.0,0.5); conv[-1].weight.data.mul_(8.0)
    m.eval().cuda()
    x=torch.from_numpy(synth.frames_u8(np.random.default_rng(1),1,256,256)).cuda().permute(0,3,1,2).float()/255
    with torch.inference_mode(): y=m(x)
    p=y[0] if seg else y
    b=p[0,4:8].max(0).values
    print(seg, p.shape, torch.isnan(p).sum().item(), b.min().item(), b.max().item(), torch.quantile(b,0.97).item(), (b>torch.quantile(b,0.97)).sum().item())
    if seg: print('proto', y[1][-1].shape, torch.isnan(y[1][-1]).sum().item(), 'mc', p[0,8:].abs().max().item())
