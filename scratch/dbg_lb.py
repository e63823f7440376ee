import numpy as np, torch, sys
sys.path.insert(0,'.')
from object_detectors_amd import hip_ops as ops, synth
from oracle import cv2_resize as OC
h,w,sh,sw=33,77,96,128
rng=np.random.default_rng(h*10007+w)
frames=synth.frames_u8(rng,2,h,w)
got,_=ops.letterbox(torch.from_numpy(frames).cuda(),(sh,sw),pad_value=114,swap_rb=True)
got=got.cpu().numpy()
for b in range(2):
    ref,_=OC.letterbox_u8(frames[b],sh,sw,pad_value=114,swap_rb=True)
    bad=np.argwhere(got[b]!=ref)
    print(b,len(bad)); 
    for c,y,x in bad[:20]: print(c,y,x,got[b][c,y,x]*255,ref[c,y,x]*255)
print(OC.letterbox_geometry(h,w,sh,sw))
