import os, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd import hip_ops as ops
def timeit(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a,b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t=sorted(a.elapsed_time(b) for a,b in e)
    return t[len(t)//2]
for (h,w,S,B) in [(1080,1920,640,64),(2160,3840,640,64),(1080,1920,1280,32)]:
    f=torch.randint(0,256,(B,h,w,3),dtype=torch.uint8,device='cuda')
    out=torch.empty((B,3,S,S),device='cuda')
    for dbg in (0,1,2,3):
        os.environ['LMI_LB_NT']=str(dbg&1); os.environ['LMI_LB_LDNT']=str(dbg>>1)
        t=timeit(lambda: ops.letterbox(f,(S,S),out=out))
        print(f'{h}x{w}->{S} B={B} dbg={dbg}: {t*1000:.1f} us')
    # references: pure write of the output, pure read of the source
    a=torch.empty(75_000_000,device='cuda'); b2=torch.empty_like(a)
    t=timeit(lambda: b2.copy_(a)); print(f'   float copy 300MB: {t*1000:.1f} us ({2*a.numel()*4/t/1e6:.0f} GB/s r+w)')
    t=timeit(lambda: out.zero_()); print(f'   zero out: {t*1000:.1f} us ({out.numel()*4/t/1e6:.0f} GB/s)')
    t=timeit(lambda: f.sum(dtype=torch.int64)); print(f'   sum src: {t*1000:.1f} us ({f.numel()/t/1e6:.0f} GB/s)')
