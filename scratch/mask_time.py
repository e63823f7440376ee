"""Stand-alone timing of the mask stages on the seg640 bench workload (HIP events, median of 20)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import hip_ops as ops, synth, _lib
L = _lib.load()
rng = np.random.default_rng(2)
B, nc, nm, S, h0, w0 = 64, 80, 32, 640, 1080, 1920
head = torch.from_numpy(synth.head(rng, B, nc, nm, S, 40)).cuda()
proto = torch.from_numpy(synth.proto(rng, B, nm, S // 4, S // 4)).cuda()
det, count, anchor, fin = ops.nms_v8(head, 0.25, 0.45, nc=nc, finalize=dict(model_hw=(S, S), orig_hw=(h0, w0), thres=0.25))
total = int(fin["slot_base"][-1].item())
out = torch.empty((total, h0, w0), dtype=torch.uint8, device="cuda")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a,b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t=sorted(a.elapsed_time(b) for a,b in e)
    return t[len(t)//2]*1e3
ws = ops._ws.get(proto.device, "mask", int(L.lmi_mask_workspace_bytes(total, 160, 160)))
st = torch.cuda.current_stream().cuda_stream
args = (B, nm, 160, 160, det.data_ptr(), 300, 38, fin["slot_map"].data_ptr(), fin["slot_base"].data_ptr(), h0, w0, None, None)
logits = lambda: _lib.check(L.lmi_mask_logits_ex(proto.data_ptr(), *args, total, ws.data_ptr(), ws.numel(), 0, st))
ovf = torch.zeros(1, dtype=torch.int32, device="cuda")
up = lambda: _lib.check(L.lmi_mask_upsample_ex(ws.data_ptr(), *args, out.data_ptr(), 2, total, 0, ovf.data_ptr(), int(os.environ.get("LMI_FLAGS", "0")), st))
logits(); up(); torch.cuda.synchronize()
print(f"masks {total}: logits {timeit(logits):.1f} us, upsample (fill + box) {timeit(up):.1f} us")
