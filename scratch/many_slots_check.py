"""One-off scale check: more than 65535 mask slots in one call (B=512 dense images, small frames) equals eight B=64 runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import hip_ops as ops, synth
rng = np.random.default_rng(12)
nc, nm, S, h0, w0 = 80, 32, 640, 90, 160
h64 = synth.head(rng, 64, nc, nm, S, 200, cluster=(2, 5))
p64 = synth.proto(rng, 64, nm, S // 4, S // 4)
heads = [np.roll(h64, 37 * q, axis=2) for q in range(8)]
protos = [np.roll(p64, 3 * q, axis=3) for q in range(8)]
big_h = torch.from_numpy(np.concatenate(heads)).cuda(); big_p = torch.from_numpy(np.concatenate(protos)).cuda()
fz = dict(model_hw=(S, S), orig_hw=(h0, w0), thres=0.25)
det, count, anchor, fin = ops.nms_v8(big_h, 0.25, 0.45, nc=nc, finalize=fz)
total = int(fin["slot_base"][-1].item())
for dt in (torch.uint8, torch.float32):
    masks, ov = ops.process_mask(big_p, det, fin["slot_map"], fin["slot_base"], (h0, w0), total, dt)
    torch.cuda.synchronize()
    print(str(dt), "B=512:", total, "masks, overflow", int(ov.item()), flush=True)
    ok = total > 65535
    for q in range(8):
        d, c, a, f = ops.nms_v8(big_h[64 * q:64 * q + 64].contiguous(), 0.25, 0.45, nc=nc, finalize=fz)
        t = int(f["slot_base"][-1].item())
        m, _ = ops.process_mask(big_p[64 * q:64 * q + 64].contiguous(), d, f["slot_map"], f["slot_base"], (h0, w0), t, dt)
        torch.cuda.synchronize()
        s0 = int(fin["slot_base"][64 * q].item()); s1 = int(fin["slot_base"][64 * q + 64].item())
        same = (s1 - s0 == t) and torch.equal(masks[s0:s1], m)
        ok &= bool(same)
        if not same: print("quarter", q, "DIFFERENT", flush=True)
    print("ok" if ok else "MISMATCH", flush=True)
    if not ok: sys.exit(1)
