"""One-off scale check: the seg640 chain at B=256 (25 GB of masks, > 2^32 bytes, ~12 k slots) equals four B=64 runs on the
same inputs, bit for bit.  usage: python scratch/big_batch_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import hip_ops as ops, synth
rng = np.random.default_rng(11)
nc, nm, S, h0, w0 = 80, 32, 640, 1080, 1920
h64 = synth.head(rng, 64, nc, nm, S, 40)
p64 = synth.proto(rng, 64, nm, S // 4, S // 4)
# four different quarters: roll the anchors / shift the prototypes so that the quarters differ
heads = [np.roll(h64, 37 * q, axis=2) for q in range(4)]
protos = [np.roll(p64, 3 * q, axis=3) for q in range(4)]
big_h = torch.from_numpy(np.concatenate(heads)).cuda(); big_p = torch.from_numpy(np.concatenate(protos)).cuda()
det, count, anchor, fin = ops.nms_v8(big_h, 0.25, 0.45, nc=nc, finalize=dict(model_hw=(S, S), orig_hw=(h0, w0), thres=0.25))
total = int(fin["slot_base"][-1].item())
masks, ov = ops.process_mask(big_p, det, fin["slot_map"], fin["slot_base"], (h0, w0), total)
torch.cuda.synchronize()
print("B=256:", total, "masks,", masks.numel() / 1e9, "GB, overflow", int(ov.item()), flush=True)
ok = True
for q in range(4):
    d, c, a, f = ops.nms_v8(big_h[64 * q:64 * q + 64].contiguous(), 0.25, 0.45, nc=nc, finalize=dict(model_hw=(S, S), orig_hw=(h0, w0), thres=0.25))
    t = int(f["slot_base"][-1].item())
    m, _ = ops.process_mask(big_p[64 * q:64 * q + 64].contiguous(), d, f["slot_map"], f["slot_base"], (h0, w0), t)
    torch.cuda.synchronize()
    s0 = int(fin["slot_base"][64 * q].item()); s1 = int(fin["slot_base"][64 * q + 64].item())
    same = (s1 - s0 == t) and torch.equal(count[64 * q:64 * q + 64], c) and torch.equal(anchor[64 * q:64 * q + 64] * (torch.arange(300, device="cuda")[None] < c[:, None]), a * (torch.arange(300, device="cuda")[None] < c[:, None])) and torch.equal(masks[s0:s1], m)
    print("quarter", q, "slots", s0, s1, "equal" if same else "DIFFERENT", "mask pixels set", int(m.sum(dtype=torch.int64).item()), flush=True)
    ok &= bool(same)
    del m
print("ok" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
