"""What a bf16 MFMA for the mask logits would cost in parity ("no bf16 MFMA variant was tried against
the IoU bar").  CPU only: the logits of the bench's synthetic prototypes and coefficients are computed (a) in fp32,
(b) with both operands rounded to bf16 and fp32 accumulation -- what v_mfma_f32_32x32x16_bf16 computes -- and (c) as the
three-product split (hi*hi + hi*lo + lo*hi) that recovers most of fp32; each is upsampled and thresholded by the oracle's
process_mask_native and compared per mask with the fp32 result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import synth
from oracle import masks as OM

rng = np.random.default_rng(2)
n, nm, mh, mw, h0, w0 = 48, 32, 160, 160, 1080, 1920
proto = synth.proto(rng, 1, nm, mh, mw)[0]
coef = rng.normal(0, 0.5, (n, nm)).astype(np.float32)
cx, cy = rng.uniform(200, w0 - 200, n), rng.uniform(150, h0 - 150, n)
bw, bh = rng.uniform(60, 700, n), rng.uniform(60, 500, n)
boxes = np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1).astype(np.float32)

def bf16(x):
    return torch.from_numpy(x).bfloat16().float().numpy()

def masks_from(logits_fn):
    # process_mask_native with the contraction replaced: monkeypatch-free restatement through its pieces
    top, bottom, left, right = OM.mask_crop_window(mh, mw, h0, w0)
    lg = logits_fn().reshape(n, mh, mw)[:, top:bottom, left:right]
    up = torch.nn.functional.interpolate(torch.from_numpy(np.ascontiguousarray(lg))[None], (h0, w0), mode="bilinear", align_corners=False)[0].numpy()
    return (up * OM.crop_indicator(boxes, h0, w0)) > 0

P = proto.reshape(nm, -1)
ref = masks_from(lambda: coef @ P)
assert np.array_equal(ref, OM.process_mask_native(proto, coef, boxes, (h0, w0)) > 0)
ch, ph = bf16(coef), bf16(P)
cl, pl = bf16(coef - ch), bf16(P - ph)
for name, fn in (("bf16 x bf16, fp32 accumulate", lambda: ch @ ph),
                 ("split bf16, 3 products", lambda: ch @ ph + ch @ pl + cl @ ph)):
    got = masks_from(fn)
    ious = np.array([OM.mask_iou(got[i], ref[i]) for i in range(n) if ref[i].any()])
    diff = (got != ref).sum()
    print(f"{name}: mask IoU min {ious.min():.6f} median {np.median(ious):.6f}; masks below 0.999: {(ious < 0.999).sum()} of {len(ious)}; "
          f"pixels that differ {diff} of {ref.sum()} set ({diff / ref.sum():.2e})")
