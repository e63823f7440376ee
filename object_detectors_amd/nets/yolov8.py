"""YOLOv8 detect / segment network in plain PyTorch (random-init or loaded from our own checkpoint
format).  This is the `forward` of the path (object_detectors/yolov8_lmi/model.py:75-77 calls
`AutoBackend.__call__`); it stays PyTorch-ROCm and is NOT part of the pre+post metric.  It exists
so the wrapper can run end to end without the `ultralytics` package and so tests/bench can feed
the post-processing kernels a head output of the real architecture and shape.

Output contract consumed by `postprocess` (model.py:198-209), same as ultralytics in eval mode:
  detect : Tensor (B, 4+nc, A)
  segment: (Tensor (B, 4+nc+32, A), (feature list, mask coefficients, proto (B,32,S/4,S/4)))
with boxes as cx,cy,w,h in input pixels and sigmoid class scores.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn as nn

# depth multiple, width multiple, max channels
SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768), "l": (1.0, 1.0, 512), "x": (1.0, 1.25, 512)}


class ConvBNAct(nn.Module):
    def __init__(self, c1, c2, k=1, s=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)
        self.act = nn.SiLU(inplace=True) if act else nn.Identity()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class Bottleneck(nn.Module):
    def __init__(self, c, shortcut=True):
        super().__init__()
        self.cv1 = ConvBNAct(c, c, 3)
        self.cv2 = ConvBNAct(c, c, 3)
        self.add = shortcut

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y


class C2f(nn.Module):
    def __init__(self, c1, c2, n=1, shortcut=False):
        super().__init__()
        self.c = c2 // 2
        self.cv1 = ConvBNAct(c1, 2 * self.c, 1)
        self.cv2 = ConvBNAct((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, shortcut) for _ in range(n))

    def forward(self, x):
        y = list(self.cv1(x).chunk(2, 1))
        for m in self.m:
            y.append(m(y[-1]))
        return self.cv2(torch.cat(y, 1))


class SPPF(nn.Module):
    def __init__(self, c1, c2, k=5):
        super().__init__()
        self.cv1 = ConvBNAct(c1, c1 // 2, 1)
        self.cv2 = ConvBNAct(c1 // 2 * 4, c2, 1)
        self.m = nn.MaxPool2d(k, 1, k // 2)

    def forward(self, x):
        x = self.cv1(x)
        y1 = self.m(x)
        y2 = self.m(y1)
        return self.cv2(torch.cat((x, y1, y2, self.m(y2)), 1))


class Proto(nn.Module):
    def __init__(self, c1, c_=256, c2=32):
        super().__init__()
        self.cv1 = ConvBNAct(c1, c_, 3)
        self.upsample = nn.ConvTranspose2d(c_, c_, 2, 2, 0, bias=True)
        self.cv2 = ConvBNAct(c_, c_, 3)
        self.cv3 = ConvBNAct(c_, c2, 1)

    def forward(self, x):
        return self.cv3(self.cv2(self.upsample(self.cv1(x))))


class DetectHead(nn.Module):
    """Anchor-free head: per level a box branch (4*reg_max DFL bins) and a class branch."""
    reg_max = 16

    def __init__(self, nc, ch, nm=0, task="detect", kpt_shape=(17, 3)):
        super().__init__()
        # nm extra channels per anchor: 32 mask coefficients (segment), nkpt*ndim keypoint values (pose), 1 angle (obb)
        self.nc, self.nm, self.nl = nc, nm, len(ch)
        self.task, self.kpt_shape = task, tuple(kpt_shape)
        c2, c3 = max(16, ch[0] // 4, self.reg_max * 4), max(ch[0], min(nc, 100))
        self.cv2 = nn.ModuleList(nn.Sequential(ConvBNAct(c, c2, 3), ConvBNAct(c2, c2, 3), nn.Conv2d(c2, 4 * self.reg_max, 1)) for c in ch)
        self.cv3 = nn.ModuleList(nn.Sequential(ConvBNAct(c, c3, 3), ConvBNAct(c3, c3, 3), nn.Conv2d(c3, nc, 1)) for c in ch)
        if nm:
            c4 = max(ch[0] // 4, nm)
            if task == "segment":
                self.proto = Proto(ch[0], 256 if ch[0] >= 64 else max(ch[0], 32), nm)
            self.cv4 = nn.ModuleList(nn.Sequential(ConvBNAct(c, c4, 3), ConvBNAct(c4, c4, 3), nn.Conv2d(c4, nm, 1)) for c in ch)
        self.strides = (8, 16, 32)
        self.register_buffer("bins", torch.arange(self.reg_max, dtype=torch.float32), persistent=False)

    def bias_init(self):
        for a, b, s in zip(self.cv2, self.cv3, self.strides):
            a[-1].bias.data[:] = 1.0
            b[-1].bias.data[: self.nc] = math.log(5 / self.nc / (640 / s) ** 2)

    def forward(self, feats):
        bs = feats[0].shape[0]
        box, cls, anchors, strides = [], [], [], []
        for i, f in enumerate(feats):
            h, w = f.shape[2:]
            box.append(self.cv2[i](f).view(bs, 4 * self.reg_max, -1))
            cls.append(self.cv3[i](f).view(bs, self.nc, -1))
            sy, sx = torch.meshgrid(torch.arange(h, device=f.device, dtype=f.dtype) + 0.5,
                                    torch.arange(w, device=f.device, dtype=f.dtype) + 0.5, indexing="ij")
            anchors.append(torch.stack((sx, sy), 0).view(2, -1))
            strides.append(torch.full((1, h * w), float(self.strides[i]), device=f.device, dtype=f.dtype))
        box, cls = torch.cat(box, 2), torch.cat(cls, 2)
        anchors, strides = torch.cat(anchors, 1), torch.cat(strides, 1)
        # DFL: expectation over the bins, then ltrb distances -> cx,cy,w,h
        dist = box.view(bs, 4, self.reg_max, -1).softmax(2)
        dist = (dist * self.bins.to(dist.dtype).view(1, 1, -1, 1)).sum(2)
        lt, rb = dist.chunk(2, 1)
        x1y1, x2y2 = anchors.unsqueeze(0) - lt, anchors.unsqueeze(0) + rb
        dbox = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1) * strides
        y = torch.cat((dbox, cls.sigmoid()), 1)
        if not self.nm:
            return y
        mc = torch.cat([self.cv4[i](f).view(bs, self.nm, -1) for i, f in enumerate(feats)], 2)
        if self.task == "pose":     # keypoints: (raw*2 + anchor - 0.5) * stride, visibility through a sigmoid
            nk, nd = self.kpt_shape
            k = mc.view(bs, nk, nd, -1)
            xy = (k[:, :, :2] * 2.0 + (anchors.view(1, 1, 2, -1) - 0.5)) * strides.view(1, 1, 1, -1)
            k = torch.cat((xy, k[:, :, 2:].sigmoid()), 2) if nd == 3 else xy
            return torch.cat((y, k.reshape(bs, nk * nd, -1)), 1), (feats, mc)
        if self.task == "obb":      # angle in [-pi/4, 3pi/4)
            angle = (mc.sigmoid() - 0.25) * math.pi
            return torch.cat((y, angle), 1), (feats, angle)
        p = self.proto(feats[0])
        return torch.cat((y, mc), 1), (feats, mc, p)


class YoloV8(nn.Module):
    def __init__(self, scale="n", nc=80, seg=False, names: Dict[int, str] = None, task: str = None, kpt_shape=(17, 3)):
        super().__init__()
        task = task or ("segment" if seg else "detect")
        if task not in ("detect", "segment", "pose", "obb"):
            raise ValueError(f"unknown task {task}")
        seg = task == "segment"
        d, w, mc = SCALES[scale]
        ch = lambda c: int(math.ceil(min(c, mc) * w / 8) * 8)
        n = lambda k: max(round(k * d), 1)
        c1, c2, c3, c4, c5 = ch(64), ch(128), ch(256), ch(512), ch(1024)
        self.b0, self.b1 = ConvBNAct(3, c1, 3, 2), ConvBNAct(c1, c2, 3, 2)
        self.b2 = C2f(c2, c2, n(3), True)
        self.b3 = ConvBNAct(c2, c3, 3, 2)
        self.b4 = C2f(c3, c3, n(6), True)
        self.b5 = ConvBNAct(c3, c4, 3, 2)
        self.b6 = C2f(c4, c4, n(6), True)
        self.b7 = ConvBNAct(c4, c5, 3, 2)
        self.b8 = C2f(c5, c5, n(3), True)
        self.b9 = SPPF(c5, c5)
        self.up = nn.Upsample(scale_factor=2, mode="nearest")
        self.n12 = C2f(c5 + c4, c4, n(3))
        self.n15 = C2f(c4 + c3, c3, n(3))
        self.n16 = ConvBNAct(c3, c3, 3, 2)
        self.n18 = C2f(c3 + c4, c4, n(3))
        self.n19 = ConvBNAct(c4, c4, 3, 2)
        self.n21 = C2f(c4 + c5, c5, n(3))
        extra = {"detect": 0, "segment": 32, "pose": kpt_shape[0] * kpt_shape[1], "obb": 1}[task]
        self.head = DetectHead(nc, (c3, c4, c5), extra, task, kpt_shape)
        self.head.bias_init()
        self.nc, self.seg, self.scale, self.task, self.kpt_shape = nc, seg, scale, task, list(kpt_shape)
        self.names = names or {i: f"class{i}" for i in range(nc)}
        self.fp16 = False
        self.stride = 32

    def forward(self, x):
        x = self.b2(self.b1(self.b0(x)))
        p3 = self.b4(self.b3(x))
        p4 = self.b6(self.b5(p3))
        p5 = self.b9(self.b8(self.b7(p4)))
        n4 = self.n12(torch.cat((self.up(p5), p4), 1))
        n3 = self.n15(torch.cat((self.up(n4), p3), 1))
        m4 = self.n18(torch.cat((self.n16(n3), n4), 1))
        m5 = self.n21(torch.cat((self.n19(m4), p5), 1))
        return self.head([n3, m4, m5])


def save_checkpoint(path: str, scale="n", nc=80, seg=False, names=None, seed=0, task=None, kpt_shape=(17, 3)) -> None:
    """Random-init weights in this package's checkpoint format (tensors + plain metadata only, so it
    loads with torch.load(weights_only=True))."""
    torch.manual_seed(seed)
    m = YoloV8(scale, nc, seg, names, task, kpt_shape)
    torch.save({"format": "lmi-yolov8-torch", "scale": scale, "nc": nc, "seg": bool(m.seg), "task": m.task,
                "kpt_shape": [int(v) for v in m.kpt_shape],
                "names": {int(k): str(v) for k, v in m.names.items()}, "state_dict": m.state_dict()}, path)


def load_checkpoint(path: str, device, fp16=False) -> YoloV8:
    ck = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(ck, dict) or ck.get("format") != "lmi-yolov8-torch":
        raise ValueError(f"{path} is not an lmi-yolov8-torch checkpoint (ultralytics .pt files need the ultralytics package)")
    m = YoloV8(ck["scale"], ck["nc"], ck["seg"], ck["names"], ck.get("task"), tuple(ck.get("kpt_shape", (17, 3))))
    m.load_state_dict(ck["state_dict"])
    m.eval().to(device)
    if fp16:
        m.half()
    m.fp16 = bool(fp16)
    return m
