"""Candidate filter alone (lmi_nms_decode_ex), cold heads (3 rotating buffers), HIP events, median of 40.
   python scratch/decode_time.py            (the library's own choice of waves per workgroup)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd import hip_ops as ops, _lib
L = _lib.load()
dev = torch.device("cuda", 0)

def med(fn, n=40):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2] * 1e3

for B, C, A in ((64, 84, 8400), (64, 116, 8400), (32, 116, 33600), (16, 116, 8400), (1, 84, 8400), (96, 84, 8400)):
    heads = [torch.rand((B, C, A), device=dev) * 0.2 for _ in range(3)]
    nc = 80
    ws = torch.empty(int(L.lmi_nms_workspace_bytes(B, A)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    k = [0]
    def run():
        k[0] = (k[0] + 1) % 3
        rc = L.lmi_nms_decode_ex(heads[k[0]].data_ptr(), 0, B, nc, C - 4 - nc, A, 0.25, ws.data_ptr(), ws.numel(), st)
        assert rc == 0, rc
    us = med(run)
    byt = B * (nc * A * 4)
    print(f"B={B} C={C} A={A}: {us:.1f} us  class rows {byt/1e6:.0f} MB -> {byt/us/1e6:.2f} TB/s", flush=True)
