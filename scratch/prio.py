"""Does a high-priority main stream keep the NMS -> logits chain from slowing down beside the forked letterbox?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "seg640"
cfg = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
frames, head, proto = bench.make_inputs(cfg, 2, dev)
pipe = bench.Pipeline(cfg, frames, head, proto)
for _ in range(3):
    pipe.step()
torch.cuda.synchronize()
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range", lo, hi)
high = torch.cuda.Stream(priority=-1)
low_side = torch.cuda.Stream(priority=0)
K = 80
def run(main_stream, side):
    old = pipe.side
    if side is not None:
        pipe.side = side
    with torch.cuda.stream(main_stream):
        for _ in range(5):
            pipe.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            pipe.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    pipe.side = old
    return dt / K * 1e3
for name, ms, sd in (("default/default", torch.cuda.current_stream(), None), ("high main / low side", high, low_side),
                     ("default/default", torch.cuda.current_stream(), None), ("high main / low side", high, low_side)):
    print(f"{wl} {name}: {run(ms, sd):.4f} ms/step", flush=True)
