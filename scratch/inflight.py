"""Two batches in flight on two streams vs one: does the chip overlap the latency/ALU-bound kernels of one step
with the HBM-bound fill of the other?  usage: python scratch/inflight.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "seg640"
cfg = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
frames, head, proto = bench.make_inputs(cfg, 2, dev)
pipes = [bench.Pipeline(cfg, frames, head, proto) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
for p, s in zip(pipes, streams):
    with torch.cuda.stream(s):
        for _ in range(3):
            p.step()
torch.cuda.synchronize()
K = 60
for n in (1, 2, 3, 1, 2, 3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        with torch.cuda.stream(streams[k % n]):
            pipes[k % n].step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{wl}: {n} in flight: {dt / K * 1e3:.4f} ms/step  {cfg['batch'] * K / dt:.0f} img/s", flush=True)
