"""Does a high-priority main stream let the NMS -> logits chain run beside the forked letterbox instead of behind it?
Eager launches and graph replays, default vs high-priority main stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "seg640"
cfg = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
pool = [bench.make_inputs(cfg, 2 + 7919 * i, dev) for i in range(3)]
pipe = bench.Pipeline(cfg, pool)
pipe.size_masks()
for i in range(3):
    pipe.step(i)
torch.cuda.synchronize()
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")
high = torch.cuda.Stream(priority=-1)
K = 60
def run(main_stream, graph):
    with torch.cuda.stream(main_stream):
        gs = None
        if graph:
            gs = []
            for i in range(3):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=main_stream):
                    pipe.step(i)
                gs.append(g)
        def one(k):
            if gs: gs[k % 3].replay()
            else: pipe.step(k % 3)
        for k in range(6): one(k)
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            for k in range(K): one(k)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / K * 1e3)
    return sorted(ts)[len(ts) // 2]
for graph in (False, True):
    for name, ms in (("default main", torch.cuda.Stream()), ("high-priority main", high), ("default main", torch.cuda.Stream()), ("high-priority main", high)):
        print(f"{wl} {'graph' if graph else 'eager'} {name}: {run(ms, graph):.4f} ms/step", flush=True)
