#!/usr/bin/env python3
"""bench.py -- pre+post images/sec of the YOLOv8 path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input that is already resident
in HBM: letterbox(64 x 1080p frames) -> NMS(head) -> scale/clip/threshold -> process_mask(proto).
The network forward is excluded (it is not part of the pre+post metric); `segments` are not
computed (return_segments=False).  Weak scaling: every rank processes its own batch per step, with
no collective inside the timed region; RCCL only reduces the timings / counters afterwards.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

The timed region replays one captured HIP graph per step (`--eager` launches every kernel from the host instead:
seg640 1.273 vs 1.298 ms/step); the roofline kernel is then timed with HIP events in an eager pass of the same steps
right after the timed region.  Prints ONE JSON line on rank 0 (see the keys below); diagnostics go to stderr.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (batch, frame_h, frame_w, model size, nc, nm, objects/img, description)
    "seg640": dict(batch=64, h=1080, w=1920, size=640, nc=80, nm=32, objects=40,
                   desc="YOLOv8-seg 640 bs64, synthetic 1080p frames, head (64,116,8400), proto (64,32,160,160)"),
    "det640": dict(batch=64, h=1080, w=1920, size=640, nc=80, nm=0, objects=40,
                   desc="YOLOv8s detect 640 bs64, synthetic 1080p frames, head (64,84,8400)"),
    "seg1280": dict(batch=32, h=1080, w=1920, size=1280, nc=80, nm=32, objects=60,
                    desc="YOLOv8x-seg 1280 bs32, synthetic 1080p frames, head (32,116,33600), proto (32,32,320,320)"),
    "det640_4k": dict(batch=64, h=2160, w=3840, size=640, nc=80, nm=0, objects=40,
                      desc="YOLOv8m detect 640, synthetic 4K frames, 64 per step per GPU"),
    "seg640_dense": dict(batch=16, h=1080, w=1920, size=640, nc=80, nm=32, objects=110, cluster=(8, 16),
                         desc="YOLOv8l-seg 640 dense scenes (>= 1000 candidates per image), 16 per step per GPU, head (16,116,8400)"),
}
CONF, IOU, MAX_DET, MASK_CONF = 0.25, 0.45, 300, 0.25
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_inputs(cfg, seed, device):
    from object_detectors_amd import synth
    rng = np.random.default_rng(seed)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    frames = torch.randint(0, 256, (cfg["batch"], cfg["h"], cfg["w"], 3), dtype=torch.uint8, device=device, generator=g)
    head = torch.from_numpy(synth.head(rng, cfg["batch"], cfg["nc"], cfg["nm"], cfg["size"], cfg["objects"],
                                       cluster=cfg.get("cluster", (5, 16)))).to(device)
    proto = None
    if cfg["nm"]:
        proto = torch.from_numpy(synth.proto(rng, cfg["batch"], cfg["nm"], cfg["size"] // 4, cfg["size"] // 4)).to(device)
    return frames, head, proto


class Pipeline:
    """The hot path on one GPU, with every output buffer preallocated after the first pass."""

    def __init__(self, cfg, frames, head, proto):
        from object_detectors_amd import hip_ops
        self.ops, self.cfg = hip_ops, cfg
        self.frames, self.head, self.proto = frames, head, proto
        self.S = cfg["size"]
        self.lb_out = torch.empty((cfg["batch"], 3, self.S, self.S), dtype=torch.float32, device=frames.device)
        self.mask_out = None
        self.capacity = 0
        self.nms_out = None
        self.fin_out = None
        self.side = torch.cuda.Stream(device=frames.device)
        self.decode_done = torch.cuda.Event()

    def step(self, time_upsample=None, time_letterbox=None, fork=True):
        """One pass of the hot path over the batch.  The letterbox and the NMS chain do not depend on each other.
        Side by side from the start they gain nothing (both the letterbox and the NMS candidate filter are
        HBM-bound, and the sort + suppression workgroups that follow starve behind the letterbox's), so the
        letterbox is started on a second stream at the moment the candidate filter is done: the latency-bound
        sort + suppression pass (one workgroup per image) then shares the chip with it (det640: 0.181 -> 0.169 ms,
        seg640: 1.312 -> 1.284 ms per step; scratch/fork_sort.py).  fork=False: everything on one stream."""
        ops, cfg = self.ops, self.cfg
        main = torch.cuda.current_stream()
        if not fork:
            self._letterbox(time_letterbox)
            self.nms_out = ops.nms_v8(self.head, CONF, IOU, nc=cfg["nc"], max_det=MAX_DET, out=self.nms_out)
        else:
            self.nms_out = ops.nms_v8(self.head, CONF, IOU, nc=cfg["nc"], max_det=MAX_DET, out=self.nms_out, decode_done=self.decode_done)
            self.side.wait_event(self.decode_done)
            with torch.cuda.stream(self.side):
                self._letterbox(time_letterbox)
        return self._post(time_upsample, join=(lambda: main.wait_stream(self.side)) if fork else None)

    def _letterbox(self, time_letterbox):
        if time_letterbox is not None:
            time_letterbox[0].record()
        self.ops.letterbox(self.frames, (self.S, self.S), pad_value=114, swap_rb=True, out=self.lb_out)
        if time_letterbox is not None:
            time_letterbox[1].record()

    def _post(self, time_upsample=None, join=None):
        ops, cfg = self.ops, self.cfg
        det, count, anchor = self.nms_out
        self.fin_out = fin = ops.finalize_detections(det, count, cfg["nc"], (self.S, self.S), (cfg["h"], cfg["w"]), MASK_CONF, out=self.fin_out)
        masks = None
        if join is not None and self.proto is None:
            join()
        if self.proto is not None:
            if self.mask_out is None:  # first pass: size the mask buffer from the (deterministic) survivor count
                self.capacity = int(fin["slot_base"][-1].item())
                self.mask_out = torch.empty((max(self.capacity, 1), cfg["h"], cfg["w"]), dtype=torch.uint8, device=self.frames.device)
            masks, ovf = ops.process_mask(self.proto, det, fin["slot_map"], fin["slot_base"], (cfg["h"], cfg["w"]),
                                          self.capacity, torch.uint8, out=self.mask_out, split_events=time_upsample,
                                          # the side stream joins BEFORE the upsample: nothing may run beside the zero fill of
                                          # the masks, whose write pattern collapses under concurrent traffic
                                          # (profiles/r01/write_ceiling.md)
                                          before_upsample=join)
        return det, count, anchor, fin, masks


def cpu_baseline(cfg, frames, head, proto, n_img, threads, max_seconds=20.0):
    """The oracle (CPU restatement of the reference path) on the first n_img images of the workload."""
    from oracle import boxes as OB, cv2_resize as OC, masks as OM, nms as ON
    torch.set_num_threads(threads)
    f = frames[:n_img].cpu().numpy()
    p = head[:n_img].cpu().numpy()
    pr = proto[:n_img].cpu().numpy() if proto is not None else None
    S = cfg["size"]
    t0 = time.perf_counter()
    ndet = 0
    done = 0
    for i in range(n_img):
        if time.perf_counter() - t0 > max_seconds:
            break
        done += 1
        OC.letterbox_u8(f[i], S, S, pad_value=114, swap_rb=True)
        (det, _), = ON.non_max_suppression(p[i:i + 1], CONF, IOU, nc=cfg["nc"], max_det=MAX_DET)
        if len(det):
            boxes = OB.scale_boxes((S, S), det[:, :4], (cfg["h"], cfg["w"], 3))
            M = OB.class_filter(det[:, 4], det[:, 5], np.full(cfg["nc"], MASK_CONF))
            if pr is not None:
                m = OM.process_mask_native(pr[i], det[:, 6:], boxes, (cfg["h"], cfg["w"]))[M]
                m = m.astype(np.uint8)
            ndet += int(M.sum())
    dt = time.perf_counter() - t0
    return done / dt, dt, done


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="seg640", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-images", type=int, default=64, help="at most this many images of the workload are timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="stop the CPU sample after this much wall time")
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--eager", action="store_true", help="launch every kernel of every step from the host instead of replaying "
                    "one captured HIP graph per step (the default; the per-kernel roofline timing then comes from an eager, "
                    "event-timed pass right after the timed region)")
    ap.add_argument("--graph", action="store_true", help=argparse.SUPPRESS)   # the default since r01; kept for old command lines
    ap.add_argument("--no-fork", action="store_true", help="run the letterbox and the NMS chain on one stream (measurement)")
    ap.add_argument("--mask-logits", default="rows", choices=["rows", "mfma"],
                    help="coefficient x prototype product: row-sparse vector-ALU kernel (default) or dense fp32 MFMA tiles "
                         "(measurement of the MFMA variant; sets LMI_MASK_LOGITS for liblmi_hip)")
    args = ap.parse_args()
    if args.mask_logits == "mfma":
        os.environ["LMI_MASK_LOGITS"] = "mfma"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with torch.distributed.run for N>1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("LMI_BENCH_FORCE_DIST"):   # the variable: a one-rank rehearsal of the multi-GPU path on a one-GPU box
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    cfg = WORKLOADS[args.workload]
    t_gen = time.perf_counter()
    frames, head, proto = make_inputs(cfg, args.seed + 1000 * rank, device)
    pipe = Pipeline(cfg, frames, head, proto)
    fork = not args.no_fork
    log(f"[bench r{rank}] inputs ready in {time.perf_counter() - t_gen:.1f}s: {cfg['desc']}")

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        out = pipe.step(fork=fork)
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    lb_evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    graph = None
    if not args.eager:
        try:   # one step = one HIP graph (both streams of the fork are captured through the event dependencies)
            graph = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread may touch the device while this thread captures
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                out = pipe.step(fork=fork)
            graph.replay()
            barrier()
        except Exception as e:   # noqa: BLE001 -- a capture problem must not cost the measurement
            log(f"[bench r{rank}] HIP graph capture failed ({type(e).__name__}: {e}); eager launches instead")
            graph = None
            torch.cuda.synchronize()
            out = pipe.step(fork=fork)
            barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        if graph is not None:
            graph.replay()
        else:
            # HIP events go on the stream each kernel is launched on: the upsample on the main stream, the
            # letterbox (the dominant kernel of the detect workloads) on the side stream
            out = pipe.step(time_upsample=evs[k] if proto is not None else None, time_letterbox=lb_evs[k], fork=fork)
    barrier()
    elapsed = time.perf_counter() - t0
    if graph is not None:   # events cannot be recorded inside a captured graph: time the kernel in eager passes
        for k in range(args.steps):
            pipe.step(time_upsample=evs[k] if proto is not None else None, time_letterbox=lb_evs[k], fork=fork)
        barrier()
    det, count, anchor, fin, masks = out
    n_surv = int(fin["keep_count"].sum().item())
    n_kept = int(count.sum().item())
    from object_detectors_amd import shard
    chk = shard.checksum_indices(torch.cat([anchor[b, :int(c)] for b, c in enumerate(count.tolist())]))
    # the only communication of the run: per-shard counters gathered over RCCL (backend nccl) for the report
    reports = shard.gather_report({"elapsed": elapsed, "images": cfg["batch"] * args.steps, "survivors": n_surv, "kept": n_kept,
                                   "chk_hi": chk >> 31, "chk_lo": chk & (2 ** 31 - 1)}, device=device)
    elapsed = max(r["elapsed"] for r in reports)
    total_images = int(sum(r["images"] for r in reports))
    value = total_images / elapsed

    # ---- roofline of the dominant kernel (rank 0's own launches) -------------------------------
    H0, W0, S, B = cfg["h"], cfg["w"], cfg["size"], cfg["batch"]
    if proto is not None:
        from object_detectors_amd.roofline import mask_upsample_bytes
        # lmi_mask_upsample is two passes since r01: zeros for every live slot at the memset rate, then the chunks
        # that meet a box.  The stage is rated as ONE unit: its algorithmic bytes (every output byte once + the
        # logits read once) over the HIP-event time around both launches; in the rocprof summary the averages of
        # the listed kernels add up to avg_launch_ms (less the launch gaps).
        single = os.environ.get("LMI_MASK_UPSAMPLE") == "single"
        kernel = "mask_upsample_kernel<uint8>" if single else "mask_zero_fill_kernel+mask_box_kernel<uint8>"
        ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        alg = mask_upsample_bytes(n_surv, H0, W0, S // 4, S // 4)
    else:
        from object_detectors_amd.roofline import letterbox_bytes
        kernel = "letterbox_tables_kernel+letterbox_kernel<float,128>"   # both launches of lmi_letterbox_u8_uniform sit between the events
        ms = float(np.mean([a.elapsed_time(b) for a, b in lb_evs]))
        alg = B * letterbox_bytes(H0, W0, S, S, 4)
    achieved = alg / (ms * 1e-3) / 1e9
    # HBM traffic per launch from the PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    # collected separately, gfx950 corrections applied; see profiles/traffic_<workload>.json)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
    if os.path.isfile(tpath):
        with open(tpath) as f:
            tk = json.load(f).get("kernels", {})
        parts = [tk.get(k.split("<")[0], {}).get("traffic") for k in kernel.split("+")]
        traffic = sum(parts) if all(p is not None for p in parts) else None
    # measured device-to-device copy bandwidth on this box (read + write bytes), quoted beside the spec peak
    src = torch.empty(1 << 30, dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    dst.copy_(src)
    ce = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in ce:
        a.record(); dst.copy_(src); b.record()
    torch.cuda.synchronize()
    copy_gbs = 2 * src.numel() / (sorted(a.elapsed_time(b) for a, b in ce)[2] * 1e-3) / 1e9
    del src, dst
    roof = {"kernel": kernel, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "avg_launch_ms": round(ms, 4),
            "algorithmic_bytes_per_launch": int(alg), "measured_copy_GBs": round(copy_gbs, 1),
            "frac_of_measured_copy": round(achieved / copy_gbs, 4)}

    res = None
    if rank == 0:
        cpu = None
        if args.cpu_images > 0 and world == 1:   # the CPU leg is timed at N=1 only
            threads = min(16, os.cpu_count() or 1)
            ips, dt, n_done = cpu_baseline(cfg, frames, head, proto, min(args.cpu_images, B), threads, args.cpu_seconds)
            cpu = {"value": round(ips, 3), "unit": "images/s", "cores": threads, "kind": "port",
                   "sample": f"first {n_done} images of the same workload through the numpy/torch-CPU oracle "
                             f"(letterbox+NMS+scale/clip/threshold+process_mask), {dt:.1f}s"}
        res = {
            "metric": "pre+post images/sec per node, YOLOv8-seg 640 bs64; % HBM roofline",
            "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8->f32 (integer resize, fp32 NMS/masks)", "data": "synthetic",
            "config": {"workload": cfg["desc"], "name": args.workload, "per_gpu_batch": B, "conf": CONF, "iou": IOU,
                       "max_det": MAX_DET, "survivors_per_batch": n_surv, "nms_kept_per_batch": n_kept,
                       "per_rank": [{"images": int(r["images"]), "survivors": int(r["survivors"]), "kept_checksum": (int(r["chk_hi"]) << 31) | int(r["chk_lo"])} for r in reports],
                       "mask_dtype": "uint8", "forward": "excluded", "launch": "hip graph replay" if graph is not None else "eager", "streams": "letterbox forked beside NMS sort+suppress" if fork else "one", "mask_logits": args.mask_logits, "sharding": f"{world} x independent batches, no collective on the hot path"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return res


if __name__ == "__main__":
    main()
