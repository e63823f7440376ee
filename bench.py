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

Sturdiness (round 2): the block of K timed steps -- barrier + synchronize, K steps, barrier + synchronize -- is
repeated `--blocks` times (default 30) and the line reports the MEDIAN block (max over ranks per block first); the
spread is in `config.block_ms`.  Steps rotate through `--pool` distinct, device-resident input batches (default 3: the source rows of two 1080p batches would still fit the 256 MB Infinity Cache),
each with its own captured graph.  `--images N` switches to the strong-scaling form of BASELINE config 4: N frames
in all, sharded by `shard.shard_bounds` (N/world per rank, 64 per step), `"scaling": "strong"`.
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
METRIC = "pre+post images/sec per node, YOLOv8-seg 640 bs64; % HBM roofline"   # BASELINE.json's metric (the seg640 workload)
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
    """The hot path on one GPU over a pool of input batches, with every output buffer preallocated."""

    def __init__(self, cfg, pool):
        from object_detectors_amd import hip_ops
        self.ops, self.cfg, self.pool = hip_ops, cfg, pool
        dev = pool[0][0].device
        self.S = cfg["size"]
        self.lb_out = torch.empty((cfg["batch"], 3, self.S, self.S), dtype=torch.float32, device=dev)
        self.mask_out = None
        self.capacity = 0
        self.nms_out = None
        self.fin_out = None
        self.side = torch.cuda.Stream(device=dev)
        self.decode_done = torch.cuda.Event()
        self.logits = "rows"
        self.mask_dtype = torch.uint8
        # plan / execute (--lb-plan): the letterbox's coefficient tables depend on the geometry alone, so they can be built
        # once (hip_ops.letterbox_tables) and every step's letterbox is ONE launch.  Not the default: the timed steps rebuild
        # them in every call (the table pass hides beside the NMS sort pass: seg640 1.2035 with against 1.2046 ms/step without)
        self.lb_tables = None

    def size_masks(self):
        """Survivor count of every batch of the pool (one eager pass each, host sync) -> one mask buffer that holds any."""
        if not self.cfg["nm"]:
            return
        cap = 1
        for i in range(len(self.pool)):
            fin = self._nms(i, fork=False)[3]
            cap = max(cap, int(fin["slot_base"][-1].item()))
        self.capacity = cap
        self.mask_out = torch.empty((cap, self.cfg["h"], self.cfg["w"]), dtype=self.mask_dtype, device=self.lb_out.device)

    def _nms(self, i, fork):
        cfg = self.cfg
        fin = dict(model_hw=(self.S, self.S), orig_hw=(cfg["h"], cfg["w"]), thres=MASK_CONF, out=self.fin_out,
                   with_slots=bool(cfg["nm"]))   # detection models need no mask slots: NMS + finalize is then ONE sort launch
        r = self.ops.nms_v8(self.pool[i][1], CONF, IOU, nc=cfg["nc"], max_det=MAX_DET, out=self.nms_out,
                            decode_done=self.decode_done if fork else None, finalize=fin)
        self.nms_out, self.fin_out = r[:3], r[3]
        return r

    def step(self, i=0, time_upsample=None, time_letterbox=None, fork=True, join=True):
        """One pass of the hot path over batch i of the pool.  The letterbox and the NMS chain do not depend on each
        other.  Side by side from the start they gain nothing (both the letterbox and the NMS candidate filter are
        HBM-bound, and the sort + suppression workgroups that follow starve behind the letterbox's), so the
        letterbox is started on a second stream at the moment the candidate filter is done: the latency-bound
        sort + suppression pass (one workgroup per image; scale/clip, class threshold and mask slots run in its
        epilogue) then shares the chip with it.  fork=False: everything on one stream."""
        main = torch.cuda.current_stream()
        if not fork:
            self._letterbox(i, time_letterbox)
            self._nms(i, fork=False)
        else:
            # The sort pass is enqueued BEFORE the letterbox on purpose.  Of the candidate filter's two successors in the
            # graph the one enqueued first starts ~5 us sooner; with the letterbox first the sort workgroups queue behind
            # its tiles: seg640 1.206 -> 1.234, det640 0.124 -> 0.148 ms per step (profiles/r03/fill_pacing.md).
            self._nms(i, fork=True)
            self.side.wait_event(self.decode_done)
            with torch.cuda.stream(self.side):
                self._letterbox(i, time_letterbox)
        # join=False (detection workloads inside a block graph): the next step's candidate filter need not wait for this
        # step's letterbox -- the letterboxes stay ordered on the side stream, the block's last step joins
        do_join = fork and (join or self.cfg["nm"])
        return self._post(i, time_upsample, join=(lambda: main.wait_stream(self.side)) if do_join else None)

    def run_block(self, steps):
        """`steps` consecutive steps (batch k % pool of step k), every one with its own fork; detection workloads join
        the letterbox stream once, at the end of the block -- the next step's candidate filter need not wait for this
        step's letterbox (the letterboxes stay ordered on their own stream).  Segmentation steps join before their zero
        fill as always: nothing runs beside it.  (Tried for them and dropped: the next step's NMS chain on a third
        stream beside this step's box pass, outputs double-buffered -- the box pass shares the chip with nothing:
        seg640 1.244 against 1.247 ms/step, seg1280 1.062 against 1.060.)"""
        for k in range(steps):
            self.step(k % len(self.pool), join=(k == steps - 1))

    def _letterbox(self, i, time_letterbox):
        if time_letterbox is not None:
            time_letterbox[0].record()
        self.ops.letterbox(self.pool[i][0], (self.S, self.S), pad_value=114, swap_rb=True, out=self.lb_out, tables=self.lb_tables)
        if time_letterbox is not None:
            time_letterbox[1].record()

    def _post(self, i, time_upsample=None, join=None):
        ops, cfg = self.ops, self.cfg
        det, count, anchor = self.nms_out
        fin = self.fin_out
        proto = self.pool[i][2]
        masks = None
        if join is not None and proto is None:
            join()
        if proto is not None:
            masks, ovf = ops.process_mask(proto, det, fin["slot_map"], fin["slot_base"], (cfg["h"], cfg["w"]),
                                          self.capacity, self.mask_dtype, out=self.mask_out, split_events=time_upsample,
                                          # the side stream joins BEFORE the upsample: nothing may run beside the zero fill of
                                          # the masks, whose write pattern collapses under concurrent traffic
                                          # (profiles/r01/write_ceiling.md)
                                          before_upsample=join, logits=self.logits)
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


def event_median_us(fn, n=40, warm=4):
    """Median of n HIP-event-timed calls of fn on the current stream, in microseconds."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2] * 1e3


def nms_targets(device):
    """North-star target 2 (NMS under 50 us per image at 8400 anchors): `lmi_nms_v8` alone, HIP events around the call
    (candidate filter + sort/suppress), detection head (B,84,8400), conf 0.25, iou 0.45, max_det 300."""
    from object_detectors_amd import hip_ops as ops, synth
    out = {"head": "(B,84,8400) fp32, conf %.2f iou %.2f max_det %d; typical = 40 objects/image (~400 candidates), "
                   "dense = 110 objects/image (>= 1000 candidates); HIP events around lmi_nms_v8, median of 40" % (CONF, IOU, MAX_DET)}
    cases = [("us_per_image_b1", 1, 40, (5, 16), False), ("us_per_image_b1_agnostic", 1, 40, (5, 16), True),
             ("us_per_image_b1_dense", 1, 110, (8, 16), False), ("us_per_image_b1_dense_agnostic", 1, 110, (8, 16), True),
             ("us_per_call_b64", 64, 40, (5, 16), False), ("us_per_call_b16_dense", 16, 110, (8, 16), False),
             ("us_per_call_b16_dense_agnostic", 16, 110, (8, 16), True)]
    for name, B, objects, cluster, agn in cases:
        rng = np.random.default_rng(1)
        p = torch.from_numpy(synth.head(rng, B, 80, 0, 640, objects, cluster=cluster)).to(device)
        res = ops.nms_v8(p, CONF, IOU, nc=80, agnostic=agn, max_det=MAX_DET)
        out[name] = round(event_median_us(lambda: ops.nms_v8(p, CONF, IOU, nc=80, agnostic=agn, max_det=MAX_DET, out=res)), 1)
        if B == 1:
            out[name.replace("us_per_image", "candidates_kept")] = [int((p[:, 4:].amax(1) > CONF).sum().item()), int(res[1][0].item())]
    return out


def letterbox_targets(device, S=640, B=64):
    """North-star target 1 (letterbox >= 95 % of the HBM roofline): `lmi_letterbox_u8_uniform` alone on cold frames (the
    calls rotate through 3 distinct source batches, so no source row is still in the 256 MiB Infinity Cache), fp32 NCHW
    out.  frac_spec = SURVEY 8(d)'s tap-count bytes over time over 8 TB/s; frac_row_granular = whole source rows that
    hold a tap + output (the bytes that must cross the HBM interface) over the same; traffic_ratio = PMC bytes
    (profiles/traffic_det640*.json) over the tap-count bytes."""
    from object_detectors_amd import hip_ops as ops
    from object_detectors_amd.roofline import letterbox_bytes, letterbox_row_granular_bytes
    out = {"what": f"{B} frames -> ({B},3,{S},{S}) fp32, cold sources (3 rotating batches), HIP events, median of 40; "
                   "*_tables_ready: the geometry's tables built once (hip_ops.letterbox_tables), one launch per call"}
    g = torch.Generator(device=device)
    g.manual_seed(11)
    for tag, h, w, tfile in (("1080p", 1080, 1920, "traffic_det640.json"), ("4k", 2160, 3840, "traffic_det640_4k.json")):
        fs = [torch.randint(0, 256, (B, h, w, 3), dtype=torch.uint8, device=device, generator=g) for _ in range(3)]
        o = torch.empty((B, 3, S, S), dtype=torch.float32, device=device)
        k = [0]

        tabs = ops.letterbox_tables((h, w), (S, S), device)

        def rot(tables=tabs):
            k[0] = (k[0] + 1) % 3
            ops.letterbox(fs[k[0]], (S, S), out=o, pad_value=114, swap_rb=True, tables=tables)
        us = event_median_us(lambda: rot(None))                             # the ordinary call: table pass + tile kernel
        out["us_%s_tables_ready" % tag] = round(event_median_us(rot), 1)    # LMI_LB_TABLES_READY: the tile kernel alone
        alg, rowg = B * letterbox_bytes(h, w, S, S, 4), B * letterbox_row_granular_bytes(h, w, S, S, 4)
        out["us_" + tag] = round(us, 1)
        out["frac_spec_" + tag] = round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        out["frac_row_granular_" + tag] = round(rowg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        out["GBs_row_granular_" + tag] = round(rowg / (us * 1e-6) / 1e9, 1)
        tpath = os.path.join(ROOT, "profiles", tfile)
        if os.path.isfile(tpath):
            with open(tpath) as f:
                tr = json.load(f).get("kernels", {}).get("letterbox_kernel", {}).get("traffic")
            out["traffic_ratio_" + tag] = round(tr / alg, 3) if tr else None
        del fs, o
    return out


def host_cpus():
    """What the host offers: the machine's CPU count and model, and the share this process may actually use (affinity
    mask, cgroup quota, the box's OMP_NUM_THREADS: a one-GPU lease of the pool is allotted 16 CPUs of a much larger host)."""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    share = os.cpu_count() or 1
    try:
        share = min(share, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            share = min(share, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    omp = os.environ.get("OMP_NUM_THREADS", "")
    if omp.isdigit() and int(omp) > 0:
        share = min(share, int(omp))
    return {"cpu_model": model, "box_cpus": os.cpu_count(), "usable_cpus": share}


def free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N ...` without a launcher: run `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a child process.
    The parent only counts devices (which does not initialise the GPU) and forwards the child's return code."""
    import subprocess
    have = torch.cuda.device_count()
    if have < n and not os.environ.get("LMI_BENCH_SHARE_DEVICE"):
        log(f"[bench] --gpus {n} but only {have} GPU(s) visible")
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *argv]
    log("[bench] launching: " + " ".join(cmd))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="seg640", choices=sorted(WORKLOADS))
    ap.add_argument("--blocks", type=int, default=30, help="the block of --steps timed steps is repeated this many times; the median block is reported")
    ap.add_argument("--pool", type=int, default=3, help="distinct device-resident input batches the steps rotate through")
    ap.add_argument("--images", type=int, default=0, help="strong scaling: this many frames IN ALL, sharded over the ranks "
                    "(shard.shard_bounds), per-GPU batch per step; --steps is then derived from the shard (BASELINE config 4: "
                    "--workload det640_4k --images 8192)")
    ap.add_argument("--cpu-images", type=int, default=64, help="at most this many images of the workload are timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="stop each CPU sample (all usable CPUs, then one thread) after this much wall time")
    ap.add_argument("--no-targets", dest="targets", action="store_false",
                    help="skip the `nms` and `letterbox` objects of the line (NMS latency per image, letterbox alone on cold frames)")
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--eager", action="store_true", help="launch every kernel of every step from the host instead of replaying "
                    "one captured HIP graph per step (the default; the per-kernel roofline timing then comes from an eager, "
                    "event-timed pass right after the timed region)")
    ap.add_argument("--graph", action="store_true", help=argparse.SUPPRESS)   # the default since r01; kept for old command lines
    ap.add_argument("--no-pipeline-steps", dest="pipeline_steps", action="store_false",
                    help="inside a block graph, join the letterbox stream at the end of every step instead of only at the block's end")
    ap.add_argument("--no-block-graph", dest="block_graph", action="store_false",
                    help="replay one HIP graph per step instead of one graph per timed block of --steps steps")
    ap.add_argument("--no-fork", action="store_true", help="run the letterbox and the NMS chain on one stream (measurement)")
    ap.add_argument("--lb-plan", action="store_true",
                    help="build the letterbox's coefficient tables once per geometry (hip_ops.letterbox_tables) instead of in every call "
                         "(one launch per letterbox instead of two; measurement -- the default rebuilds them in every timed step)")
    ap.add_argument("--mask-dtype", default="uint8", choices=["uint8", "float32"],
                    help="dtype the masks are stored in: uint8 (the headline: SURVEY 8(a) a7 allows it) or float32, the "
                         "reference's own 0./1. masks and the wrapper's default (four times the bytes)")
    ap.add_argument("--mask-logits", default="rows", choices=["rows", "mfma"],
                    help="coefficient x prototype product: row-sparse vector-ALU kernel (default) or dense fp32 MFMA tiles "
                         "(measurement of the MFMA variant; the LMI_MASK_LOGITS_MFMA flag of lmi_process_mask_ex)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        # invoked plainly with --gpus N: start the N ranks ourselves (one process per GPU) in a FRESH child, before this
        # process has touched the GPU (it never does), pass the child's output through and leave with its status
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    # stdout carries ONE line, the JSON: RCCL and gloo print banners on the C-level stdout when a process group comes up
    # ("RCCL version : ...", "[Gloo] Rank 0 is connected to ..."), so file descriptor 1 is pointed at stderr for the whole
    # run and the line goes out through a duplicate of the original descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with torch.distributed.run for N>1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # LMI_BENCH_SHARE_DEVICE=1: a rehearsal of the N-rank launch on a box with fewer GPUs -- every rank runs on cuda:0 and
    # the end-of-run gather goes over gloo (RCCL refuses two ranks on one device).  Its numbers mean nothing (the ranks
    # share one GPU); it proves the launcher, the sharding and the gather.  The line says so in config.sharding.
    share = bool(os.environ.get("LMI_BENCH_SHARE_DEVICE"))
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    gather_dev = None if share else device
    dist = None
    if world > 1 or os.environ.get("LMI_BENCH_FORCE_DIST"):   # the variable: a one-rank rehearsal of the multi-GPU path on a one-GPU box
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from object_detectors_amd import shard
    cfg = WORKLOADS[args.workload]
    B = cfg["batch"]
    steps, scaling = args.steps, "weak"
    if args.images > 0:   # strong scaling: the shard of this rank, in whole batches
        if args.images % (world * B):
            raise SystemExit(f"--images {args.images} must be a multiple of ranks x per-GPU batch = {world * B}")
        lo, hi = shard.shard_bounds(args.images, world, rank)
        steps, scaling = (hi - lo) // B, "strong"
    t_gen = time.perf_counter()
    pool = [make_inputs(cfg, args.seed + 1000 * rank + 7919 * i, device) for i in range(max(args.pool, 1))]
    frames, head, proto = pool[0]
    pipe = Pipeline(cfg, pool)
    pipe.logits = args.mask_logits
    pipe.mask_dtype = torch.float32 if args.mask_dtype == "float32" else torch.uint8
    if args.lb_plan:
        pipe.lb_tables = pipe.ops.letterbox_tables((cfg["h"], cfg["w"]), (cfg["size"], cfg["size"]), device)
    pipe.size_masks()
    fork = not args.no_fork
    log(f"[bench r{rank}] inputs ready in {time.perf_counter() - t_gen:.1f}s: {cfg['desc']} (pool of {len(pool)} batches)")

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(max(args.warmup, 1)):
        out = pipe.step(k % len(pool), fork=fork)
    if fork and args.pipeline_steps and args.block_graph and not args.eager:
        pipe.run_block(2 * len(pool))   # eager once: every stream's workspace exists before the capture
    barrier()
    graphs = None
    block_graph = None
    if not args.eager:
        try:   # one step = one HIP graph per pool entry (both streams of the fork are captured through the event dependencies)
            graphs = []
            for i in range(len(pool)):
                g = torch.cuda.CUDAGraph()
                # thread_local: RCCL's watchdog thread may touch the device while this thread captures
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    out = pipe.step(i, fork=fork)
                g.replay()
                graphs.append(g)
            barrier()
            if args.block_graph and steps <= 512:
                # the whole timed block as ONE graph (the steps still rotate through the pool, every step with its own
                # fork and join): between two graph launches the GPU idles ~15 us, 10 % of a detection step
                bg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(bg, capture_error_mode="thread_local"):
                    if args.pipeline_steps and fork:
                        pipe.run_block(steps)
                    else:
                        for k in range(steps):
                            out = pipe.step(k % len(pool), fork=fork)
                bg.replay()
                block_graph = bg
                barrier()
        except Exception as e:   # noqa: BLE001 -- a capture problem must not cost the measurement
            log(f"[bench r{rank}] HIP graph capture failed ({type(e).__name__}: {e}); eager launches instead")
            graphs = None
            block_graph = None
            torch.cuda.synchronize()
            out = pipe.step(0, fork=fork)
            barrier()

    # ---- the timed region: EXACTLY `steps` steps between barrier + synchronize, repeated `blocks` times -----------
    block_s = []
    for _ in range(max(args.blocks, 1)):
        barrier()
        t0 = time.perf_counter()
        for k in range(steps if block_graph is None else 0):
            if graphs is not None:
                graphs[k % len(pool)].replay()
            else:
                out = pipe.step(k % len(pool), fork=fork)
        if block_graph is not None:
            block_graph.replay()
        barrier()
        block_s.append(time.perf_counter() - t0)
    # max over ranks per block, then the median block
    block_all = shard.gather_vector(torch.tensor(block_s, dtype=torch.float64), device=gather_dev)
    per_block = torch.stack(block_all).max(dim=0).values
    elapsed = float(per_block.median())

    # ---- what the last timed step left behind must equal a fresh eager pass over the same batch (guards the graph /
    # stream plumbing: the buffers are overwritten by every step) ------------------------------------------------------
    last = (steps - 1) % len(pool)
    got_lb = pipe.lb_out.clone()
    got_det, got_count = pipe.nms_out[0].clone(), pipe.nms_out[1].clone()
    got_mask_sum = int(pipe.mask_out.sum(dtype=torch.int64).item()) if pipe.mask_out is not None else 0
    ref = pipe.step(last, fork=False)
    torch.cuda.synchronize()
    cnt = ref[1].tolist()
    replay_ok = bool(torch.equal(got_lb, pipe.lb_out) and torch.equal(got_count, ref[1])
                     and all(torch.equal(got_det[b, :c], ref[0][b, :c]) for b, c in enumerate(cnt))
                     and (pipe.mask_out is None or got_mask_sum == int(pipe.mask_out.sum(dtype=torch.int64).item())))
    if not replay_ok:
        raise RuntimeError("bench: the timed replay's outputs differ from an eager pass over the same batch")

    # ---- per-kernel timing with HIP events on the stream each kernel runs on (eager passes, after the timed region) ----
    n_ev = 40   # the fill's duration varies by +-2.3 % from launch to launch (rocprof: 936-1070 us); 40 launches put the mean within 0.4 %
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
    lb_evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
    for k in range(n_ev):
        out = pipe.step(k % len(pool), time_upsample=evs[k] if proto is not None else None, time_letterbox=lb_evs[k], fork=fork)
    barrier()
    # `out` belongs to pool entry (n_ev - 1) % len(pool); counters below are per pool entry
    surv, kept, chks = [], [], []
    for i in range(len(pool)):
        det, count, anchor, fin, masks = pipe.step(i, fork=False)
        torch.cuda.synchronize()
        surv.append(int(fin["keep_count"].sum().item()))
        kept.append(int(count.sum().item()))
        chks.append(shard.checksum_indices(torch.cat([anchor[b, :int(c)] for b, c in enumerate(count.tolist())])))
    steps_per_entry = [len(range(i, steps, len(pool))) for i in range(len(pool))]
    n_surv_total = sum(s * n for s, n in zip(surv, steps_per_entry))
    chk = sum(chks) % (2 ** 61 - 1)
    # the only communication of the run: per-shard counters gathered over RCCL (backend nccl) for the report
    reports = shard.gather_report({"images": B * steps, "survivors": n_surv_total, "kept": sum(k * n for k, n in zip(kept, steps_per_entry)),
                                   "chk_hi": chk >> 31, "chk_lo": chk & (2 ** 31 - 1)}, device=gather_dev)
    total_images = int(sum(r["images"] for r in reports))
    value = total_images / elapsed

    # ---- roofline of the dominant kernel (rank 0's own launches) -------------------------------
    H0, W0, S = cfg["h"], cfg["w"], cfg["size"]
    if proto is not None:
        from object_detectors_amd.roofline import mask_upsample_bytes_exact
        # lmi_mask_upsample is two passes since r01: zeros for every live slot at the memset rate, then the chunks
        # that meet a box.  The stage is rated as ONE unit: its algorithmic bytes (every output byte once + the
        # logits rows x columns under each box read once) over the HIP-event time around both launches; in the
        # rocprof summary the averages of the listed kernels add up to avg_launch_ms (less the launch gap).
        kernel = "mask_zero_fill_kernel+mask_box_kernel<%s>" % ("uint8" if args.mask_dtype == "uint8" else "float")
        ev_ms = [a.elapsed_time(b) for a, b in evs]
        ms = float(np.mean(ev_ms))
        if os.environ.get("LMI_BENCH_DEBUG_EVENTS"):
            log("[bench] upsample event times (ms): " + " ".join(f"{t:.4f}" for t in ev_ms))
        algs = []
        for i in range(len(pool)):
            det, count, anchor, fin, masks = pipe.step(i, fork=False)
            torch.cuda.synchronize()
            total = int(fin["slot_base"][-1].item())
            rows = det.view(-1, det.shape[-1])[fin["slot_map"][:total].long()][:, :4].cpu().numpy()
            algs.append(mask_upsample_bytes_exact(rows, H0, W0, S // 4, S // 4, out_elem=pipe.mask_out.element_size()))
        alg = float(np.mean([algs[k % len(pool)] for k in range(n_ev)]))
    else:
        from object_detectors_amd.roofline import letterbox_bytes
        kernel = "letterbox_tables_kernel+letterbox_kernel<float,128>"   # both launches of lmi_letterbox_u8_uniform sit between the events
        ev_ms = [a.elapsed_time(b) for a, b in lb_evs]
        ms = float(np.mean(ev_ms))
        alg = B * letterbox_bytes(H0, W0, S, S, 4)
    achieved = alg / (ms * 1e-3) / 1e9
    # HBM traffic per launch from the PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    # collected separately, gfx950 corrections applied; see profiles/traffic_<workload>.json)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
    if os.path.isfile(tpath) and args.mask_dtype == "uint8":   # the PMC passes were taken on the headline's uint8 masks
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
            "launch_ms_min_median_max": [round(float(x), 4) for x in np.percentile(ev_ms, [0, 50, 100])],
            "algorithmic_bytes_per_launch": int(alg), "measured_copy_GBs": round(copy_gbs, 1),
            "frac_of_measured_copy": round(achieved / copy_gbs, 4), "frac_of_achievable_6300": round(achieved / 6300.0, 4)}

    # ---- the north star's other two targets, measured beside the headline (rank 0, after the timed region) ----
    nms_t = lb_t = None
    if rank == 0 and args.targets:
        nms_t = nms_targets(device)
        lb_t = letterbox_targets(device)

    res = None
    if rank == 0:
        cpu = None
        if args.cpu_images > 0 and world == 1:   # the CPU leg is timed at N=1 only
            hc = host_cpus()
            threads = hc["usable_cpus"]     # every CPU this process may use (all cores AND one thread)
            log(f"[bench] host: {hc['cpu_model']}, {hc['box_cpus']} CPUs in the machine, {threads} usable by this process")
            ips, dt, n_done = cpu_baseline(cfg, frames, head, proto, min(args.cpu_images, B), threads, args.cpu_seconds)
            ips1, dt1, n1 = cpu_baseline(cfg, frames, head, proto, min(args.cpu_images, B), 1, args.cpu_seconds)
            torch.set_num_threads(threads)
            cpu = {"value": round(ips, 3), "unit": "images/s", "cores": threads, "kind": "port",
                   "value_1thread": round(ips1, 3), "box_cpus": hc["box_cpus"], "cpu_model": hc["cpu_model"],
                   "sample": f"first {n_done} images of the same workload through the numpy/torch-CPU oracle "
                             f"(letterbox+NMS+scale/clip/threshold+process_mask; a reference-equivalent restatement, cv2 / ultralytics "
                             f"are not installed) on {threads} threads = every CPU this process may use, {dt:.1f}s; then {n1} images on 1 thread, {dt1:.1f}s"}
        pb = sorted(per_block.tolist())
        res = {
            "metric": METRIC if args.workload == "seg640" else f"pre+post images/sec per node, {cfg['desc']}; % HBM roofline",
            "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "u8->f32 (integer resize, fp32 NMS/masks)", "data": "synthetic",
            "config": {"workload": cfg["desc"], "name": args.workload, "per_gpu_batch": B, "conf": CONF, "iou": IOU,
                       "max_det": MAX_DET, "survivors_per_batch": surv, "nms_kept_per_batch": kept,
                       "timing": f"median of {len(pb)} blocks of {steps} steps, each bracketed by barrier + synchronize",
                       "block_ms": {"min": round(pb[0] * 1e3, 4), "median": round(elapsed * 1e3, 4), "max": round(pb[-1] * 1e3, 4)},
                       "input_pool": f"{len(pool)} distinct device-resident batches, rotated every step",
                       "total_images": args.images if args.images > 0 else None,
                       "per_rank": [{"images": int(r["images"]), "survivors": int(r["survivors"]), "kept_checksum": (int(r["chk_hi"]) << 31) | int(r["chk_lo"])} for r in reports],
                       "mask_dtype": args.mask_dtype, "forward": "excluded", "launch": ("one hip graph per block of %d steps" % steps) if block_graph is not None else "one hip graph replay per step" if graphs is not None else "eager",
                       "streams": "letterbox forked beside NMS sort+suppress+finalize" if fork else "one", "mask_logits": args.mask_logits,
                       "letterbox_tables": "built once per geometry (--lb-plan), one launch per step" if args.lb_plan else "rebuilt by every call (two launches)",
                       "replay_equals_eager": replay_ok,
                       "step_overlap": ("letterbox stream joined once per block: step k+1's candidate filter may run beside step k's letterbox"
                                        if (block_graph is not None and args.pipeline_steps and fork and not cfg["nm"]) else "none: every step joins its streams"),
                       "sharding": f"{world} x independent batches, no collective on the hot path"
                                   + (" -- REHEARSAL: all ranks share cuda:0 (LMI_BENCH_SHARE_DEVICE), the value is not a multi-GPU number" if share else "")},
            "roofline": roof, "cpu_baseline": cpu, "nms": nms_t, "letterbox": lb_t,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return res


if __name__ == "__main__":
    main()
