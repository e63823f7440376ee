#!/bin/bash
# instruction mix / stall picture of the seg640 kernels: one PMC pass (8 SQ counters + GRBM)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq_util; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-images 0 --blocks 1 > $O/run.json 2> $O/run.err || { echo failed; tail -5 $O/run.err; exit 1; }
python3 - <<PY
import csv, glob, collections
f = max(glob.glob("$O/**/*counter_collection.csv", recursive=True), key=lambda p: __import__("os").path.getmtime(p))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    agg[n[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in agg.items():
    if not any(k in n for k in ("mask_", "letterbox_k", "nms_")): continue
    m = {k: sum(v) / len(v) for k, v in c.items()}
    print(n)
    print("   " + "  ".join(f"{k}={v:.3g}" for k, v in sorted(m.items())))
PY
