#!/bin/bash
# MFMA utilisation of the dense mask GEMM variant: one PMC pass of bench.py --mask-logits mfma
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mfma_util; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-images 0 --mask-logits mfma > $O/run.json 2> $O/run.err
echo rc=$?; tail -2 $O/run.err
cd $R && python3 - $O <<'PY'
import csv,glob,collections,sys
fs=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    n=r["Kernel_Name"]
    if "mask_gemm" in n or "mask_logits" in n: agg[n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k, {c: sum(x)/len(x) for c,x in v.items()})
PY
