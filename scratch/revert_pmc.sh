#!/bin/bash
# PMC pass over revert_masks: where the waves' cycles go.  usage: bash scratch/revert_pmc.sh <tag> [uint8|float32]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/revert_pmc_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc ${PMC:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT} --output-format csv -d $O -- python3 $R/scratch/revert_pmc_run.py $2 > $O/run.log 2> $O/run.err
echo rc=$?; tail -2 $O/run.err
cd $R && python3 - $O <<'PY'
import csv,glob,collections,sys
fs=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    n=r["Kernel_Name"]
    if "revert" in n: agg[n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    d={c: sum(x)/len(x) for c,x in v.items()}
    wc=d.get("SQ_WAVE_CYCLES",1)
    print(k); 
    for c,x in d.items(): print(f"   {c:24s} {x:14.0f}  {x/wc:6.3f} of wave cycles")
PY
