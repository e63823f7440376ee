#!/bin/bash
# PMC passes over the mask upsample (mask_box_kernel): where the waves' cycles go.  usage: bash scratch/box_pmc.sh <tag>
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  O=$R/gpurun_out/box_pmc_$1_$(echo $pass | cut -d" " -f1); mkdir -p $O
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $O -- python3 $R/scratch/mask_time.py > $O/run.log 2> $O/run.err
  echo "rc=$? ($pass)"; tail -1 $O/run.err
  python3 - $O <<'PY'
import csv,glob,collections,sys
fs=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    n=r["Kernel_Name"]
    if "mask_box" in n or "mask_logits" in n: agg[n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    d={c: sum(x)/len(x) for c,x in v.items()}
    wc=d.get("SQ_WAVE_CYCLES",0)
    print(k)
    for c,x in d.items(): print(f"   {c:24s} {x:16.0f}" + (f"  {x/wc:6.3f} of wave cycles" if wc else ""))
PY
done
find $R/gpurun_out -name "*.db" -delete
