#!/bin/bash
# usage: pmc.sh <outdir-tag> <kernel-substring> <script.py> <counter> [counter...]
TAG=$1; KSUB=$2; SCRIPT=$3; shift 3
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $O -- python3 $R/$SCRIPT > $O/run.out 2> $O/run.err
echo "rocprof rc=$?"; tail -3 $O/run.err
cd $R && python3 - "$O" "$KSUB" <<'PY'
import csv,glob,collections,sys
fs=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
if not fs: print("no counter csv"); sys.exit(0)
agg=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if sys.argv[2] in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print(k, len(v), sum(v)/len(v))
PY
