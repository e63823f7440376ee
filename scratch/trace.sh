#!/bin/bash
# usage: trace.sh <tag> [bench args]
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-images 0 "$@" > $O/bench.json 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
python3 - $O <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"])>0.2: print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1000:8.1f} us  {r["Percentage"]}%')
PY
