#!/bin/bash
# usage (on the GPU box): bash scratch/prof_all.sh <tag prefix, e.g. r2_>   -- prof.sh + a plain bench line for the five workloads
set -u
P=$1
R=$GRAFT_REPO_ROOT
for w in seg640 det640 seg1280 det640_4k seg640_dense; do
  echo "== $w"; 
  bash $R/scratch/prof.sh $P$w --workload $w > $R/gpurun_out/prof_$P$w.log 2>&1 || { echo "prof $w failed"; tail -5 $R/gpurun_out/prof_$P$w.log; exit 1; }
  cd $R && timeout -k 10 200 python bench.py --workload $w > gpurun_out/bench_$P$w.json 2> gpurun_out/bench_$P$w.err || { echo "bench $w failed"; tail -5 gpurun_out/bench_$P$w.err; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/bench_$P$w.json')); print('$w', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
