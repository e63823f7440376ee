#!/bin/bash
# usage: collect_profiles.sh <round-dir, e.g. r02> <tag prefix, e.g. r2_>   -- after `scratch/prof.sh <prefix><w> --workload <w>` and
# `python bench.py --workload <w> > gpurun_out/bench_<prefix><w>.json` on the GPU box: copy the summaries that are judged into profiles/
set -eu
R=$1; P=$2
cd "$(dirname "$0")/.."
mkdir -p profiles/$R
for w in seg640 det640 seg1280 det640_4k seg640_dense; do
  d=gpurun_out/prof_$P$w
  f=$(ls -t $(find $d/trace -name "*kernel_stats.csv") | head -1)
  python - "$f" profiles/$R/${w}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
out = [rows[0]]
for r in rows[1:]:
    r[0] = r[0].replace("(anonymous namespace)::", "")
    if len(r[0]) > 160: r[0] = r[0][:157] + "..."   # torch's elementwise kernel names run to kilobytes
    out.append(r)
csv.writer(open(sys.argv[2], "w", newline="")).writerows(out)
PY
  cp $d/trace.json profiles/$R/${w}_bench_under_rocprof.json
  cp gpurun_out/bench_$P$w.json profiles/$R/${w}_bench.json
  python profiles/make_traffic.py $d $w $R
done
