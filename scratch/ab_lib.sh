#!/bin/bash
# A/B of library variants under scratch/variants/*.so on one box: each variant is copied over liblmi_hip.so (of the
# box's scratch copy of the tree) and bench.py is run on the given workloads.  usage: scratch/ab_lib.sh "seg640 det640" a.so b.so ...
# The baseline library is restored at the end.
set -e
W="$1"; shift
LIB=object_detectors_amd/liblmi_hip.so
cp $LIB /tmp/lmi_keep.so
for rep in 1 2; do
  for v in "$@"; do
    cp scratch/variants/$v $LIB
    for w in $W; do
      timeout -k 10 200 python bench.py --workload $w --steps 20 --warmup 5 --cpu-images 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$w', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['replay_equals_eager'])" || { cp /tmp/lmi_keep.so $LIB; exit 1; }
    done
  done
done
cp /tmp/lmi_keep.so $LIB
