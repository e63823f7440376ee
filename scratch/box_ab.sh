#!/bin/bash
# A/B of mask_box_kernel variants (scratch/variants/*.so): mask stages alone + the seg640 step
LIB=object_detectors_amd/liblmi_hip.so
cp $LIB /tmp/lmi_keep.so
for rep in 1 2; do
  for v in "$@"; do
    cp scratch/variants/$v.so $LIB
    echo "$v: $(timeout -k 10 100 python scratch/mask_time.py 2>/dev/null | tail -1)"
    timeout -k 10 200 python bench.py --cpu-images 0 --no-targets 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'seg640', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['replay_equals_eager'])" || { cp /tmp/lmi_keep.so $LIB; exit 1; }
  done
done
cp /tmp/lmi_keep.so $LIB
