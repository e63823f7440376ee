#!/bin/bash
# A/B of bench.py with and without the letterbox forked beside the NMS sort pass
for w in det640 seg640 det640_4k seg1280; do
  for f in "" "--no-fork"; do
    timeout -k 10 200 python bench.py --workload $w --steps 40 --warmup 5 --cpu-images 0 $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', '$f' or 'fork', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" || exit 1
  done
done
