"""Print one step's kernel timeline (start offset, duration, gap to the previous end) from a rocprofv3 kernel trace.
usage: python scratch/timeline.py gpurun_out/trace_TAG [step-index-from-end]"""
import csv, glob, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=__import__("os").path.getmtime)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "")) for r in csv.DictReader(open(f))]
rows.sort()
# a step starts at each nms_decode_filter
starts = [i for i, r in enumerate(rows) if "nms_decode_filter" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = starts[-k - 1], starts[-k]
t0 = rows[a][0]
prev_end = t0
for s, e, n, st in rows[a:b]:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")[:48]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {n}")
    prev_end = max(prev_end, e)
print(f"step span {(rows[b][0] - t0) / 1e3:.1f} us")
