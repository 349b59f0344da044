#!/usr/bin/env python3
"""Timeline of the last complete frame in a rocprofv3 kernel trace (tools/trace_frame.py): start / end of every kernel
relative to the frame's first kernel."""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "pool_collapse_kernel" in r["Kernel_Name"] and "true>" in r["Kernel_Name"].replace(" ", "")]
# a frame ends with the (empty) direct-item launch; take the one before the last
last, prev = ends[-2], ends[-3]
frame = rows[prev + 1:last + 1]
t0 = min(int(r["Start_Timestamp"]) for r in frame)
for r in frame:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:8.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:8.1f}  q{r.get("Queue_Id", "?")}  {r["Kernel_Name"][:90]}')
