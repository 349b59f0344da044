#!/usr/bin/env python3
"""Timeline of the last complete frame in a rocprofv3 kernel trace (tools/trace_frame.py): start / end of every kernel
relative to the frame's first kernel.  A frame ends with its persistent kernel (pool_collapse_kernel<.., true> or pipe_kernel)."""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def frame_kernel(r):
    k = r["Kernel_Name"].replace(" ", "")
    return "pool_collapse_kernel" in k or "pipe_kernel<" in k
# (the serial kernel may be followed by its second launch -- the direct items without a row slot --: the frame ends with the last of them)
ends = [i for i, r in enumerate(rows) if frame_kernel(r) and not (i + 1 < len(rows) and frame_kernel(rows[i + 1]))]
last, prev = ends[-2], ends[-3]
frame = rows[prev + 1:last + 1]
t0 = min(int(r["Start_Timestamp"]) for r in frame)
print(f"frame period {(int(rows[ends[-2]]['End_Timestamp']) - int(rows[ends[-3]]['End_Timestamp'])) / 1e3:.1f} us")
for r in frame:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:8.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:8.1f}  q{r.get("Queue_Id", "?")}  {r["Kernel_Name"][:90]}')
