#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE counter CSVs of two separate rocprofv3 --pmc passes into per-kernel HBM traffic.

    python tools/pmc_to_traffic.py <dir with the FETCH_SIZE pass> <dir with the WRITE_SIZE pass> <out.json>

Units and gfx950 correction as /opt/skills/guides/MI355X_MICROARCH.md (section HBM) prescribes: both counters are in
KiB; FETCH_SIZE reports exactly half the bytes of a wide (16 B/lane) coalesced read stream on gfx950, so it is
doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.  Values are means per dispatch.
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def per_kernel(root, counter):
    acc = defaultdict(list)
    for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    rd = 2.0 * fetch.get(k, 0.0) * 1024
    wr = write.get(k, 0.0) * 1024
    out[k] = {"dispatches": nf.get(k, nw.get(k, 0)), "fetch_size_kib_raw": fetch.get(k), "write_size_kib": write.get(k),
              "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes_per_dispatch": rd + wr}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE doubled (gfx950)",
           "kernels": out}, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k[:50]: round(v["hbm_bytes_per_dispatch"] / 1e6, 1) for k, v in out.items()}))
