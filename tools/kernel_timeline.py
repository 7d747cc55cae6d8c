#!/usr/bin/env python3
"""Print the device timeline (queue, start, duration, gap) of the last N kernel dispatches of a
rocprofv3 --kernel-trace CSV: tools/kernel_timeline.py <dir-or-csv> [N]"""
import csv
import glob
import os
import sys

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
if os.path.isdir(path):
    path = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(path, newline="")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("q%-3s %-28s start %9.1f us  dur %8.1f us  gap-after-prev-end %8.1f us" % (
        r["Queue_Id"], r["Kernel_Name"][:28], (st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3))
    prev_end = max(prev_end, en)
