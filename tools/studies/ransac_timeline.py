#!/usr/bin/env python3
"""Study: where a RANSAC batch's time goes on the GPU.  Reads a rocprofv3 kernel trace (csv) of `bench.py --no-cpu-baseline
--no-operators` and prints, for the stretch between two phase-1 scoring dispatches, every kernel / copy with its start offset,
duration and the idle gap before it.   python tools/studies/ransac_timeline.py <kernel_trace.csv> [<memory_copy_trace.csv>]"""
import csv
import sys

rows = []
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name") or ("copy " + r.get("Direction", "?"))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0][-40:]))
rows.sort()
score = [i for i, r in enumerate(rows) if "k_ransac_score_fast" in r[2]]
# take a window in the middle of the run: from one long scoring dispatch to the one two batches later
longs = [i for i in score if rows[i][1] - rows[i][0] > 1_000_000]
a, b = longs[len(longs) // 2], longs[len(longs) // 2 + 2]
t0 = rows[a][0]
prev_end = rows[a][0]
busy = 0
for s, e, n in rows[a:b]:
    print("%9.1f us  dur %8.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n))
    busy += e - s
    prev_end = max(prev_end, e)
print("window %.1f us, busy %.1f us" % ((rows[b][0] - t0) / 1e3, busy / 1e3))
