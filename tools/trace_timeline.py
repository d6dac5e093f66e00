#!/usr/bin/env python3
"""Prints the kernel timeline of the LAST MSM in a rocprofv3 kernel-trace CSV (start, duration, gap to the previous kernel)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "k_points_to_mont"
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]][-1]
t0 = int(rows[idx]["Start_Timestamp"]); prev = t0
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void porla::", "").replace("porla::", "")[:30]
    print("%-30s start %8.1f us  dur %7.1f us  gap %6.1f  wg %s grid %s" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3,
          r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
    prev = e
print("total %.1f us" % ((prev - t0) / 1e3))
