#!/usr/bin/env python3
"""Concurrency picture of a pipelined run from a rocprofv3 kernel-trace CSV: per queue the kernels of a time window, and how
much of the window has 0 / 1 / 2+ kernels executing."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "porla" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
mid = rows[n // 2:]                       # steady state
t0 = int(mid[0]["Start_Timestamp"])
win = 8_000_000                           # 8 ms
sel = [r for r in mid if int(r["Start_Timestamp"]) - t0 < win]
ev = []
for r in sel:
    ev.append((int(r["Start_Timestamp"]) - t0, 1)); ev.append((int(r["End_Timestamp"]) - t0, -1))
ev.sort()
cur = 0; last = 0; hist = {}
for t, d in ev:
    hist[cur] = hist.get(cur, 0) + (t - last); last = t; cur += d
tot = sum(hist.values())
print("concurrency histogram over %.2f ms:" % (tot / 1e6), {k: round(v / tot, 3) for k, v in sorted(hist.items())})
qs = sorted(set(r.get("Queue_Id", "?") for r in sel))
for r in sel[:70]:
    name = r["Kernel_Name"].split("(")[0].replace("void porla::", "").replace("porla::", "")[:26]
    print("q%-3s %-26s %9.1f -> %9.1f us (%.1f)" % (qs.index(r.get("Queue_Id", "?")), name, (int(r["Start_Timestamp"]) - t0) / 1e3,
                                              (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
