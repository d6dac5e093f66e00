#!/usr/bin/env python3
"""Timeline of the LAST compute_multi_exp call on host buffers from rocprofv3 --kernel-trace --memory-copy-trace CSVs:
copies and kernels per stream, relative to the first copy of the call.
  rocprofv3 --kernel-trace --memory-copy-trace -d DIR -o hb --output-format csv -- python3 tools/bench_host_boundary.py --json 20
  python tools/trace_host_boundary.py DIR/hb_kernel_trace.csv DIR/hb_memory_copy_trace.csv"""
import csv, sys
k = list(csv.DictReader(open(sys.argv[1])))
m = list(csv.DictReader(open(sys.argv[2])))
ev = []
for r in k:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0].replace("void porla::", "").replace("porla::", "")[:28],
               r.get("Queue_Id", "?")))
for r in m:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "?").replace("MEMORY_COPY_", ""), "s" + r.get("Stream_Id", "?")))
ev.sort()
# the last call: the last 8 long host-to-device copies (4 ranges x scalars + points) and everything after the first of them
big = [i for i, e in enumerate(ev) if e[2] == "C" and "HOST_TO_DEVICE" in e[3] and e[1] - e[0] >= 50000]    # >= 50 us: an input range
ncopies = int(sys.argv[3]) if len(sys.argv) > 3 else 8
i0 = big[-ncopies]
t0 = ev[i0][0]
for s, e, kind, name, q in ev[i0:]:
    print("%9.1f -> %9.1f us  (%8.1f)  %s q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, kind, q, name))
