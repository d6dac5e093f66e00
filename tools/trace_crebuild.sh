#!/bin/bash
# kernel timeline of the two-stream crebuild chain (tools/bench_crebuild.py): which kernels overlap
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_crebuild -o run -- python3 /root/repo/tools/bench_crebuild.py --steps 2 --warmup 1 --no-cpu > $OUT/trace_crebuild.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/root/repo/gpurun_out/trace_crebuild/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open("/root/repo/gpurun_out/r03_f_crebuild_all_kernels.csv", "w") as fo:
    t00 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        fo.write("%s,%s,%.1f,%.1f\n" % (r["Kernel_Name"].split("(")[0].replace("void porla::", "")[:40], r["Queue_Id"],
                                       (int(r["Start_Timestamp"]) - t00) / 1e3, (int(r["End_Timestamp"]) - t00) / 1e3))
side = [r for r in rows if r["Queue_Id"] != rows[-1]["Queue_Id"] and "k_mac" in r["Kernel_Name"]]
last = side[-34:]                      # the MAC encodes of the last two-stream step: 2 x (load + 15 stages + finish)
w0, w1 = int(last[0]["Start_Timestamp"]), int(last[-1]["End_Timestamp"])
win = [r for r in rows if int(r["End_Timestamp"]) > w0 and int(r["Start_Timestamp"]) < w1]
print("two-stream step: MAC chain %.1f us; kernels overlapping it:" % ((w1 - w0) / 1e3))
for r in win:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void porla::", "").replace("porla::", "")[:34]
    print("%-34s q %s  start %9.1f  end %9.1f  dur %8.1f us" % (name, r["Queue_Id"], (s - w0) / 1e3, (e - w0) / 1e3, (e - s) / 1e3))
PY
rm -rf $OUT/trace_crebuild
