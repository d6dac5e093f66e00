#!/bin/bash
# kernel timeline of ONE porla_kzg_crebuild_stage_device step (2^15 rows): which kernels overlap, per queue
# usage (GPU box): tools/trace_crebuild_stage.sh <out-file>
OUT=${1:-gpurun_out/crebuild_stage_timeline.txt}
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_cs
PORLA_CREBUILD_ONLY_STAGE=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_cs -o run -- python3 $ROOT/tools/bench_crebuild.py --steps 3 --warmup 1 --no-cpu > /tmp/trace_cs.txt 2>&1
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/trace_cs/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the start of its MAC side (k_mac_load30<...>, the unscaled load) or of its data side, whichever came first
ml = [i for i, r in enumerate(rows) if "k_mac_load30<" in r["Kernel_Name"] and "quad" not in r["Kernel_Name"]][-1]
first_pass = [i for i, r in enumerate(rows) if "k_icc_split30" in r["Kernel_Name"] and i <= ml + 3]
start = min(ml, first_pass[-1] if first_pass else ml)
if first_pass and len(first_pass) >= 2 and ml - first_pass[-2] <= 3:
    start = min(start, first_pass[-2])
sel = rows[start:]
t0 = int(sel[0]["Start_Timestamp"])
with open(sys.argv[1], "w") as fo:
    for r in sel:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void porla::", "").replace("porla::", "")[:40]
        fo.write("%-40s q %-3s start %9.1f us  end %9.1f  dur %8.1f  vgpr %s lds %s grid %s\n" % (
            name, r["Queue_Id"], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?"),
            r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
print(open(sys.argv[1]).read())
PY
rm -rf /tmp/trace_cs
