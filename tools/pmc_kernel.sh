#!/bin/bash
# SQ issue / wait counters of one kernel of one bench workload: two rocprofv3 --pmc passes, summed per instantiation.
# usage (GPU box): tools/pmc_kernel.sh <workload> <kernel-name-substring> <out-file> [extra bench args]
W=$1; K=$2; OUT=${3:-gpurun_out/pmc_kernel.txt}; shift 3
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
B="SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
rm -rf /tmp/pmc_k_*
for pass in A B; do
  eval "C=\$$pass"
  rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_k_$pass -o run -- python3 $ROOT/bench.py --workload $W --no-cpu --no-pmc --legs-out "" "$@" > /tmp/pmc_k_$pass.txt 2>&1
done
cd $ROOT
python3 - "$OUT" "$K" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("/tmp/pmc_k_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void porla::", "")
        if sys.argv[2] not in k: continue
        agg[(k, row["Counter_Name"])][0] += float(row["Counter_Value"]); agg[(k, row["Counter_Name"])][1] += 1
with open(sys.argv[1], "w") as fo:
    per = {}
    for k, v in sorted(agg.items()):
        per[k] = v[0] / v[1]
        fo.write("%-60s %-22s per dispatch %.0f\n" % (k[0][:60], k[1], v[0] / v[1]))
    for name in sorted({k[0] for k in per}):
        g = lambda c: per.get((name, c))
        if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
            fo.write("%s: of a wave's life: VALU active %.1f %%, waiting (SQ_WAIT_ANY) %.1f %%, LDS active %.1f %%; %.0f vector / %.0f scalar / %.0f LDS instructions per wave\n"
                     % (name[:60], 100 * g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"), 100 * (g("SQ_WAIT_ANY") or 0) / g("SQ_WAVE_CYCLES"),
                        100 * (g("SQ_ACTIVE_INST_LDS") or 0) / g("SQ_WAVE_CYCLES"), g("SQ_INSTS_VALU") / g("SQ_WAVES"),
                        (g("SQ_INSTS_SALU") or 0) / g("SQ_WAVES"), (g("SQ_INSTS_LDS") or 0) / g("SQ_WAVES")))
print(open(sys.argv[1]).read())
PY
rm -rf /tmp/pmc_k_*
