#!/bin/bash
# where the ICC encode kernels' wave-cycles go: issue / wait counters of both reduced-radix kernels (PORLA_ICC_SPLIT=1 / 0),
# two SQ passes each (8 SQ slots per pass on gfx950) + GRBM_GUI_ACTIVE for the clock the chip held
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out
rocprofv3 -L > $OUT/rocprof_counters_list.txt 2>&1
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
B="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM"
for mode in 1 0; do
  export PORLA_ICC_SPLIT=$mode
  for pass in A B; do
    eval "C=\$$pass"
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_stall_${mode}_$pass -o run -- python3 /root/repo/bench.py --workload icc --steps 3 --warmup 1 --no-cpu > $OUT/pmc_stall_${mode}_$pass.txt 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections
for mode in (1, 0):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("/root/repo/gpurun_out/pmc_stall_%d_*/**/*counter_collection.csv" % mode, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if "k_icc_" not in k or "twiddle" in k: continue
            agg[(k[:70], row["Counter_Name"])][0] += float(row["Counter_Value"]); agg[(k[:70], row["Counter_Name"])][1] += 1
    for k, v in sorted(agg.items()):
        print("split=%d" % mode, k[0], k[1], "per dispatch %.0f" % (v[0] / v[1]))
PY
tail -2 $OUT/pmc_stall_1_B.txt
rm -rf $OUT/pmc_stall_?_?
