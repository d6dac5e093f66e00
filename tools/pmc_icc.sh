#!/bin/bash
# instruction counters of the ICC encode kernels (both field forms): SQ_INSTS_VALU / LDS / SALU, wave cycles
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out
for mode in 1 0; do
  export PORLA_ICC_F30=$mode
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $OUT/pmc_icc_$mode -o run -- python3 /root/repo/bench.py --workload icc --steps 3 --warmup 1 --no-cpu > $OUT/pmc_icc_$mode.txt 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for mode in (1, 0):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("/root/repo/gpurun_out/pmc_icc_%d/**/*counter_collection.csv" % mode, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if "k_icc_fused" not in k: continue
            agg[(k[:60], row["Counter_Name"])][0] += float(row["Counter_Value"]); agg[(k[:60], row["Counter_Name"])][1] += 1
    for k, v in sorted(agg.items()):
        print("f30=%d" % mode, k[0], k[1], "per dispatch %.0f" % (v[0] / v[1]))
PY
rm -rf $OUT/pmc_icc_1 $OUT/pmc_icc_0
