#!/bin/bash
# issue counters of the MAC-side stage kernels (bench.py --workload mac_encode: 2^15 MACs, both curves): vector instructions per
# butterfly and the share of the SIMD-cycles they occupy -- a stage is one wave per SIMD, so this is where its time goes.
# Two SQ passes.  usage (GPU box): tools/pmc_mac.sh <out-file>
OUT=${1:-gpurun_out/pmc_mac.txt}
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
B="SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
rm -rf /tmp/pmc_mac_*
for pass in A B; do
  eval "C=\$$pass"
  rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_mac_$pass -o run -- python3 $ROOT/bench.py --workload mac_encode --no-cpu --legs-out "" > /tmp/pmc_mac_$pass.txt 2>&1
done
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("/tmp/pmc_mac_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void porla::", "")
        if "k_mac_stage30_quad" not in k: continue
        agg[(k, row["Counter_Name"])][0] += float(row["Counter_Value"]); agg[(k, row["Counter_Name"])][1] += 1
with open(sys.argv[1], "w") as fo:
    per = {}
    for k, v in sorted(agg.items()):
        per[k] = v[0] / v[1]
        fo.write("%-50s %-22s per dispatch %.0f\n" % (k[0], k[1], v[0] / v[1]))
    for name in sorted({k[0] for k in per}):
        v, g, w = per.get((name, "SQ_INSTS_VALU")), per.get((name, "GRBM_GUI_ACTIVE")), per.get((name, "SQ_WAVES"))
        if v:
            # 2^14 butterflies per dispatch, four lanes each: SQ_INSTS_VALU counts wave instructions
            fo.write("%s: %.0f vector instructions per wave (= per butterfly's quad: one stream for its four lanes), %d waves\n" % (name, v / max(w, 1), w or 0))
        if v and g:
            fo.write("%s: vector issue occupies %.1f %% of the SIMD-cycles at 4 cycles per instruction (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)\n"
                     % (name, 100.0 * v * 4 / (g / 8 * 1024)))
print(open(sys.argv[1]).read())
PY
rm -rf /tmp/pmc_mac_*
