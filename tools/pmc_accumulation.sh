#!/bin/bash
# issue / wait counters of the two accumulation kernels (k_bucket_sum30 in a blocking 2^20-pair MSM, k_fb_commit in a 2^17-row
# commitment batch): vector instructions per wave and per addition, and where the wave-cycles go.  Two SQ passes (8 slots per pass).
# usage (GPU box): tools/pmc_accumulation.sh <out-file>
OUT=${1:-gpurun_out/pmc_accumulation.txt}
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
B="SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
rm -rf /tmp/pmc_acc_*
for pass in A B; do
  eval "C=\$$pass"
  rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_acc_msm_$pass -o run -- python3 $ROOT/tools/blocking_loop.py 20 3 > /tmp/pmc_acc_msm_$pass.txt 2>&1
  rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_acc_fb_$pass -o run -- python3 $ROOT/bench.py --workload kzg_commit --no-cpu --steps 2 --warmup 1 > /tmp/pmc_acc_fb_$pass.txt 2>&1
done
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("/tmp/pmc_acc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void porla::", "")
        if "k_bucket_sum30" not in k and "k_fb_commit<" not in k: continue
        agg[(k[:40], row["Counter_Name"])][0] += float(row["Counter_Value"]); agg[(k[:40], row["Counter_Name"])][1] += 1
with open(sys.argv[1], "w") as fo:
    per = {}
    for k, v in sorted(agg.items()):
        per[k] = v[0] / v[1]
        fo.write("%-40s %-22s per dispatch %.0f\n" % (k[0], k[1], v[0] / v[1]))
    # additions per dispatch: MSM 15 windows x 2^20 digits - 15 x 2^16 first entries (copies); commitments 2^17 rows x 128 x 15
    adds = {"k_bucket_sum30<porla::Bn254G1>": 15 * (1 << 20) - 15 * (1 << 16), "k_fb_commit<porla::Bn254G1, false>": (1 << 17) * 128 * 15}
    for name, n in adds.items():
        v = per.get((name[:40], "SQ_INSTS_VALU"))
        if v:
            fo.write("%s: %.0f vector instructions per addition (SQ_INSTS_VALU x 64 / %d additions)\n" % (name, v * 64 / n, n))
        g = per.get((name[:40], "GRBM_GUI_ACTIVE"))
        if v and g:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs, a wave instruction occupies its SIMD for >= 4 cycles
            fo.write("%s: vector issue occupies %.1f %% of the SIMD-cycles at 4 cycles per instruction (multiply-adds take 4.4-4.6)\n"
                     % (name, 100.0 * v * 4 / (g / 8 * 1024)))
print(open(sys.argv[1]).read())
PY
rm -rf /tmp/pmc_acc_*
