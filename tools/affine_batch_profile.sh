#!/bin/bash
# tools/affine_batch_ubench (batched-affine best case vs the XYZZ mixed addition): rates, then the instruction and byte counters
# of the same kernels under rocprofv3 (separate passes: FETCH_SIZE and WRITE_SIZE cannot share one on gfx950)
OUT=$PWD/gpurun_out
BIN=$PWD/tools/affine_batch_ubench
cd /tmp && export TMPDIR=/tmp
$BIN 128 > $OUT/r03_d_affine_batch_rates.json 2> $OUT/r03_d_affine_batch_rates.err
cat $OUT/r03_d_affine_batch_rates.json
for pass in "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_aff_$tag -o run -- $BIN 64 > $OUT/pmc_aff_$tag.txt 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("/root/repo/gpurun_out/pmc_aff_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if k not in ("k_xyzz_stream", "k_affine_batch", "k_inv_only"): continue
        # the ubench launches every kernel 4 times per k (1 warm + 3 timed), k = 8, 16, 32, 64: keep the launches by grid-independent key
        agg[(k, row["Counter_Name"])][0] += float(row["Counter_Value"]); agg[(k, row["Counter_Name"])][1] += 1
lanes = 256 * 4 * 256
adds = {"k_xyzz_stream": 2 * (8 + 16 + 32 + 64) * lanes * 4, "k_affine_batch": (8 + 16 + 32 + 64) * lanes * 4, "k_inv_only": 4 * lanes * 4}
print("kernel counter total per_add_or_inversion   (all launches of the k = 8..64 sweep; per-lane units for SQ_INSTS_* x 64)")
for (k, c), v in sorted(agg.items()):
    per = v[0] / adds[k]
    if c.startswith("SQ_INSTS"): per *= 64            # wave instructions -> lane instructions per addition
    if c in ("FETCH_SIZE", "WRITE_SIZE"): per *= 1024 * (2 if c == "FETCH_SIZE" else 1)   # KiB -> bytes; FETCH_SIZE doubled (gfx950)
    print(k, c, "%.0f" % v[0], "%.1f" % per)
PY
rm -rf $OUT/pmc_aff_*
