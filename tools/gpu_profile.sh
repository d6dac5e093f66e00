#!/bin/bash
# Runs on the GPU box (through gpurun): GPU parity tests, the default bench line, the rocprofv3 kernel-trace summary of
# the same bench command, and two separate PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950).
# Usage: [WORKLOAD=bn254_msm|kzg_commit|secp256k1_msm|icc] [SKIP_TESTS=1] tools/gpu_profile.sh <tag>     outputs under gpurun_out/<tag>/
set -u
TAG=${1:-run}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
WORKLOAD=${WORKLOAD:-bn254_msm}
BENCH="python3 $PWD/bench.py --workload $WORKLOAD ${BENCH_EXTRA:-}"
SHORT="--steps 10 --warmup 2"
if [ "${SKIP_TESTS:-0}" != 1 ]; then (timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5) > "$OUT/pytest.txt"; fi
(timeout 600 $BENCH 2>/dev/null | grep '^{') > "$OUT/bench_n1.json"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- $BENCH $SHORT --no-cpu > "$OUT/bench_under_rocprof.txt" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o run -- $BENCH --no-cpu --steps 3 --warmup 1 > "$OUT/pmc_fetch.txt" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o run -- $BENCH --no-cpu --steps 3 --warmup 1 > "$OUT/pmc_write.txt" 2>&1
cd "$OUT"
find . -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
python3 - <<'EOF'
import csv, glob, collections, json
out = {}
for name in ("fetch", "write"):
    files = glob.glob("pmc_%s/**/*counter_collection.csv" % name, recursive=True)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?")
            if k.startswith("(anonymous namespace)::"):      # kernels of an unnamed namespace (kzg_abi.hip)
                k = k[len("(anonymous namespace)::"):]
            k = k.split("(")[0]
            agg[(k, row.get("Counter_Name"))][0] += float(row.get("Counter_Value", 0))
            agg[(k, row.get("Counter_Name"))][1] += 1
    out[name] = {"%s|%s" % k: {"sum": v[0], "dispatches": v[1], "per_dispatch": v[0] / max(v[1], 1)} for k, v in agg.items()}
json.dump(out, open("pmc_summary.json", "w"), indent=1)
EOF
# keep the merge-back small: raw traces stay on the box
rm -rf "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write"
ls -la "$OUT"
