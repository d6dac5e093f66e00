#!/bin/bash
# refresh of the two MSM workloads' profile files + the default line after a change that touches only k_bucket_sum30 (the other
# workloads' kernels, and their files of the same set, are unchanged).  Usage (through gpurun): tools/gpu_profile_msm.sh <tag>
TAG=${1:-r05_zz}
for w in bn254_msm secp256k1_msm; do
  extra=""
  [ "$w" = bn254_msm ] && extra="--no-legs --no-commits --no-host-boundary"
  WORKLOAD=$w SKIP_TESTS=1 BENCH_EXTRA="$extra" bash tools/gpu_profile.sh ${TAG}_$w > gpurun_out/${TAG}_$w.log 2>&1
  echo "$w done: $(head -c 200 gpurun_out/${TAG}_$w/bench_n1.json)"
done
python3 bench.py > gpurun_out/${TAG}_bench_default_line.json 2> gpurun_out/${TAG}_bench_default_line.err
echo "default line: $(head -c 200 gpurun_out/${TAG}_bench_default_line.json)"
