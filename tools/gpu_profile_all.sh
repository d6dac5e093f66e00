#!/bin/bash
# the round's final profile set: for each BASELINE workload the bench line, the rocprofv3 kernel-trace summary of the same
# command and the two PMC passes (tools/gpu_profile.sh); then the default line with every leg, and the crebuild chain.
# Usage (through gpurun): tools/gpu_profile_all.sh <tag>      outputs under gpurun_out/<tag>_<workload>/ and gpurun_out/<tag>_*.json
TAG=${1:-r03_z}
for w in bn254_msm kzg_commit secp256k1_msm icc audit_combine client_mac_batch ipa_commits mac_encode server_mix; do
  extra=""
  [ "$w" = bn254_msm ] && extra="--no-legs --no-commits --no-host-boundary"
  WORKLOAD=$w SKIP_TESTS=1 BENCH_EXTRA="$extra" bash tools/gpu_profile.sh ${TAG}_$w > gpurun_out/${TAG}_$w.log 2>&1
  echo "$w done: $(head -c 300 gpurun_out/${TAG}_$w/bench_n1.json)"
done
python3 bench.py > gpurun_out/${TAG}_bench_default_line.json 2> gpurun_out/${TAG}_bench_default_line.err
echo "default line: $(head -c 200 gpurun_out/${TAG}_bench_default_line.json)"
python3 bench.py --workload crebuild > gpurun_out/${TAG}_bench_crebuild.json 2> gpurun_out/${TAG}_bench_crebuild.err
echo "crebuild: $(head -c 200 gpurun_out/${TAG}_bench_crebuild.json)"
