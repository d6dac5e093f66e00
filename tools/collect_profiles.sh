#!/bin/bash
# copies the summaries of a tools/gpu_profile_all.sh run from gpurun_out/ (scratch) into profiles/ (tracked), and refreshes
# profiles/pmc_latest*.json, which bench.py reads for roofline.traffic.   usage: tools/collect_profiles.sh <tag>
TAG=${1:-r03_z}
for w in bn254_msm kzg_commit secp256k1_msm icc audit_combine client_mac_batch ipa_commits mac_encode server_mix; do
  d=gpurun_out/${TAG}_$w
  [ -d "$d" ] || continue
  cp $d/bench_n1.json profiles/${TAG}_bench_n1_$w.json
  cp $d/kernel_stats.csv profiles/${TAG}_kernel_stats_$w.csv
  cp $d/pmc_summary.json profiles/${TAG}_pmc_fetch_write_$w.json
  grep -v "^$" $d/bench_under_rocprof.txt | grep "^{" > profiles/${TAG}_bench_under_rocprof_$w.txt
  if [ "$w" = bn254_msm ]; then cp $d/pmc_summary.json profiles/pmc_latest.json; else cp $d/pmc_summary.json profiles/pmc_latest_$w.json; fi
done
for f in bench_default_line bench_crebuild; do
  [ -f gpurun_out/${TAG}_$f.json ] && grep "^{" gpurun_out/${TAG}_$f.json > profiles/${TAG}_$f.json
done
# the default run's whole stdout: the `LEG <name> {...}` lines carry every leg's full object, the last line is the compact one
[ -f gpurun_out/${TAG}_bench_default_line.json ] && cp gpurun_out/${TAG}_bench_default_line.json profiles/${TAG}_bench_default_stdout.txt
ls -la profiles/${TAG}_* | wc -l
