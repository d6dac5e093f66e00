# compute_multi_exp on host buffers at 2^20 pairs under the range-pipeline knobs (each line its own process: the knobs are read once)
# (profiles/r02_g_host_boundary_sweep.txt also holds the two variants that were measured and removed: an upload thread, unequal ranges)
run() { echo "== $*"; env "$@" timeout -k 10 120 python tools/bench_host_boundary.py --json 20 | grep -o '"ms": [0-9.]*, "Mmul_s": [0-9.]*, "pair_ranges": [0-9]*'; }
run PORLA_MSM_SHARED_BUCKETS=0
run PORLA_MSM_SHARED_BUCKETS=1
run PORLA_MSM_MULTI_C=16
run PORLA_MSM_MULTI_C=17
run PORLA_MSM_PIPELINE=2
run PORLA_MSM_PIPELINE=3
run PORLA_MSM_PIPELINE=6 PORLA_MSM_MIN_RANGE=65536
run PORLA_MSM_PIPELINE=8 PORLA_MSM_MIN_RANGE=65536
