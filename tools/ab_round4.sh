#!/bin/bash
# same-box A/B of the accumulation kernels: A = the round-3 library (8643627), B = this tree's (porla_amd/_ab/lib{A,B}.so)
# blocking 2^20-pair BN254 MSMs with HIP events around every kernel, then the 2^17-row commitment batch (kernel time of k_fb_commit)
export PORLA_LOOP_PROFILE=1
tools/ab_lib.sh 'python3 tools/blocking_loop.py 20 30 2>/dev/null | grep -v amdgpu.ids; python3 bench.py --workload kzg_commit --no-cpu --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(\"kzg_commit\", d[\"value\"], \"commits/s  k_fb_commit ms\", d[\"roofline\"][\"kernel_ms\"])"' ${1:-2}
