#!/bin/bash
# one line per run: headline (two in flight) and blocking 2^20 BN254 MSM + the secp256k1 MSM, for tools/ab_lib.sh
python bench.py --workload bn254_msm --no-legs --no-commits --no-config3 --no-cpu --no-pmc --legs-out "" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bn254 2^20: %.1f Mmul/s %.4f ms/step, blocking %.4f ms, kernel %.4f' % (d['value'], d['ms_per_step'], d['blocking_ms_per_step'], d['roofline']['kernel_ms']))"
python bench.py --workload secp256k1_msm --no-cpu --no-pmc --legs-out "" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('secp256k1 2^20: %.1f Mmul/s %.4f ms/step' % (d['value'], d['ms_per_step']))"
