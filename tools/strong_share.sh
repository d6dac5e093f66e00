#!/bin/bash
# one rank's share of a strong-scaled 2^20-pair MSM on one GPU: per-step time (three in flight), blocking time and the per-kernel
# breakdown of bench.py --workload strong_2p20 at 2^17 .. 2^20 pairs
for L in ${@:-17 18 19 20}; do
python bench.py --workload strong_2p20 --log2n $L --no-cpu --no-pmc --legs-out "" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); rl=d.get('roofline') or {}
print('2^$L: %.4f ms/step (%.1f Mmul/s), blocking %s ms, shape %s, kernels %s' % (d['ms_per_step'], d['value'], d.get('blocking_ms_per_step'), d['config'].get('shape'), rl.get('all_kernels_ms')))"
done
