#!/bin/bash
# one line per workload (ICC encode, KZG commitments, IPA commitments, CRebuild stage) for tools/ab_lib.sh
for w in icc kzg_commit ipa_commits crebuild; do
python bench.py --workload $w --no-cpu --no-pmc --legs-out "" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w: %s %s, %.4f ms/step, bit-exact %s' % (d['value'], d['unit'], d['ms_per_step'], d.get('bit_exact_vs_oracle')))"
done
