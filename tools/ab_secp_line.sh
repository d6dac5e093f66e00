#!/bin/bash
# one line per secp256k1 workload (MSM, IPA commitments, MAC encode both curves) for tools/ab_lib.sh
python bench.py --workload secp256k1_msm --no-cpu --no-pmc --legs-out "" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('secp256k1 2^20: %.1f Mmul/s %.4f ms/step kernel %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
python bench.py --workload ipa_commits --no-cpu --no-pmc --legs-out "" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ipa_commits: %.0f commits/s %.4f ms/step' % (d['value'], d['ms_per_step']))"
python bench.py --workload mac_encode --no-cpu --no-pmc --legs-out "" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('mac_encode:', {c: v['ms_per_step'] for c, v in d['curves'].items()})"
