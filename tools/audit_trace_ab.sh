#!/bin/bash
for v in A B; do
  cp porla_amd/_ab/lib$v.so porla_amd/libmultiexp.so
  echo "== $v"
  PORLA_AUDIT_TRACE=1 python3 tools/bench_audit_flow.py --no-cpu 2> /tmp/tr_$v.txt > /dev/null
  python3 - $v <<'PY'
import sys,re,collections
acc=collections.defaultdict(list)
for l in open('/tmp/tr_%s.txt'%sys.argv[1]):
    m=re.match(r'\[audit\] (.*?)\s+([\d.]+) us',l)
    if m: acc[m.group(1).strip()].append(float(m.group(2)))
tot=0
for k,v in acc.items():
    v=v[len(v)//2:]
    med=sorted(v)[len(v)//2]; tot+=med
    print("  %-28s median %.1f us  (n=%d)"%(k,med,len(v)))
print("  sum of medians %.1f us"%tot)
PY
done
cp porla_amd/_ab/libB.so porla_amd/libmultiexp.so
