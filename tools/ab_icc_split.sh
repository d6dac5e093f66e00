# same-box A/B of the two reduced-radix ICC kernels: icc30_split.hip.h (default) against icc30.hip.h (PORLA_ICC_SPLIT=0)
bench() { python bench.py --workload icc --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'])"; }
for r in 1 2 3; do
  echo "== split (default)"; bench
  echo "== PORLA_ICC_SPLIT=0"; PORLA_ICC_SPLIT=0 bench
done
