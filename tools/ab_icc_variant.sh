# same-box A/B of a compile-time variant of the ICC kernel: rebuilds icc.o on the GPU box with the given define
bench() { python bench.py --workload icc --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; }
DEF=${1:--DPORLA_ICC_TW_PREFETCH}
echo "== as built"; bench; bench
cd porla_amd/csrc && rm -f _build/icc.o && make CXXFLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include $DEF" ../libmultiexp.so > /dev/null 2>&1; cd ../..
echo "== $DEF"; bench; bench
timeout -k 10 300 python -m pytest tests/test_icc_gpu.py -x -q 2>&1 | tail -1
