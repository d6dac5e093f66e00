bench() { python bench.py --workload icc --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; }
echo "== tile 512 (as built)"; bench; bench
cd porla_amd/csrc && rm -f _build/icc.o && make CXXFLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include -DPORLA_ICC_TILE=1024" ../libmultiexp.so > /dev/null 2>&1; cd ../..
echo "== tile 1024"; bench; bench
cd porla_amd/csrc && rm -f _build/icc.o && make CXXFLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include -DPORLA_ICC_TILE=256" ../libmultiexp.so > /dev/null 2>&1; cd ../..
echo "== tile 256"; bench; bench
