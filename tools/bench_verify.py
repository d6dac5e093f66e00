"""verify_proof latency (host pairing check) on the golden openings: best of 3 x 200 calls"""
import json, time, sys
sys.path.insert(0,'.')
from porla_amd import multiexp as mx
G=json.load(open('./tests/golden/bn254_golden.json'))
kz=G['kzg']
mx.init_key(bytes.fromhex(kz['tau']), bytes.fromhex(kz['alpha']))
blob=mx.init_SRS(kz['n']); mx.init_SRS_from_data(kz['n'], blob)
ops=[[bytes.fromhex(op[k]) for k in ("commitment","H","point","claim")] for c in kz['cases'] for op in c['open']]
for _ in range(20): mx.verify_proof(*ops[0])
best=1e9
for rep in range(3):
    t=time.perf_counter()
    for i in range(200): mx.verify_proof(*ops[i%len(ops)])
    best=min(best,(time.perf_counter()-t)/200*1e3)
print("verify_proof: %.3f ms"%best)
