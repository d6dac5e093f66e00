#!/usr/bin/env python3
"""Generates tests/golden/icc_golden.json with the Python big-int restatement (oracle/icc_py.py:crebuild, align) of the data
half of Server::CRebuild_Cached (porla/Server/Server.hpp:1487-1833) and of the align_MAC scalar derivation (:531-541):
N in {8, 16, 64} rows x 4 columns, for both group orders -- input rows, the X rows after EVERY stage (N = 8, 16), the final X and Y rows
(values < LCM), the rows mod p_icc and the alignment scalars mod q.  The reference holds no vectors for this path and NTL is
absent from this image, so the fixture pins the restatement across implementations (Python here, the C oracle
in Z/LCM, the HIP engine on residue pairs), not against NTL."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import icc_py  # noqa: E402


def rows_for(n, ncols, seed):
    vals = []
    i = 0
    while len(vals) < n * ncols:
        d = hashlib.sha256(b"porla-icc-golden" + seed.to_bytes(4, "little") + i.to_bytes(8, "little")).digest()
        vals.append(int.from_bytes(d, "little"))
        i += 1
    return [vals[r * ncols:(r + 1) * ncols] for r in range(n)]


def hx(v, nbytes):
    return v.to_bytes(nbytes, "little").hex()


out = {"_format": "integers as little-endian hex: 32-byte input chunks, 64-byte values < LCM, 32-byte residues / scalars",
       "cases": []}
for curve in ("bn254", "secp256k1"):
    for n, ws in ((8, 0), (16, 5), (64, 33)):
        rows = rows_for(n, 4, n)
        rows[1][2] = 0                       # a zero symbol and an all-ones chunk
        rows[2][0] = 2**256 - 1
        trace = []
        X, Y = icc_py.crebuild(rows, curve, ws, trace)
        al = [icc_py.align(r, curve) for r in X]
        out["cases"].append({
            "curve": curve, "n": n, "ncols": 4, "write_step": ws,
            "rows": [[hx(v, 32) for v in r] for r in rows],
            # every stage for the two small sizes; the 64-row case keeps its last stage only (fixture size)
            "X_after_stage": [[[hx(v, 64) for v in r] for r in st] for st in (trace if n <= 16 else trace[-1:])],
            "X": [[hx(v, 64) for v in r] for r in X],
            "Y": [[hx(v, 64) for v in r] for r in Y],
            "X_mod_p_icc": [[hx(v, 32) for v in a[0]] for a in al],
            "X_align_scalars": [[hx(v, 32) for v in a[1]] for a in al],
        })
json.dump(out, open(os.path.join(HERE, "icc_golden.json"), "w"), indent=0)
print("wrote icc_golden.json:", len(out["cases"]), "cases")
