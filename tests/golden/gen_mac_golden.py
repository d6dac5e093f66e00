#!/usr/bin/env python3
"""Generates tests/golden/mac_golden.json with the Python big-int restatement (oracle/icc_py.py:mac_crebuild) of the MAC
halves of Server::CRebuild_Cached (porla/Server/Server.hpp:1523-1536, 1590-1609, 1658-1676).  The reference holds no
vectors for this path and its providers (gnark / libsecp256k1 / NTL) cannot run in this image, so the fixture pins the
restatement against itself across implementations (Python here, C oracle, HIP engine), not against the reference."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import icc_py  # noqa: E402

G = {"bn254": (1, 2),
     "secp256k1": (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
                   0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)}


def pb(p):
    return (bytes(64) if p is None else p[0].to_bytes(32, "big") + p[1].to_bytes(32, "big")).hex()


out = {"cases": []}
for curve in ("bn254", "secp256k1"):
    for n, ws in ((4, 0), (8, 5)):
        macs = [icc_py.ec_mul(curve, G[curve], 0xC0FFEE + 1009 * i * i) for i in range(n)]
        macs[1] = None
        X, Y = icc_py.mac_crebuild(macs, curve, ws)
        out["cases"].append({"curve": curve, "n": n, "write_step": ws, "macs": [pb(p) for p in macs],
                             "X": [pb(p) for p in X], "Y": [pb(p) for p in Y]})
json.dump(out, open(os.path.join(HERE, "mac_golden.json"), "w"), indent=0)
print("wrote", len(out["cases"]), "cases")
