#!/usr/bin/env python3
"""Writes tests/golden/eip197_kat.json: known answers of the alt_bn128 pairing check (EIP-197) from an INDEPENDENT implementation
-- go-ethereum's bn256Pairing precompile vectors (core/vm/testdata/precompiles/bn256Pairing.json: jeff1 .. jeff6, empty_data,
one_point, two_point_match_2), the external pin for the G2 encoding, the G2 generator and the pairing behind verify_proof
(porla/main.go:177-193).  The reference holds no vectors for this path and gnark cannot run here.

There is no network in the build container, so the jeff vectors below were typed in from the published file; this script does not
trust them: every point must lie on its curve, every G2 point in the order-r subgroup, and the expected result must be what the
Python big-int pairing of oracle/bn254_pairing_py.py (a third formulation, independent of both geth's and the engine's) computes --
for the five "true" vectors a mistyped digit cannot survive that (a wrong point is off the curve; a wrong point ON the curve gives
product != 1), and the "false" vector jeff6 is checked to be jeff1 with its second G1 point negated, which is how it was built.
Run from the repo root:  python tests/golden/gen_eip197_kat.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bn254_py as o  # noqa: E402
import bn254_pairing_py as pp  # noqa: E402

G2 = ("198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c21800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed"
      "090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa")
JEFF1_A = ("1c76476f4def4bb94541d57ebba1193381ffa7aa76ada664dd31c16024c43f593034dd2920f673e204fee2811c678745fc819b55d3e9d294e45c9b03a76aef41"
           "209dd15ebff5d46c4bd888e51a93cf99a7329636c63514396b4a452003a35bf704bf11ca01483bfa8b34b43561848d28905960114c8ac04049af4b6315a41678"
           "2bb8324af6cfc93537a2ad1a445cfd0ca2a71acd7ac41fadbf933c2a51be344d120a2a4cf30c1bf9845f20c6fe39e07ea2cce61f0c9bb048165fe5e4de877550")
Q_1213 = ("1213d2149b006137fcfb23036606f848d638d576a120ca981b5b1a5f9300b3ee2276cf730cf493cd95d64677bbb75fc42db72513a4c1e387b476d056f80aa75f"
          "21ee6226d31426322afcda621464d0611d226783262e21bb3bc86b537e986237096df1f82dff337dd5972e32a8ad43e28a78a96a823ef1cd4debe12b6552ea5f")
VECTORS = [
    ("jeff1", JEFF1_A + "111e129f1cf1097710d41c4ac70fcdfa5ba2023c6ff1cbeac322de49d1b6df7c2032c61a830e3c17286de9462bf242fca2883585b93870a73853face6a6bf411" + G2, 1),
    ("jeff2", "2eca0c7238bf16e83e7a1e6c5d49540685ff51380f309842a98561558019fc0203d3260361bb8451de5ff5ecd17f010ff22f5c31cdf184e9020b06fa5997db84" + Q_1213 +
              "06967a1237ebfeca9aaae0d6d0bab8e28c198c5a339ef8a2407e31cdac516db922160fa257a5fd5b280642ff47b65eca77e626cb685c84fa6d3b6882a283ddd1" + G2, 1),
    ("jeff3", "0f25929bcb43d5a57391564615c9e70a992b10eafa4db109709649cf48c50dd216da2f5cb6be7a0aa72c440c53c9bbdfec6c36c7d515536431b3a865468acbba"
              "2e89718ad33c8bed92e210e81d1853435399a271913a6520736a4729cf0d51eb01a9e2ffa2e92599b68e44de5bcf354fa2642bd4f26b259daa6f7ce3ed57aeb3"
              "14a9a87b789a58af499b314e13c3d65bede56c07ea2d418d6874857b70763713178fb49a2d6cd347dc58973ff49613a20757d0fcc22079f9abd10c3baee24590"
              "1b9e027bd5cfc2cb5db82d4dc9677ac795ec500ecd47deee3b5da006d6d049b811d7511c78158de484232fc68daf8a45cf217d1c2fae693ff5871e8752d73b21" + G2, 1),
    ("jeff4", "2f2ea0b3da1e8ef11914acf8b2e1b32d99df51f5f4f206fc6b947eae860eddb6068134ddb33dc888ef446b648d72338684d678d2eb2371c61a50734d78da4b72"
              "25f83c8b6ab9de74e7da488ef02645c5a16a6652c3c71a15dc37fe3a5dcb7cb122acdedd6308e3bb230d226d16a105295f523a8a02bfc5e8bd2da135ac4c245d"
              "065bbad92e7c4e31bf3757f1fe7362a63fbfee50e7dc68da116e67d600d9bf6806d302580dc0661002994e7cd3a7f224e7ddc27802777486bf80f40e4ca3cfdb"
              "186bac5188a98c45e6016873d107f5cd131f3a3e339d0375e58bd6219347b008122ae2b09e539e152ec5364e7e2204b03d11d3caa038bfc7cd499f8176aacbee"
              "1f39e4e4afc4bc74790a4a028aff2c3d2538731fb755edefd8cb48d6ea589b5e283f150794b6736f670d6a1033f9b46c6f5204f50813eb85c8dc4b59db1c5d39"
              "140d97ee4d2b36d99bc49974d18ecca3e7ad51011956051b464d9e27d46cc25e0764bb98575bd466d32db7b15f582b2d5c452b36aa394b789366e5e3ca5aabd4"
              "15794ab061441e51d01e94640b7e3084a07e02c78cf3103c542bc5b298669f211b88da1679b0b64a63b7e0e7bfe52aae524f73a55be7fe70c7e9bfc94b4cf0da" + Q_1213, 1),
    ("jeff5", "20a754d2071d4d53903e3b31a7e98ad6882d58aec240ef981fdf0a9d22c5926a29c853fcea789887315916bbeb89ca37edb355b4f980c9a12a94f30deeed3021" + Q_1213 +
              "1abb4a25eb9379ae96c84fff9f0540abcfc0a0d11aeda02d4f37e4baf74cb0c11073b3ff2cdbb38755f8691ea59e9606696b3ff278acfc098fa8226470d03869"
              "217cee0a9ad79a4493b5253e2e4e3a39fc2df38419f230d341f60cb064a0ac290a3d76f140db8418ba512272381446eb73958670f00cf46f1d9e64cba057b53c"
              "26f64a8ec70387a13e41430ed3ee4a7db2059cc5fc13c067194bcc0cb49a98552fd72bd9edb657346127da132e5b82ab908f5816c826acb499e22f2412d1a2d7"
              "0f25929bcb43d5a57391564615c9e70a992b10eafa4db109709649cf48c50dd2198a1f162a73261f112401aa2db79c7dab1533c9935c77290a6ce3b191f2318d" + G2, 1),
    ("jeff6", JEFF1_A + "111e129f1cf1097710d41c4ac70fcdfa5ba2023c6ff1cbeac322de49d1b6df7c103188585e2364128fe25c70558f1560f4f9350baf3959e603cc91486e110936" + G2, 0),
    ("empty_data", "", 1),
    ("one_point", "%064x%064x" % (1, 2) + G2, 0),
    ("two_point_match_2", "%064x%064x" % (1, 2) + G2 + "%064x%064x" % (1, o.P - 2) + G2, 1),
]


def parse(h):
    raw = bytes.fromhex(h)
    assert len(raw) % 192 == 0
    pairs = []
    for k in range(len(raw) // 192):
        c = raw[192 * k:192 * (k + 1)]
        x, y = int.from_bytes(c[:32], "big"), int.from_bytes(c[32:64], "big")
        p = None if x == 0 and y == 0 else (x, y)
        pairs.append((p, pp.g2_from_eip197(c[64:])))
    return pairs


def main():
    assert bytes.fromhex(G2) == pp.g2_to_eip197(pp.G2_GEN) and pp.g2_in_subgroup(pp.G2_GEN)
    out = {"comment": "go-ethereum bn256Pairing precompile vectors (EIP-197 input layout: 192 bytes per pair), each re-validated by "
                      "tests/golden/gen_eip197_kat.py: points on their curves, G2 points in the order-r subgroup, expected result "
                      "recomputed with the Python big-int pairing of oracle/bn254_pairing_py.py", "vectors": []}
    for name, h, want in VECTORS:
        pairs = parse(h)
        for p, q in pairs:
            assert o.is_on_curve(p) and pp.g2_in_subgroup(q), name
        assert pp.pairing_product_is_one(pairs) == bool(want), name
        out["vectors"].append({"name": name, "input": h, "pairs": len(pairs), "expected": want})
        print("%-18s %d pair(s) -> %d  ok" % (name, len(pairs), want))
    j1, j6 = parse(VECTORS[0][1]), parse(VECTORS[5][1])
    assert j6[0] == j1[0] and j6[1][1] == j1[1][1] and j6[1][0] == o.g1_neg(j1[1][0])     # jeff6 = jeff1 with -P2
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "eip197_kat.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path)


if __name__ == "__main__":
    main()
