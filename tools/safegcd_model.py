"""Model of porla_amd/csrc/inv30.hip.h:fe_inv_safegcd on Python integers: the same 25 rounds of 30 Bernstein-Yang division
steps on nine signed 30-bit limbs, with every value the device code keeps in a 32-bit or 64-bit register asserted to fit.
usage: safegcd_model.py [samples per modulus]"""
import random
import sys
M30 = (1 << 30) - 1
def limbs_signed(x):      # 9 limbs, 0..7 in [0,2^30), limb 8 signed
    l = [(x >> (30 * i)) & M30 for i in range(8)]
    l.append(x >> 240)
    return l
def val(l): return sum(v << (30 * i) for i, v in enumerate(l))
def i32(x):
    assert -(1 << 31) <= x < (1 << 31), x
    return x
def i64(x):
    assert -(1 << 63) <= x < (1 << 63), x
    return x
def inv(V, p):
    INV = (-pow(p, -1, 1 << 30)) % (1 << 30)          # -p^-1 mod 2^30
    P = limbs_signed(p)
    f, g = limbs_signed(p), limbs_signed(V)
    d, e = [0] * 9, [1] + [0] * 8
    delta = 1
    for it in range(25):
        f0 = (f[0] | (f[1] << 30)) & 0xffffffff
        g0 = (g[0] | (g[1] << 30)) & 0xffffffff
        u, v, q, r = 1, 0, 0, 1
        for i in range(30):
            godd = g0 & 1
            sw = godd and delta > 0
            nf = g0 if sw else f0
            ng = (-f0) & 0xffffffff if sw else g0
            nu, nv, nq, nr = (q, r, -u, -v) if sw else (u, v, q, r)
            if sw: delta = -delta
            if godd:
                ng = (ng + nf) & 0xffffffff; nq += nu; nr += nv
            delta += 1
            g0 = ng >> 1; f0 = nf
            u, v, q, r = i32(nu * 2), i32(nv * 2), i32(nq), i32(nr)
        # update f, g
        cf = i64(u * f[0] + v * g[0]); cg = i64(q * f[0] + r * g[0])
        assert cf & M30 == 0 and cg & M30 == 0
        cf >>= 30; cg >>= 30
        nfl, ngl = [0] * 9, [0] * 9
        for i in range(1, 9):
            cf = i64(cf + u * f[i] + v * g[i]); cg = i64(cg + q * f[i] + r * g[i])
            nfl[i - 1] = cf & M30; cf >>= 30
            ngl[i - 1] = cg & M30; cg >>= 30
        nfl[8] = i32(cf); ngl[8] = i32(cg)
        # update d, e
        sd = d[8] < 0; se = e[8] < 0
        md = (u if sd else 0) + (v if se else 0)
        me = (q if sd else 0) + (r if se else 0)
        cd = i64(u * d[0] + v * e[0]); ce = i64(q * d[0] + r * e[0])
        # subtract k p: k = (cd + md p0) * p0^-1 mod 2^30 = -(...)*INV
        kd = (-(((cd + md * P[0]) & M30) * INV)) & M30    # k with (cd + md p0 - k p0) = 0 mod 2^30  -> k = x * p0^-1 = -x*INV
        ke = (-(((ce + me * P[0]) & M30) * INV)) & M30
        md -= kd; me -= ke
        i32(md); i32(me)
        cd = i64(cd + md * P[0]); ce = i64(ce + me * P[0])
        assert cd & M30 == 0 and ce & M30 == 0, (cd & M30, ce & M30)
        cd >>= 30; ce >>= 30
        ndl, nel = [0] * 9, [0] * 9
        for i in range(1, 9):
            cd = i64(cd + u * d[i] + v * e[i] + md * P[i]); ce = i64(ce + q * d[i] + r * e[i] + me * P[i])
            ndl[i - 1] = cd & M30; cd >>= 30
            nel[i - 1] = ce & M30; ce >>= 30
        ndl[8] = i32(cd); nel[8] = i32(ce)
        f, g, d, e = nfl, ngl, ndl, nel
        assert -2 * p < val(d) < p and -2 * p < val(e) < p
    assert val(g) == 0 and val(f) in (1, -1), (val(f), val(g))
    dv = val(d)
    if f[8] < 0: dv = -dv
    return dv % p
rnd = random.Random(1)
for p in (21888242871839275222246405745257275088696311157297823662689037894645226208583, (1 << 256) - (1 << 32) - 977):
    for V in [1, 2, p - 1, p - 2, (p - 1) // 2, 1 << 255 if (1 << 255) < p else 5, 3] + [rnd.randrange(1, p) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3000)]:
        V %= p
        assert inv(V, p) * V % p == 1
print("safegcd model: ok")
