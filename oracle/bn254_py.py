"""ORACLE (test infrastructure only) -- Python big-int restatement of the BN254 / KZG
plug-in that Porla reaches through libmultiexp.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
file.  The product path (porla_amd/, include/) never does.

Every function follows porla/main.go of the reference (cgo wrapper over gnark-crypto
v0.6.0, which is NOT vendored under /root/reference and cannot be fetched here).  The
arithmetic is the public definition of BN254 (alt_bn128): y^2 = x^3 + 3 over Fp,
G = (1, 2), prime order r.  PARITY PIN: the reference holds no tests / golden vectors
for this path -> "parity unpinned" w.r.t. gnark itself; what is pinned
is (1) the EIP-196 known answers for 2G / 3G, (2) r*G = infinity, (3) the in-reference
identity compute_digest(f) == alpha * compute_digest_from_srs(f) (main.go:81-88 vs :114)
and (4) the KZG opening identity C - y*G == (tau - z)*H.

Byte formats (gnark-crypto v0.6.0 ecc/bn254/marshal.go, restated from its published
format): G1 uncompressed = X||Y, 32-byte big-endian each, top two bits of byte 0 are the
flag bits (00 = uncompressed); the point at infinity is 64 zero bytes (BN254 has no spare
bit for an "uncompressed infinity" flag).  G1 compressed = X with flags 10 (Y is the
lexicographically smallest root) / 11 (largest) / 01 (infinity).  fr/fp SetBytes reduce
modulo the field order.
"""
import hashlib

P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
B = 3
G1 = (1, 2)
INF = None  # affine infinity

M_MASK = 0xC0
M_UNCOMPRESSED = 0x00
M_COMPRESSED_SMALLEST = 0x80
M_COMPRESSED_LARGEST = 0xC0
M_COMPRESSED_INFINITY = 0x40


# ---------------------------------------------------------------- Fp / Fr helpers
def fp_inv(a):
    return pow(a, P - 2, P)


def fp_sqrt(a):
    """p = 3 mod 4 -> a^((p+1)/4); returns None when a is not a square."""
    s = pow(a, (P + 1) // 4, P)
    return s if s * s % P == a % P else None


def fr_from_bytes(b):
    """fr.Element.SetBytes (main.go:34,77,127,...): big-endian, reduced mod r."""
    return int.from_bytes(bytes(b), "big") % R


def fr_to_bytes(x):
    return int(x % R).to_bytes(32, "big")


# ---------------------------------------------------------------- G1 affine group law
def is_on_curve(pt):
    if pt is INF:
        return True
    x, y = pt
    return (y * y - x * x * x - B) % P == 0


def g1_neg(pt):
    if pt is INF:
        return INF
    return (pt[0], (-pt[1]) % P)


def g1_double(pt):
    if pt is INF:
        return INF
    x, y = pt
    if y == 0:
        return INF
    lam = 3 * x * x * fp_inv(2 * y) % P
    x3 = (lam * lam - 2 * x) % P
    return (x3, (lam * (x - x3) - y) % P)


def g1_add(p1, p2):
    """bn254.G1Affine.Add (main.go:200)."""
    if p1 is INF:
        return p2
    if p2 is INF:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return INF
        return g1_double(p1)
    lam = (y2 - y1) * fp_inv(x2 - x1) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


# Jacobian (a = 0) used only to make the oracle's scalar-mult tolerable in Python.
def _jac_double(p):
    X, Y, Z = p
    if Z == 0 or Y == 0:
        return (1, 1, 0)
    A = X * X % P
    Bq = Y * Y % P
    C = Bq * Bq % P
    D = 2 * ((X + Bq) * (X + Bq) - A - C) % P
    E = 3 * A % P
    F = E * E % P
    X3 = (F - 2 * D) % P
    Y3 = (E * (D - X3) - 8 * C) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def _jac_add_affine(p, q):
    """p Jacobian + q affine (q != INF)."""
    X1, Y1, Z1 = p
    if Z1 == 0:
        return (q[0], q[1], 1)
    x2, y2 = q
    Z1Z1 = Z1 * Z1 % P
    U2 = x2 * Z1Z1 % P
    S2 = y2 * Z1 * Z1Z1 % P
    if U2 == X1:
        if S2 == Y1:
            return _jac_double(p)
        return (1, 1, 0)
    H = (U2 - X1) % P
    HH = H * H % P
    I = 4 * HH % P
    J = H * I % P
    r = 2 * (S2 - Y1) % P
    V = X1 * I % P
    X3 = (r * r - J - 2 * V) % P
    Y3 = (r * (V - X3) - 2 * Y1 * J) % P
    Z3 = ((Z1 + H) * (Z1 + H) - Z1Z1 - HH) % P
    return (X3, Y3, Z3)


def _jac_to_affine(p):
    X, Y, Z = p
    if Z == 0:
        return INF
    zi = fp_inv(Z)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def g1_mul(pt, k):
    """bn254.G1Affine.ScalarMultiplication(p, k) with k a plain integer (main.go:59,87,99,212)."""
    k = int(k)
    if pt is INF or k == 0:
        return INF
    if k < 0:
        return g1_mul(g1_neg(pt), -k)
    acc = (1, 1, 0)
    for bit in bin(k)[2:]:
        acc = _jac_double(acc)
        if bit == "1":
            acc = _jac_add_affine(acc, pt)
    return _jac_to_affine(acc)


# ---------------------------------------------------------------- (un)marshal
def g1_marshal(pt):
    """G1Affine.Marshal() (main.go:88,115,137,...): 64 bytes X||Y BE; infinity = zeros."""
    if pt is INF:
        return bytes(64)
    return pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")


def g1_unmarshal(buf):
    """G1Affine.Unmarshal on a 64-byte buffer (main.go:130,144-145,198-199,...).

    Flag bits 00: X, Y <- SetBytes (reduced mod p).  (0,0) is the point at infinity.
    Flag bits 10/11/01 mean a *compressed* encoding in gnark; Porla never produces those
    in 64-byte buffers, the oracle decodes them the gnark way for completeness.
    """
    buf = bytes(buf)
    flags = buf[0] & M_MASK
    if flags == M_UNCOMPRESSED:
        x = int.from_bytes(buf[:32], "big") % P
        y = int.from_bytes(buf[32:64], "big") % P
        if x == 0 and y == 0:
            return INF
        return (x, y)
    if flags == M_COMPRESSED_INFINITY:
        return INF
    return g1_decompress(buf[:32])


def g1_compress(pt):
    """G1Affine.Bytes(): 32 bytes, flags in the top two bits (SRS wire format)."""
    if pt is INF:
        return bytes([M_COMPRESSED_INFINITY]) + bytes(31)
    x, y = pt
    out = bytearray(x.to_bytes(32, "big"))
    out[0] |= M_COMPRESSED_LARGEST if y > (P - 1) // 2 else M_COMPRESSED_SMALLEST
    return bytes(out)


def g1_decompress(buf):
    buf = bytearray(buf[:32])
    flags = buf[0] & M_MASK
    if flags == M_COMPRESSED_INFINITY:
        return INF
    buf[0] &= 0x3F
    x = int.from_bytes(buf, "big") % P
    y = fp_sqrt((x * x * x + B) % P)
    if y is None:
        raise ValueError("not on curve")
    largest = y > (P - 1) // 2
    if largest != (flags == M_COMPRESSED_LARGEST):
        y = P - y
    return (x, y)


# ---------------------------------------------------------------- the plug-in functions
def multi_exp(scalars, points, length):
    """compute_multi_exp (main.go:118-138): sum (s_i mod r) * P_i, naive double-and-add."""
    acc = INF
    for i in range(length):
        s = fr_from_bytes(scalars[32 * i:32 * i + 32])
        pt = g1_unmarshal(points[64 * i:64 * i + 64])
        acc = g1_add(acc, g1_mul(pt, s))
    return g1_marshal(acc)


class KZG:
    """Process-global state of main.go:18-29 as an object."""

    def __init__(self):
        self.tau = None
        self.alpha = None
        self.n = 0
        self.srs_g1 = []
        self.h_mac = INF

    def init_key(self, tau_bytes, alpha_bytes):
        """main.go:31-40."""
        self.tau = fr_from_bytes(tau_bytes)
        self.tau_bi = int.from_bytes(bytes(tau_bytes), "big")
        self.alpha = fr_from_bytes(alpha_bytes)

    def init_srs(self, n, h_scalar=1):
        """main.go:42-60: SRS.G1[i] = tau^i * G (kzg.NewSRS).  h_MAC = random * G1[0]; the
        oracle takes the 'random' scalar as an argument (the reference draws it from
        crypto/rand, so complements are non-reproducible by design)."""
        self.n = n
        self.srs_g1 = []
        t = 1
        for _ in range(n):
            self.srs_g1.append(g1_mul(G1, t))
            t = t * self.tau_bi % R
        self.h_mac = g1_mul(self.srs_g1[0], h_scalar % R)

    def srs_g1_blob(self):
        """G1 part of SRS.WriteTo (main.go:48): 4-byte BE count + n compressed points."""
        return len(self.srs_g1).to_bytes(4, "big") + b"".join(g1_compress(p) for p in self.srs_g1)

    def srs_blob(self):
        """the whole SRS.WriteTo blob (main.go:48; 32 n + 132 bytes, Client.hpp:350-357): the G1 part, then G2[0] = the G2
        generator and G2[1] = tau * generator, 64 bytes compressed each (oracle/bn254_pairing_py.py)"""
        import bn254_pairing_py as pp
        return self.srs_g1_blob() + pp.srs_g2_blob(self.tau)

    def verify_proof_with_pairing(self, commitment, proof_h, point, claim):
        """kzg.Verify (main.go:177-193) as gnark states it: e(C - y G1, G2) * e(-H, tau G2 - z G2) == 1, on the Python pairing"""
        import bn254_pairing_py as pp
        c, hh = g1_unmarshal(commitment), g1_unmarshal(proof_h)
        z, y = fr_from_bytes(point), fr_from_bytes(claim)
        lhs = g1_add(c, g1_neg(g1_mul(G1, y)))
        q = pp.g2_add(pp.g2_mul(pp.G2_GEN, self.tau), pp.g2_neg(pp.g2_mul(pp.G2_GEN, z)))
        return pp.pairing_product_is_one([(lhs, pp.G2_GEN), (g1_neg(hh), q)])

    def init_srs_from_points(self, pts):
        self.n = len(pts)
        self.srs_g1 = list(pts)

    def _poly(self, data):
        return [fr_from_bytes(data[32 * i:32 * i + 32]) for i in range(self.n)]

    def compute_digest(self, data):
        """main.go:70-89: alpha * f(tau) * G1[0]; Polynomial[i] is the coefficient of X^i."""
        f = self._poly(data)
        fx = 0
        for c in reversed(f):
            fx = (fx * self.tau + c) % R
        fx = fx * self.alpha % R
        return g1_marshal(g1_mul(self.srs_g1[0], fx))

    def compute_digest_complement(self, data):
        """main.go:91-101."""
        return g1_marshal(g1_mul(self.h_mac, fr_from_bytes(data)))

    def compute_digest_from_srs(self, data):
        """main.go:103-116: kzg.Commit = MSM of the coefficients against SRS.G1."""
        f = self._poly(data)
        acc = INF
        for c, g in zip(f, self.srs_g1):
            acc = g1_add(acc, g1_mul(g, c))
        return g1_marshal(acc)

    def create_proof(self, z_u64, data):
        """main.go:153-175: (commitment, H, point, claim).  kzg.Open: y = f(z), h = (f - y)/(X - z)
        by synthetic division (degree n-2), H = Commit(h)."""
        f = self._poly(data)
        z = z_u64 % R
        commitment = self.compute_digest_from_srs(data)
        y = 0
        for c in reversed(f):
            y = (y * z + c) % R
        # synthetic division of f - y by (X - z): h[n-2] = f[n-1]; h[i-1] = f[i] + z*h[i]
        h = [0] * (self.n - 1)
        carry = 0
        for i in range(self.n - 1, 0, -1):
            carry = (f[i] + z * carry) % R
            h[i - 1] = carry
        acc = INF
        for c, g in zip(h, self.srs_g1):
            acc = g1_add(acc, g1_mul(g, c))
        return commitment, g1_marshal(acc), fr_to_bytes(z), fr_to_bytes(y)

    def verify_proof_with_tau(self, commitment, proof_h, point, claim):
        """Pairing-free restatement of kzg.Verify (main.go:177-193), valid because the oracle
        knows tau: e(C - y*G, G2) == e(H, (tau - z)*G2)  <=>  C - y*G == (tau - z)*H."""
        c = g1_unmarshal(commitment)
        hh = g1_unmarshal(proof_h)
        z = fr_from_bytes(point)
        y = fr_from_bytes(claim)
        lhs = g1_add(c, g1_neg(g1_mul(G1, y)))
        rhs = g1_mul(hh, (self.tau - z) % R)
        return lhs == rhs


def add_point(a, b):
    """main.go:195-202."""
    return g1_marshal(g1_add(g1_unmarshal(a), g1_unmarshal(b)))


def mult_point(a, s):
    """main.go:204-214."""
    return g1_marshal(g1_mul(g1_unmarshal(a), fr_from_bytes(s)))


def neg_point(a):
    """main.go:216-222."""
    return g1_marshal(g1_neg(g1_unmarshal(a)))


def set_inf_point():
    """main.go:224-230."""
    return bytes(64)


def compare_commitment(a, b):
    """main.go:140-151."""
    return g1_unmarshal(a) == g1_unmarshal(b)


# ---------------------------------------------------------------- synthetic inputs (SURVEY s8d cfg 2)
def synth_scalar(i):
    """s_i = SHA-256("porla-msm-sc" || LE32(i)) as 32 raw big-endian bytes (~81% are >= r)."""
    return hashlib.sha256(b"porla-msm-sc" + int(i).to_bytes(4, "little")).digest()


def synth_point_scalar(i):
    """k_i with P_i = k_i * G: SHA-256("porla-msm-pt" || LE32(i)) mod r."""
    return int.from_bytes(hashlib.sha256(b"porla-msm-pt" + int(i).to_bytes(4, "little")).digest(), "big") % R
