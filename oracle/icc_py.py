"""ORACLE (test infrastructure only) -- Python big-int restatement of Porla's ICC encode arithmetic.

Follows the reference loop for loop (all arithmetic there is NTL ZZ / ZZ_p, which is absent from this image, so
the reference path cannot be run here; it also holds no tests for it -> "parity unpinned"):
  constants                porla/Utils/utils.h:27-43   (GENERATOR, PRIME_MODULUS = 207*2^248+1, LCM = p_icc * q)
  w                        porla/Server/Server.hpp:214-216   w = GENERATOR^((p-1)/(2N)) mod p
  height                   Server.hpp:219                   ceil(log2 N) + 1
  CRebuild_Cached X part   Server.hpp:1548-1687             stages s = 1..height-1, m = 2^s, m2 = m/2,
                                                            v = w^(N/m2); for j < m2: vi = v^j; for k = j; k < N; k += m:
                                                            t = vi*X[k+m2]; u = X[k]; X[k] = (u+t) % LCM; X[k+m2] = (u-t) % LCM
  init scaling (Y part)    Server.hpp:1494,1512-1522        wt = w^reverse_bits(write_step % N, height-1); Y = X*wt (unreduced)
  mix                      Server.hpp:1269-1278             one stage between two length-`length` blocks, v = w^(N/length)
  align_MAC scalar part    Server.hpp:531-541 (KZG) / 495-504 (IPA)   mod = A % p; c = (mod - A) % q; A = mod
  reverse_bits             utils.h:81-91
Quirks kept on purpose: w has order N (not 2N); natural-order input, no bit-reversal permutation;
butterflies reduce mod LCM so one pass carries the value mod p_icc and mod q; Y = X * wt is NOT reduced before stage 1.
Self-check (tests): result mod p_icc equals an independent O(N^2) evaluation of the same linear network.
"""
P_ICC = 207 * 2**248 + 1
GENERATOR = 37724658858582113439798596500054279666200959181261379108294206582568298678
Q_BN254 = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q_SECP256K1 = 115792089237316195423570985008687907852837564279074904382605163141518161494337
LCM = {"bn254": P_ICC * Q_BN254, "secp256k1": P_ICC * Q_SECP256K1}
Q = {"bn254": Q_BN254, "secp256k1": Q_SECP256K1}
assert LCM["bn254"] == int("2049369031155707573937272810025244064710333118140408897690954651424664974620215782673575413484558574566298823256897068805013612518402283464943595715297281")
assert LCM["secp256k1"] == int("10841469693352021873483684275893008392101031472050201500515861578010683886271238884283113399568804205471204971859923723932950084770981108620251449466962241")


def reverse_bits(x, n):
    r = 0
    for _ in range(n):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def height_of(n):
    return (n - 1).bit_length() + 1  # ceil(log2 n) + 1


def root_w(n):
    return pow(GENERATOR, (P_ICC - 1) // (2 * n), P_ICC)


def crebuild(rows, curve="bn254", write_step=0, trace=None):
    """rows: N lists of ncols ints (< 2^256).  Returns (X, Y) as in CRebuild_Cached, values in [0, LCM)."""
    n = len(rows)
    lcm = LCM[curve]
    h = height_of(n)
    w = root_w(n)
    wt = pow(w, reverse_bits(write_step % n, h - 1), P_ICC)
    X = [list(r) for r in rows]
    Y = [[v * wt for v in r] for r in rows]
    for part in (X, Y):
        for s in range(1, h):
            m, m2 = 1 << s, 1 << (s - 1)
            v = pow(w, n // m2, P_ICC)
            vi = 1
            for j in range(m2):
                for k in range(j, n, m):
                    a, b = part[k], part[k + m2]
                    for c in range(len(a)):
                        t = vi * b[c]
                        u = a[c]
                        a[c] = (u + t) % lcm
                        b[c] = (u - t) % lcm
                vi = vi * v % P_ICC
            if trace is not None and part is X:
                trace.append([list(r) for r in X])
    return X, Y


def mix(a0, a1, n_total, curve="bn254"):
    """Server::mix data part: A[i] = A0[i] + v^i*A1[i], A[i+len] = A0[i] - v^i*A1[i] (mod LCM), v = w^(N/len)."""
    length = len(a0)
    lcm = LCM[curve]
    v = pow(root_w(n_total), n_total // length, P_ICC)
    out = [None] * (2 * length)
    vi = 1
    for i in range(length):
        lo, hi = [], []
        for c in range(len(a0[i])):
            val = vi * a1[i][c]
            lo.append((a0[i][c] + val) % lcm)
            hi.append((a0[i][c] - val) % lcm)
        out[i], out[i + length] = lo, hi
        vi = vi * v % P_ICC
    return out


def align(row, curve="bn254"):
    """align_MAC scalar part: returns (row mod p_icc, alignment scalars c_i = (mod - A) % q)."""
    q = Q[curve]
    mods = [a % P_ICC for a in row]
    cs = [(m - a) % q for m, a in zip(mods, row)]
    return mods, cs


def linear_network_matrix(n):
    """Independent O(N^2) description of the X-part network mod p_icc: out = M * in, by pushing unit vectors
    through a *recursive* (decimation) formulation rather than the iterative loops above."""
    w = root_w(n)

    def transform(vec):
        size = len(vec)
        if size == 1:
            return list(vec)
        # last stage (m = size) combines the transforms of the lower and upper halves computed on m2 = size/2
        half = size // 2
        lo = transform(vec[:half])
        hi = transform(vec[half:])
        v = pow(w, n // half, P_ICC)
        out = [0] * size
        for j in range(half):
            t = pow(v, j, P_ICC) * hi[j] % P_ICC
            out[j] = (lo[j] + t) % P_ICC
            out[j + half] = (lo[j] - t) % P_ICC
        return out

    return transform


# ---------------------------------------------------------------- MAC halves of CRebuild_Cached ("FFT in the exponent")
# Server.hpp:1523-1536 (X = MAC_U, Y = wt * MAC_U) and :1590-1609 / :1658-1676 (tm = vi*MAC[k+m2]; MAC[k] = um + tm;
# MAC[k+m2] = um - tm), vi = v^j mod p_icc handed to the group as an integer and reduced mod the group order there.
CURVE_P = {"bn254": 21888242871839275222246405745257275088696311157297823662689037894645226208583,
           "secp256k1": 2**256 - 2**32 - 977}


def ec_add(curve, a, b):
    """affine addition on y^2 = x^3 + b (a = 0); None = infinity"""
    p = CURVE_P[curve]
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % p == 0:
            return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, p) % p
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, p) % p
    x = (lam * lam - a[0] - b[0]) % p
    return (x, (lam * (a[0] - x) - a[1]) % p)


def ec_neg(curve, a):
    return None if a is None else (a[0], (-a[1]) % CURVE_P[curve])


def ec_mul(curve, a, k):
    k %= Q[curve]
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = ec_add(curve, acc, acc)
        if bit == "1":
            acc = ec_add(curve, acc, a)
    return acc


def mac_crebuild(macs, curve="bn254", write_step=0):
    """macs: N affine points (x, y) or None.  Returns (X, Y) MAC arrays after the butterfly network."""
    n = len(macs)
    h = height_of(n)
    w = root_w(n)
    wt = pow(w, reverse_bits(write_step % n, h - 1), P_ICC)
    X = list(macs)
    Y = [ec_mul(curve, m, wt) for m in macs]
    for part in (X, Y):
        for s in range(1, h):
            m, m2 = 1 << s, 1 << (s - 1)
            v = pow(w, n // m2, P_ICC)
            vi = 1
            for j in range(m2):
                for k in range(j, n, m):
                    tm = ec_mul(curve, part[k + m2], vi)
                    um = part[k]
                    part[k] = ec_add(curve, um, tm)
                    part[k + m2] = ec_add(curve, um, ec_neg(curve, tm))
                vi = vi * v % P_ICC
    return X, Y


def audit_combine(rows, coeffs, curve="bn254"):
    """Server::audit row combine (Server.hpp:790-828): B_j = sum_i coeff_i * rows[i][j] (exact), then the scalar part of
    align_MAC on B (Server.hpp:531-541).  Returns (B, B mod p_icc, alignment scalars)."""
    ncols = len(rows[0]) if rows else 0
    B = [0] * ncols
    for r, c in zip(rows, coeffs):
        for j in range(ncols):
            B[j] += c * r[j]
    mods, cs = align(B, curve)
    return B, mods, cs


def mac_mix(a0, a1, n_total, curve="bn254"):
    """Server::mix MAC part (Server.hpp:1281-1318): A[i] = A0[i] + v^i*A1[i], A[i+len] = A0[i] - v^i*A1[i], v = w^(N/len)."""
    length = len(a0)
    v = pow(root_w(n_total), n_total // length, P_ICC)
    out = [None] * (2 * length)
    vi = 1
    for i in range(length):
        tm = ec_mul(curve, a1[i], vi)
        out[i] = ec_add(curve, a0[i], tm)
        out[i + length] = ec_add(curve, a0[i], ec_neg(curve, tm))
        vi = vi * v % P_ICC
    return out


def hadd(data, mac, n_total, write_step, curve="bn254"):
    """Server::HAdd up to the level bookkeeping (Server.hpp:1388-1428): wt = w^reverse_bits(write_step % N, height-1);
    data_B2[i] = data[i] * wt (integer product, :1396-1398), MAC_B2 = wt * MAC (:1400-1417), then align_MAC(data_B2, .) (:1428):
    data_B2[i] <- data_B2[i] % p_icc and the scalars c_i whose commitment is MAC_align_B2.
    Returns (data_B2 aligned, c, MAC_B2, wt).  Client::HAdd (Client.hpp:996-1014) is the MAC_B2 part alone."""
    wt = pow(root_w(n_total), reverse_bits(write_step % n_total, height_of(n_total) - 1), P_ICC)
    data_b2 = [d * wt for d in data]
    mods, cs = align(data_b2, curve)
    return mods, cs, ec_mul(curve, mac, wt), wt


def hrebuild(levels, level, n_total, curve="bn254", mac=False):
    """Server::HRebuildX / HRebuildY (Server.hpp:1329-1386) and Client::HRebuildX / Y (Client.hpp:978-994) on one family of levels:
    levels[i] = list of 2 * 2^i rows (first half resident, second half incoming); for i < level the halves of level i are mixed
    (Server::mix :1269-1318 / Client::mix Client.hpp:921-976) into the incoming half of level i + 1; then level `level`'s incoming
    half is copied over its resident half.  In place; mac = True: rows are affine points (MAC commitments / alignments / complements)."""
    for i in range(level):
        ln = 1 << i
        out = (mac_mix if mac else mix)(levels[i][:ln], levels[i][ln:2 * ln], n_total, curve)
        levels[i + 1][2 * ln:4 * ln] = out
    top = 1 << level
    levels[level][:top] = levels[level][top:2 * top]
    return levels
