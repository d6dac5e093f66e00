"""Shared test helpers: oracle loading (CPU checker) and the synthetic input generator of SURVEY.md s8(d)."""
import ctypes
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "liboracle.so")
if ORACLE_DIR not in sys.path:
    sys.path.insert(0, ORACLE_DIR)

_oracle = None


def oracle():
    """The C restatement (oracle/*_ref.c), built on demand with gcc."""
    global _oracle
    if _oracle is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
        if not os.path.exists(ORACLE_SO) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in srcs):
            subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
        _oracle = ctypes.CDLL(ORACLE_SO)
    return _oracle


def ncpu():
    """threads of the CPU restatement: every hardware thread this process may use (hardware_concurrency()),
    i.e. the CPU count cut down to the scheduler affinity and to a cgroup CPU quota where one is set (a GPU box hands a
    one-GPU job 16 of its 256 hardware threads: 256 runnable threads on them only thrash)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), None])):
        try:
            quota, period = parse(open(path).read())
            if period is None:
                period = open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    return max(1, n)


def synth_scalars(n, start=0):
    """s_i = SHA-256("porla-msm-sc" || LE32(i)), raw 32 big-endian bytes (~81 % are >= r)."""
    return b"".join(hashlib.sha256(b"porla-msm-sc" + (start + i).to_bytes(4, "little")).digest() for i in range(n))


def synth_point_scalars(n, start=0):
    return b"".join(hashlib.sha256(b"porla-msm-pt" + (start + i).to_bytes(4, "little")).digest() for i in range(n))


def synth_points(n, start=0):
    """P_i = (SHA-256("porla-msm-pt" || LE32(i)) mod r) * G as 64-byte X||Y big-endian (oracle fixed-base)."""
    out = ctypes.create_string_buffer(64 * n)
    oracle().oracle_bn254_fixed_base(synth_point_scalars(n, start), ctypes.c_size_t(n), out, ncpu())
    return out.raw


def synth_inputs(n, start=0):
    return synth_scalars(n, start), synth_points(n, start)


def oracle_msm(scalars, points, n, naive=False, threads=None):
    out = ctypes.create_string_buffer(64)
    oracle().oracle_bn254_multi_exp(bytes(scalars), bytes(points), ctypes.c_size_t(n), out,
                                    threads or ncpu(), 1 if naive else 0)
    return out.raw


def cached_inputs(n):
    """synth_inputs(n) cached under /tmp (generation of 2^20 points costs tens of CPU-seconds)."""
    path = os.path.join(os.environ.get("PORLA_CACHE", "/tmp"), "porla_bn254_inputs_%d.bin" % n)
    if os.path.exists(path) and os.path.getsize(path) == 96 * n:
        raw = open(path, "rb").read()
        return raw[:32 * n], raw[32 * n:]
    sc, pt = synth_inputs(n)
    try:
        with open(path + ".tmp", "wb") as f:
            f.write(sc)
            f.write(pt)
        os.replace(path + ".tmp", path)
    except OSError:
        pass
    return sc, pt


# ---------------------------------------------------------------- secp256k1 (IPA path) inputs
SECP_N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


def secp_bench_scalars(n, start=0):
    """generate_scalar of the reference's bench: SHA-256("ecmult" || LE32(i)) (bench_ecmult.c:233-247)"""
    return b"".join(hashlib.sha256(b"ecmult" + (start + i).to_bytes(4, "little")).digest() for i in range(n))


def secp_bench_points(n):
    """P_i = 2^i * G, normalised (bench_ecmult.c:328-337), 64-byte x||y big-endian"""
    out = ctypes.create_string_buffer(64 * n)
    oracle().oracle_secp256k1_doubling_chain(ctypes.c_size_t(n), out)
    return out.raw


def secp_bench_expected(scalars, n):
    """(sum s_i * 2^i) * G like the bench teardown (bench_ecmult.c:258-270) -- independent of any bucket method"""
    acc = 0
    for i in range(n):
        acc = (acc + (int.from_bytes(scalars[32 * i:32 * i + 32], "big") % SECP_N) * pow(2, i, SECP_N)) % SECP_N
    out = ctypes.create_string_buffer(64)
    oracle().oracle_secp256k1_mul_g_batch(acc.to_bytes(32, "big"), ctypes.c_size_t(1), out, 1)
    return out.raw


def oracle_secp_msm(scalars, points, n, naive=False, threads=None):
    out = ctypes.create_string_buffer(64)
    oracle().oracle_secp256k1_multi(bytes(scalars), bytes(points), ctypes.c_size_t(n), out, threads or ncpu(), 1 if naive else 0)
    return out.raw


def oracle_commit_batch(curve, rows, n_rows, n_coeffs, base, row_stride=None, naive=False, threads=None):
    """row-by-row commitments against a fixed base (compute_digest_from_srs / compute_commitment restatement)"""
    out = ctypes.create_string_buffer(64 * max(n_rows, 1))
    fn = getattr(oracle(), "oracle_%s_commit_batch" % curve)
    fn(bytes(rows), ctypes.c_size_t(n_rows), ctypes.c_size_t(n_coeffs), ctypes.c_size_t(row_stride or 32 * n_coeffs),
       bytes(base), out, threads or ncpu(), 1 if naive else 0)
    return out.raw[:64 * n_rows]
