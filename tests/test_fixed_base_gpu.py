"""Parity of the batched fixed-base commitment kernels (porla_amd/csrc/fixed_base.hip.h) against the oracle, through the
C ABI (GPU box only).

What the reference does row by row: compute_digest_from_srs (porla/main.go:103-116; callers
porla/Server/Server.hpp:550-560, 1077-1078, 2061-2062) and, for IPA, compute_commitment
(porla/Client/Client.hpp:374-406 -> secp256k1_ecmult_multi_var over the fixed generators).  Bar: bit-exact 64-byte
points for every row.
"""
import hashlib

import pytest

from tests import common

pytestmark = pytest.mark.gpu

R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
TAU = bytes.fromhex("ffeeddccbbaa99887766554433221100")     # TAU_KEY, config.hpp:39
ALPHA = bytes.fromhex("00112233445566778899aabbccddeeff")   # SECRET_KEY, config.hpp:38


@pytest.fixture(scope="module")
def mx():
    from porla_amd import multiexp
    return multiexp


def rows_bytes(n_rows, n_coeffs, seed=b"porla-row"):
    return b"".join(hashlib.sha256(seed + i.to_bytes(4, "little")).digest() for i in range(n_rows * n_coeffs))


@pytest.fixture(scope="module")
def srs128(mx):
    """the Porla SRS (n = NUM_CHUNKS = 128) as raw 64-byte points, from the oracle's KZG state"""
    import ctypes
    o = common.oracle()
    o.oracle_kzg_init_key(TAU, ctypes.c_size_t(len(TAU)), ALPHA, ctypes.c_size_t(len(ALPHA)))
    o.oracle_kzg_init_srs(ctypes.c_size_t(128), (1).to_bytes(32, "big"))
    raw = ctypes.create_string_buffer(64 * 128)
    o.oracle_kzg_srs_g1_raw(raw)
    return raw.raw


@pytest.mark.parametrize("c", [4, 8, 11, 13])
@pytest.mark.parametrize("n_rows", [1, 3, 70])
def test_bn254_commit_matches_oracle_every_window(mx, srs128, c, n_rows):
    """every table window width x the slice/lane configurations (1 row -> 128 slices, 70 rows -> 128 slices, ...)"""
    fb = mx.FixedBase("bn254", srs128, 128, window_bits=c)
    assert fb.info()["window_bits"] == c
    rows = rows_bytes(n_rows, 128, b"w%d" % c)
    got = fb.commit_host(rows, n_rows, 128)
    assert got == common.oracle_commit_batch("bn254", rows, n_rows, 128, srs128)
    fb.close()


def test_bn254_commit_edge_scalars_and_short_rows(mx, srs128):
    """zero / r-1 / r / 2^256-1 coefficients (SetBytes reduction, main.go:110), all-zero row -> 64 zero bytes,
    n_coeffs < n_points (the 127-coefficient quotient of create_proof, main.go:164-170), padded row stride"""
    fb = mx.FixedBase("bn254", srs128, 128, window_bits=9)
    vals = [0, 1, 2, R - 1, R, R + 1, (1 << 256) - 1, 1 << 128, (1 << 255) + 12345, 5 * R, 5 * R + 7, 1 << 253]
    row0 = b"".join(vals[i % len(vals)].to_bytes(32, "big") for i in range(128))
    row1 = bytes(32 * 128)
    row2 = (R - 1).to_bytes(32, "big") * 128
    rows = row0 + row1 + row2
    got = fb.commit_host(rows, 3, 128)
    assert got == common.oracle_commit_batch("bn254", rows, 3, 128, srs128, naive=True)
    assert got[64:128] == bytes(64)
    got127 = fb.commit_host(rows, 3, 127, row_stride=4096)
    assert got127 == common.oracle_commit_batch("bn254", rows, 3, 127, srs128, row_stride=4096)
    assert got127 != got
    fb.close()


def test_bn254_base_with_infinity_and_repeats(mx, srs128):
    """a base holding the point at infinity and a repeated point (tau = 0 would give G1[i>0] = O)"""
    base = srs128[:64] + bytes(64) + srs128[:64] + srs128[64:128]
    fb = mx.FixedBase("bn254", base, 4, window_bits=6)
    rows = rows_bytes(5, 4, b"inf")
    assert fb.commit_host(rows, 5, 4) == common.oracle_commit_batch("bn254", rows, 5, 4, base, naive=True)
    fb.close()


def test_kzg_batch_equals_per_row_calls(mx, srs128):
    """porla_kzg_commit_batch_host over R rows == R calls of the reference symbol compute_digest_from_srs"""
    mx.init_key(TAU, ALPHA)
    blob = mx.init_SRS(128)
    mx.init_SRS_from_data(128, blob)
    n_rows = 9
    rows = rows_bytes(n_rows, 128, b"kzg")
    got = mx.kzg_commit_batch_host(rows, n_rows)
    want = common.oracle_commit_batch("bn254", rows, n_rows, 128, srs128)
    assert got == want
    for r in (0, 4, 8):
        assert mx.compute_digest_from_srs(rows[4096 * r:4096 * r + 4096]) == want[64 * r:64 * r + 64]


def test_bn254_device_rows_large_batch_default_window(mx, srs128):
    """device-resident rows, default (largest) window, 4096 rows; spot rows against the naive oracle too"""
    import torch
    fb = mx.FixedBase("bn254", srs128, 128)
    info = fb.info()
    assert info["window_bits"] >= 8
    n_rows = 4096
    rows = rows_bytes(n_rows, 128, b"big")
    d_rows = torch.frombuffer(bytearray(rows), dtype=torch.uint8).cuda()
    d_out = torch.empty(64 * n_rows, dtype=torch.uint8, device="cuda")
    fb.commit_device(d_rows.data_ptr(), n_rows, 128, d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = bytes(d_out.cpu().numpy())
    want = common.oracle_commit_batch("bn254", rows, n_rows, 128, srs128)
    assert got == want
    assert got[:128] == common.oracle_commit_batch("bn254", rows, 2, 128, srs128, naive=True)
    fb.close()


def test_linearity_at_full_row_count(mx, srs128):
    """size-independent property at a batch the oracle cannot finish in seconds: Commit(a) + Commit(b) == Commit(a + b
    mod r) row-wise, checked by a second commitment and the engine's own add_point on sampled rows"""
    import torch
    fb = mx.FixedBase("bn254", srs128, 128)
    n_rows = 1 << 15
    a = rows_bytes(n_rows, 128, b"lin-a")
    b = rows_bytes(n_rows, 128, b"lin-b")
    d_a = torch.frombuffer(bytearray(a), dtype=torch.uint8).cuda()
    d_b = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
    d_oa = torch.empty(64 * n_rows, dtype=torch.uint8, device="cuda")
    d_ob = torch.empty(64 * n_rows, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    fb.commit_device(d_a.data_ptr(), n_rows, 128, d_oa.data_ptr(), s)
    fb.commit_device(d_b.data_ptr(), n_rows, 128, d_ob.data_ptr(), s)
    torch.cuda.synchronize()
    oa, ob = bytes(d_oa.cpu().numpy()), bytes(d_ob.cpu().numpy())
    sample = [0, 1, 777, n_rows // 2, n_rows - 1]
    srows = b""
    for r in sample:
        for i in range(128):
            x = int.from_bytes(a[4096 * r + 32 * i:4096 * r + 32 * i + 32], "big")
            y = int.from_bytes(b[4096 * r + 32 * i:4096 * r + 32 * i + 32], "big")
            srows += ((x + y) % R).to_bytes(32, "big")
    sums = fb.commit_host(srows, len(sample), 128)
    for k, r in enumerate(sample):
        assert mx.bn254_add(oa[64 * r:64 * r + 64], ob[64 * r:64 * r + 64]) == sums[64 * k:64 * k + 64]
    # and the sampled rows themselves against the oracle
    for r in sample:
        assert oa[64 * r:64 * r + 64] == common.oracle_commit_batch("bn254", a[4096 * r:4096 * r + 4096], 1, 128, srs128)
    fb.close()


@pytest.mark.parametrize("c", [5, 12])
def test_secp256k1_pedersen_commit(mx, c):
    """IPA twin: 128 fixed generators (here 2^i * G as in bench_ecmult.c:328-337), rows of 256-bit chunks incl. values >= n"""
    base = common.secp_bench_points(128)
    fb = mx.FixedBase("secp256k1", base, 128, window_bits=c)
    n_rows = 6
    rows = bytearray(rows_bytes(n_rows, 128, b"secp"))
    rows[0:32] = (common.SECP_N + 5).to_bytes(32, "big")
    rows[32:64] = ((1 << 256) - 1).to_bytes(32, "big")
    rows[64:96] = bytes(32)
    rows[96:128] = (common.SECP_N - 1).to_bytes(32, "big")
    got = fb.commit_host(bytes(rows), n_rows, 128)
    assert got == common.oracle_commit_batch("secp256k1", bytes(rows), n_rows, 128, base)
    assert got[:64] == common.oracle_commit_batch("secp256k1", bytes(rows), 1, 128, base, naive=True)
    fb.close()


def test_client_side_digest_and_complement_batches(mx):
    """porla_kzg_digest_batch / porla_kzg_complement_batch == the reference symbols compute_digest /
    compute_digest_complement called row by row (porla/Client/Client.hpp:408-455 -> main.go:70-101), and the digest equals
    the oracle's; Client::initialize builds every block MAC as digest + complement (Client.hpp:216-226)"""
    import ctypes
    import torch
    mx.init_key(TAU, ALPHA)
    mx.init_SRS(128)
    n_rows = 300
    rows = bytearray(rows_bytes(n_rows, 128, b"digest"))
    rows[0:32] = (R + 5).to_bytes(32, "big")          # a coefficient >= r (SetBytes reduces)
    rows[4096:4096 + 4096] = bytes(4096)               # an all-zero block -> infinity
    rows = bytes(rows)
    d_rows = torch.frombuffer(bytearray(rows), dtype=torch.uint8).cuda()
    d_out = torch.empty(64 * n_rows, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    mx.kzg_digest_batch_device(d_rows.data_ptr(), n_rows, d_out.data_ptr(), s)
    torch.cuda.synchronize()
    got = bytes(d_out.cpu().numpy())
    o = common.oracle()
    o.oracle_kzg_init_key(TAU, ctypes.c_size_t(16), ALPHA, ctypes.c_size_t(16))
    o.oracle_kzg_init_srs(ctypes.c_size_t(128), (1).to_bytes(32, "big"))
    for r in (0, 1, 2, 150, 299):
        want = ctypes.create_string_buffer(64)
        o.oracle_kzg_compute_digest(rows[4096 * r:4096 * r + 4096], want)
        assert got[64 * r:64 * r + 64] == want.raw
        assert got[64 * r:64 * r + 64] == mx.compute_digest(rows[4096 * r:4096 * r + 4096])
    assert got[64:128] == bytes(64)
    # complements: 16-byte PRF outputs left-padded to 32 bytes
    prf = [hashlib.sha256(b"prf" + i.to_bytes(4, "little")).digest()[:16] for i in range(n_rows)]
    scal = b"".join(bytes(16) + p for p in prf)
    d_sc = torch.frombuffer(bytearray(scal), dtype=torch.uint8).cuda()
    mx.kzg_complement_batch_device(d_sc.data_ptr(), n_rows, d_out.data_ptr(), s)
    torch.cuda.synchronize()
    comp = bytes(d_out.cpu().numpy())
    for r in (0, 7, 299):
        assert comp[64 * r:64 * r + 64] == mx.compute_digest_complement(prf[r])
    # the MAC batch: digest + complement joined as the client does with add_point (Client.hpp:229-236), every row; scalar 3 is
    # zero (MAC = digest), row 1 is the zero block (MAC = complement), and the eval switch off gives the same bytes
    scal = bytearray(scal)
    scal[32 * 3:32 * 4] = bytes(32)
    scal[32 * 5:32 * 6] = (R + 9).to_bytes(32, "big")          # a scalar >= r
    d_sc = torch.frombuffer(scal, dtype=torch.uint8).cuda()
    d_mac = torch.empty(64 * n_rows, dtype=torch.uint8, device="cuda")
    mx.kzg_complement_batch_device(d_sc.data_ptr(), n_rows, d_out.data_ptr(), s)
    mx.kzg_mac_batch_device(d_rows.data_ptr(), d_sc.data_ptr(), n_rows, d_mac.data_ptr(), s)
    torch.cuda.synchronize()
    comp = bytes(d_out.cpu().numpy())
    macs = bytes(d_mac.cpu().numpy())
    for r in range(n_rows):
        assert macs[64 * r:64 * r + 64] == mx.bn254_add(got[64 * r:64 * r + 64], comp[64 * r:64 * r + 64]), r
    assert macs[64 * 3:64 * 4] == got[64 * 3:64 * 4] and macs[64:128] == comp[64:128]
    assert macs[64 * 5:64 * 6] == mx.bn254_add(got[64 * 5:64 * 6], mx.compute_digest_complement((9).to_bytes(16, "big")))
    mx.kzg_mac_batch_device(d_rows.data_ptr(), d_sc.data_ptr(), 0, d_mac.data_ptr(), s)       # empty batch: nothing to do
    # the same three batches on host buffers
    assert mx.kzg_digest_batch_host(rows, n_rows) == got
    assert mx.kzg_complement_batch_host(bytes(scal), n_rows) == comp
    assert mx.kzg_mac_batch_host(rows, bytes(scal), n_rows) == macs
    assert mx.kzg_mac_batch_host(rows, bytes(scal), 1) == macs[:64] and mx.kzg_mac_batch_host(b"", b"", 0) == b""
    # more than one staging chunk (16 384 blocks) against the device entry
    n_big = 16384 + 77
    big = hashlib.shake_256(b"bigrows").digest(4096 * 64) * (n_big // 64 + 1)
    big = big[:4096 * n_big]
    big_sc = hashlib.shake_256(b"bigsc").digest(32 * n_big)
    d_big = torch.frombuffer(bytearray(big), dtype=torch.uint8).cuda()
    d_bsc = torch.frombuffer(bytearray(big_sc), dtype=torch.uint8).cuda()
    d_bout = torch.empty(64 * n_big, dtype=torch.uint8, device="cuda")
    mx.kzg_mac_batch_device(d_big.data_ptr(), d_bsc.data_ptr(), n_big, d_bout.data_ptr(), s)
    torch.cuda.synchronize()
    assert mx.kzg_mac_batch_host(big, big_sc, n_big) == bytes(d_bout.cpu().numpy())


def test_host_batch_larger_than_the_staging_chunks(mx, srs128):
    """commit_host stages host rows in 64 MiB chunks on two streams (16 384 rows of 128 coefficients each): three chunks with a
    ragged last one give the bytes of the device entry on the same rows; first, last and chunk-boundary rows against the oracle"""
    import torch
    mx.init_key(TAU, ALPHA)
    blob = mx.init_SRS(128)
    mx.init_SRS_from_data(128, blob)
    n_rows = 2 * 16384 + 1234
    rows = hashlib.shake_256(b"chunks").digest(4096 * 97) * (n_rows // 97 + 1)
    rows = bytearray(rows[:4096 * n_rows])
    for r in (0, 16383, 16384, 32767, 32768, n_rows - 1):          # make the probed rows distinct
        rows[4096 * r:4096 * r + 32] = hashlib.sha256(b"row%d" % r).digest()
    rows = bytes(rows)
    got = mx.kzg_commit_batch_host(rows, n_rows)
    d_rows = torch.frombuffer(bytearray(rows), dtype=torch.uint8).cuda()
    d_out = torch.empty(64 * n_rows, dtype=torch.uint8, device="cuda")
    mx.kzg_commit_batch_device(d_rows.data_ptr(), n_rows, d_out.data_ptr(), 0)
    torch.cuda.synchronize()
    assert got == bytes(d_out.cpu().numpy())
    for r in (0, 16383, 16384, 32767, 32768, n_rows - 1):
        assert got[64 * r:64 * r + 64] == common.oracle_commit_batch("bn254", rows[4096 * r:4096 * r + 4096], 1, 128, srs128), r
    assert mx.kzg_commit_batch_host(rows, n_rows) == got           # and again through the warm staging buffers


@pytest.mark.parametrize("n_coeffs", [1, 7, 8, 9, 100, 1024, 1025])
def test_digest_batch_row_lengths(mx, n_coeffs):
    """the digest batch's evaluation kernels at row lengths other than the reference's 128 (init_SRS takes any size): lengths that
    are not a multiple of the eight lanes of a row, the longest row the dot-product kernel stages in LDS (1024) and one beyond it
    (Horner form); coefficients >= r and all-ones among them; a key change between two batches must not reuse the powers of tau"""
    import torch
    n_rows = 19
    rows = bytearray(hashlib.shake_256(b"rowlen%d" % n_coeffs).digest(32 * n_coeffs * n_rows))
    rows[0:32] = b"\xff" * 32
    rows[32 * (n_coeffs - 1):32 * n_coeffs] = (R + 1).to_bytes(32, "big")
    rows = bytes(rows)
    s = torch.cuda.current_stream().cuda_stream
    d_rows = torch.frombuffer(bytearray(rows), dtype=torch.uint8).cuda()
    d_out = torch.empty(64 * n_rows, dtype=torch.uint8, device="cuda")
    for key in ((TAU, ALPHA), (ALPHA, TAU)):
        mx.init_key(*key)
        mx.init_SRS(n_coeffs)
        mx.kzg_digest_batch_device(d_rows.data_ptr(), n_rows, d_out.data_ptr(), s)
        torch.cuda.synchronize()
        got = bytes(d_out.cpu().numpy())
        for r in range(n_rows):
            assert got[64 * r:64 * r + 64] == mx.compute_digest(rows[32 * n_coeffs * r:32 * n_coeffs * (r + 1)]), (n_coeffs, r)
    mx.init_key(TAU, ALPHA)
    mx.init_SRS(128)


def test_eight_threads_call_the_plugin_concurrently(mx, srs128):
    """the server calls compute_digest_from_srs / add_point / mult_point / neg_point from 8 pool threads at once
    (porla/Server/Server.hpp:1054-1078, 1530-1535, 1600-1608); ctypes releases the GIL, so these really overlap"""
    import threading
    mx.init_key(TAU, ALPHA)
    blob = mx.init_SRS(128)
    mx.init_SRS_from_data(128, blob)
    rows = rows_bytes(8, 128, b"thr")
    want = common.oracle_commit_batch("bn254", rows, 8, 128, srs128)
    errors = []

    def worker(t):
        try:
            row = rows[4096 * t:4096 * t + 4096]
            for _ in range(25):
                c = mx.compute_digest_from_srs(row)
                if c != want[64 * t:64 * t + 64]:
                    errors.append("commit %d" % t)
                k = (t + 2).to_bytes(32, "big")
                p = mx.bn254_mult(c, k)
                acc = mx.bn254_set_infinity()
                for _ in range(t + 2):
                    acc = mx.bn254_add(acc, c)
                if acc != p or mx.bn254_add(p, mx.bn254_neg(p)) != bytes(64):
                    errors.append("point ops %d" % t)
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]


def test_release_device_memory_and_rebuild(mx, srs128):
    """porla_kzg_release_device_memory frees the SRS table; the next commitment rebuilds it and gives the same bytes"""
    import torch
    from porla_amd import lib
    rows = common.synth_scalars(128 * 3, 555)
    a = mx.kzg_commit_batch_host(rows, 3)
    free0 = torch.cuda.mem_get_info()[0]
    assert lib.porla_kzg_release_device_memory() == 0
    assert torch.cuda.mem_get_info()[0] > free0 + (1 << 30)          # the table alone is tens of GB
    assert mx.kzg_commit_batch_host(rows, 3) == a
    # the client-side batches keep tables and the powers of tau on the device too: same bytes before and after a release
    mx.init_key(TAU, ALPHA)
    blob = mx.init_SRS(128)
    d_rows = torch.frombuffer(bytearray(rows), dtype=torch.uint8).cuda()
    d_sc = torch.frombuffer(bytearray(b"".join(bytes(16) + hashlib.sha256(b"rel%d" % i).digest()[:16] for i in range(3))), dtype=torch.uint8).cuda()
    outs = []
    for _ in range(2):
        d_out = torch.zeros(64 * 3 * 2, dtype=torch.uint8, device="cuda")
        mx.kzg_digest_batch_device(d_rows.data_ptr(), 3, d_out.data_ptr(), 0)
        mx.kzg_mac_batch_device(d_rows.data_ptr(), d_sc.data_ptr(), 3, d_out.data_ptr() + 192, 0)
        torch.cuda.synchronize()
        outs.append(bytes(d_out.cpu().numpy()))
        assert lib.porla_kzg_release_device_memory() == 0
    assert outs[0] == outs[1] and outs[0][:64] == mx.compute_digest(rows[:4096])
    mx.init_SRS_from_data(128, blob)


@pytest.mark.parametrize("n_rows", [1, 2, 31, 32, 33, 63, 64, 65])
def test_few_rows_single_launch_path_and_its_boundary(mx, srs128, n_rows):
    """<= 64 host rows take the single-launch kernel (k_fb_commit_small: digits, gather, LDS fold, slices folded by the last
    block, sums polled from pinned memory); 65 rows the batch kernels -- same bytes either way, every table window"""
    rows = rows_bytes(n_rows, 128, b"few%d" % n_rows)
    want = common.oracle_commit_batch("bn254", rows, n_rows, 128, srs128)
    for c in (5, 12, 16):
        fb = mx.FixedBase("bn254", srs128, 128, window_bits=c)
        assert fb.commit_host(rows, n_rows, 128) == want
        # short rows (the 127-coefficient quotient of create_proof) and a padded stride
        short = b"".join(rows[4096 * r:4096 * r + 32 * 127] + bytes(32) for r in range(n_rows))
        assert fb.commit_host(short, n_rows, 127, row_stride=4096) == common.oracle_commit_batch("bn254", short, n_rows, 127, srs128, row_stride=4096)
        fb.close()
    # edge coefficients through the small path: zero row, r - 1, r, 2^256 - 1
    vals = [0, 1, R - 1, R, R + 1, (1 << 256) - 1, 5 * R + 7, 1 << 253]
    row = b"".join(vals[i % len(vals)].to_bytes(32, "big") for i in range(128))
    edge = bytes(4096) + row
    fb = mx.FixedBase("bn254", srs128, 128, window_bits=13)
    got = fb.commit_host(edge, 2, 128)
    assert got[:64] == bytes(64) and got == common.oracle_commit_batch("bn254", edge, 2, 128, srs128)
    fb.close()


@pytest.mark.parametrize("n_rows", [256, 257, 1025])
def test_host_and_device_normalisation_meet_at_256_rows(mx, srs128, n_rows):
    """batches of <= 256 rows are normalised on the host (engine.hpp:HOST_FINISH_MAX_ROWS: one batched inversion in 64-bit limbs,
    0.3 us per row), larger ones by k_fb_finish (4 rows per lane share one division-step inversion, fe_inv_safegcd: 0.09 ms
    fixed) -- level at 256 rows on a fast host, earlier on a slow one; rows with a zero sum on both sides; slices folded by k_fb_fold_quad in either case"""
    rows = bytearray(rows_bytes(n_rows, 128, b"finish%d" % n_rows))
    rows[4096 * 7:4096 * 8] = bytes(4096)
    rows[4096 * (n_rows - 1):4096 * n_rows] = bytes(4096)
    rows = bytes(rows)
    fb = mx.FixedBase("bn254", srs128, 128)
    got = fb.commit_host(rows, n_rows, 128)
    fb.close()
    assert got[64 * 7:64 * 8] == bytes(64) and got[-64:] == bytes(64)
    assert got == common.oracle_commit_batch("bn254", rows, n_rows, 128, srs128)


def test_secp256k1_few_rows(mx):
    pts = common.secp_bench_points(128)
    rows = common.secp_bench_scalars(128 * 5, start=77)
    fb = mx.FixedBase("secp256k1", pts, 128, window_bits=10)
    assert fb.commit_host(rows, 5, 128) == common.oracle_commit_batch("secp256k1", rows, 5, 128, pts)
    fb.close()


def test_sixteen_threads_coalesce_and_a_release_in_between(mx, srs128):
    """compute_digest_from_srs from 16 threads (calls that meet are committed in one launch) while another thread frees the
    device copies of the KZG state twice: every result still equals the oracle (the table is rebuilt under the same locks)"""
    import threading
    from porla_amd import lib
    mx.init_key(TAU, ALPHA)
    blob = mx.init_SRS(128)
    mx.init_SRS_from_data(128, blob)
    T = 16
    rows = rows_bytes(T, 128, b"coal")
    want = common.oracle_commit_batch("bn254", rows, T, 128, srs128)
    errors = []

    def worker(t):
        try:
            for _ in range(40):
                if mx.compute_digest_from_srs(rows[4096 * t:4096 * t + 4096]) != want[64 * t:64 * t + 64]:
                    errors.append(t)
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    def releaser():
        import time
        for _ in range(2):
            time.sleep(0.01)
            if lib.porla_kzg_release_device_memory() != 0:
                errors.append("release")

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)] + [threading.Thread(target=releaser)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]
    c, h, z, y = mx.create_proof(99, rows[:4096])           # two rows in one batch
    assert c == want[:64] and mx.verify_proof(c, h, z, y)


@pytest.mark.parametrize("n_rows,n_coeffs", [(70, 128), (300, 37), (1100, 128)])
def test_secp256k1_mid_size_batches_through_slice_fold_and_device_normalisation(mx, n_rows, n_coeffs):
    """the special-form field through the kernels of a mid-size batch: slice partials in the reduced-radix memory form folded four
    lanes per addition (k_fb_fold_quad; 37 coefficients: 32 slices), rows normalised on the host (<= 256) or by k_fb_finish's
    division-step inversion (fe_inv_safegcd); a zero row and a base point at infinity included"""
    base = bytearray(common.secp_bench_points(128))
    base[64 * 5:64 * 6] = bytes(64)
    fb = mx.FixedBase("secp256k1", bytes(base), 128, window_bits=9)
    stride = 32 * n_coeffs
    rows = bytearray(rows_bytes(n_rows, n_coeffs, b"secp-mid%d" % n_rows))
    rows[stride * 3:stride * 4] = bytes(stride)
    got = fb.commit_host(bytes(rows), n_rows, n_coeffs)
    fb.close()
    assert got[64 * 3:64 * 4] == bytes(64)
    assert got == common.oracle_commit_batch("secp256k1", bytes(rows), n_rows, n_coeffs, bytes(base))
