"""GPU box: the multi-rank path of bench.py end to end with 2 ranks -- range sharding, per-rank partial Jacobian, all_gather
of the 96-byte partials, host fold, MAX-over-ranks timing, whole-job oracle check.  The box has one GPU, so the two ranks
share it and the partials travel over gloo (PORLA_DIST_BACKEND=gloo); the driver's 8-GPU runs use nccl = RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest

from tests import common

pytestmark = pytest.mark.gpu


LINE_LIMIT = 6000      # the driver keeps an 8 KB tail of stdout: the one JSON line must fit in it whole


def _the_line(r, rc=0):
    """the ONE stdout line that starts with `{` (the last line of stdout), within the size the driver can parse"""
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == rc and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.rstrip().splitlines()[-1] == lines[0]
    assert len(lines[0]) < LINE_LIMIT, len(lines[0])
    return json.loads(lines[0])


def _check_two_rank_line(r, legs_path):
    d = _the_line(r)
    full = json.load(open(legs_path))
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["bit_exact_vs_oracle"] is True          # the 2 * 2^20-pair whole-job MSM against the oracle
    assert d["cpu_baseline"] is None                 # reported at N = 1 only
    assert d["roofline"]["kernel"] and d["value"] > 0 and d["value"] == full["value"]
    assert 0 < d["roofline"]["int_multiplier"]["frac"] <= 1.0
    assert d["legs_failed"] == []
    # the N-rank preflight ran before any timed region: sum_g (g + 1) G == N (N + 1) / 2 G on every rank
    assert d["preflight"]["ok"] is True and d["preflight"]["ranks"] == 2
    # ONE 2^20-pair MSM split over the ranks (strong scaling at the metric's own size), bit-exact
    st = d["strong_2p20"]
    assert st["scaling"] == "strong" and st["pairs_total"] == 1 << 20 and st["pairs_per_gpu"] == 1 << 19
    assert st["bit_exact"] is True and st["value"] > 0
    assert full["strong_2p20"]["config"]["pairs_per_gpu"] == 1 << 19 and full["strong_2p20"]["bit_exact_vs_oracle"] is True
    return d, full


def test_bench_gpus_2_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` run DIRECTLY (no torchrun around it, WORLD_SIZE unset): the process starts the two ranks itself as
    a child torch.distributed.run, relays the one JSON line and the exit code.  BASELINE config 3 at full size: ONE 2^24-pair job,
    2^23 pairs per rank (Client.hpp:761-787's range split), the whole job against the oracle."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PORLA_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    legs = str(tmp_path / "legs.json")
    cmd = [sys.executable, os.path.join(common.ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-commits",
           "--legs-out", legs]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=common.ROOT)
    d, full = _check_two_rank_line(r, legs)
    c3 = d["config3"]
    assert c3["scaling"] == "strong" and c3["pairs_total"] == 1 << 24 and c3["pairs_per_gpu"] == 1 << 23
    assert c3["bit_exact"] is True and c3["value"] > 0
    assert full["config3"]["config"]["pairs_per_gpu"] == 1 << 23 and full["config3"]["bit_exact_vs_oracle"] is True
    assert d["secp256k1_msm"]["value"] > 0 and d["icc"]["value"] > 0


def test_two_rank_bench_line_under_an_external_torchrun(tmp_path):
    """the driver's form: the ranks started by torch.distributed.run around bench.py"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PORLA_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(common.ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-commits", "--no-legs", "--legs-out", str(tmp_path / "legs.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=common.ROOT)
    _check_two_rank_line(r, str(tmp_path / "legs.json"))


def test_a_failed_rank_preflight_stops_the_job_before_any_timed_region(tmp_path):
    """the N-rank preflight (every rank contributes (g + 1) G, the fold must be N (N + 1) / 2 G everywhere): a rank whose partial is
    wrong (test hook) makes every rank's fold differ -- each names itself, the job exits 4, and no JSON line is printed"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PORLA_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PORLA_BENCH_PREFLIGHT_BREAK="1")
    cmd = [sys.executable, os.path.join(common.ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-commits", "--no-legs",
           "--legs-out", str(tmp_path / "legs.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=common.ROOT)
    assert r.returncode != 0, r.stdout[-1000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "rank preflight" in r.stderr and "rank 0 of 2" in r.stderr and "rank 1 of 2" in r.stderr and "no timed region is entered" in r.stderr


def test_a_failed_baseline_leg_fails_the_run(tmp_path):
    """a BASELINE-config leg that throws is named in `legs_failed` and the run exits non-zero -- the line is still printed"""
    env = dict(os.environ, PORLA_BENCH_FAIL_LEG="icc")
    r = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-commits", "--no-config3",
                        "--no-cpu", "--no-host-boundary", "--legs-out", str(tmp_path / "legs.json")],
                       capture_output=True, text=True, timeout=600, env=env, cwd=common.ROOT)
    d = _the_line(r, rc=3)
    assert d["legs_failed"] == ["icc"] and "injected" in d["icc"]["error"] and d["value"] > 0


def test_single_rank_bench_line_prices_the_kernel_it_timed(tmp_path):
    """N = 1, the default workload with its audit-size and host-boundary legs: one JSON line, the roofline's two fractions in
    (0, 1] -- the multiplication count must be the timed 2^20-pair MSM's, not that of an audit-size MSM run after it --
    and every leg bit-exact"""
    legs = str(tmp_path / "legs.json")
    r = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-commits", "--log2job", "22",
                        "--legs-out", legs], capture_output=True, text=True, timeout=600, cwd=common.ROOT)
    c = _the_line(r)
    # the compact line (what the driver parses): the contract's keys, roofline + cpu_baseline, every leg summarised
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "bit_exact_vs_oracle", "blocking_ms_per_step", "legs_failed"):
        assert k in c, k
    assert c["roofline"]["kernel"] == "k_bucket_sum30" and 0 < c["roofline"]["frac"] <= 1.0 and c["roofline"]["bound"] == "hbm"
    assert 0.3 < c["roofline"]["int_multiplier"]["frac"] <= 1.0
    assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["value"] > 0 and c["cpu_baseline"]["cores"] >= 1
    for leg in ("secp256k1_msm", "icc", "config3", "audit_combine", "kzg_audit", "mac_encode", "server_mix"):
        assert set(c[leg]) >= {"value", "unit", "ms_per_step", "frac", "int_frac", "traffic_ratio", "cpu", "bit_exact"}, leg
        assert c[leg]["bit_exact"] is True and c[leg]["value"] > 0, leg
    # the headline kernel's counter traffic comes from two rocprofv3 --pmc passes made IN this run (or says why not)
    full_rl = json.load(open(legs))["roofline"]
    assert c["roofline"]["traffic_in_run"] is True, full_rl["traffic_source"]
    assert 10 < c["roofline"]["traffic"] / (96 * (1 << 20)) < 60 and full_rl["traffic_committed_passes"] > 0
    d = json.load(open(legs))                       # the full result: what the line summarises
    assert d["value"] == c["value"] and d["roofline"]["frac"] == c["roofline"]["frac"]
    assert d["n_gpus"] == 1 and d["bit_exact_vs_oracle"] is True and d["vs_baseline"] is None
    rf = d["roofline"]
    assert rf["kernel"] == "k_bucket_sum30" and 0 < rf["frac"] <= 1.0 and 0.3 < rf["int_multiplier"]["frac"] <= 1.0
    assert abs(rf["achieved"] - 96 * (1 << 20) / (rf["kernel_ms"] * 1e-3) / 1e9) < 0.01 * rf["achieved"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert d["host_boundary"]["same_result"] is True
    assert "in this run" in rf["int_multiplier"]["peak_source"] and d["fe_mul_peak"]["bn254"]["mul_G_s"] > 50
    # the blocking call's own per-kernel breakdown next to the pipelined one
    bk = d["blocking_kernels_ms"]
    assert bk["ms_per_step"]["bucket_sum"] > 0 and bk["sum_ms_per_step"] < 2 * d["blocking_ms_per_step"]
    # every other BASELINE configuration as a leg with its own roofline / cpu_baseline / oracle check
    for leg, kernel in (("secp256k1_msm", "k_bucket_sum30"), ("icc", "k_icc_split30"), ("config3", "k_bucket_sum30")):
        assert d[leg]["bit_exact_vs_oracle"] is True, leg
        assert d[leg]["roofline"]["kernel"] == kernel and 0 < d[leg]["roofline"]["frac"] <= 1.0
        assert d[leg]["cpu_baseline"]["kind"] == "port" and d[leg]["cpu_baseline"]["value"] > 0
    assert d["config3"]["scaling"] == "strong" and d["config3"]["config"]["pairs_total"] == 1 << 22
    for group in d["audit_size_msm"].values():
        if isinstance(group, dict):
            assert all(v["bit_exact_vs_oracle"] for v in group.values())
            assert all(v["pair_bit_exact_vs_oracle"] for v in group.values())     # the audit's pair of MSMs in one launch
    # the audit side: the HBM-bound row combine and one whole audit behind porla_kzg_audit_device
    ac = d["audit_combine"]
    assert ac["bit_exact_vs_oracle"] is True and ac["roofline"]["bound"] == "hbm" and 0.3 < ac["roofline"]["frac"] <= 1.0
    assert abs(ac["roofline"]["achieved"] - ac["roofline"]["algorithmic_bytes"] / ac["kernels_ms"]["audit_accumulate"] / 1e6) < 0.02 * ac["roofline"]["achieved"]
    ka = d["kzg_audit"]
    assert ka["bit_exact_vs_oracle"] is True and ka["unit"] == "audits/s" and ka["value"] > 100 and "True" in ka["client_checks"]


def test_client_mac_batch_and_host_rows_lines(tmp_path):
    """the two figures outside the BASELINE configurations that ride on the default line: the client's block MACs in one batch
    (checked on the oracle's arithmetic, roofline of the evaluation kernel with the committed counter traffic) and the commit
    batch from pageable host rows (same bytes as from device rows)"""
    r = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--workload", "client_mac_batch", "--legs-out",
                        str(tmp_path / "a.json")], capture_output=True, text=True, timeout=600, cwd=common.ROOT)
    d = _the_line(r)
    assert d["bit_exact_vs_oracle"] is True and d["unit"] == "blocks/s" and d["value"] > 1e6
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"] == "k_kzg_eval_rows_lazy" and 0.1 < rf["frac"] <= 1.0
    assert rf["traffic"] is None or 0.9 < rf["traffic"] / rf["algorithmic_bytes_per_launch"] < 1.5
    assert d["separate_batches_ms"]["digest_batch"] > 0 and d["kernels_ms"]["fb_commit"] > 0
    r = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--workload", "kzg_commit", "--log2rows", "15", "--steps", "3",
                        "--warmup", "1", "--legs-out", str(tmp_path / "b.json")], capture_output=True, text=True, timeout=600,
                       cwd=common.ROOT)
    _the_line(r)
    d = json.load(open(str(tmp_path / "b.json")))
    assert d["bit_exact_vs_oracle"] is True
    assert d["host_rows"]["same_bytes_as_device_rows"] is True and d["host_rows"]["commits_per_s"] > 1e5
