"""bench.py's own N-rank launcher, the parts that need no GPU: `python bench.py --gpus N` with WORLD_SIZE unset
must become a launcher BEFORE anything touches a device, refuse a node with fewer devices than ranks (RCCL: one device per rank),
and hand a failing rank's exit code up instead of hanging."""
import os
import subprocess
import sys
import time

import pytest
import torch

from tests import common

BENCH = os.path.join(common.ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "PORLA_DIST_BACKEND")}
    env.update(kw)
    return env


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="needs a node with fewer than 2 GPUs")
def test_more_ranks_than_devices_is_refused_at_once():
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300, env=_env(), cwd=common.ROOT)
    assert r.returncode == 2 and "one device per rank" in r.stderr, r.stderr[-1000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]       # no line that could be taken for a measurement
    assert time.time() - t0 < 120


@pytest.mark.skipif(torch.cuda.is_available(), reason="the failing-rank case: a container without a GPU")
def test_failing_ranks_end_the_launch_with_their_exit_code():
    """two ranks over gloo in a container without a GPU: every rank fails (there is no device and NO CPU fallback), torchrun ends
    the group, the launcher returns non-zero and prints no JSON line"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-legs", "--no-commits"],
                       capture_output=True, text=True, timeout=600, env=_env(PORLA_DIST_BACKEND="gloo", PORLA_DIST_TIMEOUT_S="60"),
                       cwd=common.ROOT)
    assert r.returncode != 0, r.stdout[-1000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_the_launcher_runs_before_any_device_call():
    """static check of the order in main(): the self-launch happens right after argument parsing, ahead of the first torch.cuda /
    library call of the process (a process that has initialised the GPU must not be replaced, and the launcher never needs one)"""
    src = open(BENCH).read()
    body = src[src.index("def main():"):]
    launch = body.index("launch_ranks(args.gpus")
    for needle in ("measure_fe_mul_peak() if", "torch.cuda.set_device", "from porla_amd import multiexp"):
        assert launch < body.index(needle), needle
    lr = src[src.index("def launch_ranks"):src.index("def main():")]
    assert "subprocess.Popen" in lr and "os.exec" not in lr
    # the launcher itself makes no torch.cuda call at all (the device count comes from a short-lived child) and picks no port
    assert "torch.cuda." not in lr and "import torch\n" not in lr and "s.bind(" not in lr and "--standalone" in lr


# ---- the ONE line the driver parses: compact, <= 6 000 bytes whatever the legs carry
def _canned_full_result():
    import json
    path = os.path.join(common.ROOT, "profiles", "r04_z_bench_default_line.json")     # round 4's 23 KB line: the one the driver could not parse
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])


def test_compact_line_fits_the_drivers_tail_and_keeps_the_contract():
    import json
    sys.path.insert(0, common.ROOT)
    import bench
    full = _canned_full_result()
    assert len(json.dumps(full)) > 20000
    c = bench.compact_line(full, "bench_legs.json")
    text = json.dumps(c)
    assert len(text) < 6000 and bench.LINE_LIMIT == 6000
    d = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "bit_exact_vs_oracle", "blocking_ms_per_step", "legs_failed"):
        assert k in d, k
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"]
    rl = d["roofline"]
    assert rl["bound"] == "hbm" and rl["kernel"] == "k_bucket_sum30" and rl["frac"] == full["roofline"]["frac"]
    assert rl["peak"] == 8000.0 and rl["unit"] == "GB/s" and rl["traffic"] == full["roofline"]["traffic"]
    assert rl["int_multiplier"]["frac"] == full["roofline"]["int_multiplier"]["frac"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 16 and cb["value"] == full["cpu_baseline"]["value"] and len(cb["sample"]) <= 80
    assert d["config"]["workload"].startswith("KZG scheme, single 2^20-point")
    for leg in ("kzg_commits", "secp256k1_msm", "icc", "config3", "ipa_commits", "mac_encode", "server_mix"):
        assert set(d[leg]) >= {"value", "unit", "ms_per_step", "frac", "int_frac", "traffic_ratio", "cpu", "bit_exact"}, leg
        assert d[leg]["value"] == full[leg]["value"] and d[leg]["bit_exact"] is full[leg]["bit_exact_vs_oracle"]
    assert d["config3"]["scaling"] == "strong" and d["config3"]["pairs_total"] == 1 << 24
    # counter traffic of the headline kernel / its algorithmic bytes: 96 B x 2^20 per launch
    assert abs(d["icc"]["traffic_ratio"] - 5.1) < 0.2 and abs(d["kzg_commits"]["traffic_ratio"] - 56) < 2


def test_compact_line_survives_oversized_and_failed_legs():
    import json
    sys.path.insert(0, common.ROOT)
    import bench
    full = _canned_full_result()
    full["config"]["workload"] = "x" * 5000
    full["cpu_baseline"]["sample"] = "y" * 9000
    full["icc"] = {"error": "RuntimeError(" + "z" * 4000 + ")", "bit_exact_vs_oracle": None}
    full["legs_failed"] = ["icc"]
    full["preflight"] = {"ok": True, "ranks": 8, "fold": "sum", "first_fold_ms": 1.0, "collective": "ncclAllGather from C++"}
    full["strong_2p20"] = dict(full["config3"], config={"pairs_total": 1 << 20, "pairs_per_gpu": 1 << 17})
    c = bench.compact_line(full, None)
    text = json.dumps(c)
    assert len(text) < 6000
    assert c["legs_failed"] == ["icc"] and "RuntimeError" in c["icc"]["error"] and c["preflight"]["ranks"] == 8
    assert c["strong_2p20"]["pairs_per_gpu"] == 1 << 17 and c["strong_2p20"]["scaling"] == "strong"


def test_emit_prints_the_compact_line_last_and_writes_the_legs_file(tmp_path, capsys):
    import json
    sys.path.insert(0, common.ROOT)
    import bench
    full = _canned_full_result()
    legs = str(tmp_path / "bench_legs.json")
    bench.emit(full, legs)
    out = capsys.readouterr().out.rstrip().splitlines()
    json_lines = [l for l in out if l.startswith("{")]
    assert len(json_lines) == 1 and out[-1] == json_lines[0] and len(out[-1]) < 6000
    assert json.loads(out[-1])["legs_file"] == "bench_legs.json"
    assert any(l.startswith("LEG mac_encode {") for l in out)
    assert json.load(open(legs)) == full            # nothing is lost: the full result is beside bench.py
    # a single-leg workload's line is the leg itself while it fits
    bench.emit(full["icc"], str(tmp_path / "icc.json"), single_leg=True)
    out = capsys.readouterr().out.rstrip().splitlines()
    assert len(out) == 1 and json.loads(out[0]) == full["icc"]


def test_round5_result_keeps_warmup_and_wall_time_on_the_line():
    """the final round-5 default run (LEG lines + compact line, profiles/r05_z_bench_default_stdout.txt): rebuilding the compact line
    from the full legs gives the line that was printed; it says how long the warm-up really was and what the run cost in wall time"""
    import json
    sys.path.insert(0, common.ROOT)
    import bench
    lines = open(os.path.join(common.ROOT, "profiles", "r05_z_bench_default_stdout.txt")).read().splitlines()
    printed = json.loads([l for l in lines if l.startswith("{")][-1])
    full = json.loads([l for l in lines if l.startswith("LEG headline ")][0][len("LEG headline "):])
    for l in lines:
        if l.startswith("LEG ") and not l.startswith("LEG headline "):
            name, obj = l[4:].split(" ", 1)
            full[name] = json.loads(obj)
    again = bench.compact_line(full, printed.get("legs_file"))
    assert len(json.dumps(printed)) < 6000
    for k in ("value", "ms_per_step", "roofline", "cpu_baseline", "kzg_commits", "icc", "config3", "run_wall_s", "legs_failed"):
        assert again[k] == printed[k], k
    cfg = printed["config"]
    assert cfg["min_warmup_s"] == 0.25 and cfg["warmup_steps_run"] >= printed["warmup"]
    assert 0 < printed["run_wall_s"] < 300          # the default run finishes within minutes
    assert all(full[leg]["wall_s"] < 60 for leg in ("kzg_commits", "secp256k1_msm", "icc", "config3"))


def test_counter_pass_children_do_not_extend_their_warmup():
    """the `rocprofv3 --pmc` child runs of bench.py must run exactly their W warm-up steps (a warm-up floor under the counters once
    took a 4-minute timeout per pass): the child arguments switch it off, and the floor's pace is taken with the device drained"""
    src = open(BENCH).read()
    child = src[src.index("def traffic_in_run"):src.index("def msm_fe_mults")]
    assert '"--min-warm-s", "0"' in child and '"--no-pmc"' in child
    timed = src[src.index("    def timed("):src.index("    def breakdown(")]
    assert timed.index("torch.cuda.synchronize()") < timed.index("per = max(")
