#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X commitment engine on BASELINE.json's metric, one rank per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload bn254_msm|kzg_commit|secp256k1_msm|icc|config3|crebuild|audit_combine|kzg_audit]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Default workload (the headline metric): a "step" is ONE 2^20-pair BN254 G1 MSM per GPU (BASELINE.json config 2, "KZG
scheme, single 2^20-point BN254 G1 MSM on 1 MI355X"), inputs resident in HBM in the reference's wire format (32-B
big-endian scalars + 64-B X||Y points, porla/main.go:118-138) when the timed region starts.  With N > 1 every rank owns
its own 2^20 pairs (input-range sharding, weak scaling), produces one partial Jacobian sum, the 96-byte partials are
exchanged with ONE ncclAllGather issued from C++ inside libmultiexp.so (porla_dist_*; RCCL over xGMI) and folded with N-1 group
additions on every host: the whole job is one N*2^20-pair MSM per step.  Steps are independent MSMs;
`--in-flight 2` (default) keeps two of them in flight on two streams through the two-phase API
(porla_bn254_msm_device_begin/_end) -- the audit issues its MSMs in pairs (Server.hpp:900-901) -- so the latency-bound tail of
one MSM (bucket reduction, host fold) overlaps the bucket accumulation of the next; every step is still one complete MSM whose
result is produced and checked, and K steps = K results inside the timed region (`--in-flight 1` = blocking calls).

The same JSON line carries every other BASELINE.json configuration as a LEG -- an object with its own value, roofline,
cpu_baseline and bit_exact_vs_oracle, timed separately after the headline region, never as `value`:
  kzg_commits     2^17 rows x 128 coefficients per GPU against the resident SRS (the second half of the metric, "KZG commits/s";
                  compute_digest_from_srs hoisted over rows), with the per-call figures of the real symbol from a plain-C caller
  secp256k1_msm   2^20-point secp256k1 ecmult_multi per GPU, bench_ecmult.c's inputs                        (config 4)
  icc             2^15 rows x 128 columns ICC encode (X part + alignment scalars) per GPU = 2^22 elements      (config 5)
  config3         ONE 2^24-pair BN254 MSM over all ranks: 2^24 / N pairs per rank (strong scaling), partials exchanged and
                  folded as above, the whole job checked against the oracle                                   (config 3)
  audit_combine   Server::audit's row combine + align_MAC scalars on an 8 GiB level store: 2^18 challenged rows (the path's one
                  HBM-bound kernel: GB/s against the 8 TB/s peak) and the audit's own 3 200 rows (ms per call); N = 1 only  (s8 f-4)
  kzg_audit       ONE server-side KZG audit at the reference's size (3 200 challenged rows of a 2^15-block level resident in HBM):
                  row combine, the two MSMs over the challenged MACs, align_MAC's commitment and create_proof behind
                  porla_kzg_audit_device; audits/s, checked against what the client verifies; N = 1 only            
  ipa_commits     the IPA build's twin of kzg_commits: 2^17 rows x 128 coefficients against the 128 secp256k1 generators
                  (Client::compute_commitment, Client.hpp:374-406, hoisted over blocks); N = 1 only                  (s8 f-1)
  mac_encode      the MAC halves of CRebuild_Cached for 2^15 MACs, both curves; N = 1 only                            (s8 f-2)
  server_mix      Server::mix in one call (data rows + MAC commitments + MAC alignments of two 2^12-row blocks), both curves; N = 1 only
  client_mac_batch  the block MACs of Client::initialize (digest + complement + add_point per block) for 2^17 blocks resident in
                  HBM behind porla_kzg_mac_batch_device; blocks/s, three blocks checked on the oracle's arithmetic; N = 1 only (s8 a4)
and, for the headline MSM: `blocking_ms_per_step` + `blocking_kernels_ms` (one MSM in flight: what a caller that waits for
every result sees, with its own per-kernel breakdown), `audit_size_msm` (128 / 1 408 / 3 200 pairs: the sizes the reference
issues), `host_boundary` (compute_multi_exp on caller-owned pageable host buffers, PCIe included).
`--workload X` prints leg X alone as the line (`crebuild`: the device-resident last encode stage, tools/bench_crebuild.py).

The LAST stdout line is ONE compact JSON object (rank 0; <= 6 000 bytes: compact_line); the full result goes to bench_legs.json
(--legs-out) and to earlier `LEG <name> {...}` lines.  `roofline` is for the dominant kernel of the workload, timed with HIP events on
the launch stream inside the library over the timed region; `traffic` of the headline and of the BASELINE-configuration legs comes from
two rocprofv3 --pmc passes made in this run (pmc_in_run: child runs of the same workload after the timed regions), of the other legs
from the committed passes (profiles/pmc_latest*.json: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, KiB -> bytes, per launch);
`int_multiplier.peak` is measured in this run on this box (porla_amd/fe30_check --peak: back-to-back products of the field
form the kernels use); `cpu_baseline` is the oracle (CPU restatement -- NOT gnark / libsecp256k1 / NTL) timed on this box's
host cores.
"""
import argparse
import ctypes
import subprocess
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
MSM_BYTES_PER_PAIR = 96       # 32-B scalar + 64-B affine point
COMMIT_BYTES_PER_ROW = 4096 + 64  # 128 x 32-B coefficients in, 64-B point out (SRS table resident)
# field products a MAC butterfly EXECUTES, averaged over the 15 launches of a 2^15-MAC encode (mac_fft.hip.h): stage 1 two additions
# (14 fe_mul each); stages 2..11 the wave-uniform sparse ladder: 128 doublings (9) + ~43 + 8 table + 2 butterfly additions;
# stages 12..15 the fixed-window ladder: 132 doublings + ~62 + 7 + 2 additions
MAC_FE_MULTS_PER_BUTTERFLY = (28 + 10 * (128 * 9 + 53 * 14) + 4 * (132 * 9 + 71 * 14)) / 15.0
ICC_BYTES_PER_ELEMENT = 64    # 32 B in + 32 B out
# fallback when the in-run measurement is unavailable: back-to-back 256-bit modular products, G/s per GPU, in the reduced-radix
# form of fe30.hip.h (profiles/r01_l_ubench_fe30.txt): BN254 180.7 G products/s and 213.6 G squares/s -> 186.5 for the 8M + 2S
# mix of a mixed addition; secp256k1 (special-form fold) 194.2 and 226.2 -> 199.9
FE_MUL_PEAK_FALLBACK = {"bn254": 186.5, "secp256k1": 199.9}
WORKLOAD_FIELD = {"bn254_msm": "bn254", "kzg_commit": "bn254", "secp256k1_msm": "secp256k1", "config3": "bn254", "strong_2p20": "bn254"}
PMC_FILE = {"bn254_msm": "pmc_latest.json", "config3": "pmc_latest.json"}

# legs of the default line that measure a BASELINE.json configuration (or the metric's second half): one of them throwing fails the run
BASELINE_CONFIG_LEGS = ("kzg_commits", "secp256k1_msm", "icc", "config3", "strong_2p20")

KERNEL_SYMBOL = {  # profile slot -> substring of the kernel symbol in the rocprofv3 output
    "bucket_sum": "k_bucket_sum30", "tree_levels": "k_tree_level", "tree_tail": "k_tree_tail", "partition_sort": "k_partition_sort",
    "fb_commit": "k_fb_commit", "digits_partition": "k_digits_partition", "points_to_mont": "k_points_to_mont",
    "icc_fused": "k_icc_split30", "icc_stages_r4": "k_icc_stages", "icc_stages_r2": "k_icc_stages", "icc_load": "k_icc_load", "icc_finish": "k_icc_finish",
    "audit_accumulate": "k_audit_accumulate<8>",     # the large-challenge instantiation (8 row slices per block)
    "kzg_eval_rows": "k_kzg_eval_rows_lazy",
    "mac_stage": "k_mac_stage30_quad", "icc_mix": "k_icc_mix30", "mac_mix": "k_mac_mix_quad",
}

LINE_LIMIT = 6000             # bytes: the driver keeps an 8 KB tail of stdout; the LAST line must fit in it whole
LEG_KEYS = ("kzg_commits", "secp256k1_msm", "icc", "config3", "strong_2p20", "audit_combine", "kzg_audit", "client_mac_batch",
            "ipa_commits", "mac_encode", "server_mix")


def _short(s, n=80):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 1] + "~"


def _compact_roofline(rl):
    if not isinstance(rl, dict):
        return None
    out = {k: rl.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms")}
    out["traffic_in_run"] = "collected in this run" in (rl.get("traffic_source") or "")
    im = rl.get("int_multiplier")
    if isinstance(im, dict):
        out["int_multiplier"] = {"achieved": im.get("achieved"), "peak": im.get("peak"), "unit": "G fe_mul/s", "frac": im.get("frac")}
    return out


def _compact_cpu(cpu):
    if not isinstance(cpu, dict):
        return None
    out = {k: cpu.get(k) for k in ("value", "unit", "cores", "kind")}
    out["sample"] = _short(cpu.get("sample"))
    return out


def _traffic_ratio(rl):
    """counter bytes per launch / algorithmic bytes per launch (achieved x kernel time), or None.  A kernel that is one of several
    dependent launches over the same array (a stage of the MAC-side network) states what ONE launch has to touch
    (`touched_bytes_per_launch`): the ratio is against that -- the call's algorithmic bytes divided by its launches would call
    every multi-pass method wasteful by construction"""
    try:
        if rl.get("touched_bytes_per_launch"):
            return round(rl["traffic"] / rl["touched_bytes_per_launch"], 2)
        for k in ("algorithmic_bytes_per_launch", "algorithmic_bytes"):        # legs that state the launch's bytes themselves
            if rl.get(k):
                return round(rl["traffic"] / rl[k], 2)
        return round(rl["traffic"] / (rl["achieved"] * rl["kernel_ms"] * 1e6), 2)
    except (TypeError, KeyError, ZeroDivisionError):
        return None


def _compact_leg(leg):
    if not isinstance(leg, dict):
        return leg
    if "error" in leg:
        return {"error": _short(leg["error"], 160), "bit_exact": None}
    rl = leg.get("roofline") if isinstance(leg.get("roofline"), dict) else {}
    cpu = leg.get("cpu_baseline") if isinstance(leg.get("cpu_baseline"), dict) else None
    out = {"value": leg.get("value"), "unit": leg.get("unit"), "ms_per_step": leg.get("ms_per_step"), "frac": rl.get("frac"),
           "int_frac": (rl.get("int_multiplier") or {}).get("frac"), "traffic_ratio": _traffic_ratio(rl),
           "cpu": {"value": cpu.get("value"), "cores": cpu.get("cores"), "kind": cpu.get("kind")} if cpu else None,
           "bit_exact": leg.get("bit_exact_vs_oracle")}
    if leg.get("scaling") == "strong":
        out["scaling"] = "strong"
        cfg = leg.get("config") or {}
        out["pairs_total"], out["pairs_per_gpu"] = cfg.get("pairs_total"), cfg.get("pairs_per_gpu")
        if "blocking_ms_per_step" in leg:
            out["blocking_ms_per_step"] = leg["blocking_ms_per_step"]
    return out


def compact_line(full, legs_file=None):
    """the ONE line the driver parses: the contract's keys for the headline workload + roofline + cpu_baseline, and per leg only
    {value, unit, ms_per_step, frac, int_frac, traffic_ratio, cpu, bit_exact}.  Everything else (per-kernel breakdowns, notes,
    sample descriptions, host-boundary and audit-size figures) is in `legs_file` (bench_legs.json) and on the earlier `LEG ...`
    stdout lines.  json.dumps of the result is <= LINE_LIMIT bytes by construction (checked; the config strings are cut further
    if a future key pushes it over)."""
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    cfg = full.get("config") or {}
    out["config"] = {k: (_short(v, 160) if isinstance(v, str) else v) for k, v in cfg.items()
                     if isinstance(v, (str, int, float, bool)) or v is None}
    out["roofline"] = _compact_roofline(full.get("roofline"))
    out["cpu_baseline"] = _compact_cpu(full.get("cpu_baseline"))
    out["bit_exact_vs_oracle"] = full.get("bit_exact_vs_oracle")
    for k in ("blocking_ms_per_step", "blocking_Mmul_s", "preflight", "run_wall_s"):
        if k in full:
            out[k] = full[k]
    hb = full.get("host_boundary")
    if isinstance(hb, dict):
        out["host_boundary"] = {k: hb.get(k) for k in ("ms", "Mmul_s", "same_result") if k in hb}
    for k in LEG_KEYS:
        if k in full:
            out[k] = _compact_leg(full[k])
    fp = full.get("fe_mul_peak")
    if isinstance(fp, dict):
        out["fe_mul_peak"] = {f: (v.get("mix_8M_2S") if isinstance(v, dict) else v) for f, v in fp.items()}
    out["legs_failed"] = full.get("legs_failed", [])
    if legs_file:
        out["legs_file"] = legs_file
    if len(json.dumps(out)) > LINE_LIMIT:
        for k, v in out["config"].items():
            out["config"][k] = _short(v, 60)
        if out.get("cpu_baseline"):
            out["cpu_baseline"]["sample"] = _short(out["cpu_baseline"]["sample"], 40)
    if len(json.dumps(out)) > LINE_LIMIT:      # last resort: legs down to value + bit_exact
        for k in LEG_KEYS:
            if isinstance(out.get(k), dict):
                out[k] = {kk: out[k].get(kk) for kk in ("value", "unit", "frac", "bit_exact", "error") if kk in out[k]}
    return out


def emit(full, legs_out, single_leg=False):
    """write the full result to `legs_out`, print every leg's full object on a `LEG <name> {...}` line (not a JSON line: the
    driver and the tests take the LAST line that starts with `{`), then the compact line as the last line of stdout"""
    legs_file = None
    if legs_out:
        try:
            with open(legs_out + ".tmp", "w") as f:
                json.dump(full, f, indent=1)
            os.replace(legs_out + ".tmp", legs_out)
            legs_file = os.path.basename(legs_out)
        except OSError as e:
            print("warning: cannot write %s: %s" % (legs_out, e), file=sys.stderr)
    if single_leg:
        # `--workload X`: the leg IS the line; printed whole while it fits, compact otherwise
        text = json.dumps(full)
        if len(text) <= LINE_LIMIT:
            print(text, flush=True)
            return full
    head = {k: v for k, v in full.items() if k not in LEG_KEYS}
    print("LEG headline " + json.dumps(head))
    for k in LEG_KEYS:
        if k in full:
            print("LEG %s %s" % (k, json.dumps(full[k])))
    out = compact_line(full, legs_file)
    print(json.dumps(out), flush=True)
    return out


CURVE_TAGS = {"bn254": ("Bn254",), "secp256k1": ("Secp256k1",)}      # template arguments that name the curve in a kernel symbol


def pmc_traffic(slot, workload, curve=None):
    """HBM-side bytes per launch of `slot`'s kernel from the committed PMC summary of this workload, or None; `curve`: only the
    instantiations of that curve (a leg that runs both curves in one process has both in its counter file)"""
    path = os.path.join(ROOT, "profiles", PMC_FILE.get(workload, "pmc_latest_%s.json" % workload))
    sym = KERNEL_SYMBOL.get(slot, slot)
    try:
        d = json.load(open(path))
        tags = CURVE_TAGS.get(curve, ("",))
        fetch = [v for k, v in d["fetch"].items() if sym in k and "FETCH_SIZE" in k and any(t in k for t in tags)]
        write = [v for k, v in d["write"].items() if sym in k and "WRITE_SIZE" in k and any(t in k for t in tags)]
        if not fetch or not write:
            return None
        # per launch, averaged over every instantiation of the kernel that ran (the ICC encode launches a first-pass and a
        # last-pass variant): sum / dispatches.  rocprofv3 reports KiB; gfx950 FETCH_SIZE counts 128-B requests as 64 B ->
        # doubled (MI355X_MICROARCH.md, HBM)
        per = lambda rows: sum(r["sum"] for r in rows) / max(1, sum(r["dispatches"] for r in rows))
        return int((2 * per(fetch) + per(write)) * 1024)
    except (OSError, KeyError, ValueError):
        return None


def pmc_in_run(symbol, bench_args, timeout_s=240):
    """HBM-side bytes per launch of the kernel whose name contains `symbol`, collected NOW: two child runs of this script under
    `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950; the program itself follows `--`, no trace domain is
    combined with the counters).  Returns (bytes per launch, launches counted) or (None, reason).  rocprofv3 reports KiB;
    gfx950's FETCH_SIZE counts a 128-byte request as 64 bytes -> doubled (MI355X_MICROARCH.md, HBM)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCPROFILER", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself under a profiler"
    per = {}
    launches = 0
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="porla_pmc_")
        try:
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "run", "--", sys.executable,
                   os.path.abspath(__file__)] + bench_args
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=dict(child_env(), TMPDIR="/tmp"))
            tot, cnt = 0.0, 0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if symbol in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                        tot += float(row.get("Counter_Value", 0))
                        cnt += 1
            if cnt == 0:
                return None, "no %s rows for %s (rocprofv3 rc %d)" % (counter, symbol, r.returncode)
            per[counter] = tot / cnt
            launches = cnt
        except Exception as e:  # noqa: BLE001
            return None, "%s pass failed: %r" % (counter, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int((2 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024), launches


def child_env():
    """environment for the measurement children (plain-C harness, host-boundary script): the parent's, minus a preloaded profiler"""
    return {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCPROF", "ROCPROFILER", "ROCP_"))}


def measure_fe_mul_peak():
    """back-to-back product / square rates of both base fields on THIS box, now (a child process, ~0.1 s of GPU time), as the
    8M + 2S mix of a mixed addition: G fe_mul/s per field, or the committed round-1 figures when the tool is missing"""
    exe = os.path.join(ROOT, "porla_amd", "fe30_check")
    try:
        r = subprocess.run([exe, "--peak"], capture_output=True, text=True, timeout=120, env=child_env())
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        out = {}
        for field in ("bn254", "secp256k1"):
            # the multiplier's ceiling = the best rate over the occupancies measured (4 waves per SIMD: what a kernel of the
            # accumulation's register footprint can hold; 8: the most the chip holds)
            per = {w: 10.0 / (8.0 / d[w][field]["mul_G_s"] + 2.0 / d[w][field]["sqr_G_s"]) for w in ("waves_4", "waves_8")}
            best = max(per, key=per.get)
            out[field] = {"mix_8M_2S": round(per[best], 2), "at_4_waves_per_simd": round(per["waves_4"], 2),
                          "at_8_waves_per_simd": round(per["waves_8"], 2), "mul_G_s": d[best][field]["mul_G_s"],
                          "sqr_G_s": d[best][field]["sqr_G_s"],
                          "source": "porla_amd/fe30_check --peak in this run (best of 4 and 8 waves per SIMD: %s)" % best}
        return out
    except Exception as e:  # noqa: BLE001
        return {f: {"mix_8M_2S": v, "source": "profiles/r01_l_ubench_fe30.txt (in-run measurement failed: %r)" % (e,)}
                for f, v in FE_MUL_PEAK_FALLBACK.items()}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` started by hand (no WORLD_SIZE in the environment): start the N ranks as ONE child
    `python -m torch.distributed.run` (the driver's launcher; rendezvous `--standalone` on 127.0.0.1) BEFORE this process has touched the GPU
    (it never does: a child, never an exec), relay its output (rank 0 prints the one JSON line) and return its exit code.
    A rank that dies takes the group down through torchrun; a group that hangs is killed at PORLA_BENCH_LAUNCH_TIMEOUT_S."""
    import signal
    backend = os.environ.get("PORLA_DIST_BACKEND", "nccl")
    if backend == "nccl":
        # the device count comes from a short-lived CHILD (on some ROCm wheels counting devices initialises the HIP runtime;
        # this process must stay clear of it whatever the wheel does)
        try:
            r = subprocess.run([sys.executable, "-c", "import torch as t; print(t.cuda.device_count())"], capture_output=True,
                               text=True, timeout=300)
            have = int(r.stdout.strip().splitlines()[-1])
        except Exception:  # noqa: BLE001  (no count: the rank-side check below still refuses a rank without a device)
            have = n
        if have < n:
            print("ERROR: --gpus %d but this node shows %d GPU(s): RCCL needs one device per rank.  (To rehearse the N-rank path "
                  "with ranks sharing a device, partials over gloo on the host: PORLA_DIST_BACKEND=gloo.)" % (n, have), file=sys.stderr)
            return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
    # the ranks are host-light between launches; without this torchrun pins OMP_NUM_THREADS=1, which would starve the oracle's
    # range-split check on rank 0 (common.ncpu() still honours affinity and the cgroup quota)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
    # --standalone: torchrun's own c10d rendezvous on a port IT picks and binds (no pick-then-rebind race), on 127.0.0.1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(n), os.path.abspath(__file__)] + argv
    limit = float(os.environ.get("PORLA_BENCH_LAUNCH_TIMEOUT_S", "3000"))
    p = subprocess.Popen(cmd, env=env, cwd=ROOT, start_new_session=True)   # own process group: killed as a group, by its pgid only
    try:
        return p.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        print("ERROR: the %d-rank run did not finish within %.0f s; killing process group %d" % (n, limit, p.pid), file=sys.stderr)
        try:
            os.killpg(p.pid, signal.SIGTERM)
            p.wait(timeout=20)
        except (subprocess.TimeoutExpired, ProcessLookupError):
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        return 124
    except KeyboardInterrupt:
        try:
            os.killpg(p.pid, signal.SIGTERM)
        except ProcessLookupError:
            pass
        raise


def main():
    t_main = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="bn254_msm",
                    choices=["bn254_msm", "kzg_commit", "secp256k1_msm", "icc", "config3", "crebuild", "audit_combine", "kzg_audit",
                             "client_mac_batch", "ipa_commits", "mac_encode", "server_mix", "strong_2p20"])
    ap.add_argument("--log2n", type=int, default=20, help="MSM pairs per GPU = 2^log2n (default: the 2^20 of BASELINE.json)")
    ap.add_argument("--log2rows", type=int, default=17, help="kzg_commit rows per GPU = 2^log2rows; icc rows = 2^(log2rows-2)")
    ap.add_argument("--log2job", type=int, default=24, help="config3: pairs of the whole job = 2^log2job, split over the ranks")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline / bit-exact check legs")
    ap.add_argument("--no-commits", action="store_true", help="bn254_msm: skip the kzg_commits leg")
    ap.add_argument("--no-legs", action="store_true", help="bn254_msm: skip the secp256k1_msm / icc / config3 legs")
    ap.add_argument("--no-config3", action="store_true", help="bn254_msm: skip the config3 leg")
    ap.add_argument("--no-host-boundary", action="store_true", help="bn254_msm: skip the compute_multi_exp-on-host-buffers leg")
    ap.add_argument("--in-flight", type=int, default=2, help="bn254_msm: independent MSMs in flight (1 = blocking calls; 2 = the "
                    "audit's pair of MSMs, Server.hpp:900-901, overlapped on two streams)")
    ap.add_argument("--min-warm-s", type=float, default=0.25,
                    help="every timed region warms up for at least this long (W steps, then more of the same); 0: exactly W steps")
    ap.add_argument("--no-pmc", action="store_true", help="bn254_msm: do not collect the dominant kernel's counter traffic in this run "
                    "(two short child runs under rocprofv3 --pmc); the committed passes are reported instead")
    ap.add_argument("--legs-out", default=os.path.join(ROOT, "bench_legs.json"),
                    help="where the FULL result (every leg's per-kernel breakdown, notes, samples) is written; the last stdout line is "
                         "the compact headline object (<= %d bytes)" % LINE_LIMIT)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand with --gpus N: this process becomes the launcher of the N ranks (nothing below runs in it)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    if args.workload == "kzg_audit":
        # one whole server-side audit at the reference's size has its own driver too (tools/bench_audit_flow.py, same JSON contract)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_audit_flow.py")] + (["--no-cpu"] if args.no_cpu else []))
        sys.exit(r.returncode)
    if args.workload == "crebuild":
        # the device-resident last encode stage has its own driver (same JSON contract), run as a child: nothing here has touched
        # the GPU yet
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_crebuild.py"), "--steps", str(args.steps),
                            "--warmup", str(args.warmup)] + (["--no-cpu"] if args.no_cpu else []))
        sys.exit(r.returncode)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # the multiplier's rate on this box, before this process has any work on the GPU (rank 0 measures, one device)
    fe_peak = measure_fe_mul_peak() if rank == 0 else None

    import torch
    import torch.distributed as dist

    # PORLA_DIST_BACKEND=gloo: exercise the multi-rank path on a box with fewer GPUs than ranks (ranks share devices, the
    # 96-byte partials travel over gloo on the host); the driver's runs use nccl = RCCL over xGMI, one rank per GPU
    backend = os.environ.get("PORLA_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl" and torch.cuda.device_count() <= dev_index:
            # a rank without a device of its own leaves at once (non-zero): torchrun then ends the other ranks instead of letting
            # them wait in the rendezvous
            print("ERROR: rank %d needs cuda:%d but this node shows %d GPU(s)" % (rank, dev_index, torch.cuda.device_count()), file=sys.stderr)
            sys.exit(2)
        # a bounded rendezvous / collective wait: a rank that cannot come up makes the others fail, not hang
        wait = datetime.timedelta(seconds=float(os.environ.get("PORLA_DIST_TIMEOUT_S", "300")))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index), timeout=wait)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=wait)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")    # where collective payloads live

    from porla_amd import multiexp as mx
    from porla_amd import sharded
    mx.lib = __import__("porla_amd.loader", fromlist=["lib"]).lib
    from tests import common  # oracle access is allowed here for input generation + the cpu_baseline leg only

    # N > 1: the 96-byte partials travel through ONE ncclAllGather issued from C++ inside libmultiexp.so (porla_dist_*, RCCL bound
    # with dlopen); torch.distributed only hands the ncclUniqueId around and keeps the barrier / MAX-over-ranks timing.  Should
    # the in-library communicator fail to come up on some node, the same exchange runs through torch.distributed's nccl
    # backend (also RCCL) -- `collective` in the JSON line says which.  With PORLA_DIST_BACKEND=gloo (ranks sharing one GPU,
    # which RCCL refuses) the partials go over gloo on the host.
    collective = "none (single GPU)"
    use_cxx_dist = False
    hard_exit = False          # an in-library ncclCommInitRank timed out: its helper thread is still inside RCCL (include/porla_gpu.h)
    if world > 1:
        collective = "torch.distributed %s all_gather" % backend
        if backend == "nccl" and os.environ.get("PORLA_DIST_CXX", "1") != "0":
            def all_ok(flag):
                t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=coll_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                return int(t.item()) == 1
            # 1. every rank checks that RCCL can be bound at all BEFORE anyone enters the collective ncclCommInitRank
            #    (a rank that fails early would leave the others waiting there)
            probe = None
            try:
                probe = mx.dist_unique_id()
            except Exception as e:  # noqa: BLE001
                print("rank %d: in-library RCCL unavailable (%s)" % (rank, e), file=sys.stderr)
            if all_ok(probe is not None):
                uid = [probe if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                inited = False
                try:
                    mx.dist_init(uid[0], rank, world)
                    inited = True
                except Exception as e:  # noqa: BLE001
                    print("rank %d: ncclCommInitRank from C++ failed (%s)" % (rank, e), file=sys.stderr)
                    hard_exit = "did not return within" in str(e)
                use_cxx_dist = all_ok(inited)
                if use_cxx_dist:
                    collective = "ncclAllGather from C++ (porla_dist_*, RCCL over xGMI)"
                elif inited:
                    mx.dist_finalize()

    def fold_across_ranks(curve, part):
        if use_cxx_dist:
            return mx.dist_fold(curve, part)
        return sharded.fold_partials(curve, sharded.gather_partials(part, coll_dev))

    def rank_preflight():
        """BEFORE any timed region with N > 1: prove the gather + fold on the N real ranks.  Rank g contributes (g + 1) * G as its
        96-byte Jacobian partial (host arithmetic of the library: mult_point on the generator), the fold -- the same call the timed
        steps use -- must equal N (N + 1) / 2 * G on EVERY rank (the reference folds its 8 workers' partials the same way,
        Client.hpp:783-787).  A rank whose fold differs names itself and the whole job exits non-zero."""
        nonlocal use_cxx_dist, collective
        gen = (1).to_bytes(32, "big") + (2).to_bytes(32, "big")                      # BN254 G1 generator (1, 2)
        mine = mx.bn254_mult(gen, mx.bn254_scalar_set_int(rank + 1))
        if os.environ.get("PORLA_BENCH_PREFLIGHT_BREAK") == str(rank):              # test hook: this rank's partial is wrong
            mine = mx.bn254_mult(gen, mx.bn254_scalar_set_int(rank + 2))
        want = mx.bn254_mult(gen, mx.bn254_scalar_set_int(world * (world + 1) // 2))

        def attempt():
            t0 = time.perf_counter()
            try:
                got = fold_across_ranks("bn254", sharded.affine_to_partial(mine))
                err = None if got == want else "folded %s, expected %s" % (got.hex()[:32], want.hex()[:32])
            except Exception as e:  # noqa: BLE001
                err = "the fold raised %r" % (e,)
            ms = (time.perf_counter() - t0) * 1e3
            if err:
                print("ERROR: rank preflight (%s): rank %d of %d: %s" % (collective, rank, world, err), file=sys.stderr, flush=True)
            flags = [None] * world
            dist.all_gather_object(flags, err is None)
            return [g for g, f in enumerate(flags) if not f], ms

        bad, ms = attempt()
        first_try = None
        if bad and use_cxx_dist:
            # the in-library exchange failed on some rank: every rank drops it (the flags are the same everywhere) and the proof is
            # repeated over torch.distributed's own RCCL collectives -- the run goes on with those and says so
            first_try = {"collective": collective, "failed_ranks": bad}
            try:
                mx.dist_finalize()
            except Exception as e:  # noqa: BLE001
                print("rank %d: porla_dist_finalize after the failed preflight: %r" % (rank, e), file=sys.stderr)
            use_cxx_dist = False
            collective = "torch.distributed %s all_gather (the in-library exchange failed the preflight)" % backend
            bad, ms = attempt()
        if bad:
            if rank == 0:
                print("ERROR: rank preflight failed on rank(s) %s: no timed region is entered" % bad, file=sys.stderr, flush=True)
            dist.barrier()
            sys.exit(4)
        out = {"ok": True, "ranks": world, "fold": "sum_g (g+1) G == N(N+1)/2 G on every rank", "first_fold_ms": round(ms, 3),
               "collective": collective}
        if first_try:
            out["first_try"] = first_try
        return out

    preflight = rank_preflight() if world > 1 else None

    stream = torch.cuda.current_stream().cuda_stream
    # every timed region warms up for at least this long (W steps, then more of the same until the time has passed) -- see timed();
    # --warmup 0 switches it off
    leg_warm_s = args.min_warm_s
    head_warm_s = args.min_warm_s

    def to_dev(b):
        return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(step, drain=None, min_warm_s=0.0):
        """W warmup + K timed steps, barrier + synchronize on both sides, MAX over ranks; returns (seconds, kernel ms, last
        result).  `drain` (pipelined steps) retires whatever is still in flight: the timed region contains K complete steps.
        `min_warm_s`: the warm-up goes on -- with further untimed steps of the same kind -- until the device has been busy for that
        long.  A timed region starts after seconds of host work (input generation, the previous leg's CPU baseline) with the GPU idle
        and its clocks down, and W = 5 steps of a 1.4 ms MSM (7 ms) do not bring them back: the same 20 steps measure 716-719 Mmul/s
        after 5 warm-up steps and 757-763 after 200 (same box; the accumulation kernel 1.01 against 0.94 ms) -- the latter is the
        sustained rate the 100-step runs see.  The line says how many warm-up steps ran (`warmup_steps_run`)."""
        res = None
        t_w = time.perf_counter()
        for _ in range(args.warmup):
            res = step()
        timed.warm_steps = args.warmup
        if min_warm_s and args.warmup:
            # the SAME number of extra steps on every rank (a step may contain a collective): from this rank's pace, MAX over ranks.
            # The pace is taken with the device drained: a step that only enqueues returns in microseconds (and would ask for
            # thousands of extra steps)
            torch.cuda.synchronize()
            per = max((time.perf_counter() - t_w) / args.warmup, 1e-4)
            extra = max(0, min(500, int((min_warm_s - (time.perf_counter() - t_w)) / per) + 1))
            if world > 1:
                t = torch.tensor([extra], dtype=torch.int64, device=coll_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                extra = int(t.item())
            for _ in range(extra):
                res = step()
            timed.warm_steps += extra
        if drain:
            res = drain() or res
        sync()
        # HIP events around the DOMINANT kernel only inside the timed region (every recorded kernel costs ~10 us of idle
        # GPU around it); the per-kernel breakdown comes from three extra, untimed steps with events around every kernel
        mx.profile_enable(2)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = step() or res
        if drain:
            res = drain() or res
        sync()
        el = time.perf_counter() - t0
        prof = mx.profile_get()
        mx.profile_enable(True)
        for _ in range(3):
            step()
        if drain:
            drain()
        sync()
        prof_all = mx.profile_get()
        mx.profile_enable(False)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        timed.totals = {name: ms / 3 for name, ms, cnt in prof_all}   # per step, all launches of the kernel (breakdown pass)
        kern = {name: ms / max(cnt, 1) for name, ms, cnt in prof_all}
        kern.update({name: ms / max(cnt, 1) for name, ms, cnt in prof})  # the dominant kernel: timed region's own average
        timed.dominant = [name for name, ms, cnt in prof]
        return el, kern, res

    def breakdown(step, reps=3):
        """per-kernel HIP-event times (ms per launch, ms per step) of `reps` untimed steps issued one at a time"""
        sync()
        mx.profile_enable(True)
        for _ in range(reps):
            step()
            torch.cuda.synchronize()
        prof = mx.profile_get()
        mx.profile_enable(False)
        return ({name: round(ms / max(cnt, 1), 4) for name, ms, cnt in prof},
                {name: round(ms / reps, 4) for name, ms, cnt in prof})

    def roofline(kern, algo_bytes_per_launch, workload, fe_mults_per_launch=None):
        if not kern:
            return None
        dom = max(kern, key=kern.get)
        ach = algo_bytes_per_launch / (kern[dom] * 1e-3) / 1e9
        r = {"bound": "hbm", "kernel": KERNEL_SYMBOL.get(dom, dom), "achieved": round(ach, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
             "frac": round(ach / HBM_PEAK_GBPS, 6), "traffic": pmc_traffic(dom, workload),
             "traffic_source": "committed rocprofv3 --pmc passes (profiles/%s), not collected in this run"
                               % PMC_FILE.get(workload, "pmc_latest_%s.json" % workload),
             "kernel_ms": round(kern[dom], 4), "all_kernels_ms": {k: round(v, 4) for k, v in kern.items()}}
        if fe_mults_per_launch and fe_peak:
            # the elliptic-curve kernels are bound by the VALU's 32x32->64 multiply-add issue rate, which neither "hbm" nor "mfma"
            # names: the supplement prices the dominant kernel's 256-bit modular multiplications against the rate the
            # multiply microbenchmark sustains on this chip in this run (tools/fe30_check.hip --peak)
            g = fe_mults_per_launch / (kern[dom] * 1e-3) / 1e9
            pk = fe_peak[WORKLOAD_FIELD[workload]]
            r["int_multiplier"] = {"achieved": round(g, 2), "peak": pk["mix_8M_2S"], "unit": "G fe_mul/s (256-bit modular)",
                                   "frac": round(g / pk["mix_8M_2S"], 4), "peak_source": pk["source"],
                                   "note": "mixed addition = 8M + 2S = 10 fe_mul, a bucket's first entry a copy (0), its second 4M + 2S = 6 (the ALGORITHMIC count of the group law; since "
                                           "round 4 the kernel computes R D - Y1 PPP with one reduction, ~9.4 product-equivalents of "
                                           "instructions per addition); peak = back-to-back product rate of the field form the kernel "
                                           "uses (8M + 2S mix)"}
        return r

    def traffic_in_run(rl, child_args):
        """replace rl['traffic'] (committed passes) by two rocprofv3 --pmc passes of `child_args` made now; N = 1, rank 0 only"""
        if not rl or world != 1 or rank != 0 or args.no_pmc:
            return rl
        sym = rl["kernel"]
        t_pmc = time.perf_counter()
        got, info = pmc_in_run(sym, child_args + ["--no-cpu", "--no-pmc", "--steps", "3", "--warmup", "1", "--min-warm-s", "0", "--legs-out", ""])
        rl["traffic_passes_wall_s"] = round(time.perf_counter() - t_pmc, 1)
        if got:
            rl["traffic_committed_passes"] = rl.get("traffic")
            rl["traffic"] = got
            rl["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes collected in this run (%d launches of %s per "
                                    "pass; FETCH_SIZE doubled per the gfx950 correction)" % (info, sym))
        else:
            rl["traffic_source"] = (rl.get("traffic_source") or "") + "; in-run collection unavailable: %s" % info
        return rl

    def msm_fe_mults(n):
        # what the accumulation kernel multiplies, counted by the group law: per (sub-scalar, window) digit one mixed addition
        # (8M + 2S = 10 fe_mul) -- except a bucket's FIRST entry, which is a copy (0), and its SECOND, which meets an affine
        # accumulator (4M + 2S = 6).  Buckets filled by uniformly random digits: Poisson with mean `load`.  Zero digits (2^-c of
        # them) are not subtracted.
        import math
        c, windows, glv = mx.last_msm_shape()
        subs = 2 if glv else 1
        buckets = windows * float(1 << (c - 1))
        load = n * subs / float(1 << (c - 1))
        nonempty = buckets * (1.0 - math.exp(-load))
        two_plus = buckets * (1.0 - math.exp(-load) * (1.0 + load))
        return 10.0 * (n * subs * windows - nonempty) - 4.0 * two_plus

    def line(metric, value, unit, el, scaling, dtype, config, rl, cpu, verified, **extra):
        d = {"metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
             "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
             "dtype": dtype, "data": "synthetic", "config": config, "roofline": rl, "cpu_baseline": cpu,
             "bit_exact_vs_oracle": verified}
        d.update(extra)
        return d

    # ---------------------------------------------------------------- KZG batched commitments
    kzg_cache = []

    def kzg_setup():
        if kzg_cache:                                               # one key, one SRS per run: the legs share them
            return kzg_cache[0]
        tau = bytes.fromhex("ffeeddccbbaa99887766554433221100")     # TAU_KEY, config.hpp:39
        alpha = bytes.fromhex("00112233445566778899aabbccddeeff")   # SECRET_KEY, config.hpp:38
        mx.init_key(tau, alpha)
        blob = mx.init_SRS(128)                                     # client side (Client.hpp:348-354)
        mx.init_SRS_from_data(128, blob)                            # server side (Server.hpp:183-188)
        o = common.oracle()
        o.oracle_kzg_init_key(tau, ctypes.c_size_t(16), alpha, ctypes.c_size_t(16))
        o.oracle_kzg_init_srs(ctypes.c_size_t(128), (1).to_bytes(32, "big"))
        raw = ctypes.create_string_buffer(64 * 128)
        o.oracle_kzg_srs_g1_raw(raw)
        kzg_cache.append(raw.raw)
        return raw.raw

    def leg_kzg_commit():
        rows_n = 1 << args.log2rows
        srs_raw = kzg_setup()
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        d_rows = torch.randint(0, 256, (rows_n * 4096,), dtype=torch.uint8, device=dev, generator=g)
        d_out = torch.empty(rows_n * 64, dtype=torch.uint8, device=dev)

        def step():
            mx.kzg_commit_batch_device(d_rows.data_ptr(), rows_n, d_out.data_ptr(), stream)
            return None

        t_build = time.perf_counter()
        step()                     # first use builds the SRS window table (one-off, reported separately)
        torch.cuda.synchronize()
        build_s = time.perf_counter() - t_build
        el, kern, _ = timed(step, min_warm_s=leg_warm_s)
        # the reference's own call pattern through the real symbol (one row per compute_digest_from_srs call, 1 and 8 pool threads,
        # Server.hpp:550-560, 1054-1078): the plain-C harness in a child process, linked with -lmultiexp like the reference
        per_call = None
        harness = os.path.join(ROOT, "integration", "kzg_harness", "harness")
        if rank == 0 and os.path.exists(harness) and not args.no_cpu:
            per_call = {}
            for threads in (1, 8):
                try:
                    r = subprocess.run([harness, "bench", str(threads), "1500"], capture_output=True, text=True, timeout=180, env=child_env())
                    ln = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
                    d = json.loads(ln)
                    per_call["threads_%d" % threads] = {"commits_per_s": d["commits_per_s"], "latency_ms_per_call": d["latency_ms_per_call"],
                                                        "consistent": d["consistent"]}
                except Exception as e:  # noqa: BLE001
                    per_call["threads_%d" % threads] = {"error": repr(e)}
        # the same batch from pageable host rows through porla_kzg_commit_batch_host (INTEGRATION.md s3's call), PCIe in and out
        # included: never `value`
        host_rows = None
        if world == 1 and not args.no_host_boundary and not args.no_cpu:      # (--no-cpu: the profiler passes; only the batch kernel's launches)
            try:
                h_rows = bytes(d_rows.cpu().numpy())
                h_out = mx.kzg_commit_batch_host(h_rows, rows_n)
                t1 = time.perf_counter()
                for _ in range(3):
                    mx.kzg_commit_batch_host(h_rows, rows_n)
                h_ms = (time.perf_counter() - t1) / 3 * 1e3
                host_rows = {"ms_per_call": round(h_ms, 3), "commits_per_s": round(rows_n / h_ms * 1e3, 1),
                             "same_bytes_as_device_rows": h_out == bytes(d_out.cpu().numpy()),
                             "note": "pageable host rows in, 64-byte commitments out, 64 MiB chunks on two streams"}
                del h_rows, h_out
            except Exception as e:  # noqa: BLE001
                host_rows = {"error": repr(e)}
        cshape = mx.kzg_commit_shape()
        cpu = None
        ok = None
        if not args.no_cpu and rank == 0:
            sample = 2048
            rows = bytes(d_rows[:sample * 4096].cpu().numpy())
            cores = common.ncpu()
            t1 = time.perf_counter()
            want = common.oracle_commit_batch("bn254", rows, sample, 128, srs_raw, threads=cores)
            cpu_s = time.perf_counter() - t1
            ok = want == bytes(d_out[:sample * 64].cpu().numpy())
            cpu = {"value": round(sample / cpu_s, 1), "unit": "commits/s", "cores": cores, "kind": "port",
                   "sample": "the first %d rows, one 128-point bucket MSM per row (oracle/bn254_ref.c, CPU restatement of "
                             "compute_digest_from_srs, not gnark) over %d threads; %.2f s wall" % (sample, cores, cpu_s)}
        if host_rows and host_rows.get("same_bytes_as_device_rows") is False:
            ok = False                  # the device rows are checked against the oracle above; the host path must give the same bytes
        return line("KZG commits/s (128-coefficient rows against the resident SRS)", round(world * rows_n * args.steps / el, 1),
                    "commits/s", el, "weak", "u32x8 (256-bit modular integer)",
                    {"workload": "2^%d rows x 128 coefficients per GPU, compute_digest_from_srs hoisted over rows "
                                 "(fixed-base window table resident in HBM)" % args.log2rows,
                     "rows_per_gpu": rows_n, "sharding": "row range per rank, no collective" if world > 1 else "single GPU",
                     "table_build_s": round(build_s, 3)},
                    traffic_in_run(roofline(kern, COMMIT_BYTES_PER_ROW * rows_n, "kzg_commit", 10.0 * rows_n * 128 * cshape[1]),
                                   ["--workload", "kzg_commit", "--no-host-boundary", "--log2rows", str(args.log2rows)]), cpu, ok,
                    rows_per_gpu=rows_n, coefficients_per_row=128, per_call_compute_digest_from_srs=per_call, host_rows=host_rows,
                    table={"window_bits": cshape[0], "windows_per_coefficient": cshape[1],
                           "GiB": round(128 * cshape[1] * (1 << (cshape[0] - 1)) * 64 / 2**30, 2) if cshape[0] else None,
                           "budget": "PORLA_COMMIT_TABLE_GB (default: a fifth of the HBM)"},
                    equiv_Mmul_per_s=round(world * rows_n * 128 * args.steps / el / 1e6, 1), table_build_s=round(build_s, 3))

    # ---------------------------------------------------------------- BN254 MSM (headline)
    def leg_bn254_msm():
        n = 1 << args.log2n
        # ---- synthetic inputs (SURVEY.md s8d cfg 2): points k_i*G, scalars SHA-256 stream (81 % of them >= r)
        t0 = time.time()
        if rank == 0:
            common.cached_inputs(n)          # rank 0 generates (or finds) the cache; the others read it
        if world > 1:
            dist.barrier()
        sc0, pt = common.cached_inputs(n)
        sc = sc0 if rank == 0 else common.synth_scalars(n, start=rank * n)  # every rank: its own scalars, same base
        gen_s = time.time() - t0
        d_sc, d_pt = to_dev(sc), to_dev(pt)

        depth = max(1, min(3, args.in_flight))
        streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        state = {"k": 0, "inflight": []}

        def retire():
            # N == 1: 64-byte affine.  N > 1: this rank's partial Jacobian, ONE RCCL all_gather of N x 96 bytes, N-1 group
            # additions + one inversion on the host (porla_amd/sharded.py)
            slot = state["inflight"].pop(0)
            part = mx.msm_end(slot, partial=world > 1)
            return part if world == 1 else fold_across_ranks("bn254", part)

        def step():
            # every step is one complete 2^20-pair MSM (all kernels + host fold + result); with in_flight > 1 the next
            # MSM is enqueued on another stream before the oldest one is retired, so the latency-bound tail of one
            # (bucket reduction, host fold) overlaps the bucket accumulation of the next
            res = None
            if depth == 1:
                if world == 1:
                    return mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, stream)
                return fold_across_ranks("bn254", mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, stream, partial=True))
            if len(state["inflight"]) == depth:
                res = retire()
            slot = 1 + state["k"] % depth
            mx.msm_begin(slot, d_sc.data_ptr(), d_pt.data_ptr(), n, streams[state["k"] % depth].cuda_stream)
            state["inflight"].append(slot)
            state["k"] += 1
            return res

        def drain():
            res = None
            while state["inflight"]:
                res = retire()
            return res

        el, kern, result = timed(step, drain, min_warm_s=head_warm_s)
        warm_steps_run = timed.warm_steps
        fe_mults = msm_fe_mults(n)        # read now: the audit-size MSMs below run with another (window, GLV) shape
        # the same MSM as blocking calls (one in flight): what a caller that waits for every result sees, and ITS per-kernel
        # breakdown (the one above is taken with `depth` MSMs contending for the chip)
        blocking_ms = None
        blocking_kern = None
        if depth > 1:
            def blocking_step():
                return mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, stream, partial=world > 1)
            reps = max(3, args.steps // 4)
            for _ in range(2):
                blocking_step()
            sync()
            t_b = time.perf_counter()
            for _ in range(reps):
                blocking_step()
            torch.cuda.synchronize()
            blocking_ms = (time.perf_counter() - t_b) / reps * 1e3
            per_launch, per_step = breakdown(blocking_step)
            blocking_kern = {"ms_per_launch": per_launch, "ms_per_step": per_step,
                             "sum_ms_per_step": round(sum(per_step.values()), 4),
                             "note": "HIP events around every kernel of 3 blocking calls (each event pair adds ~10 us of idle "
                                     "GPU around its kernel; the call's wall time is blocking_ms_per_step)"}
        # ---- the sizes the reference really issues (n_points <= 3 200, Server.hpp:585-587; coefficients abs(int32), utils.h:271-275):
        # latency of one blocking call on device-resident inputs, each checked against the oracle
        audit = None
        if world == 1 and rank == 0 and not args.no_cpu:
            try:
                import random
                rnd = random.Random(5)
                a_sc = b"".join(mx.bn254_scalar_set_int(rnd.getrandbits(31)) for _ in range(3200))
                d_a = to_dev(a_sc)
                audit = {"what": "ms per blocking call on device-resident inputs: one MSM (ms) and the audit's pair of MSMs over one scalar array (pair_ms)", "abs_int32_coefficients": {}, "256_bit_scalars": {}}
                for label, host_sc, dev_sc in (("abs_int32_coefficients", a_sc, d_a), ("256_bit_scalars", sc, d_sc)):
                    for m in (128, 1408, 3200):
                        for _ in range(3):
                            r = mx.msm_device("bn254", dev_sc.data_ptr(), d_pt.data_ptr(), m, stream)
                        t1 = time.perf_counter()
                        for _ in range(20):
                            r = mx.msm_device("bn254", dev_sc.data_ptr(), d_pt.data_ptr(), m, stream)
                        ms = (time.perf_counter() - t1) / 20 * 1e3
                        audit[label][str(m)] = {"ms": round(ms, 4), "bit_exact_vs_oracle": r == common.oracle_msm(host_sc, pt, m)}
                        # the audit's PAIR (same coefficients over the commitments and over the alignment points, Server.hpp:900-901)
                        # as one call / one launch; the second point set: the same points, 64 further on
                        pb = d_pt.data_ptr() + 64 * 64
                        for _ in range(3):
                            r2 = mx.msm_pair_device("bn254", dev_sc.data_ptr(), d_pt.data_ptr(), pb, m, stream)
                        t1 = time.perf_counter()
                        for _ in range(20):
                            r2 = mx.msm_pair_device("bn254", dev_sc.data_ptr(), d_pt.data_ptr(), pb, m, stream)
                        ms2 = (time.perf_counter() - t1) / 20 * 1e3
                        audit[label][str(m)].update(pair_ms=round(ms2, 4), pair_bit_exact_vs_oracle=(
                            r2 == (r, common.oracle_msm(host_sc, pt[64 * 64:], m))))
            except Exception as e:  # noqa: BLE001
                audit = {"error": repr(e)}
        # the reference's own boundary: compute_multi_exp on caller-owned pageable HOST buffers (PCIe included; never `value`)
        host_boundary = None
        if world == 1 and rank == 0 and not args.no_host_boundary:
            # in a child process (its launches at other sizes stay out of this process's kernel statistics; a profiler
            # preloaded into this process is not handed down)
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_host_boundary.py"), "--json", str(args.log2n)],
                                   capture_output=True, text=True, timeout=600, env=child_env())
                host_boundary = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                host_boundary["same_result"] = host_boundary.pop("result") == result.hex()
            except Exception as e:  # noqa: BLE001
                host_boundary = {"error": repr(e)}
        cpu = None
        verified = None
        if rank == 0:
            if not args.no_cpu and world == 1:
                cores = common.ncpu()
                t1 = time.perf_counter()
                want = common.oracle_msm(sc, pt, n, threads=cores)
                cpu_s = time.perf_counter() - t1
                verified = (want == result)
                cpu = {"value": round(n / cpu_s / 1e6, 4), "unit": "Mmul/s", "cores": cores, "kind": "port",
                       "per_thread": round(n / cpu_s / 1e6 / cores, 4),
                       "sample": "the same 2^%d pairs, oracle/bn254_ref.c bucket MSM (signed windows) range-split over %d threads "
                                 "(CPU restatement, not gnark); %.1f s wall" % (args.log2n, cores, cpu_s),
                       "reference_indicative": {"value": 0.143, "unit": "Mmul/s per thread",
                                                "what": "the reference's OWN CPU path where it could be built: vendored libsecp256k1 "
                                                        "ecmult_multi_var, 2^18 points, one thread of the survey container (Xeon 2.1 GHz) "
                                                        "-- BASELINE.md s2; a different machine and the other curve, quoted for scale only"}}
            if not args.no_cpu and world > 1:
                # whole-job check at N > 1 (no cpu_baseline is reported there): the N * 2^20-pair MSM over every rank's
                # scalars (rank g: SHA-256 stream starting at g * 2^20) against the oracle
                all_sc = sc + b"".join(common.synth_scalars(n, start=g * n) for g in range(1, world))
                verified = common.oracle_msm(all_sc, pt * world, world * n, threads=common.ncpu()) == result
        rl_head = roofline(kern, MSM_BYTES_PER_PAIR * n, "bn254_msm", fe_mults)
        # the counter traffic of the dominant kernel from THIS run on THIS box (two short child runs under rocprofv3 --pmc, after the
        # timed region); the committed passes stay beside it for comparison
        rl_head = traffic_in_run(rl_head, ["--workload", "bn254_msm", "--no-legs", "--no-commits", "--no-host-boundary", "--log2n", str(args.log2n)])
        return line("BN254 G1 MSM Mscalar-mul/s at 2^20 pts", round(world * n * args.steps / el / 1e6, 3), "Mmul/s", el, "weak",
                    "u32x8 (256-bit modular integer)",
                    {"workload": "KZG scheme, single 2^%d-point BN254 G1 MSM per GPU, inputs resident in HBM, "
                                 "output 64-B affine point" % args.log2n,
                     "pairs_per_gpu": n, "msm_in_flight": depth, "warmup_steps_run": warm_steps_run, "min_warmup_s": head_warm_s,
                     "sharding": "input-pair range per rank + all-gather of 96-B Jacobian partials, folded on every host"
                     if world > 1 else "single GPU", "collective": collective, "input_gen_s": round(gen_s, 1)},
                    rl_head, cpu, verified,
                    result=result.hex() if result else None,
                    blocking_ms_per_step=round(blocking_ms, 4) if blocking_ms else None,
                    blocking_Mmul_s=round(world * n / blocking_ms / 1e3, 1) if blocking_ms else None,
                    blocking_kernels_ms=blocking_kern, host_boundary=host_boundary, audit_size_msm=audit)

    # ---------------------------------------------------------------- strong scaling at the metric's own size (N > 1)
    def leg_strong_2p20():
        """ONE 2^20-pair BN254 MSM over all ranks: 2^20 / N pairs per rank (the reference's own pattern: one MSM range-split over 8
        workers, Client.hpp:761-787), the same gather + fold: the honest test of north_star's '>= 6x at 8 GPUs'.
        The per-rank fixed cost (conversion, sort, reduction tree, host fold) does not shrink with the range, so this is expected to
        scale far below N: the line carries the number."""
        total = 1 << args.log2n
        lo, hi = mx.shard_range(total, rank, world)
        n_local = hi - lo
        sc, pt = common.cached_inputs(total)
        d_sc, d_pt = to_dev(sc[32 * lo:32 * hi]), to_dev(pt[64 * lo:64 * hi])

        # the headline's call pattern (two complete MSMs in flight on two streams through the two-phase API; every step still
        # produces and folds its own result) AND blocking calls: a rank's fixed latency (sort, tree, gather, host fold) is what
        # limits this leg, and the pipelined form hides the part of it that another MSM's accumulation can run under.  THREE in flight
        # at every N (the library's three begin / end slots): at 2^17 pairs per rank a step takes 0.56 / 0.40 / 0.34 ms with 1 / 2 / 3
        # in flight (same box), at 2^20 three gain 1.3 % over two
        depth = 1 if args.in_flight == 1 else 3
        streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        state = {"k": 0, "inflight": []}

        def retire():
            part = mx.msm_end(state["inflight"].pop(0), partial=world > 1)
            return part if world == 1 else fold_across_ranks("bn254", part)

        def step():
            res = None
            if len(state["inflight"]) == depth:
                res = retire()
            slot = 1 + state["k"] % depth
            mx.msm_begin(slot, d_sc.data_ptr(), d_pt.data_ptr(), n_local, streams[state["k"] % depth].cuda_stream)
            state["inflight"].append(slot)
            state["k"] += 1
            return res

        def drain():
            res = None
            while state["inflight"]:
                res = retire()
            return res

        def blocking_step():
            if world == 1:
                return mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n_local, stream)
            return fold_across_ranks("bn254", mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n_local, stream, partial=True))

        el, kern, result = timed(step, drain, min_warm_s=leg_warm_s)
        fe_mults = msm_fe_mults(n_local)
        # blocking calls: the same number of steps on every rank (each one contains the collective)
        for _ in range(2):
            blocking_step()
        sync()
        t_b = time.perf_counter()
        for _ in range(args.steps):
            r_b = blocking_step()
        sync()
        blocking_el = time.perf_counter() - t_b
        if world > 1:
            t = torch.tensor([blocking_el], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            blocking_el = float(t.item())
        verified = None
        if rank == 0 and not args.no_cpu:
            verified = common.oracle_msm(sc, pt, total, threads=common.ncpu()) == result == r_b
        rl = roofline(kern, MSM_BYTES_PER_PAIR * n_local, "strong_2p20", fe_mults)
        if rl:
            rl["traffic"] = None                      # the committed counter pass is of a 2^20-pair launch, not of this range size
            rl["traffic_source"] = "not collected at this range size"
        return line("BN254 G1 MSM Mscalar-mul/s, ONE 2^%d-pair MSM over all GPUs" % args.log2n,
                    round(total * args.steps / el / 1e6, 3), "Mmul/s", el, "strong", "u32x8 (256-bit modular integer)",
                    {"workload": "ONE 2^%d-pair BN254 G1 MSM, pair range [g n / N, (g+1) n / N) on GPU g, partials all-gathered and "
                                 "folded on every host; %d MSMs in flight per rank" % (args.log2n, depth),
                     "pairs_total": total, "pairs_per_gpu": n_local, "msm_in_flight": depth, "collective": collective},
                    rl, None, verified, result=result.hex() if result else None,
                    blocking_ms_per_step=round(blocking_el / args.steps * 1e3, 4),
                    blocking_Mmul_s=round(total * args.steps / blocking_el / 1e6, 1))

    # ---------------------------------------------------------------- config 3: ONE 2^24-pair MSM over all ranks
    def leg_config3():
        import numpy as np
        total = 1 << args.log2job
        blk = 1 << 20
        base_n = min(total, blk)
        lo, hi = mx.shard_range(total, rank, world)         # this rank's pair range of the job (porla_shard_range)
        n_local = hi - lo
        t0 = time.time()
        if rank == 0:
            common.cached_inputs(base_n)
        if world > 1:
            dist.barrier()
        _, pt = common.cached_inputs(base_n)                # the job's points: the 2^20 base points, repeated

        def job_scalars(a, b):
            # pair i of the job: 32 bytes of PCG64(seed 3 + i // 2^20) -- one generator per 2^20-pair block, so any range of
            # the job can be produced without the rest (16 M SHA-256 calls from Python would take longer than the benchmark)
            out = []
            for k in range(a // blk, (b + blk - 1) // blk):
                raw = np.random.Generator(np.random.PCG64(3 + k)).bytes(32 * blk)
                s, e = max(a, k * blk) - k * blk, min(b, (k + 1) * blk) - k * blk
                out.append(raw[32 * s:32 * e])
            return b"".join(out)

        def job_points(a, b):
            out = []
            i = a
            while i < b:
                s = i % base_n
                take = min(base_n - s, b - i)
                out.append(pt[64 * s:64 * (s + take)])
                i += take
            return b"".join(out)

        sc = job_scalars(lo, hi)
        pts = job_points(lo, hi)
        gen_s = time.time() - t0
        d_sc, d_pt = to_dev(sc), to_dev(pts)
        del sc, pts

        def step():
            if world == 1:
                return mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n_local, stream)
            return fold_across_ranks("bn254", mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n_local, stream, partial=True))

        el, kern, result = timed(step, min_warm_s=leg_warm_s)
        launches = max(1, (n_local + (1 << 22) - 1) >> 22)    # an input above 2^22 pairs runs as ranges of 2^22 into one bucket set
        fe_mults = msm_fe_mults(n_local)
        cpu = None
        verified = None
        if rank == 0 and not args.no_cpu:
            cores = common.ncpu()
            all_sc = job_scalars(0, total)
            all_pt = job_points(0, total)
            t1 = time.perf_counter()
            want = common.oracle_msm(all_sc, all_pt, total, threads=cores)
            cpu_s = time.perf_counter() - t1
            del all_sc, all_pt
            verified = want == result
            cpu = {"value": round(total / cpu_s / 1e6, 4), "unit": "Mmul/s", "cores": cores, "kind": "port",
                   "sample": "the whole 2^%d-pair job, oracle/bn254_ref.c bucket MSM range-split over %d threads (CPU restatement, "
                             "not gnark); %.1f s wall" % (args.log2job, cores, cpu_s)}
        rl3 = roofline(kern, MSM_BYTES_PER_PAIR * n_local / launches, "config3", fe_mults / launches)
        if rl3:
            # the committed counter pass is of a 2^20-pair launch; this leg's launches cover up to 2^22 pairs each: no figure
            # rather than one from a launch of another size
            rl3["traffic"] = None
            rl3["traffic_source"] = "not collected at this leg's range size (profiles/pmc_latest.json is a 2^20-pair launch)"
            # ... unless this run can collect it itself (N = 1: two short child runs of this leg under rocprofv3 --pmc)
            rl3 = traffic_in_run(rl3, ["--workload", "config3", "--log2job", str(args.log2job)])
        return line("BN254 G1 MSM Mscalar-mul/s, one 2^%d-pair job over all GPUs" % args.log2job,
                    round(total * args.steps / el / 1e6, 3), "Mmul/s", el, "strong", "u32x8 (256-bit modular integer)",
                    {"workload": "KZG audit over 2^%d blocks: ONE 2^%d-pair BN254 G1 MSM, pair range [g 2^%d / N, (g+1) 2^%d / N) on "
                                 "GPU g, inputs resident in HBM, 96-B Jacobian partials all-gathered and folded on every host"
                                 % (args.log2job, args.log2job, args.log2job, args.log2job),
                     "pairs_total": total, "pairs_per_gpu": n_local, "collective": collective, "input_gen_s": round(gen_s, 1),
                     "accumulation_launches_per_step": launches},
                    rl3, cpu, verified, result=result.hex() if result else None)

    # ---------------------------------------------------------------- secp256k1 MSM (config 4)
    def leg_secp256k1_msm():
        n = 1 << args.log2n
        t0 = time.time()
        path = os.path.join(os.environ.get("PORLA_CACHE", "/tmp"), "porla_secp_points_%d.bin" % n)
        if rank == 0 and not (os.path.exists(path) and os.path.getsize(path) == 64 * n):
            with open(path + ".tmp", "wb") as f:
                f.write(common.secp_bench_points(n))       # P_i = 2^i * G (bench_ecmult.c:328-337)
            os.replace(path + ".tmp", path)
        if world > 1:
            dist.barrier()
        pt = open(path, "rb").read()
        sc = common.secp_bench_scalars(n, start=rank * n)   # SHA-256("ecmult" || LE32(i)) (bench_ecmult.c:233-247)
        gen_s = time.time() - t0
        d_sc, d_pt = to_dev(sc), to_dev(pt)

        def step():
            if world == 1:
                return mx.msm_device("secp256k1", d_sc.data_ptr(), d_pt.data_ptr(), n, stream)
            return fold_across_ranks("secp256k1", mx.msm_device("secp256k1", d_sc.data_ptr(), d_pt.data_ptr(), n, stream, partial=True))

        el, kern, result = timed(step, min_warm_s=leg_warm_s)
        fe_mults = msm_fe_mults(n)
        cpu = None
        verified = None
        if rank == 0 and not args.no_cpu and world == 1:
            cores = common.ncpu()
            t1 = time.perf_counter()
            want = common.oracle_secp_msm(sc, pt, n, threads=cores)
            cpu_s = time.perf_counter() - t1
            closed = common.secp_bench_expected(sc, n)  # (sum s_i 2^i) * G, bench teardown (bench_ecmult.c:258-270)
            verified = (want == result) and (closed == result)
            cpu = {"value": round(n / cpu_s / 1e6, 4), "unit": "Mmul/s", "cores": cores, "kind": "port",
                   "sample": "the same 2^%d pairs, oracle/secp256k1_ref.c bucket MSM over %d threads (CPU restatement, "
                             "not libsecp256k1); %.1f s wall" % (args.log2n, cores, cpu_s)}
        # the IPA build's own sizes (NUM_CHECK_AUDIT * height = 1 408 at 2^10 blocks, 3 200 at 2^24; abs(int32) coefficients,
        # Server.hpp:838-848): one blocking MSM and the audit's pair over one scalar array, device-resident, against the oracle
        ipa_sizes = None
        if rank == 0 and world == 1 and not args.no_cpu:
            try:
                import random
                rnd = random.Random(7)
                a_sc = b"".join(rnd.getrandbits(31).to_bytes(32, "big") for _ in range(3200))
                d_a = to_dev(a_sc)
                ipa_sizes = {}
                for m in (1408, 3200):
                    pb = d_pt.data_ptr() + 64 * 64
                    for _ in range(3):
                        r1 = mx.msm_device("secp256k1", d_a.data_ptr(), d_pt.data_ptr(), m, stream)
                        r2 = mx.msm_pair_device("secp256k1", d_a.data_ptr(), d_pt.data_ptr(), pb, m, stream)
                    t1 = time.perf_counter()
                    for _ in range(20):
                        r1 = mx.msm_device("secp256k1", d_a.data_ptr(), d_pt.data_ptr(), m, stream)
                    ms1 = (time.perf_counter() - t1) / 20 * 1e3
                    t1 = time.perf_counter()
                    for _ in range(20):
                        r2 = mx.msm_pair_device("secp256k1", d_a.data_ptr(), d_pt.data_ptr(), pb, m, stream)
                    ms2 = (time.perf_counter() - t1) / 20 * 1e3
                    ok = r1 == common.oracle_secp_msm(a_sc, pt, m) and r2 == (r1, common.oracle_secp_msm(a_sc, pt[64 * 64:], m))
                    ipa_sizes[str(m)] = {"ms": round(ms1, 4), "pair_ms": round(ms2, 4), "bit_exact_vs_oracle": ok}
            except Exception as e:  # noqa: BLE001
                ipa_sizes = {"error": repr(e)}
        return line("secp256k1 MSM Mscalar-mul/s at 2^20 pts", round(world * n * args.steps / el / 1e6, 3), "Mmul/s", el, "weak",
                    "u32x8 (256-bit modular integer)",
                    {"workload": "IPA scheme, 2^%d-point secp256k1 ecmult_multi per GPU (points 2^i*G, scalars "
                                 "SHA-256(\"ecmult\"||i) as bench_ecmult.c), blocking calls, inputs resident in HBM" % args.log2n,
                     "pairs_per_gpu": n, "input_gen_s": round(gen_s, 1)},
                    traffic_in_run(roofline(kern, MSM_BYTES_PER_PAIR * n, "secp256k1_msm", fe_mults),
                                   ["--workload", "secp256k1_msm", "--log2n", str(args.log2n)]), cpu, verified,
                    result=result.hex() if result else None, audit_size_msm=ipa_sizes)

    # ---------------------------------------------------------------- ICC encode (config 5)
    def leg_icc():
        from porla_amd import icc
        n_rows, n_cols = 1 << (args.log2rows - 2), 128
        g = torch.Generator(device=dev).manual_seed(99 + rank)
        d_in = torch.randint(0, 256, (n_rows * n_cols * 32,), dtype=torch.uint8, device=dev, generator=g)
        d_al = torch.empty(32 * n_rows * n_cols, dtype=torch.uint8, device=dev)
        d_sc = torch.empty(32 * n_rows * n_cols, dtype=torch.uint8, device=dev)

        def step():
            icc.crebuild_device(d_in.data_ptr(), n_rows, n_cols, "bn254", 0, 0, 0, d_al.data_ptr(), d_sc.data_ptr(), stream=stream)
            return None

        el, kern, _ = timed(step, min_warm_s=leg_warm_s)
        total_ms = sum(timed.totals.values())
        cpu = None
        verified = None
        if rank == 0 and not args.no_cpu and world == 1:
            sample_rows = min(n_rows, 1 << 11)
            rows = bytes(d_in[:sample_rows * n_cols * 32].cpu().numpy())
            d_s_al = torch.empty(32 * sample_rows * n_cols, dtype=torch.uint8, device=dev)
            d_s_sc = torch.empty(32 * sample_rows * n_cols, dtype=torch.uint8, device=dev)
            icc.crebuild_device(d_in.data_ptr(), sample_rows, n_cols, "bn254", 0, 0, 0, d_s_al.data_ptr(), d_s_sc.data_ptr(),
                                stream=stream)
            torch.cuda.synchronize()
            L = common.oracle()
            x = ctypes.create_string_buffer(64 * sample_rows * n_cols)
            al = ctypes.create_string_buffer(32 * sample_rows * n_cols)
            scb = ctypes.create_string_buffer(32 * sample_rows * n_cols)
            cores = common.ncpu()
            t1 = time.perf_counter()
            L.oracle_icc_crebuild(rows, ctypes.c_size_t(sample_rows), ctypes.c_size_t(n_cols), 0, 0, ctypes.c_uint64(0),
                                  x, al, scb, cores)
            cpu_s = time.perf_counter() - t1
            verified = al.raw == bytes(d_s_al.cpu().numpy()) and scb.raw == bytes(d_s_sc.cpu().numpy())
            cpu = {"value": round(sample_rows * n_cols / cpu_s / 1e6, 4), "unit": "Melements/s", "cores": cores,
                   "kind": "port", "sample": "a %d-row x 128-column encode (oracle/icc_ref.c, CPU restatement of "
                   "CRebuild_Cached + align_MAC scalars in Z/LCM, not NTL) over %d threads; %.2f s wall"
                   % (sample_rows, cores, cpu_s)}
        passes = max(1, ((n_rows.bit_length() - 1) + 8) // 9)      # LDS-fused passes per encode (<= 9 stages each) = launches of the dominant kernel
        rl = traffic_in_run(roofline(kern, ICC_BYTES_PER_ELEMENT * n_rows * n_cols / passes, "icc"),
                            ["--workload", "icc", "--log2rows", str(args.log2rows)])
        if rl:
            rl["note"] = ("an encode is %d launches of the dominant kernel; `achieved` prices the encode's algorithmic bytes / %d "
                          "per launch, `traffic` is the per-launch average of the counters; the kernel is bound by VALU issue "
                          "(profiles/r03_b_icc_stall_counters.txt)" % (passes, passes))
            rl["launches_per_encode"] = {k: round(timed.totals[k] / v) for k, v in kern.items() if v > 0}
            rl["whole_encode_kernels_ms"] = round(total_ms, 4)
            rl["whole_encode_achieved_GBps"] = round(ICC_BYTES_PER_ELEMENT * n_rows * n_cols / (total_ms * 1e-3) / 1e9, 2)
        return line("ICC encode Melements/s (2^22 Fp elements)", round(world * n_rows * n_cols * args.steps / el / 1e6, 3),
                    "Melements/s", el, "weak", "u32x9 residue pair (mod p_icc, mod q), 30-bit limbs",
                    {"workload": "ICC encode (CRebuild_Cached X part + align_MAC scalars), %d rows x 128 columns per GPU, "
                                 "rows resident in HBM" % n_rows, "rows_per_gpu": n_rows, "columns": n_cols},
                    rl, cpu, verified)

    # ---------------------------------------------------------------- audit row combine: the path's HBM-bound kernel
    def leg_audit_combine():
        import importlib.util
        spec = importlib.util.spec_from_file_location("bench_audit", os.path.join(ROOT, "tools", "bench_audit.py"))
        ba = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ba)
        # an 8 GiB level store (2^20 rows of 128 x 64-byte symbols: no row is read twice from a cache), challenges of the audit's
        # own size (3 200 rows) and of 2^18 rows (2 GiB gathered: the bandwidth regime)
        res, cpu = ba.measure(20, [3200, 1 << 18], mixed=False, reps_small=100)
        small, big = res
        rl = dict(big["roofline"])
        # the committed PMC passes ran this very leg; the 2^18-row launches are the only ones of the 8-slice instantiation
        rl["traffic"] = pmc_traffic("audit_accumulate", "audit_combine")
        rl["note"] = ("algorithmic bytes = 8 192 per challenged row (+ indices, coefficients, 64 B per column out) / the accumulation "
                      "kernel's HIP-event time; random 8-KiB rows of an 8 GiB store")
        return {"metric": "audit row combine GB/s (2^18 challenged rows x 128 symbols of 64 B)", "value": rl["achieved"], "unit": "GB/s",
                "n_gpus": world, "steps": 20, "warmup": 5, "ms_per_step": big["ms_per_call_back_to_back"], "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs, exact 608-bit integer accumulation", "data": "synthetic",
                "config": {"workload": "Server::audit row combine + align_MAC scalar part (Server.hpp:790-828, 531-541), level store "
                                       "resident in HBM", "store_rows": 1 << 20, "challenged_rows": 1 << 18},
                "kernels_ms": big["kernels_ms"], "roofline": rl, "cpu_baseline": cpu,
                "audit_size": {"challenged_rows": 3200, "ms_per_call": small["ms_per_call_back_to_back"], "kernels_ms": small["kernels_ms"],
                               "achieved_GBps": small["roofline"]["achieved"]},
                "bit_exact_vs_oracle": bool(big["bit_exact_vs_oracle_1024_row_challenge"])}

    # ---------------------------------------------------------------- one whole audit (child process: tools/bench_audit_flow.py)
    def leg_kzg_audit():
        cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_audit_flow.py")] + (["--no-cpu"] if args.no_cpu else [])
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=child_env())
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            raise RuntimeError("bench_audit_flow.py failed: rc=%d %s" % (r.returncode, r.stderr[-400:]))
        return json.loads(lines[-1])

    # ---------------------------------------------------------------- the client's block MACs in batches
    def leg_client_mac_batch():
        kzg_setup()
        o = common.oracle()
        n = 1 << 17
        gen = torch.Generator(device="cuda").manual_seed(11)
        d_rows = torch.randint(0, 256, (n, 4096), dtype=torch.uint8, device="cuda", generator=gen)
        d_sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=gen)
        d_sc[:, :16] = 0                                            # 16-byte PRF outputs (Client.hpp:424-455), left-padded
        d_out = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        calls = {"digest_batch": lambda: mx.kzg_digest_batch_device(d_rows.data_ptr(), n, d_out.data_ptr(), s),
                 "complement_batch": lambda: mx.kzg_complement_batch_device(d_sc.data_ptr(), n, d_out.data_ptr(), s),
                 "mac_batch": lambda: mx.kzg_mac_batch_device(d_rows.data_ptr(), d_sc.data_ptr(), n, d_out.data_ptr(), s)}
        ms = {}
        for name, fn in calls.items():
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            ms[name] = (time.perf_counter() - t0) / 20 * 1e3
        got = bytes(d_out.cpu().numpy())                            # the MAC batch ran last
        mx.profile_enable(True)
        for _ in range(5):
            calls["mac_batch"]()
        torch.cuda.synchronize()
        prof = {k: round(t / c, 4) for k, t, c in mx.profile_get()}
        mx.profile_enable(False)
        verified = None
        if not args.no_cpu:
            # MAC = compute_digest(block) + PRF * h_MAC on the oracle's arithmetic; the hiding base is this run's (drawn by init_SRS)
            h_mac = mx.compute_digest_complement((1).to_bytes(16, "big"))
            verified = True
            for r in (0, 1, n - 1):
                want = ctypes.create_string_buffer(64)
                o.oracle_kzg_compute_digest(bytes(d_rows[r].cpu().numpy()), want)
                comp = ctypes.create_string_buffer(h_mac, 64)
                o.oracle_bn254_mult_point(comp, bytes(d_sc[r].cpu().numpy()))
                o.oracle_bn254_add_point(want, comp.raw)
                verified = verified and got[64 * r:64 * r + 64] == want.raw
        alg = n * (4096 + 32 + 64)
        ev = prof.get("kzg_eval_rows")
        return {"metric": "client block MACs/s (digest + complement + add_point per block, 2^17 blocks of 128 coefficients)",
                "value": round(n / ms["mac_batch"] * 1e3, 1), "unit": "blocks/s", "n_gpus": world, "steps": 20, "warmup": 5,
                "ms_per_step": round(ms["mac_batch"], 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "data": "synthetic",
                "dtype": "u32 limbs (29-bit columns, exact; 256-bit modular)",
                "config": {"workload": "Client::initialize's block MACs (Client.hpp:216-236, 408-455) as one batch on blocks resident "
                                       "in HBM", "blocks": n},
                "separate_batches_ms": {"digest_batch": round(ms["digest_batch"], 4), "complement_batch": round(ms["complement_batch"], 4),
                                        "note": "plus one host add_point per block (3.7 us each) without the MAC batch"},
                "kernels_ms": prof,
                "roofline": {"bound": "hbm", "kernel": "k_kzg_eval_rows_lazy", "achieved": round(n * 4128 / ev / 1e6, 1) if ev else None,
                             "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(n * 4128 / ev / 1e6 / HBM_PEAK_GBPS, 4) if ev else None,
                             "traffic": pmc_traffic("kzg_eval_rows", "client_mac_batch"),
                             "traffic_source": "committed rocprofv3 --pmc passes of this leg (profiles/pmc_latest_client_mac_batch.json)",
                             "algorithmic_bytes_per_launch": n * 4128, "algorithmic_bytes_per_batch": alg},
                "bit_exact_vs_oracle": verified}

    # ---------------------------------------------------------------- the IPA build's side of the path (N = 1 only)
    def wall_and_kernels(call, reps=20, warm=5):
        """ms per call back to back + HIP-event ms per launch / per call of every kernel of `call` (5 extra untimed calls)"""
        for _ in range(warm):
            call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        mx.profile_enable(True)
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        prof = mx.profile_get()
        mx.profile_enable(False)
        return ms, {k: round(t / max(c, 1), 4) for k, t, c in prof}, {k: round(t / 5, 4) for k, t, c in prof}

    def hbm_roofline(per_launch, per_call, algo_bytes_per_call, note, workload=None, fe_mults_per_launch=None, field=None, curve=None):
        if not per_launch:
            return None
        dom = max(per_call, key=per_call.get)
        traffic = pmc_traffic(dom, workload, curve) if workload else None
        launches = max(1, round(per_call[dom] / per_launch[dom])) if per_launch[dom] else 1
        ach = algo_bytes_per_call / launches / (per_launch[dom] * 1e-3) / 1e9 if per_launch[dom] else None
        r = {"bound": "hbm", "kernel": KERNEL_SYMBOL.get(dom, dom), "achieved": round(ach, 3) if ach else None, "peak": HBM_PEAK_GBPS,
             "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 6) if ach else None, "traffic": traffic,
             "traffic_source": ("committed rocprofv3 --pmc passes (profiles/pmc_latest_%s.json), per launch of the kernel, not "
                                "collected in this run" % workload) if traffic else "no counter pass committed for this leg",
             "kernel_ms": per_launch[dom],
             "launches_per_call": launches, "algorithmic_bytes_per_call": algo_bytes_per_call,
             "all_kernels_ms_per_call": per_call, "note": note}
        if fe_mults_per_launch and fe_peak and field and per_launch[dom]:
            # the mandated fraction prices HBM; what bounds these kernels is the integer multiplier: the same supplement as the MSM legs
            g = fe_mults_per_launch / (per_launch[dom] * 1e-3) / 1e9
            pk = fe_peak[field]
            r["int_multiplier"] = {"achieved": round(g, 2), "peak": pk["mix_8M_2S"], "unit": "G fe_mul/s (256-bit modular)",
                                   "frac": round(g / pk["mix_8M_2S"], 4), "peak_source": pk["source"]}
        return r

    SECP_G = bytes.fromhex("79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798"
                           "483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8")      # group_impl.h:28-33

    def secp_multiples_of_g(n, start):
        """n secp256k1 points k_i * G, k_i = SHA-256("ecmult" || LE32(start + i)): inputs for the IPA-side legs, produced by the
        ENGINE (rows of one coefficient against the one-point base [G]; tests/test_reference_kats_gpu.py ties that kernel to the
        reference's own hash) -- no oracle call for input generation"""
        fbg = mx.FixedBase("secp256k1", SECP_G, 1)
        try:
            return fbg.commit_host(common.secp_bench_scalars(n, start=start), n, 1)
        finally:
            fbg.close()

    def ipa_generators(n):
        # the 128 Pedersen generators of the IPA build (Client.hpp:112-117 draws them at random)
        return secp_multiples_of_g(n, 77000)

    def leg_ipa_commits():
        rows_n = 1 << args.log2rows
        gens = ipa_generators(128)
        fb = mx.FixedBase("secp256k1", gens, 128)
        try:
            g = torch.Generator(device=dev).manual_seed(4321)
            d_rows = torch.randint(0, 256, (rows_n * 4096,), dtype=torch.uint8, device=dev, generator=g)
            d_out = torch.empty(rows_n * 64, dtype=torch.uint8, device=dev)
            t_b = time.perf_counter()
            fb.commit_device(d_rows.data_ptr(), rows_n, 128, d_out.data_ptr(), stream)
            torch.cuda.synchronize()
            first_s = time.perf_counter() - t_b
            ms, per_launch, per_call = wall_and_kernels(lambda: fb.commit_device(d_rows.data_ptr(), rows_n, 128, d_out.data_ptr(), stream),
                                                        reps=max(3, min(args.steps, 20)), warm=max(1, min(args.warmup, 5)))
            info = fb.info()
            cpu = None
            ok = None
            if not args.no_cpu:
                sample = 2048
                rows = bytes(d_rows[:sample * 4096].cpu().numpy())
                cores = common.ncpu()
                t1 = time.perf_counter()
                want = common.oracle_commit_batch("secp256k1", rows, sample, 128, gens, threads=cores)
                cpu_s = time.perf_counter() - t1
                ok = want == bytes(d_out[:sample * 64].cpu().numpy())
                cpu = {"value": round(sample / cpu_s, 1), "unit": "commits/s", "cores": cores, "kind": "port",
                       "sample": "the first %d rows, one 128-point bucket MSM per row (oracle/secp256k1_ref.c, CPU restatement of "
                                 "compute_commitment, not libsecp256k1) over %d threads; %.2f s wall" % (sample, cores, cpu_s)}
            rl = hbm_roofline(per_launch, per_call, COMMIT_BYTES_PER_ROW * rows_n,
                              "algorithmic bytes = 4 096 B of coefficients + 64 B out per row (generator table resident); the kernel "
                              "is bound by the integer multiplier, as k_fb_commit of the KZG leg", workload="ipa_commits",
                              fe_mults_per_launch=10.0 * rows_n * 128 * info["windows"],
                              field="secp256k1")
            return {"metric": "IPA Pedersen commitments/s (128-coefficient rows against the 128 secp256k1 generators)",
                    "value": round(rows_n / ms * 1e3, 1), "unit": "commits/s", "ms_per_step": round(ms, 4), "scaling": "weak",
                    "dtype": "u32x9 (30-bit limbs, 256-bit modular integer)",
                    "config": {"workload": "Client::compute_commitment (Client.hpp:374-406) hoisted over 2^%d blocks: rows of 128 "
                                           "coefficients resident in HBM against the fixed generators (window table resident)" % args.log2rows,
                               "rows_per_gpu": rows_n, "table": info, "table_build_and_first_call_s": round(first_s, 3)},
                    "roofline": rl, "cpu_baseline": cpu, "bit_exact_vs_oracle": ok,
                    "equiv_Mmul_per_s": round(rows_n * 128 / ms / 1e3, 1)}
        finally:
            fb.close()

    def leg_mac_encode():
        from porla_amd import icc
        n = 1 << 15
        out = {"metric": "MAC-side ICC encode, M point butterflies/s (2^15 MACs)", "unit": "Mbutterflies/s", "scaling": "weak",
               "dtype": "u32x9 (30-bit limbs, 256-bit modular integer)",
               "config": {"workload": "the MAC halves of CRebuild_Cached (Server.hpp:1523-1536, 1590-1609, 1658-1676): 2^15 MACs "
                                      "resident in HBM through the radix-2 network in the exponent, X part; both curves", "macs": n},
               "curves": {}}
        all_ok = None
        for curve in ("bn254", "secp256k1"):
            if curve == "bn254":
                base = common.synth_points(4096, start=9000)
            else:
                base = secp_multiples_of_g(4096, 300)
            macs = (base * (n // 4096))[:64 * n]
            d_in = to_dev(macs)
            d_o = torch.empty(64 * n, dtype=torch.uint8, device=dev)
            ms, per_launch, per_call = wall_and_kernels(lambda: icc.mac_crebuild_device(d_in.data_ptr(), n, curve, 0, 0, d_o.data_ptr(), stream),
                                                        reps=10, warm=3)
            bfly = (n // 2) * 15
            cpu = None
            ok = None
            if not args.no_cpu:
                cores = common.ncpu()
                want = ctypes.create_string_buffer(64 * n)
                t1 = time.perf_counter()
                common.oracle().oracle_icc_mac_crebuild(macs, ctypes.c_size_t(n), 0 if curve == "bn254" else 1, 0, ctypes.c_uint64(0),
                                                        want, cores)
                cpu_s = time.perf_counter() - t1
                ok = want.raw == bytes(d_o.cpu().numpy())
                all_ok = ok if all_ok is None else (all_ok and ok)
                cpu = {"value": round(bfly / cpu_s / 1e6, 4), "unit": "Mbutterflies/s", "cores": cores, "kind": "port",
                       "sample": "the same 2^15-MAC encode, oracle/mac_ref.c (one scalar multiplication + two additions per butterfly, "
                                 "CPU restatement, not gnark / libsecp256k1) over %d threads; %.1f s wall" % (cores, cpu_s)}
            out["curves"][curve] = {"value": round(bfly / ms / 1e3, 3), "ms_per_step": round(ms, 4),
                                    "roofline": hbm_roofline(per_launch, per_call, 128 * n,
                                                             "algorithmic bytes = 64 B in + 64 B out per MAC; the network is a chain of "
                                                             "dependent group operations bound by the integer multiplier",
                                                             workload="mac_encode", curve=curve,
                                                             fe_mults_per_launch=(n // 2) * MAC_FE_MULTS_PER_BUTTERFLY, field=curve),
                                    "cpu_baseline": cpu, "bit_exact_vs_oracle": ok}
        for cv in out["curves"].values():
            if cv.get("roofline"):
                cv["roofline"]["touched_bytes_per_launch"] = 256 * n      # a stage reads and writes every 128-byte work point once
        out["value"] = out["curves"]["bn254"]["value"]
        out["ms_per_step"] = out["curves"]["bn254"]["ms_per_step"]
        out["roofline"] = out["curves"]["bn254"]["roofline"]
        out["cpu_baseline"] = out["curves"]["bn254"]["cpu_baseline"]
        out["bit_exact_vs_oracle"] = all_ok
        return out

    def leg_server_mix():
        from porla_amd import icc
        lib_ = __import__("porla_amd.loader", fromlist=["lib"]).lib
        length, n_cols, n_total = 1 << 12, 128, 1 << 17
        vp = ctypes.c_void_p
        out = {"metric": "Server::mix G data symbols/s (two blocks of 2^12 rows x 128 symbols + their MACs and alignments)", "unit": "Gsymbols/s",
               "scaling": "weak", "dtype": "u32x9 residue pair (data) / u32x9 30-bit limbs (points)",
               "config": {"workload": "Server::mix(is_x, level) (Server.hpp:1209-1328) in one call: data rows, MAC commitments and MAC "
                                      "alignments of two 2^12-row blocks resident in HBM", "rows_per_block": length, "columns": n_cols},
               "curves": {}}
        all_ok = None
        for curve in ("bn254", "secp256k1"):
            g = torch.Generator(device=dev).manual_seed(17)
            blocks = []
            for _ in range(2):
                t = torch.randint(0, 256, (length, n_cols, 64), dtype=torch.uint8, device=dev, generator=g)
                t[:, :, 63] &= 0x1f                                # < 2^509 < LCM of either build
                blocks.append(t)
            if curve == "bn254":
                pool = common.synth_points(2048, start=5000)
            else:
                pool = secp_multiples_of_g(2048, 300)
            arr = (pool * (4 * length // 2048))[:64 * 4 * length]
            parts = [arr[64 * length * k:64 * length * (k + 1)] for k in range(4)]
            d_parts = [to_dev(x) for x in parts]
            o_data = torch.empty(2 * length * n_cols * 64, dtype=torch.uint8, device=dev)
            o_mac, o_al = (torch.empty(128 * length, dtype=torch.uint8, device=dev) for _ in range(2))

            def call():
                rc = lib_.porla_server_mix_device(vp(blocks[0].data_ptr()), vp(blocks[1].data_ptr()), *[vp(t.data_ptr()) for t in d_parts],
                                                  length, n_cols, n_total, icc.CURVE[curve], vp(o_data.data_ptr()), vp(o_mac.data_ptr()),
                                                  vp(o_al.data_ptr()), vp(stream))
                if rc:
                    raise RuntimeError("porla_server_mix_device rc=%d" % rc)
            ms, per_launch, per_call = wall_and_kernels(call, reps=20, warm=5)
            cpu = None
            ok = None
            if not args.no_cpu:
                cores = common.ncpu()
                a0, a1 = bytes(blocks[0].cpu().numpy()), bytes(blocks[1].cpu().numpy())
                want = ctypes.create_string_buffer(2 * length * n_cols * 64)
                t1 = time.perf_counter()
                common.oracle().oracle_icc_mix(a0, a1, ctypes.c_size_t(length), ctypes.c_size_t(n_cols), ctypes.c_size_t(n_total),
                                               icc.CURVE[curve], want)
                cpu_data_s = time.perf_counter() - t1
                ok = want.raw == bytes(o_data.cpu().numpy())
                t1 = time.perf_counter()
                for o_t, (p0, p1) in ((o_mac, parts[0:2]), (o_al, parts[2:4])):
                    w = ctypes.create_string_buffer(2 * length * 64)
                    common.oracle().oracle_icc_mac_mix(p0, p1, ctypes.c_size_t(length), ctypes.c_size_t(n_total), icc.CURVE[curve], w, cores)
                    ok = ok and w.raw == bytes(o_t.cpu().numpy())
                cpu_mac_s = time.perf_counter() - t1
                all_ok = ok if all_ok is None else (all_ok and ok)
                cpu = {"value": round(2 * length * n_cols / (cpu_data_s + cpu_mac_s) / 1e9, 6), "unit": "Gsymbols/s", "cores": cores,
                       "kind": "port", "sample": "the same mix: data rows on oracle/icc_ref.c (1 thread, %.2f s), the two point arrays on "
                                                 "oracle/mac_ref.c (%d threads, %.2f s); CPU restatement, not NTL / gnark / libsecp256k1"
                                                 % (cpu_data_s, cores, cpu_mac_s)}
            data_k = {k: v for k, v in per_call.items() if "mix" in k and "mac" not in k}
            out["curves"][curve] = {"value": round(2 * length * n_cols / ms / 1e6, 3), "ms_per_step": round(ms, 4),
                                    "roofline": hbm_roofline({k: per_launch[k] for k in data_k} or per_launch, data_k or per_call,
                                                             256 * length * n_cols,
                                                             "the data kernel: 2 x 64 B in + 2 x 64 B out per butterfly; the call's wall time is "
                                                             "set by the point butterflies beside it (a chain of dependent group operations)",
                                                             workload="server_mix", curve=curve),
                                    "kernels_ms_per_call": per_call, "cpu_baseline": cpu, "bit_exact_vs_oracle": ok}
        for k in ("value", "ms_per_step", "roofline", "cpu_baseline"):
            out[k] = out["curves"]["bn254"][k]
        out["bit_exact_vs_oracle"] = all_ok
        return out

    # ---------------------------------------------------------------- the line
    legs = {"bn254_msm": leg_bn254_msm, "strong_2p20": leg_strong_2p20, "kzg_commit": leg_kzg_commit, "secp256k1_msm": leg_secp256k1_msm, "icc": leg_icc,
            "config3": leg_config3, "audit_combine": leg_audit_combine, "client_mac_batch": leg_client_mac_batch,
            "ipa_commits": leg_ipa_commits, "mac_encode": leg_mac_encode, "server_mix": leg_server_mix}
    t_leg = time.perf_counter()
    out = legs[args.workload]()
    out["wall_s"] = round(time.perf_counter() - t_leg, 1)      # this leg's share of the run (setup, timed regions, CPU baseline, counter passes)
    if rank == 0:
        print("[bench] %s: %.1f s" % (args.workload, out["wall_s"]), file=sys.stderr, flush=True)
    for k, v in (("n_gpus", world), ("steps", args.steps), ("warmup", args.warmup), ("higher_is_better", True), ("vs_baseline", None),
                 ("data", "synthetic")):
        out.setdefault(k, v)               # (a leg printed alone as the line: the contract's keys it leaves to the line)
    legs_failed = []
    if args.workload == "bn254_msm":
        # every other BASELINE.json configuration rides on the default line
        extra = []
        if not args.no_commits:
            extra.append(("kzg_commits", leg_kzg_commit))
        if world > 1:
            extra.append(("strong_2p20", leg_strong_2p20))
        if not args.no_legs:
            extra += [("secp256k1_msm", leg_secp256k1_msm), ("icc", leg_icc)]
            if world == 1:
                extra.append(("audit_combine", leg_audit_combine))
                extra.append(("kzg_audit", leg_kzg_audit))
                if not args.no_commits:
                    extra.append(("client_mac_batch", leg_client_mac_batch))
                    extra.append(("ipa_commits", leg_ipa_commits))
                extra += [("mac_encode", leg_mac_encode), ("server_mix", leg_server_mix)]
            if not args.no_config3:
                extra.append(("config3", leg_config3))
        for name, fn in extra:
            torch.cuda.empty_cache()
            try:
                if os.environ.get("PORLA_BENCH_FAIL_LEG") == name:      # test hook: the failed-leg path of this loop
                    raise RuntimeError("injected failure (PORLA_BENCH_FAIL_LEG)")
                t_leg = time.perf_counter()
                leg = fn()
                leg["wall_s"] = round(time.perf_counter() - t_leg, 1)
                if rank == 0:
                    print("[bench] %s: %.1f s" % (name, leg["wall_s"]), file=sys.stderr, flush=True)
            except Exception as e:  # noqa: BLE001  (a leg must not take the headline down with it: the line is still printed)
                if world > 1:
                    raise                      # ... except where the ranks would fall out of step
                import traceback
                traceback.print_exc()
                leg = {"error": repr(e), "bit_exact_vs_oracle": None}
                legs_failed.append(name)
            for k in ("n_gpus", "steps", "warmup", "higher_is_better", "vs_baseline", "data"):
                leg.pop(k, None)               # the leg shares the line's
            out[name] = leg
    failed = False
    if rank == 0:
        failed = out.get("bit_exact_vs_oracle") is False or any(
            isinstance(v, dict) and v.get("bit_exact_vs_oracle") is False for v in out.values())
        if fe_peak:
            out["fe_mul_peak"] = fe_peak
        # a leg that threw is named on the line; one that measures a BASELINE.json configuration also fails the run (rc 3) -- a
        # missing configuration must not pass silently.  A result that differs from the oracle fails the run with rc 1.
        out["legs_failed"] = legs_failed
        out["run_wall_s"] = round(time.perf_counter() - t_main, 1)     # this process, argument parsing to the line (per leg: wall_s in the legs file)
        if preflight:
            out["preflight"] = preflight
        emit(out, args.legs_out, single_leg=args.workload != "bn254_msm")
        if failed:
            print("ERROR: GPU result differs from the oracle", file=sys.stderr)
        if legs_failed:
            print("ERROR: legs failed: %s" % ", ".join(legs_failed), file=sys.stderr)
    if world > 1:
        if use_cxx_dist:
            mx.dist_finalize()
        dist.barrier()
        dist.destroy_process_group()
    code = 1 if failed else (3 if any(l in BASELINE_CONFIG_LEGS for l in legs_failed) else 0)
    if hard_exit:
        # the run went on over torch.distributed's collectives, but a thread of this process still sits inside RCCL: static
        # destructors must not run under it -- leave without them (everything is printed and flushed)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(code)
    if code:
        sys.exit(code)


if __name__ == "__main__":
    main()
