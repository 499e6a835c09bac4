"""GPU: bench.py prints ONE JSON line with the fields the driver's contract names."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    for k in ("valu_frac", "algorithmic_frac", "hbm_counter_frac", "hbm_8d_frac"):   # every roof side by side
        assert k in r, k
    assert "verified" in d and d["verified"] is None      # (no committed full-shape fixture for the smoke workload)
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0


def test_headline_line_takes_the_valu_roof_from_the_committed_counters():
    """The headline workload prices its roofline with the counters under profiles/rNN/counters.json: the `valu` branch
    of bench.py (the tiny workload above has no counters and only ever exercises the HBM fallback).  frac must be
    recomputable from the fields of the line itself."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "ns", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--no-e2e"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    r, v = d["roofline"], d["roofline"]["valu"]
    assert r["bound"] == "valu" and r["unit"] == "G VALU issue cycles/s"
    assert 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert "counters_match_build" in v and v["counter_file"].startswith("profiles/r")
    # frac = (4 * (insts - dual) + 2 * dual) / (SIMDs * clock * kernel seconds), per launch
    need = v["issue_cycles_half_rate"] * (v["wave_insts_per_launch"] - v["dual_rate_insts_per_launch"]) + \
        v["issue_cycles_full_rate"] * v["dual_rate_insts_per_launch"]
    had = v["simds"] * v["clock_GHz"] * 1e9 * r["kernel_ms_per_launch"] / 1e3
    assert abs(need / had - r["frac"]) < 1e-6 * max(1.0, r["frac"])
    assert 0.0 < v["useful_frac"] <= r["frac"] and v["used_columns"] <= v["swept_columns"]
    assert 0.0 <= v["dual_rate_share"] < 1.0
    if v["counters_match_build"]:
        assert r["traffic"] and r["traffic"] > 0
    assert r["hbm"]["algo_bytes_per_cell"] == 13.0
    # round 6: the copy bandwidth of THIS box beside the data sheet's peak, the gap model's own instruction floor,
    # and whether the ABI's phases run inside ROCTx ranges
    assert 1000.0 < r["hbm"]["copy_GBps_measured"] < r["hbm"]["peak_GBps"], r["hbm"]["copy_GBps_measured"]
    assert v["gap_model"] == "convex" and 23.5 + 40.0 / 13 <= v["min_insts_per_cell"] <= 23.5 + 40.0 / 4
    assert d["engine"]["roctx_ranges"] in (True, False)
    assert d["dtype"] == "int16" and d["config"]["blocks_per_gpu"] == 1000
    # the line certifies itself: blocks 0 and 999 of the timed batch against the committed oracle output
    assert d["verified"] is True and d["verified_blocks"] == [0, 999]
    assert r["valu_frac"] == r["frac"] and r["hbm_8d_frac"] > 1.0 and 0.0 < r["algorithmic_frac"] < r["frac"]


def test_a_wrong_result_fails_the_bench(tmp_path):
    """The verification is not decoration: with a fixture whose expected scores are off by one the bench exits non-zero."""
    import shutil
    fake = tmp_path / "repo"
    shutil.copytree(ROOT, fake, ignore=shutil.ignore_patterns(".git", "gpurun_out", "profiles", "__pycache__", "*.o"))
    fx = fake / "tests" / "golden" / "fullshape_oracle.json"
    j = json.loads(fx.read_text())
    for c in j["cases"]:
        if c["name"] == "c2":
            c["scores"][3] += 1
    fx.write_text(json.dumps(j))
    out = subprocess.run([sys.executable, str(fake / "bench.py"), "--workload", "c2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e"],
                         capture_output=True, text=True, timeout=600, cwd=str(fake))
    assert out.returncode != 0 and "bench verification failed" in (out.stderr + out.stdout)


def test_drb1_workload_reports_wall_seconds_and_checks_every_iteration():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "drb1", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["unit"] == "s" and d["higher_is_better"] is False and d["verified"] is True
    assert len(d["iteration_seconds"]) == 3 and abs(sum(d["iteration_seconds"]) - d["value"]) < 0.05 * d["value"] + 0.05
    assert d["config"]["blocks_per_iteration"] == [2161, 2066, 2025]


def test_the_default_order_is_spoas_and_the_old_one_is_an_option():
    """Round 6: the bench runs spoa's node order (decree S7': sxg_poa_params::mode | SXG_ORDER_SPOA, what src/smooth.cpp:764 does) by
    default and verifies itself against committed fixtures IN THAT ORDER; `--s7-order` runs the incrementally kept order of
    rounds 1-5 against its own fixtures."""
    for flag, want in ((None, "spoa"), ("--s7-order", "incremental")):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e"]
        out = subprocess.run(cmd + ([flag] if flag else []), capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert d["config"]["order"].startswith(want) and d["verified"] is True and d["verified_blocks"] == [0, 999] and d["value"] > 0, d["config"]


def test_the_banded_workloads_verify_themselves():
    """Round 6: config 3 on the `-A` path (c3b: static band, c3a: abPOA's adaptive band) carries committed fixtures -- blocks 0 and
    4999 of the 5000 -- so those lines say `verified: true` too (order s7: abPOA does not call spoa's sort)."""
    for wl in ("c3b", "c3a"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--blocks", "48", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e"],
                             capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        # (48 of the 5000 blocks here: block 0 is in the batch, block 4999 only in the full run)
        assert d["verified"] is True and d["verified_blocks"] == [0] and d["config"]["order"].startswith("incremental"), (wl, d["verified_what"])


def test_gpus_flag_refuses_to_time_fewer_gpus_than_it_claims():
    """Round 6: `python bench.py --gpus N` without a launcher starts its own N ranks (torch.distributed.run on 127.0.0.1) -- and on a
    box with fewer GPUs it says so and stops, instead of timing one GPU under `n_gpus: 1` (what round 5 did: --gpus was never
    read).  Started by a launcher with another world size it stops as well."""
    import torch
    n_dev = torch.cuda.device_count()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n_dev + 1), "--workload", "tiny", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--no-e2e"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode != 0 and "refusing to time fewer GPUs" in out.stderr, out.stderr[-1500:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--no-e2e"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr, out.stderr[-1500:]


def test_one_rank_through_the_launcher_runs_the_multi_rank_path():
    """`--gpus 1 --launch`: bench.py starts its one rank through torch.distributed.run, the rank puts a (one-rank) RCCL communicator
    on the engine handle and every step is sxg_poa_batch_execute_sharded -- size all-gather, grouped send/recv (empty), rank-0
    bookkeeping: the path an N-GPU job takes, as far as one GPU can walk it.  The line says who it saw."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launch", "--workload", "tiny", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-e2e"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert lines, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["verified"] is None or d["verified"] is True
    ex = d["exchange"]
    assert ex is not None and ex["path"].startswith("C ABI") and ex["ranks_seen"] == 1 and ex["launcher"].startswith("bench.py started"), ex
    assert ex["pack_ms"] >= 0 and ex["exchange_ms"] >= 0 and "bytes_received" in ex
