"""Fold the outputs of profiles/run_evidence.sh (gpurun_out/) into the tracked evidence:

  profiles/<round>/counters.json   what bench.py's roofline reads: per-launch VALU instructions, dual-rate issues, HBM bytes,
                                   rocprof's duration of every launch, keyed by the SHA-256 of the kernel sources
  profiles/<round>/ns_sw_*         rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE collections, SQ counters of the headline
  profiles/<round>/wl/             raw rocprofv3 outputs of the other BASELINE shapes
  profiles/<round>/bench_*.json    the bench lines of the same gpurun call, AS PRINTED (bench.py found the counters, because
                                   run_evidence.sh runs this script with --counters-only on the box before it runs bench.py)
  profiles/traffic.json            the HBM figures with the guide's gfx950 correction spelled out

Usage: python profiles/tools/collect.py r03 [--counters-only]"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
rnd = args[0] if args else "r04"
counters_only = "--counters-only" in sys.argv
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (source_hash)

CORRECTION = ("MI355X_MICROARCH.md HBM section: counters are KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide "
              "coalesced streaming reads (calibrated there for 16 B/lane; the row ring is read 8 B/lane, 512 B per wave "
              "instruction, which the guide lists as uncalibrated) -> FETCH doubled as the conservative reading; "
              "WRITE_SIZE taken as is (uncalibrated)")


def bench_line(path):
    """Last JSON line of a bench.py run (log or .json)."""
    try:
        lines = [l for l in open(path, errors="replace").read().splitlines() if l.startswith("{") and '"metric"' in l]
        return json.loads(lines[-1]) if lines else None
    except Exception:
        return None


def counter_sum(path, name):
    return sum(float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "poa_block" in r["Kernel_Name"] and r["Counter_Name"] == name)


def per_step(path, name):
    """Counter `name` summed over EVERY poa_block dispatch of the run, per step.  Round 4: a step of the headline workload is two
    launches side by side (strips of 10 and of 11 columns: the geometries are no longer merged), so 'per launch' means per
    step -- all dispatches of one pass over the batch; steps = dispatches / distinct kernels."""
    rows = [(r["Kernel_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open(path)) if "poa_block" in r["Kernel_Name"] and r["Counter_Name"] == name]
    if not rows:
        return None, 0
    kernels = len(set(k for k, _ in rows))
    steps = max(1, len(rows) // max(kernels, 1))
    return sum(v for _, v in rows) / steps, steps


stats = glob.glob(os.path.join(OUT, "prof_stats", "*kernel_stats.csv"))[0]
stat_rows = [r for r in csv.DictReader(open(stats)) if "poa_block" in r["Name"]]
row = max(stat_rows, key=lambda r: float(r["TotalDurationNs"]))   # the launch that runs longest (the others run beside it)
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(OUT, "prof_pmc_" + c, "*counter_collection.csv"))[0]
    vals[c], _ = per_step(f, c)
    shutil.copy(f, os.path.join(dst, "ns_sw_pmc_%s.csv" % c))
shutil.copy(stats, os.path.join(dst, "ns_sw_kernel_stats.csv"))
# every launch of the dominant kernel in the stats run (the first one is the warm-up: it first-touches the arenas)
tr = glob.glob(os.path.join(OUT, "prof_stats", "*kernel_trace.csv"))
launch_ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(tr[0]))
             if r["Kernel_Name"] == row["Name"]] if tr else []
F, W = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
traffic = {"ns_sw": {
    "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} --output-format csv -- python bench.py --workload ns "
               "--steps 1 --warmup 1 --no-cpu-baseline --no-e2e (profiles/run_pmc.sh)",
    "kernel": row["Name"], "kernel_ms_avg_rocprof": float(row["AverageNs"]) / 1e6,
    "kernels_of_a_step": {r["Name"]: float(r["AverageNs"]) / 1e6 for r in stat_rows},
    "FETCH_SIZE_KB_per_launch": F, "WRITE_SIZE_KB_per_launch": W, "correction": CORRECTION,
    "hbm_bytes_per_launch": (2 * F + W) * 1024, "hbm_bytes_per_launch_uncorrected": (F + W) * 1024}}
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)

agg, cnt = {}, {}
for d in sorted(glob.glob(os.path.join(OUT, "pmc16_*/"))):
    for f in glob.glob(d + "*counter_collection.csv"):
        for name in sorted(set(r["Counter_Name"] for r in csv.DictReader(open(f)) if "poa_block" in r["Kernel_Name"])):
            agg[name], cnt[name] = per_step(f, name)   # per step: every dispatch of one pass over the batch
if agg:
    json.dump({"command": "profiles/run_sq_pmc.sh (ns workload, per step = all launches of one pass, mean over %d steps, longest kernel %s)" % (max(cnt.values()), row["Name"]), "counters": agg},
              open(os.path.join(dst, "ns_sw_sq_counters.json"), "w"), indent=1)

# cells per launch of the headline workload: from the bench line of the stats run itself
d_stats = bench_line(os.path.join(OUT, "bench_stats.log")) or bench_line(os.path.join(OUT, "bench_ns_sw.json"))
counters = {"source_sha256": bench.source_hash(),
            "collected_with": "profiles/run_evidence.sh: run_pmc.sh + run_sq_pmc.sh + run_wl_pmc.sh (rocprofv3 --kernel-trace --pmc, one "
                              "counter group per run; per-launch means over the warm-up and the timed launch)",
            "workloads": {"ns_sw": {
                "kernel": row["Name"], "kernels_of_a_step": {r["Name"]: float(r["AverageNs"]) / 1e6 for r in stat_rows},
                "cells_per_launch": d_stats["config"]["cells_per_step_per_gpu"],
                "kernel_ms_avg_rocprof": float(row["AverageNs"]) / 1e6, "kernel_ms_per_launch_rocprof": launch_ms,
                "SQ_INSTS_VALU": agg.get("SQ_INSTS_VALU"), "SQ_ACTIVE_INST_VALU2": agg.get("SQ_ACTIVE_INST_VALU2"),
                "SQ_WAIT_ANY": agg.get("SQ_WAIT_ANY"), "SQ_INSTS_SALU": agg.get("SQ_INSTS_SALU"),
                "SQ_WAVE_CYCLES": agg.get("SQ_WAVE_CYCLES"), "SQ_BUSY_CYCLES": agg.get("SQ_BUSY_CYCLES"),
                "FETCH_SIZE_KB": F, "WRITE_SIZE_KB": W, "hbm_bytes_per_launch": (2 * F + W) * 1024,
                "hbm_bytes_per_launch_uncorrected": (F + W) * 1024, "correction": CORRECTION}}}
# the other BASELINE shapes (profiles/run_wl_pmc.sh): counters summed over EVERY poa_block dispatch of one step
os.makedirs(os.path.join(dst, "wl"), exist_ok=True)
for d_v in sorted(glob.glob(os.path.join(OUT, "wl_*_SQ_INSTS_VALU"))):
    wl = os.path.basename(d_v)[3:-len("_SQ_INSTS_VALU")]
    tot = {}
    for c in ("SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(os.path.join(OUT, "wl_%s_%s" % (wl, c), "*counter_collection.csv"))
        if not f:
            continue
        tot[c] = counter_sum(f[0], c)
        if c == "SQ_INSTS_VALU":   # (same pass)
            tot["SQ_ACTIVE_INST_VALU2"] = counter_sum(f[0], "SQ_ACTIVE_INST_VALU2")
        if not counters_only:
            shutil.copy(f[0], os.path.join(dst, "wl", "wl_%s_%s_counter_collection.csv" % (wl, c)))
            kt = glob.glob(os.path.join(OUT, "wl_%s_%s" % (wl, c), "*kernel_trace.csv"))
            if kt:
                shutil.copy(kt[0], os.path.join(dst, "wl", "wl_%s_%s_kernel_trace.csv" % (wl, c)))
    bl = bench_line(os.path.join(OUT, "wl_%s_SQ_INSTS_VALU.log" % wl))
    if "SQ_INSTS_VALU" not in tot or not bl:
        continue
    e = {"kernel": "every poa_block_kernel dispatch of one step (%s)" % bl["roofline"]["kernel"],
         "cells_per_launch": bl["config"]["cells_per_step_per_gpu"], "per": "step (all launches and retry rounds)",
         "SQ_INSTS_VALU": tot["SQ_INSTS_VALU"], "SQ_ACTIVE_INST_VALU2": tot.get("SQ_ACTIVE_INST_VALU2", 0.0)}
    if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
        e.update({"FETCH_SIZE_KB": tot["FETCH_SIZE"], "WRITE_SIZE_KB": tot["WRITE_SIZE"],
                  "hbm_bytes_per_launch": (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024,
                  "hbm_bytes_per_launch_uncorrected": (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024, "correction": CORRECTION})
    counters["workloads"][wl.replace("+", "_") if "+" in wl else "%s_sw" % wl] = e
    print("workload %s: VALU instrs/step %.3e, %.3f per cell" % (wl, tot["SQ_INSTS_VALU"], tot["SQ_INSTS_VALU"] / e["cells_per_launch"]))
json.dump(counters, open(os.path.join(dst, "counters.json"), "w"), indent=1)
print("rocprof kernel avg %.1f ms; HBM bytes/launch %.3e (uncorrected %.3e); VALU instrs %.3e; dual-rate counter %.3e; SQ_WAIT_ANY/SQ_WAVE_CYCLES %.3f" % (
    traffic["ns_sw"]["kernel_ms_avg_rocprof"], traffic["ns_sw"]["hbm_bytes_per_launch"], traffic["ns_sw"]["hbm_bytes_per_launch_uncorrected"],
    agg.get("SQ_INSTS_VALU", float("nan")), agg.get("SQ_ACTIVE_INST_VALU2", float("nan")),
    agg.get("SQ_WAIT_ANY", float("nan")) / max(agg.get("SQ_WAVE_CYCLES", float("nan")), 1.0)))
if counters_only:
    sys.exit(0)

# bench lines of the same call, as printed
for src in sorted(glob.glob(os.path.join(OUT, "bench_*.json"))):
    bl = bench_line(src)
    if not bl:
        continue
    name = os.path.basename(src)
    open(os.path.join(dst, name), "w").write(json.dumps(bl) + "\n")
    r = bl["roofline"] or {"kernel_ms_per_launch": 0.0, "bound": "-", "frac": 0.0}
    print("%s: %.3f %s, kernel %.1f ms, bound %s, frac %.3f, counters_match_build %s" % (
        name, bl["value"], bl["unit"], r["kernel_ms_per_launch"], r["bound"], r["frac"], r.get("valu", {}).get("counters_match_build")))
