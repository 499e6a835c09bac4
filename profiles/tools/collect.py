"""Fold the outputs of profiles/run_pmc.sh and profiles/run_sq_pmc.sh (gpurun_out/) into the tracked
evidence: profiles/<round>/ns_sw_*, profiles/<round>/counters.json (what bench.py's roofline reads, keyed by the
SHA-256 of the kernel sources it was measured on) and profiles/traffic.json.
Usage: python profiles/tools/collect.py r02"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(OUT, "prof_stats", "*kernel_stats.csv"))[0]
row = [r for r in csv.DictReader(open(stats)) if "poa_block" in r["Name"]][0]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(OUT, "prof_pmc_" + c, "*counter_collection.csv"))[0]
    rows = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "poa_block" in r["Kernel_Name"] and r["Counter_Name"] == c]
    vals[c] = sum(rows) / len(rows)   # per launch: mean over the run's launches (warm-up + timed)
    shutil.copy(f, os.path.join(dst, "ns_sw_pmc_%s.csv" % c))
shutil.copy(stats, os.path.join(dst, "ns_sw_kernel_stats.csv"))
# every launch of the dominant kernel in the stats run (the first one is the warm-up: it first-touches the arenas)
tr = glob.glob(os.path.join(OUT, "prof_stats", "*kernel_trace.csv"))
launch_ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(tr[0]))
             if r["Kernel_Name"] == row["Name"]] if tr else []
F, W = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
traffic = {"ns_sw": {
    "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} --output-format csv -- python bench.py --workload ns "
               "--steps 1 --warmup 0 --no-cpu-baseline (profiles/run_pmc.sh)",
    "kernel": row["Name"], "kernel_ms_avg_rocprof": float(row["AverageNs"]) / 1e6,
    "FETCH_SIZE_KB_per_launch": F, "WRITE_SIZE_KB_per_launch": W,
    "correction": "MI355X_MICROARCH.md HBM section: counters are KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide "
                  "coalesced streaming reads (calibrated there for 16 B/lane; the row ring is read 8 B/lane, 512 B per wave "
                  "instruction, which the guide lists as uncalibrated) -> FETCH doubled as the conservative reading; "
                  "WRITE_SIZE taken as is (uncalibrated)",
    "hbm_bytes_per_launch": (2 * F + W) * 1024, "hbm_bytes_per_launch_uncorrected": (F + W) * 1024}}
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)

agg, cnt = {}, {}
for d in sorted(glob.glob(os.path.join(OUT, "pmc16_*/"))):
    for f in glob.glob(d + "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "poa_block" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                cnt[r["Counter_Name"]] = cnt.get(r["Counter_Name"], 0) + 1
agg = {k: v / cnt[k] for k, v in agg.items()}   # per launch: mean over the run's launches
if agg:
    json.dump({"command": "profiles/run_sq_pmc.sh (ns workload, mean over %d launches, kernel %s)" % (max(cnt.values()), row["Name"]), "counters": agg},
              open(os.path.join(dst, "ns_sw_sq_counters.json"), "w"), indent=1)
b = os.path.join(OUT, "bench_ns_sw.json")
if os.path.exists(b):
    d = json.loads(open(b).read().strip().splitlines()[-1])
    sys.path.insert(0, ROOT)
    import bench
    counters = {"source_sha256": bench.source_hash(),
                "collected_with": "profiles/run_pmc.sh + profiles/run_sq_pmc.sh (rocprofv3 --kernel-trace --pmc, one counter "
                                  "group per run; per-launch means over the warm-up and the timed launch)",
                "workloads": {"ns_sw": {
                    "kernel": row["Name"], "cells_per_launch": d["config"]["cells_per_step_per_gpu"],
                    "kernel_ms_avg_rocprof": float(row["AverageNs"]) / 1e6, "kernel_ms_per_launch_rocprof": launch_ms,
                    "SQ_INSTS_VALU": agg.get("SQ_INSTS_VALU"), "SQ_INSTS_SALU": agg.get("SQ_INSTS_SALU"),
                    "SQ_WAVE_CYCLES": agg.get("SQ_WAVE_CYCLES"), "SQ_BUSY_CYCLES": agg.get("SQ_BUSY_CYCLES"),
                    "FETCH_SIZE_KB": F, "WRITE_SIZE_KB": W, "hbm_bytes_per_launch": (2 * F + W) * 1024,
                    "hbm_bytes_per_launch_uncorrected": (F + W) * 1024,
                    "correction": traffic["ns_sw"]["correction"]}}}
    # the other BASELINE shapes (profiles/run_wl_pmc.sh): counters summed over EVERY poa_block dispatch of one step
    for d_v in sorted(glob.glob(os.path.join(OUT, "wl_*_SQ_INSTS_VALU"))):
        wl = os.path.basename(d_v)[3:-len("_SQ_INSTS_VALU")]
        tot = {}
        for c in ("SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE"):
            f = glob.glob(os.path.join(OUT, "wl_%s_%s" % (wl, c), "*counter_collection.csv"))
            if not f:
                continue
            tot[c] = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "poa_block" in r["Kernel_Name"] and r["Counter_Name"] == c)
        log = os.path.join(OUT, "wl_%s_SQ_INSTS_VALU.log" % wl)
        line = [l for l in open(log).read().splitlines() if l.startswith("{") and '"metric"' in l]
        if "SQ_INSTS_VALU" not in tot or not line:
            continue
        bl = json.loads(line[-1])
        e = {"kernel": "every poa_block_kernel dispatch of one step (%s)" % bl["roofline"]["kernel"],
             "cells_per_launch": bl["config"]["cells_per_step_per_gpu"], "per": "step (all launches and retry rounds)",
             "SQ_INSTS_VALU": tot["SQ_INSTS_VALU"]}
        if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
            e.update({"FETCH_SIZE_KB": tot["FETCH_SIZE"], "WRITE_SIZE_KB": tot["WRITE_SIZE"],
                      "hbm_bytes_per_launch": (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024,
                      "hbm_bytes_per_launch_uncorrected": (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024,
                      "correction": traffic["ns_sw"]["correction"]})
        counters["workloads"][wl.replace("+", "_") if "+" in wl else "%s_sw" % wl] = e
        print("workload %s: VALU instrs/step %.3e, %.3f per cell" % (wl, tot["SQ_INSTS_VALU"], tot["SQ_INSTS_VALU"] / e["cells_per_launch"]))
    json.dump(counters, open(os.path.join(dst, "counters.json"), "w"), indent=1)
    # bench lines of those shapes (gpurun_out/bench_<workload>[_nw].json): lines printed before their counters existed are
    # restated with them, exactly as bench.py does when it finds the counters
    for key, e in counters["workloads"].items():
        name = key[:-3] if key.endswith("_sw") else key
        src = os.path.join(OUT, "bench_%s.json" % name)
        if key == "ns_sw" or not os.path.exists(src):
            continue
        txt = [l for l in open(src).read().splitlines() if l.startswith("{")]
        if not txt:
            continue
        bl = json.loads(txt[-1])
        r = bl["roofline"]
        if r.get("bound") != "valu" and r.get("valu"):
            cells = bl["config"]["cells_per_step_per_gpu"] * bl["steps"]
            k_s = bl["roofline"]["kernel_ms_total"] / 1e3 if "kernel_ms_total" in bl["roofline"] else None
            if k_s:
                ipc = e["SQ_INSTS_VALU"] / e["cells_per_launch"]
                peak = r["valu"]["peak_wave_insts_per_s"]
                r.update({"bound": "valu", "achieved": ipc * cells / k_s / 1e9, "peak": peak / 1e9, "unit": "G wave-instructions/s",
                          "frac": ipc * cells / k_s / peak})
                r["valu"].update({"wave_insts_per_cell": ipc, "counter_file": "profiles/%s/counters.json" % rnd, "counters_match_build": True,
                                  "note": "counters collected in the same gpurun call as this bench line, on the same build"})
        open(os.path.join(dst, "bench_%s.json" % name), "w").write(json.dumps(bl) + "\n")
        print("bench %s: %.1f %s, bound %s, frac %.3f" % (name, bl["value"], bl["unit"], r["bound"], r["frac"]))
    d["roofline"]["traffic"] = traffic["ns_sw"]["hbm_bytes_per_launch"]
    # the bench line of this same gpurun call was printed BEFORE these counters existed: restate its VALU roofline with them
    r = d["roofline"]
    k_s = r["kernel_ms_per_launch"] / 1e3
    ipl = agg.get("SQ_INSTS_VALU")
    if ipl and r.get("valu"):
        peak = r["valu"]["peak_wave_insts_per_s"]
        r.update({"bound": "valu", "achieved": ipl / k_s / 1e9, "peak": peak / 1e9, "unit": "G wave-instructions/s",
                  "frac": ipl / k_s / peak})
        r["valu"].update({"wave_insts_per_launch": ipl, "wave_insts_per_cell": ipl / d["config"]["cells_per_step_per_gpu"],
                          "counter_file": "profiles/%s/counters.json" % rnd, "counters_match_build": True,
                          "note": "counters collected in the same gpurun call as this bench line, on the same build"})
    open(os.path.join(dst, "bench_ns_sw.json"), "w").write(json.dumps(d) + "\n")
    print("bench: %.1f %s, kernel %.1f ms, frac %.3f" % (d["value"], d["unit"], d["roofline"]["kernel_ms_per_launch"], d["roofline"]["frac"]))
print("rocprof kernel avg %.1f ms; HBM bytes/launch %.3e (uncorrected %.3e); VALU instrs %.3e" % (
    traffic["ns_sw"]["kernel_ms_avg_rocprof"], traffic["ns_sw"]["hbm_bytes_per_launch"],
    traffic["ns_sw"]["hbm_bytes_per_launch_uncorrected"], agg.get("SQ_INSTS_VALU", float("nan"))))
