#!/bin/bash
# The whole evidence set of a round in ONE gpurun call, in the order that lets every bench line carry its counters:
#   1. rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE of the headline (run_pmc.sh), SQ counters (run_sq_pmc.sh)
#   2. counter passes of the other BASELINE shapes (run_wl_pmc.sh)
#   3. collect.py --counters-only: profiles/<round>/counters.json of THIS build, on the box
#   4. the bench lines (python bench.py ...), which now find counters measured on the build they run
# Locally afterwards: python profiles/tools/collect.py <round>   (recreates counters.json from gpurun_out/ and copies the lines)
#   usage: gpurun -- 'ROUND=r03 WLS="c2 c3b c4 ns:nw c3" bash profiles/run_evidence.sh'
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
ROUND=${ROUND:-r06}
WARMUP=1 STEPS=3 COUNTERS=" " bash $R/profiles/run_pmc.sh > $OUT/ev_stats.log 2>&1
NOSTATS=1 WARMUP=1 STEPS=1 bash $R/profiles/run_pmc.sh > $OUT/ev_pmc.log 2>&1
bash $R/profiles/run_sq_pmc.sh > $OUT/ev_sq.log 2>&1
WLS="${WLS:-c2 c2x8 c3b c3a c4 ns:nw ns4:nw c3}" bash $R/profiles/run_wl_pmc.sh > $OUT/ev_wl.log 2>&1
cd $R
python profiles/tools/collect.py $ROUND --counters-only > $OUT/ev_collect.log 2>&1
python bench.py --steps ${NS_STEPS:-5} --warmup 2 > $OUT/bench_ns_sw.json 2> $OUT/bench_ns_sw.err
for ITEM in ${WLS:-c3b c3a c4 ns:nw ns4:nw c3}; do
  WL=${ITEM%%:*}; MODE=sw; NAME=$WL; case $ITEM in *:*) MODE=${ITEM##*:}; NAME=${WL}_$MODE;; esac
  python bench.py --workload $WL --mode $MODE --steps ${WL_STEPS:-2} --warmup 1 --no-cpu-baseline --no-e2e > $OUT/bench_$NAME.json 2> $OUT/bench_$NAME.err
done
# round 5: the small-block shape at 8000 blocks (the batch fills the chip), config 2 with its end-to-end figure and CPU baseline,
# the reference's DRB1 ctest chain (wall seconds)
python bench.py --workload c2 --steps 5 --warmup 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python bench.py --workload c2x8 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_c2x8.json 2> $OUT/bench_c2x8.err
python bench.py --workload drb1 --steps 3 --warmup 1 > $OUT/bench_drb1.json 2> $OUT/bench_drb1.err
# the price of the default node order (spoa's, S7') against the incrementally kept one (--s7-order): headline and small-block shapes
LIBS="libsxgpoa.so" WLS="ns c2x8 c2" bash $R/profiles/tools/s7_ab.sh > $OUT/s7_price.txt 2>&1
tail -3 $OUT/ev_collect.log
cat $OUT/s7_price.txt
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    r = d["roofline"] or {"kernel_ms_per_launch": 0.0, "bound": "-", "frac": 0.0}
    print(sys.argv[1].split("/")[-1], round(d["value"], 3), d["unit"], "verified", d.get("verified"), "kernel ms", round(r["kernel_ms_per_launch"], 1), r["bound"], round(r["frac"], 3),
          "match", r.get("valu", {}).get("counters_match_build"), "e2e", (d.get("end_to_end") or {}).get("ratio_to_kernel_only"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
