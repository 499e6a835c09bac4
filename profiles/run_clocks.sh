# Sample shader clock / power while the headline bench runs (is the kernel power-limited?)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
( python $R/bench.py --workload ns --steps 2 --warmup 1 --no-cpu-baseline > $OUT/clk_bench.json 2>$OUT/clk_bench.err ) &
BP=$!
: > $OUT/clk_samples.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|power\|mclk\|junction" | tr '\n' ' ' >> $OUT/clk_samples.txt
  echo >> $OUT/clk_samples.txt
  sleep 1
done
tail -25 $OUT/clk_samples.txt | cut -c1-400
tail -1 $OUT/clk_bench.json | cut -c1-200
