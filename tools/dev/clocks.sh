#!/bin/bash
# usage (GPU box): tools/dev/clocks.sh [bench args] — the bench with rocm-smi polled beside it: shader clock, power, temperature
# while the step runs (is the chip at its peak clock under this load?).  gpurun_out/clocks.log
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
(python bench.py --steps 24 --warmup 2 --no-cpu-baseline --no-roofline "$@" > gpurun_out/clocks_bench.json 2>gpurun_out/clocks_bench.err) &
pid=$!
: > gpurun_out/clocks.log
while kill -0 $pid 2>/dev/null; do
  (date +%s.%N; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|hotspot)" ) | tr '\n' ' ' >> gpurun_out/clocks.log
  echo >> gpurun_out/clocks.log
  sleep 0.4
done
tail -1 gpurun_out/clocks_bench.json | cut -c1-200
grep -c . gpurun_out/clocks.log
