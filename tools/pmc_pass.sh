#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<counters of one pass>" ["<counters of another pass>" ...]
# Each pass is its own rocprofv3 --pmc run (no trace flags), summarised into gpurun_out/pmc_<tag>_<n>.md
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
n=0
for ctrs in "$@"; do
  n=$((n+1))
  d=$R/gpurun_out/pmc_${tag}_$n
  rm -rf $d
  (cd $R && timeout 600 rocprofv3 --pmc $ctrs -d $d -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline ${BENCH_ARGS} > $d.log 2>&1)
  (cd $R && python profiles/summarize.py pmc $(find $d -name "*.db" | head -1) > gpurun_out/pmc_${tag}_$n.md 2>&1; head -40 gpurun_out/pmc_${tag}_$n.md)
done
