#!/bin/bash
# usage: tools/evidence.sh <tag>   (on the GPU box)  — the per-round evidence set:
#   gpurun_out/<tag>_bench.json          python bench.py (default flags)
#   gpurun_out/<tag>_kernel_stats.md     rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/<tag>_pmc_fetch.md/_write.md   FETCH_SIZE / WRITE_SIZE, separate passes, no trace flags
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1
mkdir -p $R/gpurun_out
cd $R && timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 3000 gpurun_out/${tag}_bench.json
cd /tmp && export TMPDIR=/tmp
d=$R/gpurun_out/${tag}_stats; rm -rf $d
(cd $R && timeout 900 rocprofv3 --kernel-trace --stats -d $d -- python bench.py > $d.log 2>&1)
(cd $R && python profiles/summarize.py stats $(find $d -name "*.db" | head -1) > gpurun_out/${tag}_kernel_stats.md; head -12 gpurun_out/${tag}_kernel_stats.md)
for c in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/${tag}_pmc_$c; rm -rf $d
  (cd $R && timeout 1200 rocprofv3 --pmc $c -d $d -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $d.log 2>&1)
  lc=$(echo $c | tr A-Z a-z | sed 's/_size//')
  (cd $R && python profiles/summarize.py pmc $(find $d -name "*.db" | head -1) > gpurun_out/${tag}_pmc_$lc.md; head -8 gpurun_out/${tag}_pmc_$lc.md)
done
