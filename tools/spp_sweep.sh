#!/bin/bash
# usage: tools/spp_sweep.sh "<spp list>" "<streams list>"  — Msamples/s of bench.py over batch size and sub-batch count
for spp in $1; do for st in $2; do
  steps=$(( 1024 / spp )); [ $steps -lt 6 ] && steps=6; [ $steps -gt 64 ] && steps=64
  out=$(timeout 300 python bench.py --spp $spp --streams $st --steps $steps --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1)
  echo "spp $spp streams $st: $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step')")"
done; done
