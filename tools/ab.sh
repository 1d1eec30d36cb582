#!/bin/bash
# usage: tools/ab.sh "<bench args A>" "<bench args B>" ...  — same build, one short bench (with serialised stage times) per
# argument set (development helper for the GPU box; appends to gpurun_out/ab.log)
mkdir -p gpurun_out
for a in "$@"; do
  out=$(timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --pmc off --stage-rates $a 2>gpurun_out/ab.err | tail -1)
  echo "[$a] $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['mrays_per_s_per_stage_serialised']; print(d['value'], d['ms_per_step'], 'serial', s['stage_ms'] if s else None, 'mean', round(d['image_mean'],6))" 2>&1 | tail -1)"
done 2>&1 | tee -a gpurun_out/ab.log
