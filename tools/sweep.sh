#!/bin/bash
# usage: tools/sweep.sh "<flags A>" "<flags B>" ...   — rebuild librfwhip.so with each flag set, run a short bench
# (development helper for the GPU box; results go to gpurun_out/sweep.log).  BENCH_ARGS: extra bench.py arguments.
mkdir -p gpurun_out
for flags in "$@"; do
  RFWHIP_EXTRA_FLAGS="$flags" python -c "import __graft_entry__ as g; g.load_package().build_native(force=True)" >/dev/null 2>gpurun_out/sweep_build.err || { echo "BUILD FAIL [$flags]"; tail -3 gpurun_out/sweep_build.err; continue; }
  out=$(timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --stage-rates ${BENCH_ARGS} 2>/dev/null | tail -1)
  echo "[$flags] $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['mrays_per_s_per_stage_serialised']; print(d['value'], d['ms_per_step'], d['stage_ms_per_step'], 'serial', s['stage_ms'] if s else None, 'mean', round(d['image_mean'],5))")"
done 2>&1 | tee -a gpurun_out/sweep.log
