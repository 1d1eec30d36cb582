#!/bin/bash
# usage: tools/quick_bench.sh [bench args...] — one short bench line, condensed (development helper for the GPU box)
out=$(timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | tail -1)
echo "[$*] $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'], 'roof', d['roofline']['achieved'] if d['roofline'] else None, 'mean', round(d['image_mean'],5))")"
