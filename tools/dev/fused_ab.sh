#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/rendering-fw_amd/librfwhip.so /tmp/librfwhip_base.so
for v in "$@"; do
  cp $R/tools/dev/variants/$v.so $R/rendering-fw_amd/librfwhip.so
  echo -n "$v "; (cd $R && python bench.py --steps 6 --warmup 2 --no-cpu-baseline --pmc off $FARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['candidates_ms_per_sub_batch'])")
done
cp /tmp/librfwhip_base.so $R/rendering-fw_amd/librfwhip.so
