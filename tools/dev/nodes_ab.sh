#!/bin/bash
# usage (GPU box): tools/dev/nodes_ab.sh NAME...  — per variant library: Msamples/s, serialised stage times and node visits / triangle tests per ray
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/rendering-fw_amd/librfwhip.so /tmp/librfwhip_base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/librfwhip_base.so $R/rendering-fw_amd/librfwhip.so; else cp $R/tools/dev/variants/$v.so $R/rendering-fw_amd/librfwhip.so; fi
  echo -n "$v "; (cd $R && python bench.py --steps 4 --warmup 2 --no-cpu-baseline --pmc off --stage-rates $FARGS 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; p=r['per_ray']
print(d['value'], r['candidates_ms_per_sub_batch'], 'inner/ray %.2f tris/ray %.2f shadow inner %.2f tris %.2f primary inner %.2f' % (p['inner_nodes'], p['triangle_tests'], p['shadow_inner_nodes'], p['shadow_triangle_tests'], p['primary_inner_nodes']))")
done
cp /tmp/librfwhip_base.so $R/rendering-fw_amd/librfwhip.so
